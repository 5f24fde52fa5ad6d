// jpeg_amd/csrc/common.hpp -- shared definitions of the HIP codec core (host + device).
//
// Data model restated from the reference (common.h:52-156, coeffs.h:9-16) for the GPU:
//   * coefficients live in HBM as int16, in SCAN ORDER: block index = mcu * B + k, where k walks the
//     blocks of one MCU the way read_macroblock()/write_macroblock() do (component, v, h;
//     decoder.c:319-357, encoder.c:425-461), and in ZIG-ZAG order inside a block (one 128-byte
//     line per block).  The reference keeps int32, per component, raster order (a1 in SURVEY 8a);
//     the taps in api.hip convert back for parity tests.
//   * one batch = frames that share geometry (Motion JPEG); tables are per frame.
#pragma once

#include <stdint.h>
#include <stddef.h>

#define JH_MAX_COMP 4
#define JH_MAX_MCU_BLOCKS 10          // JPEG limit on sum(H*V) in an interleaved scan

// ---- Huffman decode look-up ---------------------------------------------------------------
#define JH_LUT_BITS 10                // fast table index width
#define JH_SUBSEQ_SHIFT 11            // log2 of the bits per speculative sub-sequence (one thread each): 2048 is the measured
                                      // best for ordinary images; JhDecodeWork.sub_shift is what the kernels use (dense data: 12)
#define JH_SUBSEQ_SHIFT_DENSE 12
#ifndef JH_CHAIN_MAX
#define JH_CHAIN_MAX 16               // steps a speculative chain may walk before it is cut
#endif
#define JH_SEQ_OWN (256 - JH_CHAIN_MAX)   // sub-sequences owned by one workgroup (the rest are ghosts)
// chains workgroup b covers sub-sequences [b * OWN, b * OWN + 256) and EMITS the entry records of
// [b * OWN + (b ? JH_CHAIN_MAX : 0), b * OWN + 256): the first JH_CHAIN_MAX belong to its predecessor

#define JH_SUB_TABLES 8               // second-level tables (64 entries) per Huffman table
#define JH_LUT_SEARCH 0xffffu         // first-level entry: more long-code prefixes than second-level tables
#define JH_LUT_INVALID 0xfffeu        // first-level entry: no code starts with these bits

// One Huffman table, decoder view (Annex C / F.2.2.3).  Entry format of fast[] and sub[][]:
//   bit 15 = 0: (total << 10) | (run << 4) | category, total = code length + category (<= 31),
//               run = zero run, 63 for the EOB of an AC table; sub[][]: 0 = no code matches
//   bit 15 = 1 (fast[] only): the code is longer than JH_LUT_BITS bits; the low bits name the
//               second-level table indexed by the next 6 bits, JH_LUT_SEARCH = canonical search,
//               JH_LUT_INVALID = no code matches
// pair[] (AC tables; 0 = none), same index as fast[]: the window's first symbol AND the one behind it, where the
// second code also lies inside the JH_LUT_BITS bits of the index (the metric stream averages 5.1 bits a symbol: two
// in three look-ups resolve a pair, and the decode loops take a third fewer trips):
//   bits 11..15 total bits of both symbols (codes + magnitude bits, <= 31)
//   bits  4..10 run1 + 1 + run2: from the zig-zag index in front of the first symbol to the coefficient of the second
//               (run2 = 63 for EOB, so that the sum says "block complete" like the single entry does)
//   bits  0..3  category of the second symbol
// A pair is only built behind a first symbol that is not EOB; the decoder uses it only if the first symbol does not
// complete the block by itself (zero run to coefficient 63 and beyond in damaged data).
// DC tables use neither pair[] nor sub[] (their rare long codes take the canonical search): the kernels' LDS copy
// of a DC table ends in front of sub[] (JH_LUT_DC_BYTES) -- that is what lets three workgroups of k_huff_write share
// a CU's LDS.
struct JhHuffLut {
	uint16_t fast[1 << JH_LUT_BITS];
	uint32_t limit[17];               // limit[l] = (maxcode[l] + 1) << (16 - l), non-decreasing; l = 1..16
	int32_t  offs[17];                // valptr[l] - mincode[l]
	uint8_t  huffval[256];
	uint8_t  pad[8];                  // the short form and the whole are multiples of 16 (table offsets are kept in units of 16)
	uint16_t sub[JH_SUB_TABLES][64];
	uint16_t pair[1 << JH_LUT_BITS];
};
#define JH_LUT_DC_BYTES 2448u         // offsetof(JhHuffLut, sub)
#define JH_LUT_PAIR_OFF 3472u         // offsetof(JhHuffLut, pair)
#define JH_PAIR_TOTAL(p) ((p) >> 11)
#define JH_PAIR_ADV(p) (((p) >> 4) & 127u)
#define JH_PAIR_CAT2(p) ((p) & 15u)

struct JhHuffEnc {                    // one Huffman table, encoder view (EHUFCO/EHUFSI, huffman.c:118-148)
	uint32_t code[256];               // (code << 8) | size ; size == 0: symbol absent
};

// ---- geometry -------------------------------------------------------------------------------
struct JhGeom {
	int P, X, Y, Nf;
	int ncomp;
	int id[JH_MAX_COMP], H[JH_MAX_COMP], V[JH_MAX_COMP];
	int Tq[JH_MAX_COMP], Td[JH_MAX_COMP], Ta[JH_MAX_COMP];
	int scan_order[JH_MAX_COMP];      // component (index into id/H/V..) of the j-th scan component
	int max_H, max_V;
	int m_x, m_y;
	int size_x, size_y;
	int bx[JH_MAX_COMP], by[JH_MAX_COMP];
	int B;                            // blocks per MCU in scan order
	int nmcu;                         // m_x * m_y
	int nblk;                         // nmcu * B
	int blk_comp[JH_MAX_MCU_BLOCKS];  // component of block k of an MCU
	int blk_h[JH_MAX_MCU_BLOCKS], blk_v[JH_MAX_MCU_BLOCKS];
	int comp_first[JH_MAX_COMP];      // first block of the component inside an MCU
	int step_x[JH_MAX_COMP], step_y[JH_MAX_COMP];   // frame.c:57-58 replication / box factors
	int out_comps;                    // 1 or 3
	int sample_bytes;                 // 1 or 2
	int Ri;
};

// per-frame quantisation tables as the kernels want them: [table][zig-zag position]
struct JhQuant {
	uint16_t q[4][64];
};

// ---- speculative Huffman decoder records ---------------------------------------------------
// Exit state of a sub-sequence: where the first symbol NOT belonging to it starts, and the
// position inside the MCU / block.  Packed in one 64-bit word so that it is read/written atomically.
//   bits  0..31 pos (bit offset in the frame's unstuffed stream)
//   bits 32..39 c   (block inside the MCU)
//   bits 40..47 z   (next zig-zag index, 0 = DC expected)
//   bit  48     reset seen inside the sub-sequence (restart boundary crossed)
struct JhSubInfo {
	unsigned long long state;
	uint32_t seg;                     // restart segment index at the exit position
	int32_t  nb;                      // blocks completed (MCU-rounded at restart boundaries)
	int32_t  dc[JH_MAX_COMP];         // sum of DC differences per component since the last reset
};

struct JhSubStart {                   // entry of a sub-sequence for the write pass, RELATIVE to its chains workgroup
	unsigned long long state;         // parse state at entry; bit 48: a restart boundary lies between the
	                                  // workgroup's first emitted sub-sequence and this one
	uint32_t seg;
	int32_t  blk;                     // blocks completed before, inside the workgroup's emitted range
	int32_t  pred[JH_MAX_COMP];       // DC sums since the range start / the last restart boundary
};

enum {                                // JhWgAgg.flags
	JH_AGG_RESET = 1,                 // a restart boundary lies inside the emitted range
	JH_AGG_GAP = 2,                   // the workgroup's path did not reach every sub-sequence of its range
};
#define JH_SEED_NONE 0xffffffffu

struct JhWgAgg {                      // one chains workgroup: totals of its emitted range (or exclusive prefix)
	int32_t  nb;
	uint32_t flags;
	int32_t  dc[JH_MAX_COMP];
	uint32_t next_seed;               // unused (kept for the 32-byte layout)
	uint32_t pad;
};

#ifdef __HIPCC__
#define JH_HD __host__ __device__
#else
#define JH_HD
#endif

// number of chains workgroups that emit something for a stream of nsub sub-sequences
JH_HD static inline uint32_t jh_num_chain_wgs(uint32_t nsub)
{
	return nsub <= 256u ? 1u : 1u + (nsub - 256u + JH_SEQ_OWN - 1u) / JH_SEQ_OWN;
}

// zig-zag position -> raster index (common.h:34-43)
static const uint8_t jh_zigzag[64] = {
	 0,  1,  8, 16,  9,  2,  3, 10, 17, 24, 32, 25, 18, 11,  4,  5,
	12, 19, 26, 33, 40, 48, 41, 34, 27, 20, 13,  6,  7, 14, 21, 28,
	35, 42, 49, 56, 57, 50, 43, 36, 29, 22, 15, 23, 30, 37, 44, 51,
	58, 59, 52, 45, 38, 31, 39, 46, 53, 60, 61, 54, 47, 55, 62, 63
};
