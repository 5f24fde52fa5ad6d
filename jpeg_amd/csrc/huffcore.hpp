// jpeg_amd/csrc/huffcore.hpp -- the per-thread Huffman decoding core shared by the synchronisation
// kernels (huffsync.hip) and the write kernel (huffdec.hip).  Device code only.
//
// One loop iteration = one Huffman symbol, DC and AC alike (read_dc / read_ac, coeffs.c:86-161):
// the fast table yields (code length, zero run, category) with EOB pre-encoded as "run 63", so
//     k = z + run;  coefficient k <- value;  z = k + 1;  z >= 64 ends the block
// covers DC (z = 0, run = 0), (run, size) pairs, ZRL (run 15, category 0) and EOB without branches.
// The bit stream is read through a small interface: RegStream (a queue of words in registers,
// refilled by one 16-byte global load per four words, a quad ahead) or GlobalStream (plain word loads,
// careful paths).  The bit
// window is three 32-bit words in registers, and the NEXT symbol's table look-up is issued before
// the current symbol's value is extracted, so no load sits on the symbol-to-symbol critical path
// except that look-up.
#pragma once

#include <hip/hip_runtime.h>
#include "common.hpp"
#include "kernels.hpp"

struct DecTables {                    // workgroup-shared decode context (LDS)
	uint32_t slot_off[9];             // byte offset of every table of the frame inside the LDS copy (DC tables are short there)
	// per block of the MCU: (DC table offset / 16) | (AC table offset / 16) << 12 | component << 24
	uint32_t info[JH_MAX_MCU_BLOCKS];
	// the same, unpacked for the hot loop (one 16-byte read): x = byte offset of the DC table inside the
	// LDS copy, y = of the AC table, z = component * zunit (load_tables: the byte offset of the component's DC
	// accumulator relative to the thread's first one, or the component itself), w = the block that follows in the MCU
	uint4 ent[JH_MAX_MCU_BLOCKS];
};

struct DecState {
	uint32_t pos;                     // bit position of the next symbol
	uint32_t seg;                     // restart segment that contains pos
	int c, z;                         // block inside the MCU, next zig-zag index (0 = DC expected)
	int reset;                        // a restart boundary was crossed
	int nb;                           // blocks completed (partial MCUs before a boundary are discarded)
	int dc[JH_MAX_COMP];              // DC difference sums since the last reset (WRITE: the predictors)
};

__device__ __forceinline__ unsigned long long pack_state(const DecState &s)
{
	return (unsigned long long)s.pos | ((unsigned long long)(s.c & 0xff) << 32) |
	       ((unsigned long long)(s.z & 0xff) << 40) | ((unsigned long long)(s.reset & 1) << 48);
}

// ---- bit-stream sources -------------------------------------------------------------------------

struct GlobalStream {
	const uint8_t *cs;
	__device__ __forceinline__ uint32_t word(uint32_t w) const
	{
		return __builtin_bswap32(reinterpret_cast<const uint32_t *>(cs)[w]);
	}
};

// The stream read straight from global memory by a thread that walks it sequentially: the two window
// words plus four look-ahead words live in registers, the four after those are already in flight
// (one 16-byte load per four words consumed, issued a full quad ahead, so its latency is never
// exposed).  No LDS: this is what lets the synchronisation and write kernels run at full occupancy.
struct RegStream {
	const uint8_t *cs;
	struct Quad {
		uint32_t x, y, z, w;
	};
	struct Cursor {
		uint32_t b, c, d;             // the words after w2
		int nq;                       // valid words among (w2, b, c, d)
		Quad ld;                      // the four words after d (raw byte order), in flight
		uint32_t nxt;                 // stream word index of ld.x
	};
	__device__ __forceinline__ uint32_t word(uint32_t w) const
	{
		return __builtin_bswap32(reinterpret_cast<const uint32_t *>(cs)[w]);
	}
	__device__ __forceinline__ Quad load4(uint32_t w) const
	{
		const uint32_t *p = reinterpret_cast<const uint32_t *>(cs) + w;    // 4-byte aligned is enough
		Quad q;
		q.x = p[0];
		q.y = p[1];
		q.z = p[2];
		q.w = p[3];
		return q;
	}
	__device__ __forceinline__ void open(Cursor &cur, uint32_t wi, uint32_t &w0, uint32_t &w1, uint32_t &w2) const
	{
		w0 = word(wi);
		w1 = word(wi + 1);
		w2 = word(wi + 2);
		cur.b = word(wi + 3);
		cur.c = word(wi + 4);
		cur.d = word(wi + 5);
		cur.nq = 4;
		cur.nxt = wi + 6;
		cur.ld = load4(cur.nxt);
	}
	__device__ __forceinline__ uint32_t next(Cursor &cur, bool cross, uint32_t w2) const
	{
		uint32_t a = cross ? cur.b : w2;
		cur.b = cross ? cur.c : cur.b;
		cur.c = cross ? cur.d : cur.c;
		cur.nq -= cross ? 1 : 0;
		if (cur.nq == 0) {
			a = __builtin_bswap32(cur.ld.x);
			cur.b = __builtin_bswap32(cur.ld.y);
			cur.c = __builtin_bswap32(cur.ld.z);
			cur.d = __builtin_bswap32(cur.ld.w);
			cur.nq = 4;
			cur.nxt += 4;
			cur.ld = load4(cur.nxt);
		}
		return a;
	}
};

// per-thread DC accumulators live in LDS as [component][thread]: one ds_add per DC symbol instead
// of a select chain over four registers
#define DCS_AT(dcs, k, stride) ((dcs)[(k) * (stride)])

__device__ __forceinline__ uint32_t bit_window(uint32_t w0, uint32_t w1, uint32_t pos)
{
	return (uint32_t)((((unsigned long long)w0 << 32 | w1) << (pos & 31u)) >> 32);
}

// F.12: extend the sign of a cat-bit magnitude field (coeffs.c:33-48); cat == 0 gives 0
__device__ __forceinline__ int extend_sign(unsigned extra, int cat)
{
	return (int)extra - ((extra >> ((cat - 1) & 31)) ? 0 : (1 << cat) - 1);
}

// fast[] and pair[] entry of the window in one word (pair in the high half): two 16-bit LDS reads off one address.
// Of a DC table only the low half means anything: its LDS copy has no pair[] (the read lands in the next table).
__device__ __forceinline__ unsigned lut_lookup(const unsigned char *tab, uint32_t win)
{
	const unsigned char *p = tab + ((win >> (32 - JH_LUT_BITS)) << 1);
	return (unsigned)*reinterpret_cast<const uint16_t *>(p) |
	       ((unsigned)*reinterpret_cast<const uint16_t *>(p + JH_LUT_PAIR_OFF) << 16);
}

// resolves a first-level entry with bit 15 set: second-level table, canonical search, or 0 = no code
__device__ __forceinline__ unsigned long_code(const unsigned char *tab, unsigned e, uint32_t win, bool isdc)
{
	const JhHuffLut *L = reinterpret_cast<const JhHuffLut *>(tab);
	e &= 0xffffu;                                          // (lut_lookup: the high half belongs to pair[])
	if (e == JH_LUT_INVALID) {
		return 0;
	}
	if (e != JH_LUT_SEARCH) {
		return L->sub[e & 0x7fffu][(win >> (26 - JH_LUT_BITS)) & 63u];
	}
	// canonical search on the left-aligned 16-bit window (more prefixes than second-level tables)
	const unsigned v = win >> 16;
	int len = JH_LUT_BITS + 1;
#pragma unroll
	for (int l = JH_LUT_BITS + 1; l < 16; ++l) {
		len += v >= L->limit[l] ? 1 : 0;
	}
	if (v >= L->limit[16]) {
		return 0;
	}
	const unsigned sym = L->huffval[(L->offs[len] + (int)(v >> (16 - len))) & 0xff];
	const unsigned cat = sym & 15u;
	const unsigned run = (!isdc && sym == 0) ? 63u : (sym >> 4);   // EOB (coeffs.c:241-243)
	return (((unsigned)len + cat) << 10) | (run << 4) | cat;
}

// Decode every symbol that STARTS before bit `stop`, beginning at st.pos (sums of DC differences and blocks only:
// the synchronisation kernels; the write pass has its own loop with the same symbol core, huffdec.hip).
//
// One loop, one symbol per trip and lane.  FAST: symbols that start at least 31 bits before the end of the restart
// segment cannot run past it, so they need no end-of-segment test; the code is software-pipelined by hand -- the
// table look-up of the NEXT symbol is issued as soon as the current symbol's length is known, the value
// extraction and the bookkeeping of the current symbol run in its shadow -- and branch-free apart from the rare
// long code.  Three stream words are kept in registers (w0:w1 hold the current bit, w2 the word after them), and
// the tables of the next block of the MCU are fetched while the current block is being decoded.  CAREFUL: one
// symbol with every test of the reference's read_block(), used near segment ends and for undecodable data, behind
// a test the whole wave takes together.  (The two used to be an inner loop and its tail: a lane that met a
// segment end early then waited for the others' inner loops and decoded the rest of its sub-sequence alone
// afterwards -- with a restart interval per MCU row the kernels took twice as long.)
template <class Stream>
__device__ __forceinline__ void decode_span_segmented(DecState &st, const uint32_t stop, const Stream &S,
                                            const uint32_t total_bits, const uint32_t *seg_start,
                                            const uint32_t nseg, const unsigned char *luts,
                                            const DecTables &tb, const int B, int *dcs, const int dcs_stride,
                                            int16_t *coefs, int blk, const int nblk)
{
	if (st.pos >= stop || st.pos >= total_bits) {
		return;
	}
#pragma unroll
	for (int k = 0; k < JH_MAX_COMP; ++k) {
		DCS_AT(dcs, k, dcs_stride) = st.dc[k];
	}
	uint32_t seg_end = min(seg_start[st.seg + 1] * 8u, total_bits);
	uint32_t pos = st.pos;
	int c = st.c, z = st.z, nb = st.nb;
	// a second symbol resolved by the same look-up starts less than JH_LUT_BITS bits behind the first: below this
	// position it starts before `stop`, i.e. belongs to this span too
	const uint32_t stop_pair = stop > (uint32_t)JH_LUT_BITS ? stop - (uint32_t)JH_LUT_BITS + 1u : 0u;
	bool primed = false;
	int cn = 0;
	uint4 ent_n = make_uint4(0, 0, 0, 0);
	uint32_t wi = 0, w0 = 0, w1 = 0, w2 = 0, win = 0, tac = 0, tab = 0, dca = 0;
	typename Stream::Cursor cur;
	cur.b = cur.c = cur.d = 0;
	cur.nq = 4;
	cur.nxt = 0;
	cur.ld.x = cur.ld.y = cur.ld.z = cur.ld.w = 0;
	unsigned e = 0;

	const bool segmented = nseg > 1;                       // restart segments: a lane may meet a segment end anywhere
	for (;;) {
		const bool go = pos < stop && pos < total_bits;
		if (__builtin_amdgcn_ballot_w64(go) == 0) {
			break;
		}
		// symbols that start before `fs` take the fast path; an undecodable code pulls `fs` back to its position
		uint32_t fs = go ? min(stop, seg_end > 31u ? seg_end - 31u : 0u) : 0u;
		if (__builtin_amdgcn_ballot_w64(pos < fs && !primed) != 0) {
			if (pos < fs && !primed) {
				const uint4 ent = tb.ent[c];
				cn = (int)ent.w;
				ent_n = tb.ent[cn];
				tac = ent.y;
				dca = ent.z;
				wi = pos >> 5;
				S.open(cur, wi, w0, w1, w2);
				win = bit_window(w0, w1, pos);
				tab = z == 0 ? ent.x : tac;
				e = lut_lookup(luts + tab, win);
				primed = true;
			}
		}
		// one fast symbol of this lane (pos < fs)
		auto symbol = [&]() {
			const bool isdc = z == 0;
			bool ok = true;
			if (e & 0x8000u) {
				e = long_code(luts + tab, e, win, isdc);
				ok = e != 0;
				fs = ok ? fs : pos;            // no code matches: the careful path ends the segment
			}
			if (ok) {
				const int total1 = (int)((e >> 10) & 31u), run = (int)((e >> 4) & 63u), cat = (int)(e & 15u);
				// the symbol behind it as well, if the table has resolved it (JhHuffLut), the first one leaves the
				// block open, and the second one still starts inside this sub-sequence
				const unsigned pr = e >> 16;
				const bool two = pr != 0 && !isdc && z + run < 63 && pos < stop_pair;
				const int total = two ? (int)JH_PAIR_TOTAL(pr) : total1;
				const uint32_t npos = pos + (uint32_t)total;
				const int k = z + (two ? (int)JH_PAIR_ADV(pr) : run);   // zero run(s) (coeffs.c:246)
				const bool done = k >= 63;
				// the next symbol: its table, its window, its look-up
				const uint32_t tabn = done ? ent_n.x : tac;
				const uint32_t nwi = npos >> 5;
				const bool cross = nwi != wi;                       // a symbol has at most 31 bits
				w0 = cross ? w1 : w0;
				w1 = cross ? w2 : w1;
				wi = nwi;
				const uint32_t nwin = bit_window(w0, w1, npos);
				const unsigned en = lut_lookup(luts + tabn, nwin);
				w2 = S.next(cur, cross, w2);
				// the current symbol's value: only the DC differences are summed here (a DC symbol is never one of two)
				const unsigned extra = __builtin_amdgcn_ubfe(win, 32u - (unsigned)total1, (unsigned)cat);
				const int val = extend_sign(extra, cat);
				atomicAdd(reinterpret_cast<int *>(reinterpret_cast<char *>(dcs) + dca), isdc ? val : 0);
				// next zig-zag index / next block
				z = done ? 0 : k + 1;
				nb += done ? 1 : 0;
				c = done ? cn : c;
				tac = done ? ent_n.y : tac;
				dca = done ? ent_n.z : dca;
				cn = done ? (int)ent_n.w : cn;
				ent_n = tb.ent[cn];
				pos = npos;
				win = nwin;
				tab = tabn;
				e = en;
			}
		};
		if (!segmented) {
			// one segment: the careful path is for the last bits of the stream only; every lane runs to its `fs`
			while (pos < fs) {
				symbol();
			}
		} else {
			// Restart segments: the loop ends as soon as a lane waits for the careful path -- a lane that met a
			// segment end would otherwise sit out the others' sub-sequences and decode the rest of its own alone
			// afterwards.
			for (;;) {
				const bool fast = pos < fs;
				if (__builtin_amdgcn_ballot_w64(fast) == 0 ||
				    __builtin_amdgcn_ballot_w64(!fast && pos < stop && pos < total_bits) != 0) {
					break;
				}
				if (fast) {
					symbol();
				}
			}
		}
		// ---- careful: one symbol for the lanes that wait for it
		if (pos >= fs && pos < stop && pos < total_bits) {
			primed = false;
			const uint32_t info = tb.info[c];
			const bool isdc = z == 0;
			const unsigned char *tb1 = luts + (((info >> (isdc ? 0 : 12)) & 0xfffu) << 4);
			const uint32_t wj = pos >> 5;
			const uint32_t wn = bit_window(S.word(wj), S.word(wj + 1), pos);
			unsigned e1 = reinterpret_cast<const uint16_t *>(tb1)[wn >> (32 - JH_LUT_BITS)];
			if (e1 & 0x8000u) {
				e1 = long_code(tb1, e1, wn, isdc);
			}
			const int total = (int)(e1 >> 10), run = (int)((e1 >> 4) & 63u), cat = (int)(e1 & 15u);
			const uint32_t npos = pos + (uint32_t)total;
			if (e1 == 0 || npos > seg_end) {
				// no code matches (the reference reads on to the end of the segment), or the data of this
				// restart segment is used up (padding bits, truncated data): decoder.c:376-381.
				// The MCU in progress is dropped (context->mblocks is not incremented), the next segment
				// starts with block 0 / DC predictors 0 (decoder.c:369-373).
				nb -= c;
				c = 0;
				z = 0;
				st.reset = 1;
#pragma unroll
				for (int k = 0; k < JH_MAX_COMP; ++k) {
					DCS_AT(dcs, k, dcs_stride) = 0;
				}
				st.seg++;
				if (st.seg >= nseg) {
					pos = total_bits;                  // the lane is done
				} else {
					pos = seg_end;
					seg_end = min(seg_start[st.seg + 1] * 8u, total_bits);
				}
			} else {
				const unsigned extra = __builtin_amdgcn_ubfe(wn, 32u - (unsigned)total, (unsigned)cat);
				const int val = extend_sign(extra, cat);
				if (isdc) {
					atomicAdd(dcs + (info >> 24) * dcs_stride, val);
				}
				const int k = z + run;
				pos = npos;
				z = k + 1;
				if (k >= 63) {
					z = 0;
					c = (c + 1 == B) ? 0 : c + 1;
					nb++;
				}
			}
		}
	}
	st.pos = pos;
	st.c = c;
	st.z = z;
	st.nb = nb;
#pragma unroll
	for (int k = 0; k < JH_MAX_COMP; ++k) {
		st.dc[k] = DCS_AT(dcs, k, dcs_stride);
	}
}

// The same with the fast symbols as an inner loop of each lane and the careful path as its tail: the form for
// streams of ONE restart segment (what is alive in the inner loop is dead in the tail, which leaves the
// synchronisation kernels at 43 instead of 74 vector registers, 8 instead of 6 waves per SIMD).  Right for any
// stream; slow on segmented ones for the reason given above.
template <class Stream>
__device__ __forceinline__ void decode_span_plain(DecState &st, const uint32_t stop, const Stream &S,
                                            const uint32_t total_bits, const uint32_t *seg_start,
                                            const uint32_t nseg, const unsigned char *luts,
                                            const DecTables &tb, const int B, int *dcs, const int dcs_stride,
                                            int16_t *coefs, int blk, const int nblk)
{
	constexpr bool WRITE = false;                          // the write pass is k_huff_write
	if (st.pos >= stop || st.pos >= total_bits) {
		return;
	}
#pragma unroll
	for (int k = 0; k < JH_MAX_COMP; ++k) {
		DCS_AT(dcs, k, dcs_stride) = st.dc[k];
	}
	uint32_t seg_end = min(seg_start[st.seg + 1] * 8u, total_bits);
	uint32_t pos = st.pos;
	int c = st.c, z = st.z, nb = st.nb;
	// a second symbol resolved by the same look-up starts less than JH_LUT_BITS bits behind the first: below this
	// position it starts before `stop`, i.e. belongs to this span too
	const uint32_t stop_pair = stop > (uint32_t)JH_LUT_BITS ? stop - (uint32_t)JH_LUT_BITS + 1u : 0u;

	for (;;) {
		// ------------------------------------------------------------------------------ fast
		const uint32_t fast_stop = min(stop, seg_end > 31u ? seg_end - 31u : 0u);
		if (pos < fast_stop) {
			const uint4 ent = tb.ent[c];
			int cn = (int)ent.w;
			uint4 ent_n = tb.ent[cn];
			uint32_t tac = ent.y, dca = ent.z;
			uint32_t wi = pos >> 5;
			typename Stream::Cursor cur;
			uint32_t w0, w1, w2;
			S.open(cur, wi, w0, w1, w2);
			uint32_t win = bit_window(w0, w1, pos);
			uint32_t tab = z == 0 ? ent.x : tac;
			unsigned e = lut_lookup(luts + tab, win);
			do {
				const bool isdc = z == 0;
				if (e & 0x8000u) {
					e = long_code(luts + tab, e, win, isdc);
					if (e == 0) {
						break;                 // no code matches: the careful path ends the segment
					}
				}
				const int total1 = (int)((e >> 10) & 31u), run = (int)((e >> 4) & 63u), cat = (int)(e & 15u);
				// the symbol behind it as well, if the table has resolved it (JhHuffLut), the first one leaves the
				// block open, and the second one still starts inside this sub-sequence
				const unsigned pr = e >> 16;
				const bool two = pr != 0 && !isdc && z + run < 63 && pos < stop_pair;
				const int total = two ? (int)JH_PAIR_TOTAL(pr) : total1;
				const uint32_t npos = pos + (uint32_t)total;
				const int k = z + (two ? (int)JH_PAIR_ADV(pr) : run);   // zero run(s) (coeffs.c:246)
				const bool done = k >= 63;
				// the next symbol: its table, its window, its look-up
				const uint32_t tabn = done ? ent_n.x : tac;
				const uint32_t nwi = npos >> 5;
				const bool cross = nwi != wi;                       // one or two symbols: at most 31 bits
				w0 = cross ? w1 : w0;
				w1 = cross ? w2 : w1;
				wi = nwi;
				const uint32_t nwin = bit_window(w0, w1, npos);
				const unsigned en = lut_lookup(luts + tabn, nwin);
				w2 = S.next(cur, cross, w2);
				// the current symbol's value (a DC symbol is never one of two)
				const unsigned extra = __builtin_amdgcn_ubfe(win, 32u - (unsigned)total1, (unsigned)cat);
				int val = extend_sign(extra, cat);
				int *acc = reinterpret_cast<int *>(reinterpret_cast<char *>(dcs) + dca);
				if (WRITE) {
					const int old = atomicAdd(acc, isdc ? val : 0); // DC un-prediction (decoder.c:351-355)
					val += isdc ? old : 0;
					if ((isdc || cat != 0) && k < 64 && blk >= 0 && blk < nblk) {
						coefs[(size_t)blk * 64 + k] = (int16_t)val;
					}
					blk += done ? 1 : 0;
				} else {
					atomicAdd(acc, isdc ? val : 0);
				}
				// next zig-zag index / next block
				z = done ? 0 : k + 1;
				nb += done ? 1 : 0;
				c = done ? cn : c;
				tac = done ? ent_n.y : tac;
				dca = done ? ent_n.z : dca;
				cn = done ? (int)ent_n.w : cn;
				ent_n = tb.ent[cn];
				pos = npos;
				win = nwin;
				tab = tabn;
				e = en;
			} while (pos < fast_stop);
		}
		if (pos >= stop) {
			break;
		}
		// --------------------------------------------------------------------------- careful
		{
			const uint32_t info = tb.info[c];
			const bool isdc = z == 0;
			const unsigned char *tab = luts + (((info >> (isdc ? 0 : 12)) & 0xfffu) << 4);
			const uint32_t wi = pos >> 5;
			const uint32_t win = bit_window(S.word(wi), S.word(wi + 1), pos);
			unsigned e = reinterpret_cast<const uint16_t *>(tab)[win >> (32 - JH_LUT_BITS)];
			if (e & 0x8000u) {
				e = long_code(tab, e, win, isdc);
			}
			const int total = (int)(e >> 10), run = (int)((e >> 4) & 63u), cat = (int)(e & 15u);
			const uint32_t npos = pos + (uint32_t)total;
			if (e == 0 || npos > seg_end) {
				// no code matches (the reference reads on to the end of the segment), or the data of this
				// restart segment is used up (padding bits, truncated data): decoder.c:376-381.
				// The MCU in progress is dropped (context->mblocks is not incremented), the next segment
				// starts with block 0 / DC predictors 0 (decoder.c:369-373).
				nb -= c;
				if (WRITE) {
					blk -= c;
				}
				c = 0;
				z = 0;
				st.reset = 1;
#pragma unroll
				for (int k = 0; k < JH_MAX_COMP; ++k) {
					DCS_AT(dcs, k, dcs_stride) = 0;
				}
				st.seg++;
				if (st.seg >= nseg) {
					pos = total_bits;
					break;
				}
				pos = seg_end;
				seg_end = min(seg_start[st.seg + 1] * 8u, total_bits);
				if (pos >= stop) {
					break;
				}
				continue;
			}
			const unsigned extra = __builtin_amdgcn_ubfe(win, 32u - (unsigned)total, (unsigned)cat);
			int val = extend_sign(extra, cat);
			int *dca = dcs + (info >> 24) * dcs_stride;
			if (isdc) {
				val += atomicAdd(dca, val);
			}
			const int k = z + run;
			if (WRITE && (isdc || cat != 0) && k < 64 && blk >= 0 && blk < nblk) {
				coefs[(size_t)blk * 64 + k] = (int16_t)val;
			}
			pos = npos;
			z = k + 1;
			if (k >= 63) {
				z = 0;
				c = (c + 1 == B) ? 0 : c + 1;
				nb++;
				if (WRITE) {
					blk++;
				}
			}
			if (pos >= stop) {
				break;
			}
		}
	}
	st.pos = pos;
	st.c = c;
	st.z = z;
	st.nb = nb;
#pragma unroll
	for (int k = 0; k < JH_MAX_COMP; ++k) {
		st.dc[k] = DCS_AT(dcs, k, dcs_stride);
	}
}


// SEGMENTED: which form (a hint from the headers -- DRI -- not a promise: both are right for every stream)
template <bool SEGMENTED, class Stream>
__device__ __forceinline__ void decode_span(DecState &st, const uint32_t stop, const Stream &S,
                                            const uint32_t total_bits, const uint32_t *seg_start,
                                            const uint32_t nseg, const unsigned char *luts,
                                            const DecTables &tb, const int B, int *dcs, const int dcs_stride,
                                            int16_t *coefs, int blk, const int nblk)
{
	if (SEGMENTED) {
		decode_span_segmented(st, stop, S, total_bits, seg_start, nseg, luts, tb, B, dcs, dcs_stride, coefs, blk, nblk);
	} else {
		decode_span_plain(st, stop, S, total_bits, seg_start, nseg, luts, tb, B, dcs, dcs_stride, coefs, blk, nblk);
	}
}

// segment that contains byte offset `byte` : largest k with seg_start[k] <= byte
__device__ __forceinline__ uint32_t find_segment(const uint32_t *seg_start, uint32_t nseg, uint32_t byte)
{
	uint32_t lo = 0, hi = nseg;   // invariant: seg_start[lo] <= byte, answer in [lo, hi)
	while (hi - lo > 1) {
		const uint32_t mid = (lo + hi) >> 1;
		if (seg_start[mid] <= byte) {
			lo = mid;
		} else {
			hi = mid;
		}
	}
	return lo;
}

// bytes the tables of a frame take in LDS: AC tables whole, DC tables in their short form (host: api.hip sizes the
// kernels' LDS with the same rule)
__device__ __forceinline__ void load_tables(const JhDecodeWork &w, const JhDecFrame &fr, unsigned char *luts, DecTables *tb,
                                            const JhGeom &g, int zunit)
{
	// read through the global pointer: indexing the by-value copy would put it in scratch
	const JhDecFrame *gf = w.frames + blockIdx.y;
	if (threadIdx.x == 0) {
		uint32_t off = 0;
		for (uint32_t s = 0; s < fr.n_lut && s < 8u; ++s) {
			bool isdc = false;
			for (int b = 0; b < g.B; ++b) {
				isdc = isdc || gf->blk_dc[b] == s;
			}
			tb->slot_off[s] = off;
			off += isdc ? JH_LUT_DC_BYTES : (uint32_t)sizeof(JhHuffLut);
		}
		tb->slot_off[fr.n_lut < 8u ? fr.n_lut : 8u] = off;
	}
	__syncthreads();
	for (uint32_t s = 0; s < fr.n_lut && s < 8u; ++s) {
		const uint32_t *src = reinterpret_cast<const uint32_t *>(w.luts + fr.lut_off + s);
		uint32_t *dst = reinterpret_cast<uint32_t *>(luts + tb->slot_off[s]);
		const int nw = (int)((tb->slot_off[s + 1] - tb->slot_off[s]) / 4);
		for (int i = threadIdx.x; i < nw; i += blockDim.x) {
			dst[i] = src[i];
		}
	}
	if (threadIdx.x < JH_MAX_MCU_BLOCKS) {
		const uint32_t odc = tb->slot_off[gf->blk_dc[threadIdx.x] & 7], oac = tb->slot_off[gf->blk_ac[threadIdx.x] & 7];
		tb->info[threadIdx.x] = (odc >> 4) | (oac >> 4) << 12 | (uint32_t)g.blk_comp[threadIdx.x] << 24;
		tb->ent[threadIdx.x] = make_uint4(odc, oac, (uint32_t)g.blk_comp[threadIdx.x] * (uint32_t)zunit,
		                                  (int)threadIdx.x + 1 >= g.B ? 0u : threadIdx.x + 1u);
	}
	__syncthreads();
}

__device__ __forceinline__ void store_info(JhSubInfo *dst, const DecState &s)
{
	JhSubInfo r;
	r.state = pack_state(s);
	r.seg = s.seg;
	r.nb = s.nb;
#pragma unroll
	for (int k = 0; k < JH_MAX_COMP; ++k) {
		r.dc[k] = s.dc[k];
	}
	*dst = r;
}

#define STATE_SYNC_MASK 0x0000ffffffffffffull      // pos, c, z (the reset flag is not part of the parse state)

struct HuffArgs {
	JhDecodeWork w;
	JhGeom g;
	int round;                         // synchronisation round (fix-up kernel)
};
