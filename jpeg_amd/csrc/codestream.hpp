// jpeg_amd/csrc/codestream.hpp -- host-side codestream layer (markers, tables, geometry).
// Serial, byte-sized work that stays on the CPU; it feeds the
// device kernels with geometry and <= 10 KB of tables per frame.
#pragma once

#include <vector>
#include <stdint.h>
#include <stddef.h>
#include "common.hpp"

struct JhHuffSpec {                   // DHT payload (common.h:85-91), flattened
	uint8_t L[16];
	uint8_t V[256];
};

struct JhHeader {
	JhGeom g;
	uint16_t Q[4][64];                // raster order, as the reference stores it (common.h:52-57)
	JhHuffSpec ht[2][4];              // [Tc][Th]; starts as the implicit MJPEG tables (common.c:90-99)
	int Ns;
	int scan_comp[JH_MAX_COMP];       // index into g.* of each scan component
	size_t ecs_offset;                // first byte after the SOS header
	bool have_sof;
	bool tolerant;                    // set by the caller before parsing: skip APP9/APP10/APP15/DNL/JPGn instead of failing
};

// Parse SOI .. SOS of one codestream held in memory (decoder.c:472-659 restated).  Returns a RET_* code.
int jh_parse_headers(const uint8_t *p, size_t n, JhHeader *h, bool tolerant = false);

// Files with several scans (non-interleaved components, decoder.c:593-633): continue behind the entropy-coded
// data of a scan.  Tables may be redefined on the way; *eoi = true when EOI came instead of another SOS.
int jh_parse_next_scan(const uint8_t *p, size_t n, size_t pos, JhHeader *h, bool *eoi);
// parse_format() (decoder.c:487-657) from a marker inside a scan's data that is neither RSTn nor EOI, as far as
// it only SKIPS things: TEM, APPn, COM, DAC and DRI segments by their length, bytes between markers
// (read_marker(), io.c:186-220).  *rst = position of the first RSTn it arrives at (the reference decodes on
// from there, decoder.c:622-633) or 0; RET_SUCCESS with *rst == 0: it arrived at EOI.  Everything else --
// table or frame segments, another SOS, markers the reference does not know -- is returned as an error.
int jh_walk_to_restart(const uint8_t *p, size_t n, size_t pos, bool tolerant, size_t *rst);
size_t jh_ecs_end(const uint8_t *p, size_t n, size_t pos);            // position of the marker that ends the ECS, or n

// Geometry from P/X/Y/components (common.c:156-195, frame.c:16-35).  interleaved = scan holds all
// components; a single-component frame is normalised to H = V = 1 over its padded block grid
// (the reference's Ns == 1 walk, decoder.c:274-302).
int jh_finish_geometry(JhGeom *g);

void jh_default_table(int Tc, int Th, JhHuffSpec *s);                 // mjpeg.h:5-82
void jh_set_qtable(uint16_t Q[64], int chroma, int q);                // encoder.c:14-65
int  jh_split_stream(const uint8_t *p, size_t n, size_t *off, size_t *len, int cap);   // Motion JPEG frame splitter
int  jh_build_declut(const JhHuffSpec *s, int tc, JhHuffLut *lut);            // huffman.c:45-148 -> decode tables
int  jh_build_enccodes(const JhHuffSpec *s, JhHuffEnc *e);            // huffman.c:228-253 first-match codes

// SOI, DQT.., SOF0, DHT.., [DRI], SOS  (encoder.c:195-400, order of produce_codestream :589-645)
void jh_produce_headers(const JhGeom &g, const uint16_t Q[4][64], const JhHuffSpec ht[2][4],
                        int restart_interval, std::vector<uint8_t> *out);
