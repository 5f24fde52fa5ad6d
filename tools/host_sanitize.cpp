// tools/host_sanitize.cpp -- the host side (marker parser, geometry, table builders, stream splitter) on damaged
// headers under AddressSanitizer / UBSan.  Host code only, no GPU:
//   g++ -std=c++17 -O1 -g -fsanitize=address,undefined -fno-sanitize-recover=undefined -Iinclude -Ijpeg_amd/csrc
//       tools/host_sanitize.cpp -x c++ jpeg_amd/csrc/codestream.cpp -o /tmp/host_sanitize
//   /tmp/host_sanitize tests/golden/*.jpg
#include <stdio.h>
#include <stdlib.h>
#include <string.h>
#include <vector>
#include "jpeg_amd.h"
#include "codestream.hpp"
static uint64_t s = 88172645463325252ull;
static uint32_t rnd() { s ^= s << 13; s ^= s >> 7; s ^= s << 17; return (uint32_t)(s >> 11); }
int main(int argc, char **argv)
{
	long total = 0, ok = 0;
	for (int a = 1; a < argc; ++a) {
		FILE *f = fopen(argv[a], "rb");
		if (!f) continue;
		std::vector<uint8_t> d(1 << 20);
		d.resize(fread(d.data(), 1, d.size(), f));
		fclose(f);
		size_t sos = 0;
		for (size_t i = 0; i + 1 < d.size(); ++i) if (d[i] == 0xff && d[i + 1] == 0xda) { sos = i; break; }
		const int iters = getenv("HOST_SAN_ITERS") ? atoi(getenv("HOST_SAN_ITERS")) : 4000;
		for (int it = 0; it < iters; ++it) {
			std::vector<uint8_t> b(d);
			const int k = 1 + rnd() % 3;
			for (int j = 0; j < k; ++j) {
				size_t p = 2 + rnd() % (sos + 12);
				b[p] = (rnd() & 1) ? (uint8_t)rnd() : (uint8_t)(b[p] ^ (1u << (rnd() % 8)));
			}
			if (rnd() % 2 == 0 && b.size() > sos + 40) {
				for (int j = 0; j < 3; ++j) {
					size_t p = sos + 20 + rnd() % (b.size() - sos - 24);
					static const uint8_t mk[] = {0x01, 0xd8, 0xe1, 0xfe, 0xdd, 0xd3, 0xc4, 0xdb, 0xd9, 0xcc};
					b[p] = 0xff;
					b[p + 1] = mk[rnd() % sizeof mk];
				}
			}
			if (rnd() % 8 == 0) b.resize(2 + rnd() % b.size());
			// exact-size heap copy so that any over-read is caught
			uint8_t *h = (uint8_t *)malloc(b.size());
			memcpy(h, b.data(), b.size());
			JhHeader hd;
			int err = jh_parse_headers(h, b.size(), &hd, (rnd() & 1) != 0);
			total++;
			if (!err) {
				for (int j = 0; j < hd.g.ncomp; ++j) hd.g.scan_order[j] = hd.Ns == hd.g.ncomp ? hd.scan_comp[j] : j;
				err = jh_finish_geometry(&hd.g);
			}
			if (!err) {
				ok++;
				for (int tc = 0; tc < 2; ++tc) for (int th = 0; th < 4; ++th) {
					JhHuffLut l; JhHuffEnc e;
					jh_build_declut(&hd.ht[tc][th], tc, &l);
					jh_build_enccodes(&hd.ht[tc][th], &e);
				}
				if (hd.Ns) {
					size_t end = jh_ecs_end(h, b.size(), hd.ecs_offset);
					bool eoi = false;
					if (end < b.size()) jh_parse_next_scan(h, b.size(), end, &hd, &eoi);
					// the walk sync() makes behind a stray marker, from anywhere in the data
					for (int j = 0; j < 8; ++j) {
						size_t rst = 0;
						const size_t pos = hd.ecs_offset + rnd() % (b.size() - hd.ecs_offset + 1);
						const int e = jh_walk_to_restart(h, b.size(), pos, (rnd() & 1) != 0, &rst);
						if (e == 0 && rst && (rst < pos || rst + 1 >= b.size() || h[rst] != 0xff)) abort();
					}
				}
			}
			size_t off[8], len[8];
			jh_split_stream(h, b.size(), off, len, 8);
			free(h);
		}
	}
	printf("sanitizer run: %ld inputs, %ld accepted\n", total, ok);
	return 0;
}
