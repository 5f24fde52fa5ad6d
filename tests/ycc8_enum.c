/* tests/ycc8_enum.c -- TEST INFRASTRUCTURE.  The product's binary32 formula for frame_to_ycc at precision 8
 * (jpeg_amd/csrc/ycc8.inc, the very text pixels.hip compiles, here with T = float and libm's exact fmaf)
 * on n interleaved (R, G, B) float triples, in place.  Built by tests/test_ycc8_cpu.py with
 * gcc -std=c99 -O2 -ffp-contract=off. */
#include <math.h>
#include <stddef.h>
#include "../jpeg_amd/csrc/ycc8.inc"

#define SPLAT1(x) (x)
YCC8_DEFINE(static inline, ycc8_scalar, float, SPLAT1, fmaf)

void ycc8_run(float *px, size_t n)
{
	for (size_t i = 0; i < n; ++i, px += 3) {
		ycc8_scalar(px[0], px[1], px[2], &px[0], &px[1], &px[2]);
	}
}
