"""The pixel kernels replace (int)roundf(x) (quantize_block imgproc.c:28, write_frame_body frame.c:375) by
(int)(x + copysignf(pred(0.5), x)) -- one add and one truncating convert (jpeg_amd/csrc/pixels.hip,
round_half_away).  That identity is checked here for EVERY binary32 value inside the int range, on the host,
with the C library's roundf() as the judge (no GPU, no oracle needed: it is a statement about IEEE arithmetic)."""
import os
import shutil
import subprocess

import pytest

SRC = r"""
#include <math.h>
#include <stdio.h>
#include <stdint.h>
#include <string.h>
int main(void)
{
	const float h = 0x1.fffffep-2f;               /* the largest float below 0.5 */
	unsigned long long bad = 0, n = 0;
	for (unsigned long long u = 0; u <= 0xffffffffull; ++u) {
		uint32_t b = (uint32_t)u;
		float x;
		memcpy(&x, &b, 4);
		if (!(fabsf(x) < 2147483000.0f)) {
			continue;                             /* outside int, infinities, NaNs */
		}
		volatile float s = x + copysignf(h, x);   /* one rounding, as v_add_f32 */
		if ((int)roundf(x) != (int)s) {
			bad++;
		}
		n++;
	}
	printf("%llu %llu\n", n, bad);
	return bad != 0;
}
"""


def test_add_and_truncate_is_round_half_away_for_every_float(tmp_path):
    if shutil.which("gcc") is None:
        pytest.skip("no gcc here")
    src = tmp_path / "rt.c"
    src.write_text(SRC)
    exe = str(tmp_path / "rt")
    subprocess.run(["gcc", "-O2", "-ffp-contract=off", "-o", exe, str(src), "-lm"], check=True, timeout=120)
    r = subprocess.run([exe], capture_output=True, text=True, timeout=300)
    n, bad = (int(v) for v in r.stdout.split())
    assert r.returncode == 0 and bad == 0 and n > 2_600_000_000
