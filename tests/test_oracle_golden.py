"""The CPU oracle (oracle/jpeg_oracle.c) against the committed golden vectors.

Every golden in tests/golden/ is an output of the compiled reference (make_golden.py); the
reference itself ships no vectors.  These tests run anywhere (no GPU,
no /root/reference) and pin the oracle: JPEG bytes, quantised coefficients and decoded samples
must be bit-identical to what the reference produced.
"""
import hashlib

import numpy as np
import pytest

from jpeg_amd.synth import synth_image
from oracle.pyoracle import pnm_body
from tests.conftest import golden_path


def md5(b):
    return hashlib.md5(b).hexdigest()


def coef_md5(img):
    h = hashlib.md5()
    for c in img.comps:
        h.update(np.ascontiguousarray(c["coef"], np.int32).tobytes())
    return h.hexdigest()


def _enc_cases():
    import json
    import os
    with open(os.path.join(os.path.dirname(__file__), "golden", "manifest.json")) as f:
        return sorted(json.load(f)["encode"].items())


def _dec_cases():
    import json
    import os
    with open(os.path.join(os.path.dirname(__file__), "golden", "manifest.json")) as f:
        return sorted(json.load(f)["decode"].items())


@pytest.mark.parametrize("name,g", _enc_cases())
def test_oracle_encode_matches_reference_golden(oracle, name, g):
    img = synth_image(g["X"], g["Y"], g["ncomp"], g["P"], g["seed"], g["range_safe"])
    assert md5(pnm_body(img, g["P"])) == g["input_md5"], "synthetic generator drifted"
    err, jpg, info = oracle.encode(img, P=g["P"], H=g["H"], V=g["V"], q=g["q"], optimize=g["optimize"])
    assert err == 0
    assert coef_md5(info) == g["coef_md5"]
    assert len(jpg) == g["jpg_len"]
    assert md5(jpg) == g["jpg_md5"]
    if "jpg_file" in g:
        with open(golden_path(g["jpg_file"]), "rb") as f:
            assert f.read() == jpg


@pytest.mark.parametrize("name,g", _enc_cases())
def test_oracle_decode_of_reference_jpeg(oracle, name, g):
    if "jpg_file" in g:
        with open(golden_path(g["jpg_file"]), "rb") as f:
            jpg = f.read()
    else:
        img = synth_image(g["X"], g["Y"], g["ncomp"], g["P"], g["seed"], g["range_safe"])
        jpg = oracle.encode(img, P=g["P"], H=g["H"], V=g["V"], q=g["q"], optimize=g["optimize"])[1]
    assert md5(jpg) == g["jpg_md5"]
    err, dec, info = oracle.decode(jpg)
    assert err == 0
    assert (info.X, info.Y, info.P) == (g["X"], g["Y"], g["P"])
    assert coef_md5(info) == g["coef_md5"]       # entropy decode inverts entropy encode exactly
    assert md5(pnm_body(dec, g["P"])) == g["dec_md5"]


@pytest.mark.parametrize("name,g", _dec_cases())
def test_oracle_decode_external_streams(oracle, name, g):
    with open(golden_path(g["jpg_file"]), "rb") as f:
        jpg = f.read()
    assert md5(jpg) == g["jpg_md5"]
    err, dec, info = oracle.decode(jpg)
    assert err == g["err"]                       # e.g. progressive -> RET_FAILURE_FILE_UNSUPPORTED
    if err == 0:
        assert (info.X, info.Y, info.Nf, info.P) == (g["X"], g["Y"], g["Nf"], g["P"])
        assert coef_md5(info) == g["coef_md5"]
        assert md5(pnm_body(dec, g["P"])) == g["dec_md5"]


def test_kat_lut_and_dct(oracle, manifest):
    kat = manifest["kat"]
    assert ["%08x" % v for v in oracle.lut().view(np.uint32).ravel()] == kat["lut_hex"]
    blk = np.array([((i * 37 + 11) % 256) - 128 for i in range(64)], np.float32)
    f = oracle.fdct(blk)
    assert ["%08x" % v for v in f.view(np.uint32)] == kat["fdct_hex"]
    assert ["%08x" % v for v in oracle.idct(f).view(np.uint32)] == kat["idct_hex"]
    # SURVEY.md section 8a KAT (first rows)
    assert kat["fdct_hex"][:8] == ["407fffe0", "c11184c6", "c26c8362", "419a337c", "361db9d0", "c1b0ac85", "41c3ef15", "c1883e70"]
    assert kat["idct_hex"][:8] == ["c2ea0000", "c2a00000", "c22c0015", "c0bfffca", "41f7fffa", "42880002", "42d1fffb", "c2e3fffd"]


def test_coefficient_coding_exhaustive(oracle):
    """encode_cat / encode_extra / decode_coeff round trip over the whole 16-bit range (Q5)."""
    L = oracle.L
    for c in list(range(-32767, 32768, 1)):
        cat = L.orc_encode_cat(c)
        extra = L.orc_encode_extra(c, cat)
        assert 0 <= cat <= 15 and 0 <= extra < (1 << cat) or cat == 0
        assert L.orc_decode_coeff(cat, extra) == c


def test_quantize_ties_round_half_away(oracle):
    Q = np.full(64, 2, np.uint16)
    f = np.zeros(64, np.float32)
    f[:8] = [1, -1, 3, -3, 5, -5, 0.999, -0.999]
    q = oracle.quantize(f, Q)
    assert q[:8].tolist() == [1, -1, 2, -2, 3, -3, 0, 0]


def test_quality_scaling(oracle):
    # encoder.c:38-65: q=50 -> table itself, q=100 -> all ones, q=1 -> clamp 255
    assert oracle.qtable(0, 50)[:4].tolist() == [16, 11, 10, 16]
    assert set(oracle.qtable(0, 100).tolist()) == {1}
    assert oracle.qtable(1, 1).max() == 255
    assert oracle.qtable(0, 90)[:4].tolist() == [3, 2, 2, 3]


def test_k2_optimiser_properties(oracle):
    rng = np.random.default_rng(7)
    for trial in range(20):
        freq = np.zeros(257, np.int64)
        n = int(rng.integers(1, 200))
        idx = rng.choice(256, n, replace=False)
        freq[idx] = rng.integers(1, 10 ** int(rng.integers(1, 7)), n)
        freq[256] = 1
        L, V = oracle.adapt_table(freq)
        assert int(L.sum()) == n                       # one code per used symbol, reserved one dropped
        assert sorted(V[:n].tolist()) == sorted(idx.tolist())
        # Kraft inequality with the reserved all-ones code point
        assert sum(int(L[i]) * 2.0 ** -(i + 1) for i in range(16)) < 1.0
        si, co = oracle.make_codes(L, V)
        assert all(si[s] > 0 for s in idx)


def test_a_scan_that_names_a_table_no_dht_defined(oracle):
    """tests/golden/damaged/undefined_ac_table_in_second_scan.jpg (tools/fuzz_corrupt.py, seed 202, case 18356): a bit
    flip in the second SOS header makes Cb use AC table 3, which no DHT segment defines.  The reference reads
    uninitialised memory there (init_context() converts the four implicit tables only, common.c:90-99, and the context
    is malloc()ed, decoder.c:665) -- its output changes from run to run; on zero pages nothing matches, read_code()
    runs to the end of each restart interval, and what stays of an interval is the DC difference of its first Cb block
    (coeffs.c:222).  The oracle takes that reading (and so does the HIP path: test_gpu_damaged.py)."""
    with open(golden_path("damaged/undefined_ac_table_in_second_scan.jpg"), "rb") as fh:
        jpg = fh.read()
    assert jpg[2861:2871] == bytes.fromhex("ffda000a020213031100"[:20])
    err, px, info = oracle.decode(jpg)
    assert err == 0
    cb, cr = info.comps[1]["coef"], info.comps[2]["coef"]
    assert [int(cb[k][0]) for k in range(6)] == [-9, 0, 0, 0, 0, 0]
    assert not cr.any()
    # ri = 3 MCUs of one Cb and one Cr block: only the first Cb block of an interval can hold anything, its DC alone
    for k in range(len(cb)):
        assert not cb[k][1:].any()
        assert k % 3 == 0 or cb[k][0] == 0
