"""Damaged codestreams on the HIP path against the oracle (which tests/test_oracle_vs_ref.py pins to the compiled
reference on the same kinds of damage).  Nothing is masked: status, every coefficient and every sample are
compared.  What the GPU path deliberately does otherwise is listed by name in DEVIATIONS below and in DESIGN.md;
a case may differ only if one of them claims it.

  * data that runs out inside an MCU (decoder.c:347-355, 376-381; coeffs.c:222)
  * a zero run that ends past coefficient 63 (coeffs.c:246-251)
  * markers that do not belong into the entropy-coded data (io.c:257-268, decoder.c:487-657)
  * a fixed-seed run of tools/fuzz_cases.py (bit flips, cuts, garbage, stray FF xx; one scan or several)
"""
import random

import numpy as np
import pytest

from jpeg_amd.synth import synth_image
from tests.damaged import zero_run_past_block_end
from tools.fuzz_cases import corrupt_case

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def hip():
    from jpeg_amd import binding
    assert binding.lib().jpeg_hip_device_count() > 0, "no HIP device: the product path has no CPU fallback"
    return binding


def _ecs_start(jpg):
    sos = jpg.index(b"\xff\xda")
    return sos + 2 + ((jpg[sos + 2] << 8) | jpg[sos + 3])


def _segments_behind(bad, pos):
    """marker codes from `pos` on, as parse_format() would meet them (segments with a length are stepped over)"""
    out = []
    i = pos
    while i + 1 < len(bad) and len(out) < 64:
        if bad[i] != 0xFF or bad[i + 1] in (0x00, 0xFF):
            i += 1
            continue
        m = bad[i + 1]
        out.append(m)
        i += 2
        if m == 0xD9:
            break
        if not (0xD0 <= m <= 0xD8 or m == 0x01) and i + 1 < len(bad):
            i += (bad[i] << 8) | bad[i + 1]
    return out


def _first_stray_marker(bad):
    i = _ecs_start(bad)
    while i + 1 < len(bad):
        if bad[i] == 0xFF and bad[i + 1] not in (0x00, 0xFF) and not 0xD0 <= bad[i + 1] <= 0xD7:
            return i
        i += 1
    return None


# Named deviations.  Each one: (name, predicate(bad, hip status, oracle status)).
def _tables_behind_a_stray_marker(bad, st, oerr):
    """D1: the scan was ended by a marker that is neither RSTn nor EOI, and a table, frame or scan header comes
    before the next RSTn/EOI.  The reference reads it and goes on with the changed tables; the GPU path made its
    decoding tables when the scan started and reports FILE_UNSUPPORTED (0x1001) for the frame.  In files of
    several scans the same happens behind a damaged DHT or SOS segment: the reference reads past it, skips to the
    next marker, and decodes the restart intervals it finds there as more of the scan before."""
    if st != 0x1001 or oerr != 0:
        return False
    at = _first_stray_marker(bad)
    if at is None:
        return False
    for m in _segments_behind(bad, at):
        if m in (0xC4, 0xDB, 0xDA) or 0xC0 <= m <= 0xCF:
            return True
        if 0xD0 <= m <= 0xD7 or m == 0xD9:
            return False
    return False


DEVIATIONS = [("tables behind a stray marker", _tables_behind_a_stray_marker)]


def _compare(hip, oracle, bad, what):
    """-> name of the deviation that explains a difference, or None if hip == oracle; asserts otherwise"""
    oerr, opx, oinfo = oracle.decode(bad)
    d = hip.Decoder()
    try:
        try:
            d.prepare([bad])
            out = d.run()
            st = d.sync()[0]
        except hip.JpegError as e:
            st, out = e.code, None
        if st == oerr:
            if st == 0:
                if max(int(np.abs(c["coef"]).max()) for c in oinfo.comps) > 32767:
                    # D4: garbage whose DC differences add up past 16 bits: the reference keeps int32 coefficients
                    # (coeffs.h:9-11), this path int16 -- no codestream of an image gets there
                    return "coefficients beyond 16 bits"
                for k, (a, c) in enumerate(zip(d.coefficients(0), oinfo.comps)):
                    df = np.nonzero((a != c["coef"]).any(axis=1))[0]
                    assert len(df) == 0, "%s: component %d, %d blocks differ, first %d" % (what, k, len(df), df[0])
                assert np.array_equal(d.to_arrays(out)[0], opx), what
            return None
        for name, claims in DEVIATIONS:
            if claims(bad, st & 0xFFFFFFFF, oerr & 0xFFFFFFFF):
                return name
        raise AssertionError("%s: status hip 0x%x, oracle 0x%x, and no documented deviation covers it" % (
            what, st & 0xFFFFFFFF, oerr & 0xFFFFFFFF))
    finally:
        d.close()


def test_data_that_runs_out_inside_an_mcu(hip, oracle):
    """Cuts all over the stream, EOI appended or not: the completed blocks of the unfinished MCU stay, the block
    the data ran out in keeps its DC difference and the AC coefficients read so far, and the next restart
    interval starts over that MCU."""
    img = synth_image(96, 64, 3, 8, seed=77)
    for H, V, ri in ((2, 2, 0), (2, 1, 2), (1, 1, 3)):
        err, jpg, _ = oracle.encode(img, P=8, H=H, V=V, q=90, optimize=1, restart_interval=ri)
        assert err == 0
        start = _ecs_start(jpg)
        for cut in range(start + 5, len(jpg) - 2, max(1, (len(jpg) - start) // 37)):
            assert _compare(hip, oracle, jpg[:cut] + b"\xff\xd9", (H, V, ri, cut)) is None
        assert _compare(hip, oracle, jpg[:start + (len(jpg) - start) // 2], "no EOI") is None
    # gray (one block per MCU) and a larger frame whose stream spans several workgroups
    g = synth_image(200, 120, 1, 8, seed=5)
    err, jpg, _ = oracle.encode(g, P=8, H=1, V=1, q=92, optimize=0, restart_interval=0)
    start = _ecs_start(jpg)
    for cut in range(start + 3, len(jpg) - 2, (len(jpg) - start) // 23):
        assert _compare(hip, oracle, jpg[:cut] + b"\xff\xd9", ("gray", cut)) is None
    big = synth_image(1280, 720, 3, 8, seed=9)
    for ri in (0, 40):
        err, jpg, _ = oracle.encode(big, P=8, H=2, V=2, q=93, optimize=0, restart_interval=ri)
        start = _ecs_start(jpg)
        for frac in (0.11, 0.5, 0.93):
            cut = start + int((len(jpg) - start) * frac)
            assert _compare(hip, oracle, jpg[:cut] + b"\xff\xd9", ("720p", ri, frac)) is None


def test_restart_intervals_cut_in_the_middle_of_the_stream(hip, oracle):
    img = synth_image(96, 64, 3, 8, seed=77)
    for H, V, ri in ((2, 2, 2), (1, 1, 1), (2, 1, 5)):
        err, jpg, _ = oracle.encode(img, P=8, H=H, V=V, q=90, optimize=0, restart_interval=ri)
        assert err == 0
        start = _ecs_start(jpg)
        marks = [i for i in range(start, len(jpg) - 1) if jpg[i] == 0xFF and 0xD0 <= jpg[i + 1] <= 0xD7]
        for m in marks[1:8]:
            for back in (1, 3, 9):
                assert _compare(hip, oracle, jpg[:m - back] + jpg[m:], (H, V, ri, m, back)) is None
        # several intervals in a row damaged, the last one too
        bad = bytearray(jpg)
        for m in marks[2:5] + marks[-1:]:
            del bad[m - 4:m - 1]
            marks = [x - 3 for x in marks]
        assert _compare(hip, oracle, bytes(bad), (H, V, ri, "several")) is None


def test_frames_in_several_scans_cut_short(hip, oracle):
    """one scan per component, or groups of components: the file ends inside one scan (or inside the headers
    between two), the scans behind it are missing and their blocks stay zero (common.c:129)"""
    img = synth_image(112, 80, 3, 8, seed=31)
    for split in (0, 1, 2):
        for ri in (0, 3):
            err, jpg = oracle.encode_scans(img, P=8, H=2, V=1, q=88, optimize=1, restart_interval=ri, redefine=1, split=split)
            assert err == 0
            start = _ecs_start(jpg)
            n = 0
            for cut in range(start + 7, len(jpg) - 2, (len(jpg) - start) // 29):
                assert _compare(hip, oracle, jpg[:cut] + b"\xff\xd9", ("scans", split, ri, cut)) is None
                n += 1
            assert n > 10


@pytest.mark.parametrize("value_cat", [0, 1])
def test_zero_run_past_the_end_of_a_block(hip, oracle, value_cat):
    """coeffs.c:246-251: the run is counted down past the end, the block is complete without EOB, the value's
    extra bits are consumed, and decoding goes on with the next block."""
    bad, want = zero_run_past_block_end(oracle, value_cat)
    assert _compare(hip, oracle, bad, ("overshoot", value_cat)) is None
    d = hip.Decoder()
    d.prepare([bad])
    d.run()
    assert d.sync() == [0]
    co = d.coefficients(0)[0]
    for blk, vals in want.items():
        for idx, v in vals.items():
            assert co[blk][idx] == v
    assert not co[0].any()
    d.close()
    # the same inside a batch
    frames = [bad, bad, bad]
    outs, st = hip.decode(frames)
    assert st == [0, 0, 0]
    opx = oracle.decode(bad)[1]
    for o in outs:
        assert np.array_equal(o, opx)


def test_markers_inside_the_data_followed_by_restart_intervals(hip, oracle):
    """read_ecs() ends at the stray marker; parse_format() skips what has a length and, at the next RSTn, decodes
    on with the MCU counter where it stood (decoder.c:622-633).  One scan and several."""
    img = synth_image(160, 96, 3, 8, seed=12)
    err, jpg, _ = oracle.encode(img, P=8, H=2, V=2, q=85, optimize=1, restart_interval=4)
    start = _ecs_start(jpg)
    marks = [i for i in range(start, len(jpg) - 1) if jpg[i] == 0xFF and 0xD0 <= jpg[i + 1] <= 0xD7]
    for stray in (b"\xff\x01", b"\xff\xfe\x00\x05abc", b"\xff\xe1\x00\x02", b"\xff\xdd\x00\x04\x00\x09", b"\xff\xd8"):
        for at in (marks[1] - 6, marks[3], (marks[5] + marks[6]) // 2):
            if jpg[at - 1] == 0xFF:
                at += 1
            bad = jpg[:at] + stray + jpg[at:]
            assert _compare(hip, oracle, bad, (stray, at)) is None
        # two of them in one file
        a, b = marks[2] - 3, marks[7] - 5
        a += jpg[a - 1] == 0xFF
        b += jpg[b - 1] == 0xFF
        bad = jpg[:a] + stray + jpg[a:b] + stray + jpg[b:]
        assert _compare(hip, oracle, bad, (stray, "twice")) is None
    err, jpg = oracle.encode_scans(img, P=8, H=2, V=1, q=85, optimize=1, restart_interval=3, redefine=0, split=0)
    start = _ecs_start(jpg)
    marks = [i for i in range(start, len(jpg) - 1) if jpg[i] == 0xFF and 0xD0 <= jpg[i + 1] <= 0xD7]
    for at in (marks[2] - 5, marks[len(marks) // 2] - 4, marks[-2] - 3):
        at += jpg[at - 1] == 0xFF
        bad = jpg[:at] + b"\xff\xe1\x00\x04xy" + jpg[at:]
        assert _compare(hip, oracle, bad, ("scans", at)) is None
    # a table behind the stray marker is the documented deviation D1
    dht = jpg[jpg.index(b"\xff\xc4"):]
    dht = dht[:2 + ((dht[2] << 8) | dht[3])]
    at = marks[4] - 4
    at += jpg[at - 1] == 0xFF
    err, jpg1, _ = oracle.encode(img, P=8, H=2, V=2, q=85, optimize=1, restart_interval=4)
    marks1 = [i for i in range(_ecs_start(jpg1), len(jpg1) - 1) if jpg1[i] == 0xFF and 0xD0 <= jpg1[i + 1] <= 0xD7]
    at = marks1[4] - 4
    at += jpg1[at - 1] == 0xFF
    assert _compare(hip, oracle, jpg1[:at] + dht + jpg1[at:], "dht") == "tables behind a stray marker"


def test_streams_the_fuzz_runs_found(hip, oracle):
    """tests/golden/damaged/: streams of tools/fuzz_corrupt.py (seeds 22 and 23) on which the write pass lost the
    blocks of an unfinished MCU to k_coef_tail_clear -- data that ends exactly where a sub-sequence does, and a
    spurious RSTn whose segment holds nothing decodable behind a segment that ended inside a block; and (seed 202) a scan
    header that names a Huffman table no DHT segment defined, and (header fuzz, seed 304) a DHT segment with more 16-bit
    codes than there is room for -- where it was the ORACLE that had to change, both times."""
    import glob
    import os
    files = sorted(glob.glob(os.path.join(os.path.dirname(__file__), "golden", "damaged", "*.jpg")))
    assert len(files) >= 3
    for path in files:
        with open(path, "rb") as fh:
            assert _compare(hip, oracle, fh.read(), os.path.basename(path)) is None


def test_scans_that_name_tables_no_dht_defined(hip, oracle):
    """golden/damaged/undefined_ac_table_in_second_scan.jpg (fuzz seed 202, case 18356: a bit flip in the second SOS
    header, Ta = 3 for Cb) and its relatives: table ids 2 and 3 without a DHT segment, for the DC or the AC symbols,
    in the luma scan or the chroma scan.  The reference reads uninitialised memory there (common.c:90-99 converts the
    implicit tables 0 / 1 only; its output changes from run to run); the oracle and this path take what zero pages
    give -- no code matches, the restart interval is used up (tests/test_oracle_golden.py has the details)."""
    import os
    with open(os.path.join(os.path.dirname(__file__), "golden", "damaged", "undefined_ac_table_in_second_scan.jpg"), "rb") as fh:
        jpg0 = fh.read()
    for off in (2867, 2869, 603):                        # Td | Ta of Cb, of Cr (second scan), of Y (first scan)
        for val in (0x13, 0x12, 0x31, 0x21, 0x33, 0x22, 0x10, 0x01, 0x30, 0x03):
            b = bytearray(jpg0)
            b[off] = val
            assert _compare(hip, oracle, bytes(b), ("undefined table", off, hex(val))) is None


def test_data_that_ends_where_a_sub_sequence_ends(hip, oracle):
    """The unstuffed data is cut to whole multiples of 256 bytes (the decoder's sub-sequences): the thread that
    meets the end is then past its own sub-sequence and no thread starts behind it."""
    img = synth_image(320, 200, 3, 8, seed=13)
    for H, V in ((2, 1), (1, 1), (2, 2)):
        err, jpg, _ = oracle.encode(img, P=8, H=H, V=V, q=88, optimize=0)
        start = _ecs_start(jpg)
        body = jpg[start:-2]
        for nsub in (3, 8, 20, 21):
            # walk the stuffed bytes until nsub * 256 payload bytes are behind us
            kept, i = 0, 0
            while kept < nsub * 256 and i < len(body):
                i += 2 if body[i] == 0xFF else 1
                kept += 1
            for extra in (0, 1):
                bad = jpg[:start] + body[:i + extra] + b"\xff\xd9"
                if bad[-3] == 0xFF:
                    continue
                assert _compare(hip, oracle, bad, ("boundary", H, V, nsub, extra)) is None


@pytest.mark.parametrize("seed", [7, 11, 2024, 22, 23])
def test_fixed_seed_fuzz(hip, oracle, seed):
    """300 damaged streams per seed: identical to the oracle, or claimed by a named deviation (counted and
    bounded: they are rare)."""
    rnd = random.Random(seed)
    seen = {}
    n = 0
    for case in range(300):
        bad, meta = corrupt_case(rnd, oracle, seed, case)
        if bad is None:
            continue
        n += 1
        name = _compare(hip, oracle, bad, (seed, case, meta))
        if name:
            seen[name] = seen.get(name, 0) + 1
    assert n > 250
    assert sum(seen.values()) <= n // 50, seen
