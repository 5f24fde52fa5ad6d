"""The CPU oracle against the LIVE compiled reference (oracle/_ref).

oracle/_ref is built from /root/reference by oracle/Makefile (unmodified sources + tap harness).
Here (build container) it is built on demand; on the GPU box the prebuilt files travel with the
tree.  Where neither exists the module skips -- test_oracle_golden.py still pins the oracle
against the committed reference outputs.
"""
import numpy as np
import pytest

from jpeg_amd.synth import synth_image


def _same_coefs(a, b):
    return len(a.comps) == len(b.comps) and all(
        (x["id"], x["H"], x["V"], x["bx"], x["by"]) == (y["id"], y["H"], y["V"], y["bx"], y["by"])
        and np.array_equal(x["coef"], y["coef"]) for x, y in zip(a.comps, b.comps))


def test_lut_is_the_reference_lut(oracle, reference):
    assert np.array_equal(oracle.lut().view(np.uint32), reference.lut().view(np.uint32))


def test_block_primitives_random(oracle, reference):
    rng = np.random.default_rng(3)
    for t in range(200):
        scale = 10.0 ** rng.integers(0, 5)
        blk = (rng.standard_normal(64) * scale).astype(np.float32)
        assert np.array_equal(oracle.fdct(blk).view(np.uint32), reference.fdct(blk).view(np.uint32))
        assert np.array_equal(oracle.idct(blk).view(np.uint32), reference.idct(blk).view(np.uint32))
        Q = rng.integers(1, 256, 64).astype(np.uint16)
        # include exact .5 ties
        if t % 4 == 0:
            blk = (rng.integers(-400, 400, 64) * Q / 2).astype(np.float32)
        assert np.array_equal(oracle.quantize(blk, Q), reference.quantize(blk, Q))


CASES = [
    # X, Y, nc, P, H, V, q, opt, safe
    (64, 64, 3, 8, 1, 1, 90, 1, False),
    (67, 35, 3, 8, 2, 2, 60, 1, False),
    (130, 18, 3, 8, 2, 1, 85, 0, False),
    (24, 90, 3, 8, 1, 2, 40, 1, False),
    (57, 43, 1, 8, 1, 1, 99, 1, False),
    (160, 96, 1, 12, 1, 1, 75, 1, True),
    (96, 64, 3, 12, 2, 2, 75, 1, True),
    (320, 200, 3, 8, 2, 2, 95, 0, False),
    (16, 16, 3, 8, 2, 2, 5, 1, False),
]


@pytest.mark.parametrize("case", CASES)
def test_encode_decode_bit_exact(oracle, reference, case):
    X, Y, nc, P, H, V, q, opt, safe = case
    img = synth_image(X, Y, nc, P, seed=X * 7 + Y, range_safe=safe)
    e1, j1, i1 = oracle.encode(img, P=P, H=H, V=V, q=q, optimize=opt)
    assert e1 == 0
    e2, j2, i2 = reference.encode(img, P=P, H=H, V=V, q=q, optimize=opt)
    assert e2 == 0
    assert _same_coefs(i1, i2)                       # quantised coefficients
    assert np.array_equal(i1.qtables[:2], i2.qtables[:2])
    assert j1 == j2                                  # whole codestream incl. DHT of the K.2 optimiser
    # float planes behind transform_frame_to_components (component[].frame_buffer, frame.c:84-132), padding included
    assert i2.planes is not None and len(i1.planes) == len(i2.planes)
    for a, b in zip(i1.planes, i2.planes):
        assert a.shape == b.shape and np.array_equal(a.view(np.uint32), b.view(np.uint32))
    e1, s1, d1 = oracle.decode(j2)
    e2, s2, d2 = reference.decode(j2)
    assert e1 == 0 and e2 == 0
    assert _same_coefs(d1, d2) and _same_coefs(d1, i2)
    assert np.array_equal(s1, s2)                    # output samples
    # float frame before rounding, visible region (padding is uninitialised in the reference)
    assert np.array_equal(d1.frame[:Y, :X].view(np.uint32), d2.frame[:Y, :X].view(np.uint32))


def test_random_noise_images(oracle, reference):
    """High-entropy content: long codes, ZRL runs, big categories."""
    rng = np.random.default_rng(11)
    for (X, Y, nc, H, V, q) in [(48, 48, 3, 2, 2, 90), (40, 24, 3, 1, 1, 100), (64, 32, 1, 1, 1, 97)]:
        img = rng.integers(96, 160, (Y, X, nc) if nc > 1 else (Y, X)).astype(np.uint8)
        e1, j1, i1 = oracle.encode(img, P=8, H=H, V=V, q=q, optimize=1)
        e2, j2, i2 = reference.encode(img, P=8, H=H, V=V, q=q, optimize=1)
        assert e1 == 0 and e2 == 0 and j1 == j2
        e1, s1, _ = oracle.decode(j1)
        e2, s2, _ = reference.decode(j1)
        assert e1 == 0 and e2 == 0 and np.array_equal(s1, s2)


def test_restart_extension_decodes_identically_with_reference(oracle, reference):
    """Restart-marker ENCODE is an extension (the reference never emits DRI/RSTn).
    Its oracle is: the reference DECODER decodes it to the same coefficients and samples as the
    no-restart stream."""
    img = synth_image(96, 80, 3, 8, seed=5)
    for (H, V, ri) in [(2, 2, 1), (2, 2, 6), (1, 1, 12), (2, 1, 5), (2, 2, 1000)]:
        e0, j0, i0 = oracle.encode(img, P=8, H=H, V=V, q=85, optimize=1, restart_interval=0)
        e1, j1, i1 = oracle.encode(img, P=8, H=H, V=V, q=85, optimize=1, restart_interval=ri)
        assert e0 == 0 and e1 == 0
        er, sr, dr = reference.decode(j1)
        e0r, s0r, d0r = reference.decode(j0)
        assert er == 0 and e0r == 0
        assert _same_coefs(dr, d0r)
        assert np.array_equal(sr, s0r)
        eo, so, do = oracle.decode(j1)
        assert eo == 0 and np.array_equal(so, sr)


def test_several_scans_decode_identically_with_reference(oracle, reference):
    """Frames coded in several scans are written by the oracle's extension writer (the reference encoder
    writes one scan).  The reference DECODER is the judge: same samples and coefficients as from the
    one-scan stream where the intervals respect its block groups, and the oracle's decoder agrees with it
    on every stream, the bug-compatible ones included (interval inside the H x V block group)."""
    img = synth_image(104, 72, 3, 8, seed=6)
    for split in (0, 1, 2):
        for (H, V, ri, opt, redefine) in [(2, 2, 0, 1, 0), (2, 1, 4, 0, 1), (1, 1, 7, 1, 0), (2, 2, 5, 1, 1), (1, 2, 2, 0, 0)]:
            e0, j0, _ = oracle.encode(img, P=8, H=H, V=V, q=85, optimize=opt)
            e1, j1 = oracle.encode_scans(img, P=8, H=H, V=V, q=85, optimize=opt, restart_interval=ri,
                                         redefine=redefine, split=split)
            assert e0 == 0 and e1 == 0
            er, sr, dr = reference.decode(j1)
            eo, so, do = oracle.decode(j1)
            assert er == 0 and eo == 0
            assert np.array_equal(so, sr) and _same_coefs(do, dr)
            if ri % (H * V) == 0:
                e0r, s0r, d0r = reference.decode(j0)
                assert e0r == 0 and np.array_equal(sr, s0r) and _same_coefs(dr, d0r)


def test_truncated_and_garbage_streams_agree(oracle, reference):
    img = synth_image(64, 48, 3, 8, seed=9)
    _, jpg, _ = reference.encode(img, P=8, H=2, V=2, q=80, optimize=1)
    # ECS cut short but EOI present: remaining blocks stay zero in both
    cut = jpg[:len(jpg) - 200] + b"\xff\xd9"
    e1, s1, d1 = oracle.decode(cut)
    e2, s2, d2 = reference.decode(cut)
    assert e1 == e2
    if e1 == 0:
        assert d1.mblocks > 0
        assert np.array_equal(s1, s2)
    # no EOI at all: both fail with an I/O error
    e1, _, _ = oracle.decode(jpg[:-2])
    e2, _, _ = reference.decode(jpg[:-2])
    assert e1 == e2 and e1 != 0


def test_zero_run_past_the_end_of_a_block_ends_the_block(oracle, reference):
    """Damaged data: a zero run that carries the coefficient pointer past 63.  The reference indexes zigzag[]
    out of range there (coeffs.c:246-247, undefined behaviour: a stray store), but its control flow is
    defined: `rem` is used up, the block is complete, decoding goes on.  The oracle follows that flow and
    leaves the store out.  With ZRL as the overshooting symbol the stray store writes a zero into a block
    that is zero (or is reset later), so the compiled reference and the oracle must agree on everything;
    with a nonzero value they must agree on every block but the damaged one."""
    from tests.damaged import zero_run_past_block_end
    for value_cat in (0, 1):
        jpg, want = zero_run_past_block_end(oracle, value_cat)
        eo, po, io = oracle.decode(jpg)
        er, pr, ir = reference.decode(jpg)
        assert eo == 0 and er == 0
        co, cr = io.comps[0]["coef"], ir.comps[0]["coef"]
        for b, vals in want.items():
            exp = np.zeros(64, np.int32)
            for k, v in vals.items():
                exp[k] = v
            assert np.array_equal(co[b], exp) and np.array_equal(cr[b], exp), (value_cat, b)
        assert not co[0].any()                           # the store is left out
        if value_cat == 0:
            assert np.array_equal(co, cr) and np.array_equal(po, pr)


def test_data_that_runs_out_inside_an_mcu(oracle, reference):
    """decoder.c:347-355, 376-381, coeffs.c:222: the blocks of the unfinished MCU that were completed stay (with
    prediction), the block the data ran out in stays with its DC difference and the AC coefficients read so
    far, the MCU is not counted, and the next restart interval starts over it.  The oracle against the compiled
    reference on streams cut at many places, with and without restart markers, EOI appended."""
    img = synth_image(96, 64, 3, 8, seed=77)
    for H, V, ri in ((2, 2, 0), (2, 1, 2), (1, 1, 3)):
        err, jpg, _ = oracle.encode(img, P=8, H=H, V=V, q=90, optimize=1, restart_interval=ri)
        assert err == 0
        sos = jpg.index(b"\xff\xda")
        start = sos + 2 + ((jpg[sos + 2] << 8) | jpg[sos + 3])
        for cut in range(start + 5, len(jpg) - 2, max(1, (len(jpg) - start) // 37)):
            bad = jpg[:cut] + b"\xff\xd9"
            eo, po, io = oracle.decode(bad)
            er, pr, ir = reference.decode(bad)
            assert eo == er
            if eo == 0:
                assert _same_coefs(io, ir), (H, V, ri, cut)
                assert np.array_equal(po, pr)
    # restart intervals whose data is cut in the middle of the stream (not only at its end)
    err, jpg, _ = oracle.encode(img, P=8, H=2, V=2, q=90, optimize=0, restart_interval=2)
    marks = [i for i in range(len(jpg) - 1) if jpg[i] == 0xFF and 0xD0 <= jpg[i + 1] <= 0xD7]
    for m in marks[1:8]:
        for back in (1, 3, 9):
            bad = jpg[:m - back] + jpg[m:]
            eo, po, io = oracle.decode(bad)
            er, pr, ir = reference.decode(bad)
            assert eo == er == 0
            assert _same_coefs(io, ir), (m, back)
            assert np.array_equal(po, pr)


def _scan_segments(jpg):
    """(start of SOS marker, end of its entropy-coded data) of every scan of a file"""
    out = []
    i = 2
    while i + 3 < len(jpg):
        m = jpg[i + 1]
        if m == 0xD9:
            break
        ln = (jpg[i + 2] << 8) | jpg[i + 3]
        if m != 0xDA:
            i += 2 + ln
            continue
        j = i + 2 + ln
        while not (jpg[j] == 0xFF and jpg[j + 1] != 0 and not 0xD0 <= jpg[j + 1] <= 0xD7):
            j += 1
        out.append((i, j))
        i = j
    return out


def test_components_coded_more_than_once(oracle, reference):
    """decoder.c:593-601 reads whatever scans come, and read_block() stores every coefficient of a block it reads
    (coeffs.c:222-229): a component coded again is written again.  The oracle against the compiled reference on
    files spliced from the scans of two images (the GPU test uses the same constructions)."""
    a = synth_image(176, 112, 3, 8, seed=41)
    b = synth_image(176, 112, 3, 8, seed=42)
    for ri in (0, 5):
        _, ja = oracle.encode_scans(a, P=8, H=2, V=2, q=80, optimize=0, restart_interval=ri, redefine=0, split=0)
        _, jb = oracle.encode_scans(b, P=8, H=2, V=2, q=80, optimize=0, restart_interval=ri, redefine=0, split=0)
        _, jfa, _ = oracle.encode(a, P=8, H=2, V=2, q=80, optimize=0, restart_interval=ri)
        _, jfb, _ = oracle.encode(b, P=8, H=2, V=2, q=80, optimize=0, restart_interval=ri)
        sa, sb, sfa, sfb = _scan_segments(ja), _scan_segments(jb), _scan_segments(jfa), _scan_segments(jfb)
        files = [
            ja[:sa[-1][1]] + jb[sb[0][0]:sb[0][1]] + ja[sa[-1][1]:],            # luma twice
            ja[:sa[-1][1]] + jfb[sfb[0][0]:sfb[0][1]] + ja[sa[-1][1]:],         # every component again
            jfa[:sfa[0][1]] + jb[sb[0][0]:sb[0][1]] + jfa[sfa[0][1]:],          # all, then luma
            jfa[:sfa[0][1]] + jfb[sfb[0][0]:sfb[0][1]] + jfa[sfa[0][1]:],       # all twice
        ]
        for jpg in files:
            eo, po, io = oracle.decode(jpg)
            er, pr, ir = reference.decode(jpg)
            assert eo == er == 0
            assert _same_coefs(io, ir) and np.array_equal(po, pr)


def test_a_dozen_scans(oracle, reference):
    """Thirteen scans in one file (the construction of tests/test_gpu_parity.py::test_a_dozen_scans): the reference
    reads as many as there are (decoder.c:593-601)."""
    imgs = [synth_image(176, 112, 3, 8, seed=61 + i) for i in range(3)]
    files = [oracle.encode_scans(im, P=8, H=2, V=2, q=80, optimize=0, restart_interval=0, redefine=0, split=0)[1] for im in imgs]
    segs = [_scan_segments(j) for j in files]
    ja, sa = files[0], segs[0]
    luma = [files[k][segs[k][0][0]:segs[k][0][1]] for k in (1, 2)]
    jpg = ja[:sa[-1][1]] + b"".join(luma[i % 2] for i in range(10)) + ja[sa[-1][1]:]
    assert len(_scan_segments(jpg)) == 13
    eo, po, io = oracle.decode(jpg)
    er, pr, ir = reference.decode(jpg)
    assert eo == er == 0
    assert _same_coefs(io, ir) and np.array_equal(po, pr)


def test_a_table_with_more_codes_than_a_size_has_room_for(oracle, reference):
    """tests/golden/damaged/oversubscribed_ac_table.jpg (tools/fuzz_headers.py, seed 304, case 734): three more 16-bit
    codes than the file's AC table 1 has room for (a damaged count in the DHT segment, whose symbols are then read out
    of the next segment's bytes, decoder.c:163-175).  generate_code_table() counts CODE in 16 bits (huffman.c:91): the
    last codes wrap to 0, 1, 2, the codes of size 16 are no longer one range, and query_code()'s search in table
    order (huffman.c:193-225) is what says which bits match.  The oracle had a range test there and lost every 16-bit
    code of the table; the HIP path, which fills its tables code by code, agreed with the reference all along."""
    from tests.conftest import golden_path
    with open(golden_path("damaged/oversubscribed_ac_table.jpg"), "rb") as fh:
        jpg = fh.read()
    eo, po, io = oracle.decode(jpg)
    er, pr, ir = reference.decode(jpg)
    assert eo == er == 0
    assert _same_coefs(io, ir) and np.array_equal(po, pr)


def test_eight_huffman_tables_in_one_scan(oracle, reference):
    """The Pillow-written CMYK golden with a DC and an AC table of its own for each of the four components
    (tests/damaged.py: table_per_component): every table id of decoder.c:176-182 in use in one scan."""
    from tests.conftest import golden_path
    from tests.damaged import table_per_component
    with open(golden_path("d_pil_cmyk.jpg"), "rb") as fh:
        jpg = fh.read()
    eight = table_per_component(jpg)
    eo, po, io = oracle.decode(eight)
    er, pr, ir = reference.decode(eight)
    assert eo == er == 0
    assert _same_coefs(io, ir) and np.array_equal(po, pr)
    assert np.array_equal(po, oracle.decode(jpg)[1])


def test_sample_precisions_beyond_12_bits(oracle, reference):
    """P 13 ... 16 (frame.c:270-275): the oracle against the compiled reference, both directions, on the
    low-contrast images the reference's DC assert leaves it at such precisions (the GPU test uses the same)."""
    from tests.damaged import deep_image
    for P in (13, 14, 15, 16):
        for nc, H, V in ((1, 1, 1), (3, 2, 2)):
            img = deep_image(P, nc, seed=P + nc)
            for q, opt in ((30, 1), (50, 0)):
                eo, jo, io = oracle.encode(img, P=P, H=H, V=V, q=q, optimize=opt)
                er, jr, ir = reference.encode(img, P=P, H=H, V=V, q=q, optimize=opt)
                assert eo == er == 0 and jo == jr, (P, nc, q, opt)
                do, po, ido = oracle.decode(jo)
                dr, pr, idr = reference.decode(jo)
                assert do == dr == 0 and np.array_equal(po, pr) and _same_coefs(ido, idr)
