"""Host logic without a GPU: jpeg_hip_parse_info() -- the marker parser (decoder.c:472-601 restated for
memory) and the geometry (common.c:156-195) -- against the golden manifest, the oracle and, on damaged
headers, the error codes the oracle (pinned to the reference) returns for the same bytes."""
import json
import os

import pytest

HERE = os.path.dirname(os.path.abspath(__file__))


@pytest.fixture(scope="module")
def hip():
    from jpeg_amd import binding
    if not os.path.exists(binding.LIB_PATH):
        import __graft_entry__
        __graft_entry__.build()
    return binding


def _golden():
    with open(os.path.join(HERE, "golden", "manifest.json")) as f:
        man = json.load(f)
    out = []
    for kind in ("decode", "encode"):
        for name, g in sorted(man[kind].items()):
            if "jpg_file" in g:
                out.append((name, g))
    return out


def _read(g):
    with open(os.path.join(HERE, "golden", g["jpg_file"]), "rb") as f:
        return f.read()


@pytest.mark.parametrize("name,g", _golden())
def test_headers_of_every_golden_stream(hip, oracle, name, g):
    jpg = _read(g)
    err, fi = hip.parse_info(jpg)
    oerr, _, info = oracle.decode(jpg)
    if g.get("err", 0) != 0:
        assert err == g["err"] == oerr            # e.g. progressive: RET_FAILURE_FILE_UNSUPPORTED
        return
    assert err == 0 and oerr == 0
    assert (fi.X, fi.Y, fi.P, fi.Nf) == (info.X, info.Y, info.P, info.Nf)
    assert (fi.m_x * (1 if fi.Nf > 1 else 1), fi.size_x, fi.size_y) == (fi.m_x, info.size_x, info.size_y)
    for k, c in enumerate(info.comps):
        assert fi.id[k] == c["id"] and fi.Tq[k] == c["Tq"]
        assert (fi.bx[k], fi.by[k]) == (c["bx"], c["by"])
        if fi.Nf > 1:
            assert (fi.H[k], fi.V[k]) == (c["H"], c["V"])
    assert fi.out_components == (1 if info.Nf == 1 else 3)
    assert fi.sample_bytes == (1 if info.P <= 8 else 2)


def test_damaged_headers_fail_as_in_the_reference(hip, oracle):
    g = dict(_golden())
    jpg = _read(g["d_pil_422_jfif"])
    sos = jpg.index(b"\xff\xda")
    # cut anywhere inside the headers: read_*() meets the end of the file
    for cut in range(2, sos + 14):
        err, _ = hip.parse_info(jpg[:cut])
        assert err != 0
        assert err == oracle.decode(jpg[:cut])[0], cut
    # markers parse_format() does not know (decoder.c:654-656), and the tolerant mode
    for marker in (b"\xff\xe9", b"\xff\xea", b"\xff\xef", b"\xff\xf3"):
        odd = jpg[:2] + marker + b"\x00\x04ab" + jpg[2:]
        assert hip.parse_info(odd)[0] == 0x1001 == oracle.decode(odd)[0]
        err, fi = hip.parse_info(odd, tolerant=True)
        assert err == 0 and fi.X > 0
    # segments it does know pass: COM, APP1, DAC, TEM
    ok = jpg[:2] + b"\xff\xfe\x00\x05abc" + b"\xff\xe1\x00\x02" + b"\xff\xcc\x00\x04\x00\x10" + b"\xff\x01" + jpg[2:]
    assert hip.parse_info(ok)[0] == 0 == oracle.decode(ok)[0]
    # frame types the reference refuses (decoder.c:545-579)
    sof = jpg.index(b"\xff\xc0")
    for m in (0xC2, 0xC3, 0xC9, 0xCA):
        bad = jpg[:sof + 1] + bytes([m]) + jpg[sof + 2:]
        assert hip.parse_info(bad)[0] == 0x1001 == oracle.decode(bad)[0]
    # a scan before any frame header (decoder.c:306-309)
    assert hip.parse_info(jpg[:2] + jpg[sos:])[0] == oracle.decode(jpg[:2] + jpg[sos:])[0] != 0
    # EOI before any scan: the reference writes the image of zero coefficients (decoder.c:603-612)
    err, fi = hip.parse_info(jpg[:sos] + b"\xff\xd9")
    assert err == 0 == oracle.decode(jpg[:sos] + b"\xff\xd9")[0] and fi.X > 0
    # ... but not without a frame header
    assert hip.parse_info(b"\xff\xd8\xff\xd9")[0] == 0x1001
    # sampling factors of 0 make the reference divide by zero (common.c:171-172): refused here
    for off in (11, 14, 17):
        bad = jpg[:sof + off] + b"\x00" + jpg[sof + off + 1:]
        assert hip.parse_info(bad)[0] == 0x1001
    # not a codestream at all
    assert hip.parse_info(b"")[0] == 0x1000
    assert hip.parse_info(b"\x00" * 64)[0] == oracle.decode(b"\x00" * 64)[0] != 0


def test_host_parser_under_sanitizers(tmp_path):
    """tools/host_sanitize.cpp: the marker parser, the table builders, the splitter and the walk behind stray markers
    on damaged golden streams under AddressSanitizer / UBSan (host code only; a short run here, 4 000 mutations per
    file when the tool is run by hand)."""
    import glob
    import shutil
    import subprocess
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    if shutil.which("g++") is None:
        pytest.skip("no g++")
    exe = str(tmp_path / "host_sanitize")
    subprocess.run(["g++", "-std=c++17", "-O1", "-g", "-fsanitize=address,undefined", "-fno-sanitize-recover=undefined",
                    "-I" + os.path.join(root, "include"), "-I" + os.path.join(root, "jpeg_amd", "csrc"),
                    os.path.join(root, "tools", "host_sanitize.cpp"), "-x", "c++",
                    os.path.join(root, "jpeg_amd", "csrc", "codestream.cpp"), "-o", exe], check=True)
    files = sorted(glob.glob(os.path.join(root, "tests", "golden", "*.jpg")))[:12] + \
        sorted(glob.glob(os.path.join(root, "tests", "golden", "damaged", "*.jpg")))
    env = dict(os.environ, HOST_SAN_ITERS="150", ASAN_OPTIONS="verify_asan_link_order=0")
    r = subprocess.run([exe] + files, env=env, capture_output=True, text=True, timeout=600)
    assert r.returncode == 0, r.stderr[-2000:]
    assert "sanitizer run:" in r.stdout
