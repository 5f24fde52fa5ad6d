"""CPU-side checks of the C-ABI boundary: the shared library loads and exports every symbol that
include/jpeg_amd.h declares.  No compute calls (those need a GPU: tests/test_gpu_*.py)."""
import ctypes as C
import os
import re

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


@pytest.fixture(scope="module")
def built_lib():
    from jpeg_amd import binding
    if not os.path.exists(binding.LIB_PATH):
        import __graft_entry__
        __graft_entry__.build()
    return binding


def test_header_symbols_are_exported(built_lib):
    hdr = open(os.path.join(ROOT, "include", "jpeg_amd.h")).read()
    declared = sorted(set(re.findall(r"\b(jpeg_hip_[a-z0-9_]+)\s*\(", hdr)))
    assert len(declared) >= 25
    L = C.CDLL(built_lib.LIB_PATH)
    for name in declared:
        assert hasattr(L, name), "missing export: " + name
    assert sorted(built_lib.EXPORTS) == declared


def test_every_entry_point_cites_the_reference(built_lib):
    hdr = open(os.path.join(ROOT, "include", "jpeg_amd.h")).read()
    for ref in ("decoder.c:364", "decoder.c:456", "encoder.c:176", "encoder.c:525", "decoder.c:661",
                "decoder.c:687", "encoder.c:647", "common.h:15-31", "decoder.c:364-388", "decoder.c:456-470",
                "encoder.c:176-193", "encoder.c:525-587"):
        assert ref in hdr


def test_struct_layouts_match_header(built_lib):
    # jpeg_hip_frame_info: 4 + 6*4 + 6 + 2*4 + 4 ints
    assert C.sizeof(built_lib.FrameInfo) == 4 * (4 + 24 + 6 + 8 + 4)
    assert C.sizeof(built_lib.Params) == 12          # == sizeof(struct params), encoder.c:68-76
    # stage level: jpeg_hip_frame_desc 4 + 3*4 ints, jpeg_hip_htable 16 + 256 bytes, jpeg_hip_scan_desc 1 + 3*4 + 1 ints
    assert C.sizeof(built_lib.FrameDesc) == 4 * 16
    assert C.sizeof(built_lib.HTable) == 272
    assert C.sizeof(built_lib.ScanDesc) == 4 * 14


def test_set_qtable_is_the_reference_scaling(built_lib, oracle):
    """jpeg_hip_set_qtable = set_qtable() + quality_to_sf() + the Annex K.1 tables (encoder.c:14-65); host only."""
    import numpy as np
    L = built_lib.lib()
    for chroma in (0, 1):
        for q in (-3, 0, 1, 2, 10, 49, 50, 51, 75, 90, 95, 99, 100, 250):
            Q = np.zeros(64, np.uint16)
            L.jpeg_hip_set_qtable(Q.ctypes.data, chroma, q)
            assert np.array_equal(Q, oracle.qtable(chroma, q)), (chroma, q)


def test_ref_glue_compiles_against_the_reference_headers(built_lib, tmp_path):
    """integration/ref_glue.c (the binding of INTEGRATION.md section 2) in strict C99 against the reference's
    own headers; linking and running need the GPU (tests/test_gpu_ref_glue.py)."""
    import shutil
    import subprocess
    if shutil.which("gcc") is None or not os.path.isdir("/root/reference"):
        pytest.skip("needs gcc and /root/reference")
    for which in ("GLUE_DECODER", "GLUE_ENCODER", "GLUE_STAGES"):
        subprocess.run(["gcc", "-std=c99", "-pedantic", "-Wall", "-Wextra", "-Werror", "-D_XOPEN_SOURCE=700",
                        "-I/root/reference", "-I" + os.path.join(ROOT, "include"), "-D" + which, "-c",
                        os.path.join(ROOT, "integration", "ref_glue.c"), "-o", str(tmp_path / (which + ".o"))],
                       check=True, timeout=120)


def test_fails_loudly_without_gpu(built_lib):
    L = built_lib.lib()
    if L.jpeg_hip_device_count() > 0:
        pytest.skip("a GPU is visible here")
    h = C.c_void_p()
    assert L.jpeg_hip_decoder_create(C.byref(h), 0) == built_lib.RET_FAILURE_LOGIC_ERROR
    assert not h
    assert L.jpeg_hip_encoder_create(C.byref(h), 0) == built_lib.RET_FAILURE_LOGIC_ERROR
    with pytest.raises(built_lib.JpegError):
        built_lib.Decoder()
    # the stage level entry points refuse as well
    import numpy as np
    fd = built_lib.frame_desc(16, 16, 8, [1], [1], [0])
    sd = built_lib.scan_desc([0], [0], [0])
    tabs = built_lib.htables([None] * 4)
    assert built_lib.read_scan(fd, sd, tabs, tabs, b"\x00\xff\xd9")[0] == built_lib.RET_FAILURE_LOGIC_ERROR
    assert built_lib.decode_pixels(fd, np.ones((4, 64), np.uint16), [np.zeros((4, 64), np.int32)])[0] == built_lib.RET_FAILURE_LOGIC_ERROR
    assert built_lib.encode_pixels(fd, np.ones((4, 64), np.uint16), np.zeros((16, 16), np.uint8))[0] == built_lib.RET_FAILURE_LOGIC_ERROR
    assert built_lib.write_scan(fd, sd, tabs, tabs, 1, [np.zeros((4, 64), np.int32)])[0] == built_lib.RET_FAILURE_LOGIC_ERROR
    # the drop-in entry points refuse too (no silent CPU path)
    assert L.jpeg_hip_process_jpeg_file(b"/nonexistent.jpg", None) == 0x1002  # RET_FAILURE_FILE_OPEN first


def test_header_is_plain_c99_and_links(built_lib, tmp_path):
    """include/jpeg_amd.h through a C compiler in strict C99 mode, linked against the library, run on the host:
    what a maintainer of the reference (C99, no C++) does first."""
    import shutil
    import subprocess
    if shutil.which("gcc") is None:
        pytest.skip("no gcc here")
    exe = str(tmp_path / "info")
    libdir = os.path.dirname(built_lib.LIB_PATH)
    subprocess.run(["gcc", "-std=c99", "-pedantic", "-Wall", "-Wextra", "-Werror", "-I" + os.path.join(ROOT, "include"),
                    os.path.join(ROOT, "examples", "info.c"), "-L" + libdir, "-ljpeg_amd", "-Wl,-rpath," + libdir,
                    "-o", exe], check=True, timeout=120)
    r = subprocess.run([exe, os.path.join(ROOT, "tests", "golden", "d_pil_422_jfif.jpg")], capture_output=True,
                       text=True, timeout=60)
    assert r.returncode == 0 and "112x80 P=8 Nf=3" in r.stdout and "[id 1 2x1 Tq 0]" in r.stdout
    r = subprocess.run([exe, os.path.join(ROOT, "tests", "golden", "d_pil_progressive.jpg")], capture_output=True,
                       text=True, timeout=60)
    assert r.returncode == 1 and "RET 0x1001" in r.stdout


def test_abi_version_matches_header_and_binding(built_lib):
    """jpeg_hip_params changed layout without a version bump.  The header, the library and the Python
    binding carry one ABI number; binding.lib() refuses a library with another."""
    hdr = open(os.path.join(ROOT, "include", "jpeg_amd.h")).read()
    want = int(re.search(r"#define JPEG_HIP_ABI_VERSION (\d+)", hdr).group(1))
    L = C.CDLL(built_lib.LIB_PATH)
    assert L.jpeg_hip_abi_version() == want == built_lib.ABI_VERSION
    L.jpeg_hip_version.restype = C.c_char_p
    assert ("ABI %d" % want).encode() in L.jpeg_hip_version()
    # struct params of the reference (encoder.c:68-76): H, V, q, optimize and nothing else
    assert C.sizeof(built_lib.Params) == 12
