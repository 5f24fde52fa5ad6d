"""SURVEY.md section 8b made real: the reference's OWN drivers on libjpeg_amd.so.

oracle/_ref/decoder_hip and oracle/_ref/encoder_hip are the unmodified decoder.c / encoder.c of the
reference (main(), parse_format() decoder.c:472-659, produce_codestream() encoder.c:589-645, every
marker parser and producer) linked with integration/ref_glue.c, which defines read_ecs (decoder.c:364),
epilogue (:456), prologue (encoder.c:176), write_ecs_dry (:525) and write_ecs (:560) with the
reference's exact signatures on top of the C-ABI of include/jpeg_amd.h.  oracle/_ref/decoder and
oracle/_ref/encoder are the same sources with the reference's own loops.  Same command lines, same
input files: the files both write must be byte-identical.

The programs are built in the build container (oracle/Makefile, needs /root/reference) and travel
with the tree; where they are absent the tests skip (they cannot be built on the GPU box)."""
import os
import subprocess

import pytest

from jpeg_amd.synth import synth_image
from oracle.pyoracle import pnm_bytes, HERE as ORACLE_DIR
from tests.conftest import golden_path

pytestmark = pytest.mark.gpu

REF = os.path.join(ORACLE_DIR, "_ref")
PROGS = {n: os.path.join(REF, n) for n in ("decoder", "encoder", "decoder_hip", "encoder_hip", "decoder_stages", "encoder_stages")}


@pytest.fixture(scope="module")
def progs():
    for p in PROGS.values():
        if not os.access(p, os.X_OK):
            pytest.skip("%s not built (oracle/Makefile needs /root/reference)" % p)
    from jpeg_amd import binding
    assert binding.lib().jpeg_hip_device_count() > 0, "no HIP device: the product path has no CPU fallback"
    return PROGS


def run(cmd, cwd):
    return subprocess.run(cmd, cwd=cwd, stdout=subprocess.PIPE, stderr=subprocess.STDOUT, timeout=300)


def read(path):
    with open(path, "rb") as f:
        return f.read()


def _manifest():
    import json
    with open(golden_path("manifest.json")) as f:
        return json.load(f)


ENC = sorted(_manifest()["encode"].items())
DEC = sorted(_manifest()["decode"].items())


@pytest.mark.parametrize("name,case", ENC, ids=[n for n, _ in ENC])
def test_encoder_and_decoder_programs(progs, tmp_path, name, case):
    """encoder_hip == encoder and decoder_hip == decoder on every encode golden (C1 .. C5 shapes included)."""
    img = synth_image(case["X"], case["Y"], case["ncomp"], case["P"], seed=case["seed"], range_safe=case["range_safe"])
    pnm = tmp_path / ("in.ppm" if case["ncomp"] == 3 else "in.pgm")
    pnm.write_bytes(pnm_bytes(img, case["P"]))
    args = ["-h", str(case["H"]), "-v", str(case["V"]), "-q", str(case["q"]), "-o", str(case["optimize"]), str(pnm)]
    r0 = run([progs["encoder"]] + args + ["ref.jpg"], tmp_path)
    r1 = run([progs["encoder_hip"]] + args + ["hip.jpg"], tmp_path)
    assert r0.returncode == 0 and r1.returncode == 0, r1.stdout[-2000:]
    assert b"Failure" not in r1.stdout, r1.stdout[-2000:]
    jpg = read(tmp_path / "ref.jpg")
    assert len(jpg) == case["jpg_len"] if "jpg_len" in case else True
    assert read(tmp_path / "hip.jpg") == jpg

    d0 = run([progs["decoder"], "ref.jpg", "ref.pnm"], tmp_path)
    d1 = run([progs["decoder_hip"], "ref.jpg", "hip.pnm"], tmp_path)
    assert d0.returncode == 0 and d1.returncode == 0, d1.stdout[-2000:]
    assert read(tmp_path / "hip.pnm") == read(tmp_path / "ref.pnm")

    # Stage by stage (SURVEY.md section 8b's table): ALL of the reference's drivers and loops, its Huffman coder included,
    # with dequantize / inverse_dct / frame_to_rgb and frame_to_ycc / forward_dct / quantize from the library under the
    # reference's own names (integration/ref_glue.c -DGLUE_STAGES): one GPU kernel per stage, the float buffers between
    # them the reference's own.
    r2 = run([progs["encoder_stages"]] + args + ["stages.jpg"], tmp_path)
    assert r2.returncode == 0 and b"Failure" not in r2.stdout, r2.stdout[-2000:]
    assert read(tmp_path / "stages.jpg") == jpg
    d2 = run([progs["decoder_stages"], "ref.jpg", "stages.pnm"], tmp_path)
    assert d2.returncode == 0, d2.stdout[-2000:]
    assert read(tmp_path / "stages.pnm") == read(tmp_path / "ref.pnm")


@pytest.mark.parametrize("name,case", DEC, ids=[n for n, _ in DEC])
def test_decoder_program_on_foreign_streams(progs, tmp_path, name, case):
    """Pillow-written files, restart markers, several scans, DHT-less Motion JPEG frames, CMYK, a progressive
    file (refused): same exit status, same output file (default name: frame.c:548-567)."""
    src = golden_path(case["jpg_file"])
    d0 = run([progs["decoder"], src], tmp_path / "a" if (tmp_path / "a").mkdir() is None else None)
    d1 = run([progs["decoder_hip"], src], tmp_path / "b" if (tmp_path / "b").mkdir() is None else None)
    assert d0.returncode == d1.returncode, d1.stdout[-2000:]
    assert (d0.returncode == 0) == (case["err"] == 0)
    outs0 = sorted(os.listdir(tmp_path / "a"))
    assert outs0 == sorted(os.listdir(tmp_path / "b"))
    for o in outs0:
        assert read(tmp_path / "a" / o) == read(tmp_path / "b" / o), o
    if case["err"] == 0:
        assert outs0 in (["output.ppm"], ["output.pgm"])
    # and stage by stage (CMYK: the four-component frame_to_rgb, frame.c:196-217)
    (tmp_path / "c").mkdir()
    d2 = run([progs["decoder_stages"], src], tmp_path / "c")
    assert d2.returncode == d0.returncode, d2.stdout[-2000:]
    assert sorted(os.listdir(tmp_path / "c")) == outs0
    for o in outs0:
        assert read(tmp_path / "a" / o) == read(tmp_path / "c" / o), o
