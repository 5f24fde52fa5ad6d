"""No fused multiply-add may slip into the float pipelines: the reference is
built -std=c99 (no contraction), so every product and every sum of the DCT, the colour maths and the box mean
rounds on its own.  Checked without a GPU, on what hipcc makes of jpeg_amd/csrc/pixels.hip for gfx950:

  * LLVM IR: no llvm.fmuladd, no `contract` / `fast` / `reassoc` flag on any floating-point instruction -- the
    back end may then fuse nothing by itself; the only fma calls are the ones the source writes out
    (quant_reciprocal / quant_divide: the steps of the correctly rounded division, which ARE fused in hipcc's
    own expansion of `/`);
  * ISA: the decode kernels for one and three components hold no fused instruction at all; the four-component
    kernel only inside the expansion of its three divisions (frame.c:208-210); no decode kernel fuses in binary64;
    every encode kernel holds exactly as many FMAs, binary32 and binary64, as its IR has fma calls (binary64: the two
    products with 0.5 of frame_to_ycc, frame.c:169-170 -- exact products, so fused or not is the same value; the
    kernels for 8-bit samples at precision 8 hold no binary64 instruction: their frame_to_ycc is ycc8.inc, whose
    fused steps are exact integer sums and the one rounding of the division, tests/test_ycc8_cpu.py)."""
import os
import re
import shutil
import subprocess

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
HIPCC = "/opt/rocm/bin/hipcc"
FLAGS = ["--offload-arch=gfx950", "-O3", "-std=c++17", "-ffp-contract=off", "-I" + os.path.join(ROOT, "include"),
         "--cuda-device-only", "-S", "-w"]
FUSED32 = re.compile(r"^\s*(v_fma_f32|v_fmac_f32|v_fmaak_f32|v_fmamk_f32|v_mad_f32|v_mac_f32|v_pk_fma_f32|v_fma_mix|v_dot)")
FUSED64 = re.compile(r"^\s*(v_fma_f64|v_fmac_f64)")


@pytest.fixture(scope="module")
def built(tmp_path_factory):
    if not os.path.exists(HIPCC) or shutil.which("make") is None:
        pytest.skip("no hipcc here")
    with open(os.path.join(ROOT, "Makefile")) as f:
        assert "-ffp-contract=off" in f.read(), "the library must be built without contraction"
    d = tmp_path_factory.mktemp("isa")
    src = os.path.join(ROOT, "jpeg_amd", "csrc", "pixels.hip")
    subprocess.run([HIPCC] + FLAGS + ["-o", str(d / "pixels.s"), src], check=True, timeout=600)
    subprocess.run([HIPCC] + FLAGS + ["-emit-llvm", "-o", str(d / "pixels.ll"), src], check=True, timeout=600)
    return (d / "pixels.s").read_text(), (d / "pixels.ll").read_text()


def kernels_isa(asm):
    """{symbol: [instruction lines]} for every kernel of the file"""
    out = {}
    for m in re.finditer(r"^(_Z\w*k_(?:decode|encode)_pixels\w+):\s*;.*?$(.*?)^\s*s_endpgm", asm, re.S | re.M):
        out[m.group(1)] = m.group(2).split("\n")
    return out


def kernels_ir(ir):
    out = {}
    for m in re.finditer(r"^define [^\n]*@(_Z\w*k_(?:decode|encode)_pixels\w+)\([^\n]*\{$(.*?)^\}", ir, re.S | re.M):
        out[m.group(1)] = m.group(2)
    return out


def test_ir_allows_no_contraction(built):
    _, ir = built
    assert "llvm.fmuladd" not in ir
    for line in ir.split("\n"):
        if re.search(r"= (fmul|fadd|fsub|fdiv|fneg)\b", line):
            assert not re.search(r"\b(contract|fast|reassoc)\b", line), line
    funcs = kernels_ir(ir)
    assert len(funcs) >= 10
    for name, body in funcs.items():
        if "k_decode_pixels" in name:
            assert "llvm.fma" not in body, name        # inlined helpers included


def test_isa_fuses_only_what_the_source_fuses(built):
    asm, ir = built
    isa, funcs = kernels_isa(asm), kernels_ir(ir)
    assert len(isa) >= 10 and set(isa) == set(funcs)
    for name, lines in isa.items():
        n32 = sum((2 if m.group(1) == "v_pk_fma_f32" else 1) for m in (FUSED32.match(l) for l in lines) if m)
        n64 = sum(1 for l in lines if FUSED64.match(l))
        ndiv = sum(1 for l in lines if re.match(r"^\s*v_div_fmas_f32\b", l))
        if "k_decode_pixels" in name:
            if "ILi4E" in name:                        # YCCK: three divisions per pixel, FMAs only inside their expansion
                assert ndiv > 0 and n32 <= 5 * ndiv, (name, n32, ndiv)
            else:
                assert n32 == 0 and ndiv == 0, (name, n32)
            assert n64 == 0, name
        else:
            body = funcs[name]
            # explicit fmaf() calls of the source, as scalars or gathered into vectors by the SLP vectoriser
            want = len(re.findall(r"call[^\n]*@llvm\.fma\.f32", body)) + \
                sum(int(n) for n in re.findall(r"call[^\n]*@llvm\.fma\.v(\d+)f32", body))
            assert want > 0 and n32 == want and ndiv == 0, (name, n32, want)
            want64 = len(re.findall(r"call[^\n]*@llvm\.fma\.f64", body)) + \
                sum(int(n) for n in re.findall(r"call[^\n]*@llvm\.fma\.v(\d+)f64", body))
            y8 = "Lb1EE" in name                        # precision 8: frame_to_ycc in binary32 (ycc8.inc), no binary64 at all
            sy = int(re.search(r"ILi\dELi\dELi(\d)E", name).group(1))
            assert n64 == want64 and want64 == (0 if "ILi1E" in name or y8 else 2 * 4 * sy * 2), \
                (name, n64, want64)                     # two per pixel, four pixels per row, one or two rows, whole and edge path
            if y8:
                assert not any(re.search(r"_f64\b", l) for l in lines), name
