"""Inline-asm MFMAs are invisible to hipcc's hazard recogniser: a compiler-generated VALU write into a
register an asm-issued MFMA is still reading goes unpadded.  scripts/audit_asm_mfma.py scans the ISA for exactly that;
here it is checked on the pattern that broke the stride-2 data gradient's asm form, and run over the ISA of every
shipped source file that issues MFMAs as asm.  CPU only (hipcc cross-compiles)."""
import os
import shutil
import subprocess
import sys
import tempfile
from concurrent.futures import ThreadPoolExecutor

import pytest

from conftest import ROOT

sys.path.insert(0, os.path.join(ROOT, "scripts"))
HIPCC = shutil.which("hipcc") or "/opt/rocm/bin/hipcc"

BROKEN = """
kern_bad:
	;;#ASMSTART
	v_mfma_f32_16x16x32_bf16 v[78:81], v[218:221], v[42:45], v[78:81]
	;;#ASMEND
	v_mov_b64_e32 v[42:43], v[110:111]
	;;#ASMSTART
	v_mfma_f32_16x16x32_bf16 v[42:45], v[46:49], v[222:225], v[42:45]
	;;#ASMEND
kern_padded:
	;;#ASMSTART
	v_mfma_f32_16x16x32_bf16 v[78:81], v[218:221], v[42:45], v[78:81]
	;;#ASMEND
	s_nop 4
	v_mov_b64_e32 v[42:43], v[110:111]
kern_branch:
	;;#ASMSTART
	v_mfma_f32_16x16x32_bf16 v[78:81], v[218:221], v[42:45], v[78:81]
	;;#ASMEND
	s_branch .LBB0_2
.LBB0_1:
	v_mov_b64_e32 v[42:43], v[110:111]
.LBB0_2:
	v_mov_b64_e32 v[218:219], v[110:111]
kern_behind_an_mfma:
	;;#ASMSTART
	v_mfma_f32_16x16x32_bf16 v[78:81], v[218:221], v[42:45], v[78:81]
	;;#ASMEND
	;;#ASMSTART
	v_mfma_f32_16x16x32_bf16 v[82:85], v[218:221], v[46:49], v[82:85]
	;;#ASMEND
	v_mov_b64_e32 v[42:43], v[110:111]
kern_result_copied_too_early:
	;;#ASMSTART
	v_mfma_f32_16x16x32_bf16 v[76:79], v[140:143], v[124:127], v[76:79]
	s_nop 4
	;;#ASMEND
	v_mov_b64_e32 v[138:139], v[78:79]
kern_result_copied_in_time:
	;;#ASMSTART
	v_mfma_f32_16x16x32_bf16 v[76:79], v[140:143], v[124:127], v[76:79]
	s_nop 7
	;;#ASMEND
	v_mov_b64_e32 v[138:139], v[78:79]
"""


def test_auditor_on_the_pattern_that_broke_the_stride2_data_gradient(tmp_path, capsys):
    import audit_asm_mfma as au
    p = tmp_path / "x.s"
    p.write_text(BROKEN)
    # kern_bad (B overwritten behind its MFMA), kern_branch's target (A overwritten), and -- round 4, conv_x3's data gradient
    # at 256 registers -- the allocator's copy of an accumulator five wait states behind the MFMA that writes it (the
    # upper half of the fragment arrived stale; eight wait states are enough)
    assert au.main(str(p)) == 3
    out = capsys.readouterr().out
    assert "kern_bad: 2 asm MFMAs, 1 VALU" in out and "kern_padded: 1 asm MFMAs, 0 VALU" in out
    assert "kern_branch: 1 asm MFMAs, 1 VALU" in out and "kern_behind_an_mfma: 2 asm MFMAs, 0 VALU" in out
    assert "kern_result_copied_too_early: 1 asm MFMAs, 1 VALU" in out and "kern_result_copied_in_time: 1 asm MFMAs, 0 VALU" in out


@pytest.mark.skipif(not os.path.exists(HIPCC), reason="hipcc not available")
def test_shipped_kernels_with_asm_mfmas_have_no_unpadded_source_overwrite():
    import audit_asm_mfma as au
    csrc = os.path.join(ROOT, "dusty-gan-v2_amd", "csrc")
    def issues_asm_mfmas(f):
        txt = open(os.path.join(csrc, f)).read()
        return ("asm volatile(" in txt and "v_mfma_" in txt) or "MfmaAsm<" in txt    # (MfmaAsm: the in-place forms of gemm_core.h)
    files = [f for f in sorted(os.listdir(csrc)) if f.endswith(".hip") and issues_asm_mfmas(f)]
    assert {"conv_direct.hip", "conv8.hip", "conv_wgrad_stream.hip", "conv_x3.hip"} <= set(files)
    assert files, "no source issues MFMAs as inline asm any more: drop this test"
    with tempfile.TemporaryDirectory() as d:
        def isa(f):
            out = os.path.join(d, f[:-4] + ".s")
            subprocess.run([HIPCC, "--offload-arch=gfx950", "-O3", "-std=c++17", "-I" + os.path.join(ROOT, "include"),
                            "--cuda-device-only", "-S", os.path.join(csrc, f), "-o", out], check=True,
                           stderr=subprocess.DEVNULL)
            return out
        with ThreadPoolExecutor(4) as ex:
            outs = list(ex.map(isa, files))
        for f, o in zip(files, outs):
            assert au.main(o) == 0, f
