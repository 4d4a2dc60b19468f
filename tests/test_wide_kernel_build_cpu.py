"""Build-time checks of k_lmm_quadform_i8w, the LMM contraction whose accumulators, LDS reads and scalar state are inline assembly that hipcc does
not model (cdna_hip_programming.md 5.7: "audit after every edit").  No GPU needed: hipcc cross-compiles to gfx950 assembly.

* the generated sub-step statements are what tools/gen_wide_substeps.py writes today (nobody edited one side only);
* the kernel's assembly holds no scratch access, no compiler-generated v_accvgpr_* (the accumulators a[0:255] are the statements' own) and no
  compiler use of the scalar registers s84 .. s99 the statements keep their ring / cursor state in."""
import os
import re
import shutil
import subprocess
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
CSRC = os.path.join(ROOT, "pyseer_amd", "csrc")
HIPCC = "/opt/rocm/bin/hipcc"


def test_generated_substeps_are_in_sync(tmp_path):
    gen = os.path.join(ROOT, "tools", "gen_wide_substeps.py")
    src = open(gen).read().replace('os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", "pyseer_amd", "csrc", "lmm_quadform_wide_substeps.inc")',
                                   repr(str(tmp_path / "out.inc")))
    assert repr(str(tmp_path / "out.inc")) in src, "the generator's output path moved: update this test"
    script = tmp_path / "gen.py"
    script.write_text(src)
    subprocess.check_call([sys.executable, str(script)], stdout=subprocess.DEVNULL)
    assert (tmp_path / "out.inc").read_text() == open(os.path.join(CSRC, "lmm_quadform_wide_substeps.inc")).read(), \
        "pyseer_amd/csrc/lmm_quadform_wide_substeps.inc is stale: run python tools/gen_wide_substeps.py"


@pytest.mark.skipif(not os.path.exists(HIPCC), reason="hipcc not installed")
def test_wide_kernel_assembly_audit(tmp_path):
    out = tmp_path / "lmm.s"
    subprocess.check_call([HIPCC, "-O3", "-std=c++17", "-fPIC", "--offload-arch=gfx950", "-ffp-contract=off", "-S", "--cuda-device-only",
                           "-o", str(out), "lmm_kernels.hip"], cwd=CSRC, stderr=subprocess.DEVNULL)
    text = out.read_text()
    m = re.search(r"^(_Z18k_lmm_quadform_i8wILi0ELi0ELi0EE\w*):[^\n]*\n(.*?)\.end_amdhsa_kernel", text, re.S | re.M)
    assert m, "k_lmm_quadform_i8w<0, 0, 0> not found in the assembly"
    body = m.group(2)
    assert "scratch_" not in body and re.search(r"\.amdhsa_private_segment_fixed_size 0\b", body), "the kernel spills"
    assert body.count("v_mfma_i32_32x32x32_i8") == 2 * 64 + 2 * 32 + 2 * 16, "stage bodies: {C = 0, accumulate} x {4, 2, 1 row sub-tiles}"
    outside, inside = [], False                         # the compiler's own instructions = everything outside ;;#ASMSTART .. ;;#ASMEND
    for ln in body.splitlines():
        if "#ASMSTART" in ln:
            inside = True
        elif "#ASMEND" in ln:
            inside = False
        elif not inside:
            outside.append(ln)
    comp = "\n".join(outside)
    assert "v_accvgpr" not in comp, "hipcc touched the accumulator registers"
    assert not re.search(r"\bs(8[4-9]|9[0-9])\b", comp), "hipcc uses a scalar register the asm statements keep state in"
    assert "s_waitcnt vmcnt" not in comp, "hipcc waits for the hand-counted LDS-DMAs"
