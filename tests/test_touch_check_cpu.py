"""tools/check_touch_regs.py (the static check behind `make check-touch`, csrc/Makefile): the scalar-cache touches of pass32_pk_f16 load into a
register nobody reads without waiting, so nothing else may write that register while a touch can be in flight.  The checker is run on two
hand-made kernels: one where the register is left alone inside the window, one where a record load lands on it."""
import os, subprocess, sys, textwrap

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
TOOL = os.path.join(ROOT, "tools", "check_touch_regs.py")

KERNEL = """\
_Z6k_{name}v:                              ; @_Z6k_{name}v
\ts_mov_b32 s3, 0
\ts_lshl_b32 s3, s4, 2
.LBB0_1:
\ts_waitcnt lgkmcnt(0)
\ts_load_dwordx16 s[36:51], s[64:65], 0x0
\t;;#ASMSTART
\ts_load_dword s3, s[30:31], 0
\t;;#ASMEND
\t;;#ASMSTART
\ts_load_dword s3, s[30:31], 64
\t;;#ASMEND
{inside}
\tv_pk_fma_f32 v[2:3], v[4:5], s[36:37], v[2:3]
\ts_waitcnt lgkmcnt(0)
\ts_cbranch_scc0 .LBB0_1
\ts_mov_b32 s3, s9
\ts_endpgm
\t.end_amdhsa_kernel
"""


def run(tmp_path, name, inside):
    p = tmp_path / (name + ".s")
    p.write_text(KERNEL.format(name=name, inside=inside))
    return subprocess.run([sys.executable, TOOL, str(p)], capture_output=True, text=True)


def test_a_register_left_alone_passes(tmp_path):
    r = run(tmp_path, "ok", "\ts_add_i32 s5, s5, 1")
    assert r.returncode == 0, r.stdout + r.stderr
    assert "1 kernels with touches checked, 0 failures" in r.stdout


def test_a_write_inside_the_window_is_reported(tmp_path):
    for inside in ("\ts_load_dwordx4 s[0:3], s[12:13], 0x8", "\tv_readlane_b32 s3, v254, 52", "\tv_cmp_gt_u32_e64 s[2:3], 32, v0"):
        r = run(tmp_path, "bad", inside)
        assert r.returncode == 1, (inside, r.stdout)
        assert "FAIL" in r.stdout and "s3" in r.stdout


def test_writes_outside_the_window_do_not_count(tmp_path):
    # the prologue's and the epilogue's uses of s3 in KERNEL (before the first touch, behind the closing wait) are not failures
    r = run(tmp_path, "edge", "\ts_nop 0")
    assert r.returncode == 0, r.stdout


CSRC = os.path.join(ROOT, "pyseer_amd", "csrc")


def test_the_library_is_linked_only_behind_the_check_of_its_own_assembly():
    """The product's own device assembly, not a hand-made kernel: `make` keeps glm_kernels.gfx950.s (same compiler, same flags as
    glm_kernels.o) and links ../libseerhip.so only once tools/check_touch_regs.py passed on it (Makefile: glm_kernels.touch_ok is a
    prerequisite of the library).  Here the checker runs again on that file -- rebuilt first if a source is newer -- and must see every
    instantiation that holds touches (Q = 1 .. 10 of the two packed passes and their fused-epilogue variants)."""
    mk = open(os.path.join(CSRC, "Makefile")).read()
    assert "../libseerhip.so: $(OBJS) glm_kernels.touch_ok" in mk, "the touch check no longer gates the link"
    r = subprocess.run(["make", "-C", CSRC, "glm_kernels.touch_ok"], capture_output=True, text=True)      # a no-op when up to date (~3 min otherwise)
    assert r.returncode == 0, r.stdout[-2000:] + r.stderr[-2000:]
    s = os.path.join(CSRC, "glm_kernels.gfx950.s")
    assert os.path.getmtime(s) >= os.path.getmtime(os.path.join(CSRC, "glm_kernels.hip"))
    r = subprocess.run([sys.executable, TOOL, s], capture_output=True, text=True)
    assert r.returncode == 0, r.stdout
    n = int(r.stdout.split(" kernels with touches checked")[0].split()[-1])
    assert n >= 40, "only %d kernels with touches found in the built assembly: the check lost sight of the passes" % n
    assert "0 failures" in r.stdout
