"""Static check of the scalar-cache touches (pass32_pk_f16, csrc/glm_kernels.hip): the touches are s_load_dword instructions into a register
nobody reads, issued without a wait -- so the register they target must stay out of every other use for the rest of the kernel (a load in
flight would overwrite whatever the compiler put there).  Reads the gfx950 assembly of the translation unit (hipcc -save-temps) and, for
every kernel that holds touches, checks that (a) they all target ONE scalar register and (b) no other instruction writes it between the first touch and the
first full scalar wait (s_waitcnt lgkmcnt(0)) behind the last one -- the window in which a touch can be in flight.  usage: python tools/check_touch_regs.py <glm_kernels-hip-amdgcn-amd-amdhsa-gfx950.s>"""
import re, sys
src = open(sys.argv[1]).read()
bad = 0; seen = 0
for m in re.finditer(r"^(_Z\w+):\s*;.*?\n(.*?)\.end_amdhsa_kernel", src, re.S | re.M):
    name, body = m.group(1), m.group(2)
    lines = body.split("\n")
    touch_regs = set(); touch_lines = set(); in_asm = False
    for i, ln in enumerate(lines):
        if "#ASMSTART" in ln: in_asm = True
        elif "#ASMEND" in ln: in_asm = False
        elif in_asm:
            t = re.match(r"\s*s_load_dword\s+s(\d+),", ln)
            if t: touch_regs.add(int(t.group(1))); touch_lines.add(i)
    if not touch_regs: continue
    seen += 1
    if len(touch_regs) != 1:
        print("FAIL %s: touches target several registers %s" % (name, sorted(touch_regs))); bad += 1; continue
    r = touch_regs.pop()
    # the window in which a touch can be in flight: from the first touch to the first full scalar wait behind the last one
    lo = min(touch_lines); hi = max(touch_lines)
    while hi < len(lines) - 1 and not re.search(r"s_waitcnt\s+(vmcnt\(\d+\)\s+)?lgkmcnt\(0\)", lines[hi]): hi += 1
    writers = []
    for i, ln in enumerate(lines):
        if i in touch_lines or i < lo or i > hi: continue
        t = re.match(r"\s*(s_\w+|v_readfirstlane_b32|v_readlane_b32|v_cmp\w*|v_cmpx\w*)\s+([^,\s]+)", ln)
        if not t: continue
        op, dst = t.group(1), t.group(2)
        if op.startswith(("s_cmp", "s_cbranch", "s_branch", "s_waitcnt", "s_nop", "s_barrier", "s_endpgm", "s_sleep", "s_setprio", "s_bitcmp", "s_setreg", "s_dcache", "s_icache", "s_sendmsg", "s_code_end", "s_trap", "s_store")): continue
        d = re.match(r"s(\d+)$", dst); dr = re.match(r"s\[(\d+):(\d+)\]$", dst)
        hit = (d and int(d.group(1)) == r) or (dr and int(dr.group(1)) <= r <= int(dr.group(2)))
        if hit: writers.append((i, ln.strip()))
    others = writers
    if others:
        print("FAIL %s: s%d (the touches' register) is also written by: %s" % (name, r, others[:3])); bad += 1
print("%d kernels with touches checked, %d failures" % (seen, bad))
sys.exit(1 if bad else 0)
