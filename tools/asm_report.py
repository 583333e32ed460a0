"""Report on a hipcc -S listing of a search kernel: what sits among the k-steps (spills, register-file moves,
full vmcnt drains, lane spills of scalars), and every vector-memory wait / atomic after the first MFMA.
    python tools/asm_report.py /tmp/qsw.s cosine_topk_walk_kernel"""
import re, sys
asm = open(sys.argv[1]).read()
pat = sys.argv[2] if len(sys.argv) > 2 else "cosine_topk"
ks = re.findall(r"^(_ZN\w*%s\w*):[^\n]*\n(.*?)s_endpgm" % pat, asm, re.S | re.M)
for name, body in ks:
    lines = body.splitlines()
    mf = [i for i, l in enumerate(lines) if "v_mfma" in l]
    if not mf:
        continue
    loop = lines[mf[0]:mf[-1] + 1]
    cnt = lambda pred, ls: sum(bool(pred(l)) for l in ls)
    print(name)
    print("  lines %d  mfma %d  (first %d last %d)" % (len(lines), len(mf), mf[0], mf[-1]))
    print("  among the k-steps: scratch %d  accvgpr %d  vmcnt(0) %d  lane-spill %d  s_nop %d  v_ insts %d" % (
        cnt(lambda l: "scratch_" in l, loop), cnt(lambda l: "v_accvgpr" in l, loop),
        cnt(lambda l: re.search(r"vmcnt\(0\)", l), loop), cnt(lambda l: "v_writelane" in l or "v_readlane" in l, loop),
        cnt(lambda l: "s_nop" in l, loop), cnt(lambda l: re.match(r"\s+v_(?!mfma)", l), loop)))
    print("  whole kernel: scratch %d  vmcnt(0) %d  calls %d" % (
        cnt(lambda l: "scratch_" in l, lines), cnt(lambda l: re.search(r"vmcnt\(0\)", l), lines),
        cnt(lambda l: "s_swappc" in l, lines)))
    if "-v" in sys.argv:
        for i, l in enumerate(lines):
            if i > mf[0] and ("global_atomic" in l or "scratch_" in l or re.search(r"s_waitcnt.*vmcnt", l) or "global_store" in l or "global_load" in l):
                print("    %5d %s" % (i, l.strip()))
