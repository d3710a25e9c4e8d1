#!/usr/bin/env python3
"""Instruction mix of a kernel's main loop from hipcc -S output.

A wave that is alone on its SIMD issues at most one instruction every ~7 clocks (tools/probes/
issue_cost.hip: s_nop, s_waitcnt, VALU alike; v_exp_f32 ~10, v_accvgpr_* ~10.5), while a
v_mfma_f32_32x32x16_bf16 occupies the matrix pipe for ~32: a loop with more than ~4 instructions
per MFMA is issue-bound.  This prints the mix of the biggest loop (label .. backward branch) and
the issue-time estimate beside the matrix-pipe time.

usage: isa_mix.py file.s kernel_name_substring [--all-blocks]
"""
import re, sys, collections

def main():
    path, key = sys.argv[1], sys.argv[2]
    lines = open(path).read().split("\n")
    start = next(i for i, l in enumerate(lines) if l.startswith("_Z") and key in l and l.rstrip().endswith(":") or (l.startswith("_Z") and key in l and ": " in l and "@" in l))
    end = next(i for i in range(start, len(lines)) if "s_endpgm" in lines[i])
    body = lines[start:end]
    # the hot loop: the label .. last backward branch span holding the most MFMAs (smallest such span)
    cands = []
    for i, l in enumerate(body):
        m = re.match(r"^(\.LBB\d+_\d+):", l)
        if not m:
            continue
        lab = m.group(1)
        tails = [j for j in range(i, len(body)) if re.match(r"\s+s_c?branch\S*\s+" + re.escape(lab) + r"\b", body[j])]
        if not tails:
            continue
        nm = sum("v_mfma" in t for t in body[i:tails[-1]])
        cands.append((nm, tails[-1] - i, i, tails[-1]))
    top = max(c[0] for c in cands)
    best = min((c for c in cands if c[0] * 2 >= top), key=lambda c: c[1])[2:]
    lo, hi = best
    # cold blocks (label .. next label): the rare O rescale (v_accvgpr traffic) and the masked-tile path
    # (bursts of v_cndmask) are left out unless --all-blocks is given
    blocks, cur = [], [lo]
    for i in range(lo + 1, hi + 1):
        if re.match(r"^(\.LBB\d+_\d+):", body[i]) or body[i].startswith("; %bb."):
            blocks.append((cur[0], i)); cur = [i]
    blocks.append((cur[0], hi + 1))
    keep = []
    for a, b in blocks:
        txt = body[a:b]
        cold = sum("v_accvgpr" in t for t in txt) > 8 or sum("v_cndmask" in t for t in txt) > 12 or any("fa2-cold" in t for t in txt)
        if cold and "--all-blocks" not in sys.argv:
            continue
        keep += txt
    cat = collections.Counter()
    ops = collections.Counter()
    for l in keep:
        l = l.strip()
        if not l or l.startswith(";") or l.startswith(".") or l.endswith(":"):
            continue
        op = l.split()[0]
        ops[op] += 1
        if op.startswith("v_mfma"): c = "mfma"
        elif op.startswith("ds_"): c = "lds"
        elif op.startswith("buffer_") or op.startswith("global_") or op.startswith("scratch_"): c = "vmem"
        elif op in ("v_exp_f32_e32", "v_exp_f32", "v_log_f32_e32", "v_rcp_f32_e32"): c = "trans"
        elif op.startswith("v_accvgpr"): c = "accvgpr"
        elif op.startswith("s_waitcnt"): c = "s_waitcnt"
        elif op.startswith("s_nop"): c = "s_nop"
        elif op.startswith("s_barrier"): c = "s_barrier"
        elif op.startswith("s_"): c = "salu"
        elif op.startswith("v_"): c = "valu"
        else: c = "other"
        cat[c] += 1
    n = sum(cat.values())
    mf = max(cat["mfma"], 1)
    cost = {"trans": 10.0, "accvgpr": 10.5}
    issue = sum(v * cost.get(k, 7.0) for k, v in cat.items())
    print("loop lines %d..%d: %d instructions, %d MFMA -> %.2f instructions per MFMA" % (lo, hi, n, cat["mfma"], n / mf))
    for k, v in cat.most_common():
        print("  %-10s %5d  (%.2f per MFMA)" % (k, v, v / mf))
    print("issue estimate %.0f clocks vs matrix pipe %.0f (34/MFMA alone on the SIMD): %.2fx" % (issue, 34.0 * mf, issue / (34.0 * mf)))
    if "--ops" in sys.argv:
        for k, v in ops.most_common(40):
            print("    %-28s %d" % (k, v))

if __name__ == "__main__":
    main()
