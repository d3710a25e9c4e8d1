#!/usr/bin/env python3
"""Static checker of a generated main-loop body (csrc/fa2_bwd_*_body.inc): executes the text of a body twice in a row
(steady state) with an in-order model of the LDS queue and verifies, for every MFMA,
  * each VGPR source that an LDS read delivers has been waited for (the counted lgkmcnt covers it), and
  * no LDS read overwrites a fragment register before every MFMA that consumes the previous fragment has issued;
for every VALU instruction that its VGPR sources are not the destination of an LDS read still in flight, that it does not
read an MFMA result before PASS_GAP later MFMAs have issued, and that an MFMA does not read a VGPR a VALU instruction
wrote fewer than 2 instructions before.  usage: check_body.py file.inc MACRO [MACRO_OF_THE_PRECEDING_BODY]"""
import re
import sys

PASS_GAP = 2


def body(text, name):
    m = re.search(r"#define " + name + r" \\\n(.*?)(?:\n\n|\Z)", text, flags=re.S)
    out = []
    for l in m.group(1).split("\n"):
        l = re.sub(r'^\s*"|\\n\\t" \\$|\\n\\t"$', "", l)
        out += l.split("\\n\\t")
    return out


def regs(tok):
    """Register numbers a token names: v<n> -> n, a<n> -> 1000 + n (accumulator registers matter where LDS reads deliver
    fragments into them: the fp8 forward's slots)."""
    m = re.match(r"([va])\[(\d+):(\d+)\]", tok)
    if m:
        base = 1000 if m.group(1) == "a" else 0
        return set(range(base + int(m.group(2)), base + int(m.group(3)) + 1))
    m = re.match(r"([va])(\d+)$", tok)
    return {(1000 if m.group(1) == "a" else 0) + int(m.group(2))} if m else set()


def check(lines, label):
    errs = []
    queue = []                 # outstanding LDS reads: (dst regs)
    pending = {}               # reg -> True while an LDS read into it is outstanding
    vm_pending = {}            # reg -> True while a buffer load into it is outstanding
    mfma_no = 0
    mfma_wrote = {}            # reg -> mfma index that last wrote it
    valu_wrote = {}            # reg -> instruction index
    last_read_by_mfma = {}     # reg -> mfma index of the last MFMA that read it
    frag_consumed = {}         # reg -> True once an MFMA has read the fragment currently in it
    for rep in (0, 1):
        for i, l in enumerate(lines):
            idx = rep * len(lines) + i
            toks = [t.strip(",") for t in l.split()]
            if not toks:
                continue
            op = toks[0]
            if op == "s_waitcnt":
                if "vmcnt" in l:               # every vmcnt wait in these bodies covers the body's own buffer loads
                    vm_pending.clear()
                m = re.search(r"lgkmcnt\((\d+)\)", l)
                if m:
                    n = int(m.group(1))
                    while len(queue) > n:
                        for r in queue.pop(0):
                            pending.pop(r, None)
            elif op.startswith("ds_read"):
                dst = regs(toks[1])
                for r in dst:
                    if rep == 1 and r in frag_consumed and not frag_consumed[r]:
                        errs.append(f"{label}: line {i}: {l}  overwrites a fragment no MFMA has consumed yet")
                    frag_consumed[r] = False
                    pending[r] = True
                queue.append(dst)
            elif op.startswith("ds_write"):
                queue.append(set())          # LDS writes share lgkmcnt with the reads and complete in order with them
                for t in toks[2:3]:
                    for r in regs(t):
                        if r in pending and rep == 1:
                            errs.append(f"{label}: line {i}: {l}  stores v{r} while an LDS read into it is in flight")
            elif op.startswith("buffer_store"):
                for r in regs(toks[1]):
                    if r in pending and rep == 1:
                        errs.append(f"{label}: line {i}: {l}  stores v{r} while an LDS read into it is in flight")
                    if r in vm_pending and rep == 1:
                        errs.append(f"{label}: line {i}: {l}  stores v{r} while a buffer load into it is in flight")
                    if r in mfma_wrote and mfma_no - 1 - mfma_wrote[r] < PASS_GAP and rep == 1:
                        errs.append(f"{label}: line {i}: {l}  stores v{r} only {mfma_no - 1 - mfma_wrote[r]} MFMAs after the MFMA that writes it")
            elif op.startswith("buffer_load"):
                for r in regs(toks[1]):
                    if r in last_read_by_mfma and mfma_no - 1 - last_read_by_mfma[r] < 1 and rep == 1:
                        errs.append(f"{label}: line {i}: {l}  loads into v{r} right behind the MFMA that reads it")
                    vm_pending[r] = True
            elif op.startswith("v_mfma"):
                dst, a, b, c = (regs(t) for t in toks[1:5])
                for r in a | b | c:
                    if r in pending and rep == 1:
                        errs.append(f"{label}: line {i}: {l}  reads v{r} before its LDS read is waited for")
                    if r in vm_pending and rep == 1:
                        errs.append(f"{label}: line {i}: {l}  reads v{r} before its buffer load is waited for")
                    if r in valu_wrote and idx - valu_wrote[r] < 2 and rep == 1:
                        errs.append(f"{label}: line {i}: {l}  reads v{r} {idx - valu_wrote[r]} instruction(s) after a VALU write")
                    frag_consumed[r] = True
                for r in dst:
                    mfma_wrote[r] = mfma_no
                mfma_no += 1
            elif op.startswith("v_"):
                srcs = set()
                for t in toks[2:]:
                    srcs |= regs(t.lstrip("-"))
                dst = regs(toks[1])
                for r in srcs | dst:
                    if r in pending and rep == 1:
                        errs.append(f"{label}: line {i}: {l}  touches v{r} while an LDS read into it is in flight")
                    if r in mfma_wrote and mfma_no - 1 - mfma_wrote[r] < PASS_GAP and rep == 1:
                        errs.append(f"{label}: line {i}: {l}  touches v{r} only {mfma_no - 1 - mfma_wrote[r]} MFMAs after the MFMA that writes it")
                for r in dst:
                    valu_wrote[r] = idx
    return errs


def main():
    text = open(sys.argv[1]).read()
    names = sys.argv[2:] or re.findall(r"#define (FA2_\w+_[CM]?BODY_\w+) ", text)
    bad = 0
    for n in names:
        e = check(body(text, n), n)
        bad += len(e)
        for x in e[:8]:
            print(x)
    print("checked", len(names), "bodies:", "OK" if not bad else f"{bad} problems")
    return 1 if bad else 0


if __name__ == "__main__":
    sys.exit(main())
