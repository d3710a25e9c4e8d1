#!/usr/bin/env python3
"""Generates the main-loop bodies of the fp8 (e4m3) forward -> csrc/fa2_fwd_fp8_body.inc.

Work split (fa2_fwd_fp8.hip): workgroup = 8 waves = 256 query rows of one head, two waves per SIMD (128 + 128 registers), a
wave owns 32 rows; d = 128.  One body = 64 keys (two 32-key blocks blk = 0, 1) for one wave; v_mfma_f32_32x32x64_f8f6f4
contracts 64 values per instruction, so

    A   S^T(j)[blk] = K(j, blk) Q^T                      2 x 2 MFMAs      K fragments: 2 ds_read_b128 each, into AGPR slots
    P   O^T[dt]    += V^T(j-2)[dt] P^T(j-2)              4 MFMAs          V^T fragments: 2 ds_read_b128 each
    VALU beside both: the softmax of the 64 keys j-1 (fma, exp, row sum, pack to e4m3: 14 instructions per 4 elements) spread
    over the whole body, and the lane maxima of the keys j behind their A chains; the body ends with the compare "does any
    row need a new softmax reference" (returned in an SGPR; the rare update is compiler code between two bodies).

This is the structure of the bf16 forward's bodies (tools/gen_fwd_body.py: a pipeline two bodies deep, every LDS read issued
ahead of its use behind a counted lgkmcnt, the next tile's LDS-DMA issued from inside the bodies, state in registers the
bodies name) with the fp8 kernel's operand layout: K rows are fed in the permuted order pi (fa2_fwd_fp8.hip) so that register r
of lane-half h of an S^T accumulator is key 16 h + r of its block -- packed four to a register, the 16 registers of the two
blocks ARE the B operand of the second product, and its A operand is two plain 16-byte reads of a V^T image.

LDS: a ring of FOUR K tiles (128 keys x 128 B), then a ring of four V^T tiles (two halves of [128 d][64 keys]); 16 KiB per
tile.  Body (tile in buffer b, half kb) reads K rows 64 kb .. of buffer b and V^T half kb of the PREVIOUS tile's buffer
(keys j - 2).  The body of a tile's second half starts with `s_waitcnt vmcnt(0); s_barrier` and issues the LDS-DMA of tile
t + 2 itself (two pieces of K, two of V^T per wave), as fillers.

Registers (kernel compiled with amdgpu_num_vgpr(32): hipcc owns v0..v31):
    a[0:64)    O^T tiles dt;   a[64:80)  Q fragments s = 0, 1 (8 each);   a[80:128)  six fragment slots of 8
    v[32:96)   S sets: S^T(parity, blk), 16 registers each;   v[96:112)  packed P: PF[parity], 8 registers (word 4 blk + w =
               keys 4 w .. 4 w + 3 of the lane's 16 of block blk)
    v[112:116) KA[s][i]: K row-read addresses;  v[116:118)  VA[i]: V^T row-read addresses;  then the softmax STATE:
               la lb (partial row sums), rm (lane maximum of the keys j), mb (reference * log2 e), th (raw-score threshold)
Operands: %[c2] (s), %[need] (=s), masked variant %[hi] (v: first masked key of the lane's row, relative to the body and to
the lane's half), %[ninf] (v); DMA bodies %[mw] (s: LDS byte address of the wave's first K piece), %[dvk], %[dvv] (v: per-lane
source offsets), %[krs], %[vrs] (s x4), %[kso], %[vso] (s: byte offsets of the wave's first pieces of tile t + 2).
"""
import argparse
import os
import re

import gen_dkdv_body as base
from gen_dkdv_body import Task

READ_AHEAD = int(os.environ.get("FA2_GEN_F8_READ_AHEAD", "4"))
READ_LATEST = int(os.environ.get("FA2_GEN_F8_READ_LATEST", "2"))
BUDGET = int(os.environ.get("FA2_GEN_F8_BUDGET", "96"))
ORDER = int(os.environ.get("FA2_GEN_F8_ORDER", "0"))       # 0: A A A A P P P P; 1: A A P P A A P P
WIN = int(os.environ.get("FA2_GEN_F8_WIN", "1"))           # half-width (gaps) of a softmax quad's window
NBUF = 4
D = 128
ROWB = 128                    # bytes per K row
KV = 128                      # keys per tile
NH = 2                        # bodies per tile
TILEB = KV * ROWB             # 16 KiB (K tile; the V^T tile is two halves of 128 x 64 B)
HALFV = 128 * 64
KS, DT = 2, 4
NS = 2 * KS + DT              # 8 MFMAs per body
NSLOT = 6
WAVES = 8
COST = {"lds": 4, "valu": 4, "exp": 8, "cvt": 4, "cmp": 8, "vmem": 12}
MFMA = "v_mfma_f32_32x32x64_f8f6f4"

V0 = 32
SET = [V0, V0 + 32]
PF = [V0 + 64, V0 + 72]
KA = V0 + 80                  # 112..115
VA = KA + 4                   # 116, 117
STATE = VA + 2                # 118: la lb rm mb th
VEND = STATE + 5
assert VEND <= 128
A_O, A_QF, A_SLOT = 0, 64, 80


def sset(par, blk): b = SET[par] + 16 * blk; return f"v[{b}:{b + 15}]"
def sreg(par, blk, r): return f"v{SET[par] + 16 * blk + r}"
def pf(par): b = PF[par]; return f"v[{b}:{b + 7}]"
def pfw(par, blk, w): return f"v{PF[par] + 4 * blk + w}"
def slot(i): b = A_SLOT + 8 * i; return f"a[{b}:{b + 7}]"
def slot_lo(i): b = A_SLOT + 8 * i; return f"a[{b}:{b + 3}]"
def slot_hi(i): b = A_SLOT + 8 * i + 4; return f"a[{b}:{b + 3}]"
def ka(s, i): return f"v{KA + 2 * s + i}"
def va(i): return f"v{VA + i}"
def l(e): return f"v{STATE + e}"
RM, MB, TH = f"v{STATE + 2}", f"v{STATE + 3}", f"v{STATE + 4}"
def qf(s): b = A_QF + 8 * s; return f"a[{b}:{b + 7}]"
def o(dt): b = A_O + 16 * dt; return f"a[{b}:{b + 15}]"


def build(par, masked, dma, nomax=False):
    """One body of parity `par`.  Gap units 0 .. NS-1; tasks with a negative release belong to the tail of the previous body
    (emitted there with the NEXT body's bases: '@N')."""
    # gap of A product (blk, s) and of P product dt
    if ORDER == 0:
        gA = lambda blk, s: KS * blk + s
        gPd = lambda dt: 2 * KS + dt
    else:
        gA = lambda blk, s: 4 * blk + s
        gPd = lambda dt: 2 + 4 * (dt // 2) + dt % 2
    mfma = [None] * NS
    tasks = []
    ctr = [0]
    busy = [-(10 ** 6)] * NSLOT

    def take(last):
        i = ctr[0] % NSLOT
        ctr[0] += 1
        f = busy[i]
        busy[i] = last
        return i, f

    def rd(text, key, consume, free_after):
        rel = max(consume - READ_AHEAD, free_after + 1)
        tasks.append(Task(text, COST["lds"], rel, max(consume - READ_LATEST, rel), "lds", key))

    def allocate(rec):
        for g in range(NS):
            for blk in (0, 1):
                for s in range(KS):
                    if gA(blk, s) == g:
                        alloc_a(rec, blk, s, g)
            for dt in range(DT):
                if gPd(dt) == g:
                    alloc_p(rec, dt, g)

    def alloc_a(rec, blk, s, g):
        if True:
            if True:
                sk, fk = take(g)
                if rec:
                    k0, k1 = ("K", blk, s, 0), ("K", blk, s, 1)
                    rd(f"ds_read_b128 {slot_lo(sk)}, {ka(s, 0)} offset:@K+{blk * 32 * ROWB}", k0, g, fk)
                    rd(f"ds_read_b128 {slot_hi(sk)}, {ka(s, 1)} offset:@K+{blk * 32 * ROWB}", k1, g, fk)
                    c = "0" if s == 0 else sset(par, blk)
                    mfma[g] = (f"{MFMA} {sset(par, blk)}, {slot(sk)}, {qf(s)}, {c}", [k0, k1])

    def alloc_p(rec, dt, g):
        if True:
            sv, fv = take(g)
            if rec:
                k0, k1 = ("VT", dt, 0), ("VT", dt, 1)
                rd(f"ds_read_b128 {slot_lo(sv)}, {va(0)} offset:@VP+{dt * 32 * 64}", k0, g, fv)
                rd(f"ds_read_b128 {slot_hi(sv)}, {va(1)} offset:@VP+{dt * 32 * 64}", k1, g, fv)
                # P of the keys j - 2: the same parity as this body's
                mfma[g] = (f"{MFMA} {o(dt)}, {slot(sv)}, {pf(par)}, {o(dt)}", [k0, k1])

    # two passes (the schedule is cyclic): the first learns which gap last consumes each slot
    allocate(False)
    for i in range(NSLOT):
        busy[i] -= NS
    ctr[0] = 0
    allocate(True)

    def valu(text, kind, rel, dl, after=None):
        t = Task(text, COST[kind], rel, dl, kind, after=after)
        tasks.append(t)
        return t

    # ---- softmax of the keys j - 1 (the other parity's S sets -> the other parity's PF).  Its consumers are the NEXT body's P
    #      MFMAs: the eight element quads of a body get staggered windows so that every gap carries the same mix.
    op = par ^ 1
    k = 0
    for blk in (0, 1):
        for w in range(4):
            rel, dl = max(0, k - WIN), min(NS - 1, k + WIN)
            k += 1
            done = []
            for e, r in enumerate((4 * w, 4 * w + 1, 4 * w + 2, 4 * w + 3)):
                # FA2_GEN_F8_ABL (timing only, wrong results): noFMA / expMOV / noADD -- what a class of the softmax's instructions costs
                f = None if "noFMA" in ABL else valu(f"v_fma_f32 {sreg(op, blk, r)}, {sreg(op, blk, r)}, %[c2], -{MB}", "valu", rel, dl)
                x = valu(f"{'v_mov_b32' if 'expMOV' in ABL else 'v_exp_f32'} {sreg(op, blk, r)}, {sreg(op, blk, r)}",
                         "valu" if "expMOV" in ABL else "exp", rel, dl, after=[f] if f else None)
                if "noADD" not in ABL:
                    valu(f"v_add_f32 {l(e & 1)}, {l(e & 1)}, {sreg(op, blk, r)}", "valu", rel, dl, after=[x])
                done.append(x)
            lo = valu(f"v_cvt_pk_fp8_f32 {pfw(op, blk, w)}, {sreg(op, blk, 4 * w)}, {sreg(op, blk, 4 * w + 1)}", "cvt", rel, dl, after=done[:2])
            valu(f"v_cvt_pk_fp8_f32 {pfw(op, blk, w)}, {sreg(op, blk, 4 * w + 2)}, {sreg(op, blk, 4 * w + 3)} op_sel:[0,0,1]", "cvt", rel, dl,
                 after=done[2:] + [lo])

    # ---- lane maxima of the keys j behind their A chains (masked variant: dead keys to -inf first), then the compare
    last = []
    prev = None
    for blk in (() if nomax else (0, 1)):
        rel = min(gA(blk, KS - 1) + 2, NS - 1)          # two later MFMAs have issued: the chain's last product has left the pipe
        masks = {}
        if masked:
            for r in range(16):
                masks[r] = valu(f"v_cmp_gt_i32 vcc, %[hi], {32 * blk + r}\n\tv_cndmask_b32 {sreg(par, blk, r)}, %[ninf], {sreg(par, blk, r)}, vcc",
                                "cmp", rel, NS - 1)
        for i in range(8):
            a, b = sreg(par, blk, 2 * i), sreg(par, blk, 2 * i + 1)
            text = f"v_max_f32 {RM}, {a}, {b}" if (blk == 0 and i == 0) else f"v_max3_f32 {RM}, {RM}, {a}, {b}"
            dep = ([prev] if prev else []) + ([masks[2 * i], masks[2 * i + 1]] if masked else [])
            prev = valu(text, "valu", rel, NS - 1, after=dep)
    if nomax:
        # the caller has shown that no score of these keys can pass its row's threshold (fa2_fwd_fp8.hip: |q| |k| bound)
        valu("s_mov_b32 %[need], 0", "valu", NS - 1, NS - 1)
    else:
        last.append(prev)
        valu(f"v_cmp_gt_f32 vcc, {RM}, {TH}\n\ts_or_b32 %[need], vcc_lo, vcc_hi", "cmp", NS - 1, NS - 1, after=last)

    # ---- LDS-DMA of tile t + 2 (second halves only): two K pieces (rows 8 w .. and 64 + 8 w ..) and two V^T pieces (d rows
    #      16 w .. of the two key halves) per wave; LDS piece p of a tensor's tile lies 1 KiB x p into it
    if dma:
        pieces = [(0, 0, "%[kso]", None), (0, 1, "s12", f"s_add_u32 s12, %[kso], {64 * ROWB}"),
                  (1, 0, "%[vso]", None), (1, 1, "s12", "s_add_u32 s12, %[vso], 64")]
        for n, (which, i, so, pre) in enumerate(pieces):
            g0 = 1 + int(n * (NS - 3) / 3)
            rs, dv = ("%[vrs]", "%[dvv]") if which else ("%[krs]", "%[dvk]")
            text = f"s_add_u32 m0, %[mw], @NB+{which * NBUF * TILEB + i * WAVES * 1024}\n\t{pre or 's_nop 0'}\n\tbuffer_load_dwordx4 {dv}, {rs}, {so} offen lds"
            tasks.append(Task(text, COST["vmem"], g0, min(g0 + 2, NS - 1), "vmem", ("dma", which, i)))
    return mfma, tasks


ABL = set(x for x in os.environ.get("FA2_GEN_F8_ABL", "").split(",") if x)


def render(par, masked, dma, budget, nomax=False):
    mfma, tasks = build(par, masked, dma, nomax)
    per_gap, load = base.place(tasks, NS, budget)
    lines, pro = base.render_lines(mfma, per_gap, NS)
    return lines, pro, load


def resolve(lines, buf, kb, barrier):
    """Substitutes the placeholders for the body of half kb of the tile in ring buffer buf; '@N ' lines (the next body's early
    reads) get the next body's bases."""
    lines = [part for l in lines for part in (l.split("\n\t") if not l.startswith("@N ") else [l])]

    def bases(b, k):
        vb = (b + NBUF - 1) % NBUF          # the keys j - 2: the same half of the previous tile
        return {"K": b * TILEB + k * 64 * ROWB, "VP": vb * TILEB + k * HALFV}      # (VA holds the V^T ring's base: ds offsets are 16 bits)
    cur = bases(buf, kb)
    nxt = bases(buf, kb + 1) if kb + 1 < NH else bases((buf + 1) % NBUF, 0)
    out = []
    if barrier:
        # FA2_GEN_F8_NOSYNC (timing experiments only -- the results are wrong): 1 = no barrier, 2 = no barrier and no DMA wait
        nosync = int(os.environ.get("FA2_GEN_F8_NOSYNC", "0"))
        out += (["s_waitcnt vmcnt(0)"] if nosync < 2 else []) + (["s_barrier"] if nosync < 1 else [])
    for l in lines:
        b = cur
        if l.startswith("@N "):
            l, b = l[3:], nxt
        l = re.sub(r"@NB\+(\d+)", lambda m: str(((buf + 2) % NBUF) * TILEB + int(m.group(1))), l)
        l = re.sub(r"@(K|VP)\+(\d+)", lambda m: str(b[m.group(1)] + int(m.group(2))), l)
        out.append(l)
    return out


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--check", action="store_true")
    ap.add_argument("--out", default=os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", "cuda_flashattention_amd", "csrc",
                                                  "fa2_fwd_fp8_body.inc"))
    args = ap.parse_args()
    chunks = ["// GENERATED by tools/gen_fwd_fp8_body.py -- do not edit.  Main-loop bodies of fa2_fwd_fp8_kernel (two waves per SIMD):\n"
              "// FA2_F8_BODY_B<ring buffer>_K<half of the tile>_<M0 plain | M1 masked | X no maxima> and the prologue FA2_F8_PRO (the early reads of the very\n"
              "// first body).  Register map, LDS map and schedule: the generator.\n",
              f"#define FA2_F8_V0 {V0}\n#define FA2_F8_SET0 {SET[0]}\n#define FA2_F8_SET1 {SET[1]}\n#define FA2_F8_PF0 {PF[0]}\n"
              f"#define FA2_F8_KA {KA}\n#define FA2_F8_VA {VA}\n#define FA2_F8_STATE {STATE}\n#define FA2_F8_VEND {VEND}\n"
              f"#define FA2_F8_A_QF {A_QF}\n#define FA2_F8_KV {KV}\n#define FA2_F8_NBUF {NBUF}\n"]
    pros = set()
    # variants: M0 plain, M1 masked (sequence tail, causal diagonal), X = plain without the lane maxima and the compare (the
    # kernel runs it for keys whose scores it has bounded below every row's threshold: |q| |k| <= m + 6)
    for tag, masked, nomax in (("M0", False, False), ("M1", True, False), ("X", False, True)):
        for kb in range(NH):
            par = kb & 1
            dma = kb == NH - 1
            nm = nomax or (os.environ.get("FA2_GEN_F8_NOMAX_ALL") == "1" and not masked)
            lines, pro, load = render(par, masked, dma, BUDGET + ((64 if ORDER == 0 else 160) if masked else 0) + (8 if dma else 0), nomax=nm)
            pros.add(tuple(pro))
            if args.check:
                print(f"kb={kb} {tag} dma={int(dma)}: {len(lines)} lines, {sum('v_mfma' in x for x in lines)} MFMAs, {len(pro)} early, "
                      f"max gap load {max(load)}, mean {sum(load) / len(load):.1f}")
                print("   load:", " ".join(str(x) for x in load))
            for buf in range(NBUF):
                body = resolve(lines, buf, kb, dma)
                chunks.append(f"#define FA2_F8_BODY_B{buf}_K{kb}_{tag} \\\n" + base.c_string(body) + "\n")
    assert len(pros) == 1, "every body must leave the same reads in flight for the next one"
    p = resolve(list(pros.pop()), NBUF - 1, NH - 1, False)       # 'next' of the last half of buffer 3 = (buffer 0, half 0)
    p.append("s_waitcnt lgkmcnt(0)")
    chunks.append("#define FA2_F8_PRO \\\n" + base.c_string(p) + "\n")
    if not args.check:
        with open(args.out, "w") as f:
            f.write("\n".join(chunks))
        print("wrote", args.out)


if __name__ == "__main__":
    main()
