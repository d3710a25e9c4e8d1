#!/usr/bin/env python3
"""Generates the hand-scheduled main-loop bodies of the one-wave-per-SIMD forward -> csrc/fa2_fwd_body.inc.

Work split (fa2_fwd1_bf16.hip): workgroup = 4 waves = 256 query rows of one head, ONE wave per SIMD (512 registers); a wave
owns 64 query rows (two 32-row blocks qb = 0, 1), so every K / V^T fragment it reads from LDS feeds two MFMAs.  One body =
one 32-key block j for one wave, a software pipeline two key blocks deep:

    A   S^T(j)[qb]  = K(j) Q[qb]^T                      2 x KS MFMAs        K fragments by ds_read_b128
    P   O^T[qb]    += V^T(j-2) P^T(j-2)[qb]             4 x DT MFMAs        V^T fragments by ds_read_b64_tr_b16
    VALU beside both: the softmax of block j-1 (fma, exp, row sum, pack to bf16: 7 instructions per element pair) anywhere in
    the body, and the lane maxima of block j behind its A chains; the body ends with the compare "does any row need a new
    softmax reference" whose result it returns in an SGPR (the rare rescale is compiler code between two bodies).

So each of the 4 VALU instructions per element has a whole body of MFMAs that do not depend on it to hide behind (the
two-wave kernel's stages put 8 of them beside the 8 S^T products and 56 beside the 8 PV products), every LDS read is issued
4-7 MFMAs ahead of its use behind a counted lgkmcnt, and nothing is left for hipcc to schedule.  Same generator core as the
backward kernels' (tools/gen_dkdv_body.py: task placement, cyclic bodies, wait derivation).

LDS: a ring of FOUR K tiles, then a ring of four V tiles; a tile is KV keys = NH key blocks (KV = 64 at d = 128, 128 at
d = 64: 16 KiB per tile either way).  Body (tile in buffer b, key block kb) reads K(b, kb) and the V rows of key block
j - 2: (b, kb - 2), or ((b + 3) % 4, kb + NH - 2) in the previous tile's buffer (always, at NH = 2).  The body of the LAST key block of a tile starts with `s_waitcnt vmcnt(0); s_barrier` (tile t + 1 has
landed for everyone; everyone is done with tile t - 2's V, the buffer tile t + 2 goes to) and then issues the LDS-DMA of
tile t + 2 itself, as fillers.

Registers (kernel compiled with amdgpu_num_vgpr(64): hipcc owns v0..v63):
    a[0 : 32 DT)        O^T tiles (qb, dt);      a[128 : 128 + 8 KS)  Q fragments (qb, s)
    v[64:128)           S sets: S^T(parity, qb), 16 registers each;  v[128:160) packed P: PF[parity][qb][sp], 4 each
    v[160:192)          eight fragment slots;  then ROFF (KS row-read addresses of the K ring), TOFFV (2 DT transposed-read
                        addresses of the V ring) -- set once by the kernel; then the softmax STATE, ten registers the kernel
                        initialises and its rare update path rewrites: l0a l0b l1a l1b (partial row sums), rm0 rm1 (lane maxima
                        of block j), mb0 mb1 (reference * log2 e of the lane's rows), th0 th1 (raw-score thresholds).  Keeping
                        them out of hipcc's hands means no copies between the plain and the masked body variants.
Operands: %[c2] (s), %[need] (=s: some lane's maximum is over its threshold), masked
variant %[hi0/1] (v: first masked key of the lane's row, relative to the block and to the lane's half), %[ninf] (v: -inf);
DMA bodies
%[mw] (s: LDS byte address of the wave's first piece), %[dvo] (v), %[krs], %[vrs] (s x4), %[kso] (s: byte offset of the
wave's first piece of tile t + 2).
"""
import argparse
import os
import re

import gen_dkdv_body as base
from gen_dkdv_body import Task, COST

READ_AHEAD = int(os.environ.get("FA2_GEN_READ_AHEAD", str(base.READ_AHEAD)))
READ_LATEST = int(os.environ.get("FA2_GEN_READ_LATEST", str(base.READ_LATEST)))
NBUF = 4
# (head_dim, row blocks per wave): what fa2_fwd1_bf16.hip instantiates
CONFIGS = ((128, 2), (64, 2), (64, 1))
COST = dict(COST, vmem=12, cmp=8)
NEG_INF = "0xff800000"
# ABLATIONS (timing only, WRONG RESULTS; tools/build_fwd_variant.sh + tools/gpu_ab_multi.py): a comma-separated subset of
#   noFMA / noEXP / noADD / noCVT (that class of the softmax's VALU instructions dropped; expMOV: v_exp_f32 -> v_mov_b32),
#   noK / noV (the K / V^T fragment reads), noDMA (the next tile's LDS-DMA)
ABL = set(x for x in os.environ.get("FA2_GEN_FWD_ABL", "").split(",") if x)


def kv_of(D):
    return 64 if D == 128 else 128


class Regs:
    """Register map for head_dim D and QBS row blocks per wave.  QBS = 2: one wave per SIMD (512 registers, hipcc owns v0..v63);
    QBS = 1: two waves per SIMD (128 + 128 registers, hipcc owns v0..v39)."""

    def __init__(self, D, QBS):
        self.D, self.KS, self.DT, self.QBS = D, D // 16, D // 32, QBS
        self.KV = kv_of(D)
        self.NH = self.KV // 32
        self.WAVES = 8 // QBS
        self.NSLOT = 8 if QBS == 2 else 6
        self.V0 = V0 = 64 if QBS == 2 else 40
        self.A_O = 0
        self.A_QF = 16 * QBS * self.DT
        self.SET = [V0, V0 + 16 * QBS]
        self.PF = [V0 + 32 * QBS, V0 + 40 * QBS]
        self.SLOT = V0 + 48 * QBS
        self.ROFF = self.SLOT + 4 * self.NSLOT
        self.TOFFV = self.ROFF + self.KS
        self.STATE = self.TOFFV + 2 * self.DT      # l (2 per row block) | rm | mb | th (1 per row block each)
        self.VEND = self.STATE + 5 * QBS
        assert self.VEND <= (256 if QBS == 2 else 128), self.VEND
        assert self.A_QF + 4 * QBS * self.KS <= (256 if QBS == 2 else 128)

    def s(self, par, qb): b = self.SET[par] + 16 * qb; return f"v[{b}:{b + 15}]"
    def sreg(self, par, qb, r): return f"v{self.SET[par] + 16 * qb + r}"
    def pf(self, par, qb, sp): b = self.PF[par] + 8 * qb + 4 * sp; return f"v[{b}:{b + 3}]"
    def pfw(self, par, qb, sp, j): return f"v{self.PF[par] + 8 * qb + 4 * sp + j}"
    def slot(self, i): b = self.SLOT + 4 * i; return f"v[{b}:{b + 3}]"
    def slot_lo(self, i): b = self.SLOT + 4 * i; return f"v[{b}:{b + 1}]"
    def slot_hi(self, i): b = self.SLOT + 4 * i + 2; return f"v[{b}:{b + 1}]"
    def roff(self, s): return f"v{self.ROFF + s}"
    def toffv(self, i): return f"v{self.TOFFV + i}"
    def l(self, qb, e): return f"v{self.STATE + 2 * qb + e}"
    def rm(self, qb): return f"v{self.STATE + 2 * self.QBS + qb}"
    def mb(self, qb): return f"v{self.STATE + 3 * self.QBS + qb}"
    def th(self, qb): return f"v{self.STATE + 4 * self.QBS + qb}"
    def qf(self, qb, s): b = self.A_QF + 4 * (qb * self.KS + s); return f"a[{b}:{b + 3}]"
    def o(self, qb, dt): b = self.A_O + 16 * (qb * self.DT + dt); return f"a[{b}:{b + 15}]"


def build(D, QBS, par, masked, dma, nomax=False):
    """One body for a key block of parity `par` (S set / PF set selection).  Gap units 0 .. NS-1; tasks with a negative
    release belong to the tail of the previous body (they are emitted there with the NEXT body's bases: '@N')."""
    R = Regs(D, QBS)
    KS, DT, NSLOT = R.KS, R.DT, R.NSLOT
    ROWB = 2 * D
    NS = QBS * (KS + 2 * DT)
    gP = QBS * KS
    QB = range(QBS)
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
        for s in range(KS):
            g = QBS * s
            sk, fk = take(g + QBS - 1)
            if rec:
                key = ("K", s)
                rd(f"ds_read_b128 {R.slot(sk)}, {R.roff(s)} offset:@K+0", key, g, fk)
                for qb in QB:
                    c = "0" if s == 0 else R.s(par, qb)
                    mfma[g + qb] = (f"v_mfma_f32_32x32x16_bf16 {R.s(par, qb)}, {R.slot(sk)}, {R.qf(qb, s)}, {c}", [key])
        for sp in (0, 1):
            for dt in range(DT):
                g = gP + QBS * (sp * DT + dt)
                sv, fv = take(g + QBS - 1)
                if rec:
                    ka, kb_ = ("VT", sp, dt, 0), ("VT", sp, dt, 1)
                    off = sp * 16 * ROWB
                    rd(f"ds_read_b64_tr_b16 {R.slot_lo(sv)}, {R.toffv(2 * dt)} offset:@VP+{off}", ka, g, fv)
                    rd(f"ds_read_b64_tr_b16 {R.slot_hi(sv)}, {R.toffv(2 * dt + 1)} offset:@VP+{off}", kb_, g, fv)
                    for qb in QB:
                        # P of block j - 2: the same parity as this block's
                        mfma[g + qb] = (f"v_mfma_f32_32x32x16_bf16 {R.o(qb, dt)}, {R.slot(sv)}, {R.pf(par, qb, sp)}, {R.o(qb, dt)}", [ka, kb_])

    # Two passes (the schedule is cyclic): the first learns which gap last consumes each slot, the second places the reads with
    # "free after the previous body's last consumer".  The rotation restarts with every body -- all bodies take their slots in
    # the same order, so the early reads a body issues for the next one are the same whatever variant follows.
    allocate(False)
    for i in range(NSLOT):
        busy[i] -= NS
    ctr[0] = 0
    allocate(True)

    def valu(text, kind, rel, dl, after=None):
        t = Task(text, COST[kind], rel, dl, kind, after=after)
        tasks.append(t)
        return t

    # ---- softmax of block j - 1 (the other parity's S set -> the other parity's PF set).  Its consumers are the NEXT body's P
    #      MFMAs, so the 16 element pairs of a body may go anywhere in it: they are given staggered windows (pair k around gap
    #      k NS / 16) so that every gap carries the same mix -- one exponential or two, never a burst of them.
    op = par ^ 1
    span = NS / (8.0 * QBS)
    k = 0
    for sp in (0, 1):
        for qb in QB:
            for j in range(4):
                c = int(k * span)
                rel, dl = max(0, c - int(span)), min(NS - 1, c + int(span) + 2)
                k += 1
                pair = []
                for e, r in enumerate((8 * sp + 2 * j, 8 * sp + 2 * j + 1)):
                    f = None if "noFMA" in ABL else valu(f"v_fma_f32 {R.sreg(op, qb, r)}, {R.sreg(op, qb, r)}, %[c2], -{R.mb(qb)}", "valu", rel, max(dl - 2, rel))
                    ex = "v_mov_b32" if "expMOV" in ABL else "v_exp_f32"
                    x = None if "noEXP" in ABL else valu(f"{ex} {R.sreg(op, qb, r)}, {R.sreg(op, qb, r)}", "valu" if "expMOV" in ABL else "exp", rel, max(dl - 1, rel), after=[f] if f else None)
                    if "noADD" not in ABL:
                        valu(f"v_add_f32 {R.l(qb, e)}, {R.l(qb, e)}, {R.sreg(op, qb, r)}", "valu", rel, dl, after=[x] if x else None)
                    if x:
                        pair.append(x)
                if "noCVT" not in ABL:
                    valu(f"v_cvt_pk_bf16_f32 {R.pfw(op, qb, sp, j)}, {R.sreg(op, qb, 8 * sp + 2 * j)}, {R.sreg(op, qb, 8 * sp + 2 * j + 1)}",
                         "cvt", rel, dl, after=pair)

    # ---- lane maxima of block j behind its A chains (masked variant: dead keys to -inf first), then the compare
    last = []
    # X variant: no lane maxima, no compare -- the kernel runs it where it does not move the reference (fa2_fwd1_bf16.hip);
    # FA2_GEN_FWD_NOMAX_ALL=1 (experiment): every plain body, the upper bound of what the X rounds can gain
    nomax = nomax or (os.environ.get("FA2_GEN_FWD_NOMAX_ALL") == "1" and not masked)
    for qb in (() if nomax else QB):
        rel = min(QBS * (KS - 1) + qb + 3, NS - 3)          # the chain's last product has left the matrix pipe
        prev = None
        masks = {}
        if masked:
            for r in range(16):
                rr = (r & 3) + 8 * (r >> 2)
                masks[r] = valu(f"v_cmp_gt_i32 vcc, %[hi{qb}], {rr}\n\tv_cndmask_b32 {R.sreg(par, qb, r)}, %[ninf], {R.sreg(par, qb, r)}, vcc",
                                "cmp", rel, NS - 3)
        for i in range(8):
            a, b = R.sreg(par, qb, 2 * i), R.sreg(par, qb, 2 * i + 1)
            text = f"v_max_f32 {R.rm(qb)}, {a}, {b}" if i == 0 else f"v_max3_f32 {R.rm(qb)}, {R.rm(qb)}, {a}, {b}"
            dep = ([prev] if prev else []) + ([masks[2 * i], masks[2 * i + 1]] if masked else [])
            prev = valu(text, "valu", rel, NS - 2, after=dep)
        last.append(prev)
    if nomax:
        valu("s_mov_b32 %[need], 0", "valu", NS - 2, NS - 1)
    elif QBS == 2:
        valu(f"v_cmp_gt_f32 vcc, {R.rm(0)}, {R.th(0)}\n\tv_cmp_gt_f32 s[10:11], {R.rm(1)}, {R.th(1)}\n\ts_or_b64 vcc, vcc, s[10:11]\n\t"
             "s_or_b32 %[need], vcc_lo, vcc_hi", "cmp", NS - 2, NS - 1, after=last)
    else:
        valu(f"v_cmp_gt_f32 vcc, {R.rm(0)}, {R.th(0)}\n\ts_or_b32 %[need], vcc_lo, vcc_hi", "cmp", NS - 2, NS - 1, after=last)

    # ---- LDS-DMA of tile t + 2 (bodies of a tile's last key block only): piece i of tensor `which` for this wave
    if dma:
        TILEB = R.KV * ROWB
        npw = TILEB // 1024 // R.WAVES           # pieces per tensor per wave
        step = R.WAVES                           # a wave's pieces are `step` apart
        rpi = 1024 // ROWB                       # rows per piece
        for which in (0, 1):
            for i in range(npw):
                rs = "%[vrs]" if which else "%[krs]"
                so = f"s_add_u32 s12, %[kso], {step * i * rpi * ROWB}" if i else "s_nop 0"
                n = which * npw + i                  # one piece every NS / (2 npw + 1) gaps, not a burst of them
                g0 = 1 + int(n * max(NS - 4, 1) / (2 * npw))
                tasks.append(Task(f"s_add_u32 m0, %[mw], @NB+{which * NBUF * TILEB + step * i * 1024}\n\t{so}\n\t"
                                  f"buffer_load_dwordx4 %[dvo], {rs}, {'s12' if i else '%[kso]'} offen lds", COST["vmem"], g0, g0 + 3, "vmem",
                                  ("dma", which, i)))
    if ABL:
        def gone(t):
            k0 = t.key[0] if isinstance(t.key, tuple) else None
            return ("noK" in ABL and k0 == "K") or ("noV" in ABL and k0 == "VT") or ("noDMA" in ABL and k0 == "dma")
        dead = set(id(t) for t in tasks if gone(t))
        tasks[:] = [t for t in tasks if id(t) not in dead]
        present = set(t.key for t in tasks)
        mfma = [(text, [k for k in needs if k in present]) for text, needs in mfma]
    return R, mfma, tasks, NS


def render(D, QBS, par, masked, dma, budget, nomax=False):
    R, mfma, tasks, NS = build(D, QBS, par, masked, dma, nomax)
    per_gap, load = base.place(tasks, NS, budget)
    lines, pro = base.render_lines(mfma, per_gap, NS)
    return R, lines, pro, load, NS


def resolve(lines, D, QBS, buf, kb, barrier):
    """Substitutes the placeholders for the body of key block kb of the tile in ring buffer buf; '@N ' lines (the next
    body's early reads) get the next key block's bases."""
    R = Regs(D, QBS)
    ROWB = 2 * D
    TILEB = R.KV * ROWB
    lines = [part for l in lines for part in (l.split("\n\t") if not l.startswith("@N ") else [l])]

    def bases(b, k):
        # key block j - 2: two blocks back in the same tile, or in the previous tile's buffer
        vb, vk = (b, k - 2) if k >= 2 else ((b + NBUF - 1) % NBUF, k + R.NH - 2)
        return {"K": b * TILEB + k * 32 * ROWB, "VP": vb * TILEB + vk * 32 * ROWB}
    cur = bases(buf, kb)
    nxt = bases(buf, kb + 1) if kb + 1 < R.NH else bases((buf + 1) % NBUF, 0)
    out = []
    if barrier:
        out += ["s_waitcnt vmcnt(0)", "s_barrier"]
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
                                                  "fa2_fwd_body.inc"))
    args = ap.parse_args()
    chunks = ["// GENERATED by tools/gen_fwd_body.py -- do not edit.  Main-loop bodies of fa2_fwd1_bf16_kernel (one wave per SIMD):\n"
              "// FA2_FWD_BODY_D<d>Q<row blocks>_B<ring buffer>_K<key block of the tile>_<M0 plain | M1 masked | X no maxima> and the prologue FA2_FWD_PRO_D<d> (the early reads\n"
              "// of the very first body).  Register map, LDS map and schedule: the generator.\n"]
    for D, QBS in CONFIGS:
        R0 = Regs(D, QBS)
        tag = f"D{D}Q{QBS}"
        budget = int(os.environ.get("FA2_GEN_BUDGET_FWD%d" % D, "24" if D == 128 else "44"))
        chunks.append(f"#define FA2_FWD_{tag}_SET0 {R0.SET[0]}\n#define FA2_FWD_{tag}_SET1 {R0.SET[1]}\n#define FA2_FWD_{tag}_PF0 {R0.PF[0]}\n"
                      f"#define FA2_FWD_{tag}_ROFF {R0.ROFF}\n#define FA2_FWD_{tag}_TOFFV {R0.TOFFV}\n#define FA2_FWD_{tag}_VEND {R0.VEND}\n"
                      f"#define FA2_FWD_{tag}_KV {R0.KV}\n#define FA2_FWD_{tag}_STATE {R0.STATE}\n#define FA2_FWD_{tag}_V0 {R0.V0}\n"
                      f"#define FA2_FWD_{tag}_A_QF {R0.A_QF}\n")
        pros = set()
        for vtag, masked, nomax in (("M0", False, False), ("M1", True, False), ("X", False, True)):
            for kb in range(R0.NH):
                par = kb & 1
                dma = kb == R0.NH - 1
                R, lines, pro, load, NS = render(D, QBS, par, masked, dma, budget + (12 if masked else 0) + (4 if dma else 0), nomax)
                pros.add(tuple(pro))
                if args.check:
                    print(f"{tag} kb={kb} {vtag} dma={int(dma)}: {len(lines)} lines, {sum('v_mfma' in l for l in lines)} MFMAs, "
                          f"{len(pro)} early, max gap load {max(load)}, mean {sum(load) / len(load):.1f}")
                    print("   load:", " ".join(str(l) for l in load))
                for buf in range(NBUF):
                    body = resolve(lines, D, QBS, buf, kb, dma)
                    chunks.append(f"#define FA2_FWD_BODY_{tag}_B{buf}_K{kb}_{vtag} \\\n" + base.c_string(body) + "\n")
        assert len(pros) == 1, "every body must leave the same reads in flight for the next one"
        p = resolve(list(pros.pop()), D, QBS, NBUF - 1, R0.NH - 1, False)       # 'next' of the last key block of buffer 3 = (buffer 0, kb 0)
        p.append("s_waitcnt lgkmcnt(0)")
        chunks.append(f"#define FA2_FWD_PRO_{tag} \\\n" + base.c_string(p) + "\n")
    if not args.check:
        with open(args.out, "w") as f:
            f.write("\n".join(chunks))
        print("wrote", args.out)


if __name__ == "__main__":
    main()
