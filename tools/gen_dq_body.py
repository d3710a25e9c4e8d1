#!/usr/bin/env python3
"""Generates the hand-scheduled main-loop body of fa2_bwd_dq_kernel -> csrc/fa2_bwd_dq_body.inc.

One body = one 64-key K/V tile for one wave that owns 64 query rows (two 32-row blocks qb = 0, 1); the tile's two
32-key blocks kb = 0, 1 are a software pipeline one half-tile deep:

    A0(t)      S^T, dP^T of key block 0          4 x KS MFMAs   | VALU beside it: P, dS, packs of key block 1 of tile t-1
    Q1(t-1)    dQ^T += K^T dS^T, key block 1 of the PREVIOUS tile (its K image is still in the LDS ring)   4 x DT MFMAs
    A1(t)      S^T, dP^T of key block 1          4 x KS MFMAs   | VALU beside it: P, dS, packs of key block 0 of tile t
    Q0(t)      dQ^T of key block 0               4 x DT MFMAs

so the 3 + 1 VALU instructions per element (fma, exp, mul, half a pack) always run beside MFMAs that do not depend on
them, and every LDS fragment read is issued several MFMAs ahead of its use behind a counted lgkmcnt -- the same
generator as the dK/dV kernel's (tools/gen_dkdv_body.py: task placement, cyclic bodies, wait derivation).

Registers (kernel compiled with amdgpu_num_vgpr(64): hipcc owns v0..v63):
    a[0:128)   dQ^T tiles (qb, dt);  a[128:192) Q fragments (qb, s);  a[192:256) dO fragments -- as before
    v[64:128)  SET0 = S^T(qb0), S^T(qb1), dP^T(qb0), dP^T(qb1) of key block 0;   v[128:192) SET1, key block 1
    v[192:224) DSF[qb][kb][sp] packed dS^T (B operands of the dQ products);  v[224:248) six fragment slots;
    v[248:256) ROFFV: the row-read addresses of the V ring (set once by the kernel: roff + 3 tiles)
Operands: %[r*] row-read, %[t*] transposed-read LDS addresses, %[c2], %[lq0/1] = L log2(e) of the lane's rows,
%[nd0/1] = tiles of -D (start values of the dP^T chains), masked variant: %[hi0/1] (bounds of this tile),
%[hp0/1] (of the previous one).  LDS: a ring of three K tiles, then a ring of three V tiles (tile t in slot t mod 3);
every offset from its address register fits the 16-bit immediate.
"""
import argparse
import os
import re

import gen_dkdv_body as base
from gen_dkdv_body import Task, COST, READ_AHEAD, READ_LATEST

NSLOT = 6
A_DQ, A_QF, A_GF = 0, 128, 192
V0 = 64


class Regs:
    def __init__(self, D):
        self.D, self.KS, self.DT = D, D // 16, D // 32
        self.SET = [V0, V0 + 64]
        self.DSF = V0 + 128
        self.SLOT = V0 + 160
        self.ROFFV = V0 + 160 + 4 * NSLOT
        self.VEND = self.ROFFV + self.KS
        assert self.VEND <= 256

    def s(self, kb, qb): b = self.SET[kb] + 16 * qb; return f"v[{b}:{b + 15}]"
    def sreg(self, kb, qb, r): return f"v{self.SET[kb] + 16 * qb + r}"
    def d(self, kb, qb): b = self.SET[kb] + 32 + 16 * qb; return f"v[{b}:{b + 15}]"
    def dreg(self, kb, qb, r): return f"v{self.SET[kb] + 32 + 16 * qb + r}"
    def dsf(self, qb, kb, sp): b = self.DSF + 4 * (4 * qb + 2 * kb + sp); return f"v[{b}:{b + 3}]"
    def dsfw(self, qb, kb, sp, j): return f"v{self.DSF + 4 * (4 * qb + 2 * kb + sp) + j}"
    def slot(self, i): b = self.SLOT + 4 * i; return f"v[{b}:{b + 3}]"
    def slot_lo(self, i): b = self.SLOT + 4 * i; return f"v[{b}:{b + 1}]"
    def slot_hi(self, i): b = self.SLOT + 4 * i + 2; return f"v[{b}:{b + 1}]"
    def roffv(self, s): return f"v{self.ROFFV + s}"
    def qf(self, qb, s): b = A_QF + 4 * (qb * self.KS + s); return f"a[{b}:{b + 3}]"
    def gf(self, qb, s): b = A_GF + 4 * (qb * self.KS + s); return f"a[{b}:{b + 3}]"
    def dq(self, qb, dt): b = A_DQ + 16 * (qb * self.DT + dt); return f"a[{b}:{b + 15}]"


def build(D, masked):
    R = Regs(D)
    KS, DT = R.KS, R.DT
    ROWB = 2 * D
    NS = 8 * KS + 8 * DT
    gA = [0, 4 * KS + 4 * DT]                # A0, A1
    gQ1p, gQ0 = 4 * KS, 8 * KS + 4 * DT      # Q1 of the previous tile, Q0 of this one
    mfma = [None] * NS
    tasks = []
    slot_ctr = [0]
    busy = [-(10 ** 6)] * NSLOT

    def take_slot(last_consumer_gap):
        i = slot_ctr[0] % NSLOT
        slot_ctr[0] += 1
        free_after = busy[i]
        busy[i] = last_consumer_gap
        return i, free_after

    def add_read(text, key, consume, free_after):
        rel = max(consume - READ_AHEAD, free_after + 1)
        dl = max(consume - READ_LATEST, rel)
        tasks.append(Task(text, COST["lds"], rel, dl, "lds", key))

    def allocate(record):
        def a_stage(kb):
            for s in range(KS):
                g = gA[kb] + 4 * s
                sk, fk = take_slot(g + 1)
                sv, fv = take_slot(g + 3)
                if record:
                    kK, kV = ("K", kb, s), ("V", kb, s)
                    add_read(f"ds_read_b128 {R.slot(sk)}, %[r{s}] offset:@K+{kb * 32 * ROWB}", kK, g, fk)
                    add_read(f"ds_read_b128 {R.slot(sv)}, {R.roffv(s)} offset:@V+{kb * 32 * ROWB}", kV, g + 2, fv)
                    c_s = "0" if s == 0 else None
                    for qb in (0, 1):
                        mfma[g + qb] = (f"v_mfma_f32_32x32x16_bf16 {R.s(kb, qb)}, {R.slot(sk)}, {R.qf(qb, s)}, {c_s or R.s(kb, qb)}", [kK])
                        c_d = f"%[nd{qb}]" if s == 0 else R.d(kb, qb)
                        mfma[g + 2 + qb] = (f"v_mfma_f32_32x32x16_bf16 {R.d(kb, qb)}, {R.slot(sv)}, {R.gf(qb, s)}, {c_d}", [kV])

        def q_stage(kb, g0, kbase):
            for sp in (0, 1):
                for dt in range(DT):
                    g = g0 + 2 * (sp * DT + dt)
                    slot, free = take_slot(g + 1)
                    if record:
                        ka, kb_ = ("KT", kb, sp, dt, 0), ("KT", kb, sp, dt, 1)
                        off = kb * 32 * ROWB + sp * 16 * ROWB
                        add_read(f"ds_read_b64_tr_b16 {R.slot_lo(slot)}, %[t{2 * dt}] offset:{kbase}+{off}", ka, g, free)
                        add_read(f"ds_read_b64_tr_b16 {R.slot_hi(slot)}, %[t{2 * dt + 1}] offset:{kbase}+{off}", kb_, g, free)
                        for qb in (0, 1):
                            mfma[g + qb] = (f"v_mfma_f32_32x32x16_bf16 {R.dq(qb, dt)}, {R.slot(slot)}, {R.dsf(qb, kb, sp)}, {R.dq(qb, dt)}",
                                            [ka, kb_])
        a_stage(0)
        q_stage(1, gQ1p, "@KP")
        a_stage(1)
        q_stage(0, gQ0, "@K")

    allocate(False)
    assert slot_ctr[0] % NSLOT == 0, slot_ctr[0]
    for i in range(NSLOT):
        busy[i] -= NS
    slot_ctr[0] = 0
    allocate(True)

    # ---- VALU: key block kb of THIS tile (kb = 0: in this body) or of the previous tile (kb = 1: described one period
    #      early, so its release is negative and its deadlines are those of Q1 in this body)
    def valu(text, kind, rel, dl, after=None):
        t = Task(text, COST[kind], rel, dl, kind, after=after)
        tasks.append(t)
        return t

    for kb in (0, 1):
        if kb == 0:
            ends = gA[0] + 4 * KS - 1          # last MFMA of A0
            use0 = gQ0
            hi = "hi"
        else:
            ends = gA[1] + 4 * KS - 1 - NS     # last MFMA of the previous body's A1
            use0 = gQ1p
            hi = "@HP"                         # the previous tile's bounds in-body, this tile's when emitted in its own body's tail
        rel = ends + 3 + int(os.environ.get('FA2_GEN_REL_EXTRA', '0'))
        for qb in (0, 1):
            for sp in (0, 1):
                use = use0 + 2 * sp * DT + qb
                # Which part of key block 1's arithmetic runs in the tail of its own body and which in the next body must
                # be the SAME in the plain and the masked variant (they follow each other in any order): the most urgent
                # quarter (qb 0, sp 0) is pinned to the tail, the rest to the next body.
                tail_part = kb == 1 and qb == 0 and sp == 0
                lo_rel = rel if (kb == 0 or tail_part) else max(rel, 0)
                cap = (lambda d, slack=0: min(d, -1 - slack)) if tail_part else (lambda d, slack=0: d)
                for j in range(4):
                    pair = []
                    for r in (8 * sp + 2 * j, 8 * sp + 2 * j + 1):
                        f = valu(f"v_fma_f32 {R.sreg(kb, qb, r)}, {R.sreg(kb, qb, r)}, %[c2], -%[lq{qb}]", "valu", lo_rel, cap(use - 6, 3))
                        e = valu(f"v_exp_f32 {R.sreg(kb, qb, r)}, {R.sreg(kb, qb, r)}", "exp", lo_rel, cap(use - 5, 2), after=[f])
                        last = e
                        if masked:
                            rr = (r & 3) + 8 * (r >> 2) + 32 * kb
                            hop = f"%[hi{qb}]" if kb == 0 else f"{hi}{qb}@"
                            last = valu(f"v_cmp_gt_i32 vcc, {hop}, {rr}\n\tv_cndmask_b32 {R.sreg(kb, qb, r)}, 0, {R.sreg(kb, qb, r)}, vcc",
                                        "mask", lo_rel, cap(use - 4, 1), after=[e])
                        m = valu(f"v_mul_f32 {R.dreg(kb, qb, r)}, {R.sreg(kb, qb, r)}, {R.dreg(kb, qb, r)}", "valu", lo_rel, cap(use - 3, 1), after=[last])
                        pair.append(m)
                    valu(f"v_cvt_pk_bf16_f32 {R.dsfw(qb, kb, sp, j)}, {R.dreg(kb, qb, 8 * sp + 2 * j)}, {R.dreg(kb, qb, 8 * sp + 2 * j + 1)}",
                         "cvt", lo_rel, cap(use - 2), after=pair)
    return R, mfma, tasks, NS


def render(D, masked):
    R, mfma, tasks, NS = build(D, masked)
    per_gap, load = base.place(tasks, NS, int(os.environ.get('FA2_GEN_BUDGET', str(base.GAP_BUDGET))))
    # the tiles a stage overwrites must have been consumed: A1 writes SET1 at gA1, A0 of the next body SET0 at NS
    gA1 = 4 * R.KS + 4 * R.DT
    for t in tasks:
        if t.kind != "lds":
            reg = int(re.search(r"v(\d+)", t.text.split("\n")[-1]).group(1))
            if R.SET[1] <= reg < R.SET[1] + 64:
                assert t.gap < gA1 - 1, (t.text, t.gap)
            if R.SET[0] <= reg < R.SET[0] + 64:
                assert t.gap < NS - 1, (t.text, t.gap)
    lines, pro = base.render_lines(mfma, per_gap, NS)
    return R, lines, pro, load, NS


def resolve(lines, D, buf):
    """Substitutes placeholders for the body of the tile in ring buffer `buf`; '@N ' lines belong to the next tile."""
    ROWB = 2 * D
    TILEB = 64 * ROWB

    def bases(b):
        return {"K": b * TILEB, "V": b * TILEB}          # V relative to ROFFV (= roff + 3 tiles)
    cur, nxt, prv = bases(buf), bases((buf + 1) % 3), bases((buf + 2) % 3)
    out = []
    barrier_done = False
    for l in lines:
        b, is_next = cur, False
        if l.startswith("@N "):
            l, b, is_next = l[3:], nxt, True
            if not barrier_done and l.startswith("ds_read"):
                out.append("s_waitcnt vmcnt(0)")          # this wave's pieces of the next tile have landed ...
                out.append("s_barrier")                   # ... and everyone's
                barrier_done = True
        l = re.sub(r"@KP\+(\d+)", lambda m: str(prv["K"] + int(m.group(1))), l)
        l = re.sub(r"@(K|V)\+(\d+)", lambda m: str(b[m.group(1)] + int(m.group(2))), l)
        l = re.sub(r"@HP(\d)@", lambda m: f"%[hi{m.group(1)}]" if is_next else f"%[hp{m.group(1)}]", l)
        out.append(l)
    assert barrier_done
    return out


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--check", action="store_true")
    ap.add_argument("--out", default=os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", "cuda_flashattention_amd", "csrc",
                                                  "fa2_bwd_dq_body.inc"))
    args = ap.parse_args()
    chunks = ["// GENERATED by tools/gen_dq_body.py -- do not edit.  Hand-placed main-loop bodies of fa2_bwd_dq_kernel:\n"
              "// FA2_DQ_BODY_D<d>_B<ring buffer>_M<masked> and the prologues FA2_DQ_PRO_D<d>_M<masked> (the early work of the\n"
              "// very first tile).  Register map and schedule: tools/gen_dq_body.py.\n"]
    for D in (128, 64):
        R0 = Regs(D)
        e0, e1 = (sorted(l for l in render(D, m)[2] if "v_cmp" not in l) for m in (False, True))
        assert e0 == e1, "plain and masked bodies must leave the same work to the next body"
        chunks.append(f"#define FA2_DQ_D{D}_SET1 {R0.SET[1]}\n#define FA2_DQ_D{D}_ROFFV {R0.ROFFV}\n#define FA2_DQ_D{D}_VEND {R0.VEND}\n")
        for masked in (0, 1):
            R, lines, pro, load, NS = render(D, bool(masked))
            if args.check:
                print(f"D={D} masked={masked}: {len(lines)} lines, {sum('v_mfma' in l for l in lines)} MFMAs, {len(pro)} early, "
                      f"max gap load {max(load)}, {sum(l > base.GAP_BUDGET for l in load)} of {NS} gaps over {base.GAP_BUDGET}")
                print("   load:", " ".join(str(l) for l in load))
            p = [l for l in resolve(pro, D, 2) if l not in ("s_waitcnt vmcnt(0)", "s_barrier")]      # 'next' of buffer 2 is buffer 0
            p.append("s_waitcnt lgkmcnt(0)")          # in steady state the previous body's last waits cover these reads
            chunks.append(f"#define FA2_DQ_PRO_D{D}_M{masked} \\\n" + base.c_string(p) + "\n")
            for buf in (0, 1, 2):
                chunks.append(f"#define FA2_DQ_BODY_D{D}_B{buf}_M{masked} \\\n" + base.c_string(resolve(lines, D, buf)) + "\n")
    if not args.check:
        with open(args.out, "w") as f:
            f.write("\n".join(chunks))
        print("wrote", args.out)


if __name__ == "__main__":
    main()
