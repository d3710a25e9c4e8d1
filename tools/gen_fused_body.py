#!/usr/bin/env python3
"""Generates the main-loop bodies of the single-kernel (five-product) backward -> csrc/fa2_bwd_fused_body.inc.

One workgroup = 4 waves = 256 keys of one head (as in fa2_bwd_dkdv_kernel), but the query gradient is formed here too:
per 32-row sub-tile and wave

    A  S'  = Q K^T - L/scale         2 x 8 MFMAs     K fragments from the K IMAGE in LDS (V moved to resident VGPRs); the two key
                                                     blocks' chains alternate, so that one Q fragment read serves both
    B  dP' = dO V^T - D              2 x 8 MFMAs
    E  dQ[q][col] += dS[q][key] K[key][col] of the PREVIOUS sub-tile: 16 MFMAs over all 256 keys of the workgroup for
       this wave's 32 columns -- dS crosses LDS once ([key][q] tile written as the packed pairs already are, read back
       transposed), K^T comes from the same K image by transposed reads
    C  dV^T += dO^T P                16 MFMAs
    D  dK^T += Q^T dS                16 MFMAs        then the packed dS pairs are written to the dS tile

Five block products (80 MFMAs per sub-tile) instead of the seven of the two-kernel backward.  Q/dO tiles of 32 rows live in
a ring of three LDS buffers, the dS tile is double buffered, one barrier per sub-tile (in front of the first read of the
next tile: vmcnt(0) covers the DMA and this wave's dQ stores, the barrier makes everyone's dS visible).

Registers (kernel compiled with amdgpu_num_vgpr(40)): a[0:128) dK^T, a[128:256) dV^T;
    v[40:104) V fragments vf[kb][s]; SACC, DPACC, PF, DSF as in the dK/dV kernel; six fragment slots; DQT = 16 registers
    holding this wave's dQ tile (the E chain accumulates on top of what the kernel put there: zeros, or the running sum
    handed over by the previous key block's workgroup); ROFFK (8: row-read addresses of the wave's K rows), DSWR (4: dS
    write addresses), DSRD (2), KT (2).
Two families: FA2_FUSED_BODY_* (the dQ tile is left in DQT for the kernel) and FA2_FUSED_CBODY_* (chained kernel: the body
itself stores the finished tile, loads the next running sum, issues the next tile's LDS-DMA, and does the hand-shake with
the neighbouring key blocks; extra operands %[dqv], %[drs], %[dso], %[lrs], %[lso] (dQ tiles), %[mw], %[mw2], %[qrs], %[grs],
%[rcrs], %[qso], %[rcso], %[dvo], %[rcvo], %[wv] (DMA), %[ctl], %[pvo], %[mso], %[need], %[pval], %[err] (progress words)).
FA2_FUSED_MBODY_* = CBODY + the causal mask behind each exponential (operands %[lo0], %[lo1]).
Operands: %[r*], %[t*] (Q/dO ring addresses), %[rc], %[c2] (scale * log2 e; %[c2p], the same in both halves of an
SGPR pair, with FA2_GEN_PK=1), %[vm] (immediate: how many of the kernel's vector-memory
operations may still be in flight when the E chain starts -- those issued after its loads into DQT; 63 = no such loads).
Same generator core as tools/gen_dkdv_body.py (cyclic bodies, counted waits derived from the issue order)."""
import argparse
import os
import re

import gen_dkdv_body as base
from gen_dkdv_body import Task, COST

# tuning switches (defaults = what is committed; tools/README.md)
base.GAP_BUDGET = int(os.environ.get("FA2_GEN_BUDGET", str(base.GAP_BUDGET)))
READ_AHEAD = int(os.environ.get("FA2_GEN_READ_AHEAD", str(base.READ_AHEAD)))
READ_LATEST = int(os.environ.get("FA2_GEN_READ_LATEST", str(base.READ_LATEST)))

V0 = 40
KIMG = 0
DSTILE = 256 * 64
COST = dict(COST, ldsw=6, vmem=int(os.environ.get("FA2_GEN_COST_VMEM", "10")), cmask=8, lds=int(os.environ.get("FA2_GEN_COST_LDS", str(COST["lds"]))))


def set_dim(d):
    """Register map, LDS map and stage table for head_dim d (128: the round-2 kernel; 64: round 4).  d = 64: a sub-tile is 40
    MFMAs -- A 8, B 8, E 8, C 8, D 8 -- and the dQ tile of a sub-tile is 32 x 64: wave w forms columns 32 (w & 1) .. + 31 over
    the 128 keys of half w >> 1 of the workgroup's 256 (the kernel puts the half into the E address registers), so there are
    TWO running sums per column block, one per half, each handed from key block to key block like d = 128's one; the
    output pass adds them."""
    global D, KS, DT, ROWB, NE, VF, SACC, DPACC, PF, DSF, SLOT, DQT, ROFFK, DSWR, DSRD, KT, VEND, A_DK, A_DV, QRING, BUFB, DSB
    global LDS_BYTES, GBAR, DQTILE, NS_, PRE, GAP_BUDGET_D, NSLOT
    D, KS, DT = d, d // 16, d // 32
    # fragment slots: the rotation must close over a sub-tile AND leave the slots of the next body's first reads free early
    # (d = 128: 64 takes, six slots; d = 64: 32 takes, eight slots -- with six the last D-stage slot would be the next A's)
    NSLOT = 6 if D == 128 else 8
    ROWB = 2 * D
    NE = 16 if D == 128 else 8            # E's k-steps of 16 keys per wave: all 256 keys (d = 128) or the wave's half (d = 64)
    VF = V0
    SACC = VF + 8 * KS
    DPACC = SACC + 32
    PF = DPACC + 32
    DSF = PF + 16
    SLOT = DSF + 16
    DQT = SLOT + 4 * NSLOT
    ROFFK = DQT + 16
    DSWR = ROFFK + KS
    DSRD = DSWR + 4
    KT = DSRD + 2
    VEND = KT + 2
    assert VEND <= 256
    A_DK, A_DV = 0, 32 * DT
    # LDS map (bytes): K image | ring of three Q / dO tiles | two dS tiles
    QRING = 256 * ROWB
    BUFB = 2 * 32 * ROWB + 256        # Q tile | dO tile | 32 x (-L/scale), 32 x (-D)
    DSB = QRING + 3 * BUFB            # two dS tiles of 256 keys x 32 q bf16
    LDS_BYTES = DSB + 2 * DSTILE
    NS_ = 4 * KS + NE + 8 * DT
    # The body's one barrier sits right behind MFMA number GBAR: everything that must be out before it (the dS tile's writes,
    # the chained form's dQ stores) has a deadline in front of it, everything that relies on it (reads of the next tile, the
    # loads of the next running sum) is released behind it.
    GBAR = int(os.environ.get("FA2_GEN_GBAR" if D == 128 else "FA2_GEN_GBAR64", str(NS_ - 8 if D == 128 else NS_ - 5)))
    DQTILE = f"v[{DQT}:{DQT + 15}]"
    PRE = "FA2_FUSED" if D == 128 else "FA2_FUSED64"
    # d = 64: the same VALU work per sub-tile beside half the MFMAs -- the gaps carry more than a 32-clock MFMA hides
    GAP_BUDGET_D = base.GAP_BUDGET if D == 128 else int(os.environ.get("FA2_GEN_BUDGET64", "44"))
# where the next tile's LDS-DMA pieces may be issued (gaps): the guide prices a piece at ~60 clocks among bare MFMAs, 100 - 185
# in a phase full of ds_read_b128, 25 - 60 in VALU-only gaps
DMA_REL = int(os.environ.get("FA2_GEN_DMA_REL", "1"))
DMA_DL = int(os.environ.get("FA2_GEN_DMA_DL", "12"))
DMA_POLICY = os.environ.get("FA2_GEN_DMA_POLICY", "")          # experiment: " nt" on the Q / dO tiles' LDS-DMA


def vf(kb, s): b = VF + 4 * (kb * KS + s); return f"v[{b}:{b + 3}]"
def sacc(kb): b = SACC + 16 * kb; return f"v[{b}:{b + 15}]"
def sreg(kb, r): return f"v{SACC + 16 * kb + r}"
def dpacc(kb): b = DPACC + 16 * kb; return f"v[{b}:{b + 15}]"
def dreg(kb, r): return f"v{DPACC + 16 * kb + r}"
def pf(kb, sp): b = PF + 4 * (2 * kb + sp); return f"v[{b}:{b + 3}]"
def pfw(kb, sp, j): return f"v{PF + 4 * (2 * kb + sp) + j}"
def dsf(kb, sp): b = DSF + 4 * (2 * kb + sp); return f"v[{b}:{b + 3}]"
def dsfw(kb, sp, j): return f"v{DSF + 4 * (2 * kb + sp) + j}"
def slot(i): b = SLOT + 4 * i; return f"v[{b}:{b + 3}]"
def slot_lo(i): b = SLOT + 4 * i; return f"v[{b}:{b + 1}]"
def slot_hi(i): b = SLOT + 4 * i + 2; return f"v[{b}:{b + 1}]"
def dk(kb, dt): b = A_DK + 16 * (kb * DT + dt); return f"a[{b}:{b + 15}]"
def dv(kb, dt): b = A_DV + 16 * (kb * DT + dt); return f"a[{b}:{b + 15}]"


set_dim(128)


# packed multiplies (v_pk_mul_f32 for the two scalings of a register pair): MEASURED SLOWER -- 971K instead of 882K cycles per
# unit at (4,16,8192,128); a v_pk_mul_f32 costs the issue port about as much as four plain multiplies.  Kept as a switch.
PK = os.environ.get("FA2_GEN_PK", "0") == "1"
# K image pre-scaled by scale * log2 e (fa2_bwd_fused.hip scales it once per unit): S' arrives in the exp2 domain and the
# multiply in front of each exponential goes away (32 of a body's ~135 VALU instructions)
KSCALED = os.environ.get("FA2_GEN_KSCALED", "0") == "1"
# stage A with the two key blocks' chains interleaved (one Q fragment read per k-step serves both, as one dO fragment does in
# stage B): 24 instead of 31 ds_read_b128 in the stage whose reads alone fill the LDS array when all four waves are in it
# (round 4, measured same-box with the dS tile's conflict-free key: 4.313 -> 4.280 ms; FA2_GEN_AILV=0 = the round-3 order)
AILV = os.environ.get("FA2_GEN_AILV", "1") == "1"
# stages C and D: the two key blocks of a d-tile in alternating order (kb 0, 1 | 1, 0 | 0, 1 ...), so that consecutive MFMAs
# always share one operand -- the transposed fragment inside a pair, the packed P / dS fragment across two pairs (operand
# switching is part of the energy of an MFMA: tools/probes/mfma_shape.hip modes 4-6)
SNAKE = os.environ.get("FA2_GEN_SNAKE", "0") == "1"
# ABLATIONS (timing only, WRONG RESULTS; var/ builds for tools/gpu_ab_multi.py): a comma-separated subset of
#   noE (E's dS / K^T reads), noAK (A's K reads), noDMA (Q / dO / row-constant LDS-DMA), noDQ (running-sum loads and stores),
#   noVALU (exp / mul / cvt), noDSW (dS tile writes), noRC (row-constant reads), noSEEN (progress prefetch)
ABL = set(x for x in os.environ.get("FA2_GEN_ABL", "").split(",") if x)


def build(chain=False, masked=False):
    NS = NS_
    gA1, gB = KS + 1, 2 * KS
    gC = gB + 2 * KS + NE
    gD = gC + 4 * DT
    # B (dP', one dO fragment per pair of MFMAs: light on the LDS) and E (four transposed reads per MFMA: with all four
    # waves in it at once, exactly what the LDS array can deliver) share the gaps between A and C: the first half of the B
    # pairs, then groups of [B pair, four E steps] -- 75 % of the array instead of 25 % followed by 100 %.
    ILV = os.environ.get("FA2_GEN_ILV", "1") == "1"
    def gBp(sidx): return (gB + 2 * sidx) if (not ILV or sidx < KS // 2) else (gB + KS + 6 * (sidx - KS // 2))
    def gEs(sidx): return (gB + 2 * KS + sidx) if not ILV else (gB + KS + 6 * (sidx // 4) + 2 + sidx % 4)
    gEend = gEs(NE - 1) + 1
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
        if rel < 0:
            rel = max(rel, GBAR - NS)            # a read of the NEXT tile: behind this body's barrier
        tasks.append(Task(text, COST["lds"], rel, max(consume - READ_LATEST, rel), "lds", key))

    def allocate(rec):
        # ---- A0 / A1 (as in the dK/dV kernel, with K fragments from the K image)
        sq, fq = take(1)
        s1, f1 = take(0)
        s0, f0 = take(1)
        if rec:
            rd(f"ds_read_b128 {slot(sq)}, %[r0] offset:@Q+0", ("Q", 0, 0), 0, fq)
            rd(f"ds_read_b128 {slot(s1)}, v{ROFFK} offset:{32 * ROWB}", ("K", 1, 0), 0, f1)
            rd(f"ds_read_b128 {slot(s0)}, v{ROFFK}", ("K", 0, 0), 1, f0)
            mfma[0] = (f"v_mfma_f32_32x32x16_bf16 {sacc(1)}, {slot(sq)}, {slot(s1)}, {sacc(0)}", [("Q", 0, 0), ("K", 1, 0), ("RCS",)])
            mfma[1] = (f"v_mfma_f32_32x32x16_bf16 {sacc(0)}, {slot(sq)}, {slot(s0)}, {sacc(0)}", [("Q", 0, 0), ("K", 0, 0)])
        for s in (range(1, KS) if AILV else ()):
            g = 2 * s
            a, fa = take(g + 1)
            b0, f0_ = take(g)
            b1, f1_ = take(g + 1)
            if rec:
                rd(f"ds_read_b128 {slot(a)}, %[r{s}] offset:@Q+0", ("Q", 0, s), g, fa)
                rd(f"ds_read_b128 {slot(b0)}, v{ROFFK + s}", ("K", 0, s), g, f0_)
                rd(f"ds_read_b128 {slot(b1)}, v{ROFFK + s} offset:{32 * ROWB}", ("K", 1, s), g + 1, f1_)
                mfma[g] = (f"v_mfma_f32_32x32x16_bf16 {sacc(0)}, {slot(a)}, {slot(b0)}, {sacc(0)}", [("Q", 0, s), ("K", 0, s)])
                mfma[g + 1] = (f"v_mfma_f32_32x32x16_bf16 {sacc(1)}, {slot(a)}, {slot(b1)}, {sacc(1)}", [("Q", 0, s), ("K", 1, s)])
        for kb in (() if AILV else (0, 1)):
            for s in range(1, KS):
                g = (1 + s) if kb == 0 else (gA1 + s - 1)
                a, fa = take(g)
                b, fb = take(g)
                if rec:
                    rd(f"ds_read_b128 {slot(a)}, %[r{s}] offset:@Q+0", ("Q", kb, s), g, fa)
                    rd(f"ds_read_b128 {slot(b)}, v{ROFFK + s} offset:{kb * 32 * ROWB}", ("K", kb, s), g, fb)
                    mfma[g] = (f"v_mfma_f32_32x32x16_bf16 {sacc(kb)}, {slot(a)}, {slot(b)}, {sacc(kb)}", [("Q", kb, s), ("K", kb, s)])
        # ---- B: dP' (dO fragment shared by the two key blocks; V fragments are resident)
        # ---- E: dQ of the previous sub-tile: 16 k-steps of 16 keys over the workgroup's 256 keys.  The K^T fragments go
        # through the regular slots; the dS fragments land in the four DSF tuples, which are idle from the previous body's D
        # stage to this body's dS packs (so E reads five MFMAs ahead instead of three).
        # (fragment slots are handed out in the order of use: the two stages interleave)
        xbusy = [gD - NS + 2 * DT - 2, gD - NS + 2 * DT - 1, gD - NS + 4 * DT - 2, gD - NS + 4 * DT - 1]      # last D-stage readers of dsf(0,0), (1,0), (0,1), (1,1)
        if SNAKE and (DT - 1) % 2 == 1:            # the last d-tile of a half runs (kb 1, kb 0)
            xbusy = [xbusy[1], xbusy[0], xbusy[3], xbusy[2]]
        for g, kind, s in sorted([(gBp(i), "B", i) for i in range(KS)] + [(gEs(i), "E", i) for i in range(NE)]):
            if kind == "B":
                sg, fg = take(g + 1)
                if rec:
                    rd(f"ds_read_b128 {slot(sg)}, %[r{s}] offset:@G+0", ("G", s), g, fg)
                    if s == 0:
                        mfma[g] = (f"v_mfma_f32_32x32x16_bf16 {dpacc(1)}, {slot(sg)}, {vf(1, s)}, {dpacc(0)}", [("G", s), ("RCD",)])
                        mfma[g + 1] = (f"v_mfma_f32_32x32x16_bf16 {dpacc(0)}, {slot(sg)}, {vf(0, s)}, {dpacc(0)}", [("G", s)])
                    else:
                        mfma[g] = (f"v_mfma_f32_32x32x16_bf16 {dpacc(0)}, {slot(sg)}, {vf(0, s)}, {dpacc(0)}", [("G", s)])
                        mfma[g + 1] = (f"v_mfma_f32_32x32x16_bf16 {dpacc(1)}, {slot(sg)}, {vf(1, s)}, {dpacc(1)}", [("G", s)])
                continue
            x = s % 4
            xr = DSF + 4 * x
            sb, fb = take(g)
            fa = xbusy[x]
            xbusy[x] = g
            if rec:
                ka, kb_ = ("DS", s), ("KT", s)
                rd(f"ds_read_b64_tr_b16 v[{xr}:{xr + 1}], v{DSRD} offset:@DSP+{1024 * s}", ("ds0", s), g, fa)
                rd(f"ds_read_b64_tr_b16 v[{xr + 2}:{xr + 3}], v{DSRD + 1} offset:@DSP+{1024 * s}", ka, g, fa)
                rd(f"ds_read_b64_tr_b16 {slot_lo(sb)}, v{KT} offset:{16 * ROWB * s}", ("kt0", s), g, fb)
                rd(f"ds_read_b64_tr_b16 {slot_hi(sb)}, v{KT + 1} offset:{16 * ROWB * s}", kb_, g, fb)
                mfma[g] = (f"v_mfma_f32_32x32x16_bf16 {DQTILE}, v[{xr}:{xr + 3}], {slot(sb)}, {DQTILE}", [ka, kb_, ("ds0", s), ("kt0", s)])
        # ---- C, D
        for nm, basep, gs, acc, frag in (("GT", "@G", gC, dv, pf), ("QT", "@Q", gD, dk, dsf)):
            for sp in (0, 1):
                for dt in range(DT):
                    g = gs + 2 * (sp * DT + dt)
                    sl, fr = take(g + 1)
                    if rec:
                        ka, kb_ = (nm, sp, dt, 0), (nm, sp, dt, 1)
                        rd(f"ds_read_b64_tr_b16 {slot_lo(sl)}, %[t{2 * dt}] offset:{basep}+{sp * 16 * ROWB}", ka, g, fr)
                        rd(f"ds_read_b64_tr_b16 {slot_hi(sl)}, %[t{2 * dt + 1}] offset:{basep}+{sp * 16 * ROWB}", kb_, g, fr)
                        k0, k1 = (1, 0) if (SNAKE and dt % 2 == 1) else (0, 1)
                        mfma[g] = (f"v_mfma_f32_32x32x16_bf16 {acc(k0, dt)}, {slot(sl)}, {frag(k0, sp)}, {acc(k0, dt)}", [ka, kb_])
                        mfma[g + 1] = (f"v_mfma_f32_32x32x16_bf16 {acc(k1, dt)}, {slot(sl)}, {frag(k1, sp)}, {acc(k1, dt)}", [ka, kb_])
        while ctr[0] % NSLOT:
            take(NS - 1)                          # skipped slot numbers: the rotation closes over a sub-tile

    allocate(False)
    for i in range(NSLOT):
        busy[i] -= NS
    ctr[0] = 0
    allocate(True)

    def valu(text, kind, rel, dl, after=None):
        t = Task(text, COST[kind], rel, dl, kind, after=after)
        tasks.append(t)
        return t

    last_p, last_d = {}, {}
    for kb in (0, 1):
        rel_exp = ((2 * KS - 2 + kb) if AILV else (KS if kb == 0 else 2 * KS - 1)) + 3
        exps = {}
        for sp in (0, 1):
            use_pf = gC + 2 * sp * DT + kb
            for j in range(4):
                pair = []
                r0 = 8 * sp + 2 * j
                a0 = SACC + 16 * kb + r0
                # both scalings of a register pair in one packed multiply (%[c2p]: c2 in both halves of an SGPR pair): the
                # loop is bound by what one wave can ISSUE, and a packed fp32 multiply issues like a plain one
                m2 = valu(f"v_pk_mul_f32 v[{a0}:{a0 + 1}], v[{a0}:{a0 + 1}], %[c2p]", "valu", rel_exp, use_pf - 5) if PK else None
                for r in (r0, r0 + 1):
                    m = None if KSCALED else (m2 or valu(f"v_mul_f32 {sreg(kb, r)}, %[c2], {sreg(kb, r)}", "valu", rel_exp, use_pf - 5))
                    e = valu(f"v_exp_f32 {sreg(kb, r)}, {sreg(kb, r)}", "exp", rel_exp, use_pf - 4, after=[m] if m else None)
                    if masked:
                        # causal: P = 0 where the lane's key lies above the row.  Row of register r within the sub-tile =
                        # (r & 3) + 8 (r >> 2) + 4 h; %[lo<kb>] (per lane) = key - 32 tile - 4 h: keep iff row >= key
                        rr = (r & 3) + 8 * (r >> 2)
                        e = valu(f"v_cmp_le_i32 vcc, %[lo{kb}], {rr}\n\tv_cndmask_b32 {sreg(kb, r)}, 0, {sreg(kb, r)}, vcc", "cmask", rel_exp,
                                 use_pf - 3, after=[e])
                    exps[r] = e
                    pair.append(e)
                valu(f"v_cvt_pk_bf16_f32 {pfw(kb, sp, j)}, {sreg(kb, 8 * sp + 2 * j)}, {sreg(kb, 8 * sp + 2 * j + 1)}", "cvt", rel_exp,
                     use_pf - 2, after=pair)
        rel_ds = gBp(KS - 1) + kb + 3
        for sp in (0, 1):
            use_ds = gD + 2 * sp * DT + kb
            for jp in (0, 1):
                cv = []
                for j in (2 * jp, 2 * jp + 1):
                    pair = []
                    r0 = 8 * sp + 2 * j
                    if PK:
                        a0, d0 = SACC + 16 * kb + r0, DPACC + 16 * kb + r0
                        t = valu(f"v_pk_mul_f32 v[{d0}:{d0 + 1}], v[{a0}:{a0 + 1}], v[{d0}:{d0 + 1}]", "valu", rel_ds, use_ds - 3,
                                 after=[exps[r0], exps[r0 + 1]])
                        pair.append(t)
                        last_p[kb] = t
                    for r in (() if PK else (r0, r0 + 1)):
                        t = valu(f"v_mul_f32 {dreg(kb, r)}, {sreg(kb, r)}, {dreg(kb, r)}", "valu", rel_ds, use_ds - 3, after=[exps[r]])
                        pair.append(t)
                        last_p[kb] = t
                    c = valu(f"v_cvt_pk_bf16_f32 {dsfw(kb, sp, j)}, {dreg(kb, 8 * sp + 2 * j)}, {dreg(kb, 8 * sp + 2 * j + 1)}", "cvt",
                             max(rel_ds, gEend), use_ds - 2, after=pair)    # the DSF tuples are E's dS fragment slots until E ends
                    cv.append(c)
                    last_d[kb] = c
                # the packed pair (4 consecutive q of this lane's key) -> the dS tile of THIS sub-tile ([key][q], 8-byte chunks
                # XOR-swizzled by the key; the address register carries the (sp, jp, lane) part, the key block is an immediate)
                b = DSF + 4 * (2 * kb + sp) + 2 * jp
                tasks.append(Task(f"ds_write_b64 v{DSWR + 2 * sp + jp}, v[{b}:{b + 1}] offset:@DSW+{kb * 32 * 64}", COST["ldsw"], rel_ds,
                                  GBAR - 1, "ldsw", ("dsw", kb, sp, jp), after=cv))
    for g4 in range(4):
        d0 = SACC + 4 * g4
        t = Task(f"ds_read_b128 v[{d0}:{d0 + 3}], %[rc] offset:@RC+{32 * g4}", COST["lds"], -READ_AHEAD - 4, -READ_LATEST, "lds",
                 ("RCS",) if g4 == 3 else ("rcs", g4))
        t.release = max(t.release, max(last_p[0].deadline, last_p[1].deadline) - NS + 1, GBAR - NS)
        t.deadline = max(t.deadline, t.release)
        tasks.append(t)
    for g4 in range(4):
        d0 = DPACC + 4 * g4
        t = Task(f"ds_read_b128 v[{d0}:{d0 + 3}], %[rc] offset:@RC+{128 + 32 * g4}", COST["lds"], gB - READ_AHEAD - 4, gB - READ_LATEST, "lds",
                 ("RCD",) if g4 == 3 else ("rcd", g4))
        t.release = max(t.release, max(last_d[0].deadline, last_d[1].deadline) - NS + 1)
        t.deadline = max(t.deadline, t.release)
        tasks.append(t)
    if chain:
        # The chained kernel keeps the running dQ sums in a layout of its own -- [tile][wave][g][lane] x 4 floats, register
        # 4 g + e of the tile = float e of the lane's 16 bytes -- so a tile moves with four 1-KiB instructions each way.
        # Stores of the tile E has just finished go in front of the barrier (whose vmcnt(0) then says they are out: the
        # kernel publishes its progress right after the body); the loads of the next tile's running sum follow the barrier
        # and have until the next body's E stage.  Out-of-range soffsets / a null descriptor make either a no-op (zeros).
        # What the kernel would otherwise do between two bodies, with the matrix pipe idle, is part of the body too: the
        # LDS-DMA of the next Q / dO tile (two 1-KiB pieces of each per wave; waves 0 and 1 also fetch the 32 + 32 row
        # constants) and the prefetch of the previous key block's progress word into v39.
        for which in (0, 1):
            for i in range(ROWB // 128):           # 1-KiB pieces of a 32-row tile per wave: two at d = 128, one at d = 64
                rs = "%[grs]" if which else "%[qrs]"
                mid = f"s_add_u32 s12, %[qso], {4096 * i}" if i else "s_nop 0"
                tasks.append(Task(f"s_add_u32 m0, %[mw], @NB+{32 * ROWB * which + 4096 * i}\n\t{mid}\n\t"
                                  f"buffer_load_dwordx4 %[dvo], {rs}, {'s12' if i else '%[qso]'} offen{DMA_POLICY} lds", COST["vmem"] + 2, DMA_REL, DMA_DL, "vmem",
                                  ("dma", which, i)))
        tasks.append(Task(f"s_cmp_lt_u32 %[wv], 2\n\ts_cbranch_scc0 4f\n\ts_mov_b64 exec, 0xffffffff\n\ts_add_u32 m0, %[mw2], @NB+{2 * 32 * ROWB}\n\t"
                          "s_nop 0\n\tbuffer_load_dword %[rcvo], %[rcrs], %[rcso] offen lds\n\ts_mov_b64 exec, -1\n\t4:", COST["vmem"] + 6, DMA_REL, DMA_DL,
                          "vmem", ("dma", "rc")))
        # the progress prefetch is issued late (its round trip is ~900 clocks, the barrier behind which it is read sits at
        # gap 72): the staler the prefetched word, the further behind its predecessor a key block has to run
        pg = int(os.environ.get("FA2_GEN_SEEN_GAP", str(GBAR - 30))) if D == 128 else int(os.environ.get("FA2_GEN_SEEN_GAP64", str(GBAR - 20)))
        tasks.append(Task("buffer_load_dword v39, off, %[ctl], %[pvo] sc1", COST["vmem"], pg, pg + 4, "vmem", ("seen",)))
        st = [Task(f"buffer_store_dwordx4 v[{DQT + 4 * g}:{DQT + 4 * g + 3}], %[dqv], %[drs], %[dso] offen offset:{1024 * g}", COST["vmem"],
                   gEend + 3, min(gEend + 8, GBAR - 1), "vmem", ("dqst", g)) for g in range(4)]
        tasks.extend(st)
        for g in range(4):
            tasks.append(Task(f"buffer_load_dwordx4 v[{DQT + 4 * g}:{DQT + 4 * g + 3}], %[dqv], %[lrs], %[lso] offen offset:{1024 * g}@LDSC",
                              COST["vmem"], GBAR, NS - 1, "vmem", ("dqld", g), after=st))
    if ABL:
        def gone(t):
            k = t.key if isinstance(t.key, tuple) else ()
            k0 = k[0] if k else None
            return (("noE" in ABL and k0 in ("ds0", "DS", "kt0", "KT")) or ("noAK" in ABL and k0 == "K") or
                    ("noDMA" in ABL and k0 == "dma") or ("noDQ" in ABL and k0 in ("dqst", "dqld")) or
                    ("noVALU" in ABL and t.kind in ("valu", "exp", "cvt", "cmask")) or ("noDSW" in ABL and k0 == "dsw") or
                    ("noRC" in ABL and k0 in ("rcs", "RCS", "rcd", "RCD")) or ("noSEEN" in ABL and k0 == "seen"))
        dead = set(id(t) for t in tasks if gone(t))
        tasks[:] = [t for t in tasks if id(t) not in dead]
        for t in tasks:
            t.after = [d for d in t.after if id(d) not in dead]
        present = set(t.key for t in tasks)
        mfma = [(text, [k for k in needs if k in present]) for text, needs in mfma]
    return mfma, tasks, NS


def render_lines(mfma, per_gap, NS):
    """base.render_lines, with LDS writes counted in the issue order too (they share lgkmcnt and return in order)."""
    for g in per_gap:
        for t in per_gap[g]:
            if t.kind == "ldsw":
                t.kind = "lds"
    # one s_waitcnt may also cover what the next four MFMAs need, as far as those reads have been in flight for two MFMAs
    # or more: 26 waits per body instead of 56, measured -0.9 % (4.409 -> 4.369 ms; merging younger reads too is slower)
    base.WAIT_LOOK = int(os.environ.get("FA2_GEN_WAIT_LOOK", "4"))
    base.WAIT_AGE = int(os.environ.get("FA2_GEN_WAIT_AGE", "3"))      # round 4: 3 measured 0.6 % faster than 2 beside the interleaved stage A
    return base.render_lines(mfma, per_gap, NS)


SPIN_LIMIT = 1 << 22

# Behind the barrier of a chained body (every wave's dQ stores are out): publish this key block's progress -- each wave's
# lane 0 writes the same word -- and make sure the previous key block has stored the running sum the loads that follow
# will fetch.  v39 holds the progress word prefetched at the top of the body; a wait that runs out (SPIN_LIMIT polls)
# raises %[err] and goes on, so that a fault ends in poisoned output, not in a hung GPU.
AFTER_BARRIER = [
    "v_readfirstlane_b32 s12, v39",
    "v_mov_b32 v39, %[pval]",
    "s_mov_b64 exec, 1",
    "buffer_store_dword v39, off, %[ctl], %[mso]",
    "s_mov_b64 exec, -1",
    "s_cmp_ge_i32 s12, %[need]",
    "s_cbranch_scc1 2f",
    f"s_mov_b32 s13, {SPIN_LIMIT}",
    "1:",
    "buffer_load_dword v39, off, %[ctl], %[pvo] sc1",
    "s_waitcnt vmcnt(0)",
    "v_readfirstlane_b32 s12, v39",
    "s_cmp_ge_i32 s12, %[need]",
    "s_cbranch_scc1 2f",
    "s_sub_u32 s13, s13, 1",
    "s_cmp_lg_u32 s13, 0",
    "s_cbranch_scc1 1b",
    "s_mov_b32 %[err], 1",
    "2:",
]


def resolve(lines, buf, par, chain=False, prologue=False):
    """Body of the sub-tile in ring buffer `buf` whose dS tile is `par`; E reads the previous sub-tile's dS tile (par ^ 1)."""
    lines = [part for l in lines for part in (l.split("\n\t") if not l.startswith("@N ") else [l])]
    def bases(b):
        return {"Q": b * BUFB, "G": b * BUFB + 32 * ROWB, "RC": b * BUFB}
    cur, nxt = bases(buf), bases((buf + 1) % 3)
    out, barrier_done, e_wait, n_mfma = [], False, False, 0
    for l in lines:
        b = cur
        is_next = l.startswith("@N ")
        if is_next:
            l, b = l[3:], nxt
            assert barrier_done or prologue or not l.startswith("ds_read"), "a read of the next tile in front of the barrier"
        if not e_wait and l.startswith("v_mfma") and l.split()[1].startswith(f"v[{DQT}:"):
            out.append("s_waitcnt vmcnt(%c[vm])")      # the running dQ sum the kernel loaded into DQT ahead of this body has landed
            e_wait = True
        if "ds_write" in l:
            assert not barrier_done, "a dS write behind the barrier that publishes the dS tile"
        if "buffer_store" in l:
            assert not barrier_done, "a dQ store behind the barrier whose vmcnt(0) the kernel's progress flag relies on"
        if "buffer_load_dwordx4 v[" in l:
            assert barrier_done and not is_next, "a dQ load in front of the barrier (its vmcnt(0) would wait for it)"
        l = re.sub(r"@NB\+(\d+)", lambda m: str(((buf + 1) % 3) * BUFB + int(m.group(1))), l)
        # E reads of the NEXT body (wrapped) read the tile this body wrote (par); in-body E reads the previous one
        l = re.sub(r"@DSP\+(\d+)", lambda m: str((par if is_next else par ^ 1) * DSTILE + int(m.group(1))), l)
        # sc1: past the CU's vector cache (a workgroup may meet the same running-sum lines twice when a head has more key
        # blocks than the XCD has CUs); the XCD's L2 serves them
        l = l.replace("@LDSC", os.environ.get("FA2_GEN_LDSC", " sc1"))
        l = re.sub(r"@DSW\+(\d+)", lambda m: str(par * DSTILE + int(m.group(1))), l)
        l = re.sub(r"@(Q|G|RC)\+(\d+)", lambda m: str(b[m.group(1)] + int(m.group(2))), l)
        out.append(l)
        if l.startswith("v_mfma"):
            n_mfma += 1
            if n_mfma == GBAR + 1:
                # the next tile's DMA (issued in front of this body) has landed, and the chained body's dQ stores are out.
                # (Letting the stores stay in flight here -- vmcnt(4) -- and publishing one body later was measured: no
                # faster per step, and a longer start-up skew along the chain.)
                out.append("s_waitcnt vmcnt(0)")
                out.append("s_barrier")
                if chain:
                    out.extend(AFTER_BARRIER)
                barrier_done = True
    assert barrier_done or prologue
    return out


def emit(check):
    """The chunks of the .inc for the head_dim set_dim() selected (or, with check, the schedule's per-gap load)."""
    mfma, tasks, NS = build()
    per_gap, load = base.place(tasks, NS, budget=GAP_BUDGET_D)
    lines, pro = render_lines(mfma, per_gap, NS)
    cm, ct, _ = build(chain=True)
    cper_gap, cload = base.place(ct, NS, budget=GAP_BUDGET_D)
    clines, cpro = render_lines(cm, cper_gap, NS)
    assert cpro == pro
    # the causal kernel's bodies for the tiles around the diagonal: rare (at most 18 per unit), so they may run over the
    # issue budget of a gap; what they leave in flight for the next body is the same as the plain ones' (they alternate)
    mm, mt, _ = build(chain=True, masked=True)
    mper_gap, mload = base.place(mt, NS, budget=GAP_BUDGET_D + 8)
    mlines, mpro = render_lines(mm, mper_gap, NS)
    assert mpro == pro
    if check:
        print("   chained load:", " ".join(str(l) for l in cload))
        print("   masked  load:", " ".join(str(l) for l in mload))
        print(f"fused D={D}: {len(lines)} lines, {sum('v_mfma' in l for l in lines)} MFMAs, {len(pro)} early, max gap load {max(load)}, "
              f"{sum(l > GAP_BUDGET_D for l in load)} of {NS} gaps over {GAP_BUDGET_D}")
        print("   load:", " ".join(str(l) for l in load))
        return []
    chunks = [f"#define {PRE}_VF {VF}\n#define {PRE}_DQT {DQT}\n#define {PRE}_ROFFK {ROFFK}\n#define {PRE}_DSWR {DSWR}\n"
              f"#define {PRE}_DSRD {DSRD}\n#define {PRE}_KT {KT}\n#define {PRE}_QRING {QRING}\n#define {PRE}_BUFB {BUFB}\n"
              f"#define {PRE}_DSB {DSB}\n#define {PRE}_DSTILE {DSTILE}\n#define {PRE}_LDS {LDS_BYTES}\n"
              + (f"#define FA2_FUSED_SPIN_LIMIT {SPIN_LIMIT}\n" if D == 128 else "")]
    p = resolve(pro, 2, 1, prologue=True)      # 'next' of (buffer 2, parity 1) = (0, 0)
    p.append("s_waitcnt lgkmcnt(0)")
    chunks.append(f"#define {PRE}_PRO \\\n" + base.c_string(p) + "\n")
    for buf in range(3):
        for par in range(2):
            chunks.append(f"#define {PRE}_BODY_B{buf}_P{par} \\\n" + base.c_string(resolve(lines, buf, par)) + "\n")
            chunks.append(f"#define {PRE}_CBODY_B{buf}_P{par} \\\n" + base.c_string(resolve(clines, buf, par, chain=True)) + "\n")
            chunks.append(f"#define {PRE}_MBODY_B{buf}_P{par} \\\n" + base.c_string(resolve(mlines, buf, par, chain=True)) + "\n")
    return chunks


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--check", action="store_true")
    ap.add_argument("--out", default=os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", "cuda_flashattention_amd", "csrc",
                                                  "fa2_bwd_fused_body.inc"))
    args = ap.parse_args()
    chunks = ["// GENERATED by tools/gen_fused_body.py -- do not edit.  Main-loop bodies of the single-kernel five-product backward:\n"
              "// FA2_FUSED_{BODY,CBODY,MBODY}_B<ring buffer>_P<dS tile parity> (atomics form / chained / chained + causal mask), prologue\n"
              "// FA2_FUSED_PRO; the same with the prefix FA2_FUSED64_ for head_dim 64.  Register and LDS maps: the generator.\n"]
    for d in (128, 64):
        set_dim(d)
        chunks += emit(args.check)
    if args.check:
        return
    with open(args.out, "w") as f:
        f.write("\n".join(chunks))
    print("wrote", args.out)


if __name__ == "__main__":
    main()
