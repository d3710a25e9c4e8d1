#!/usr/bin/env python3
"""Generates the hand-scheduled main-loop body of fa2_bwd_dkdv_kernel -> csrc/fa2_bwd_dkdv_body.inc.

One body = one 32-row sub-tile of a 64-row Q/dO tile for one wave that owns 64 keys (two 32-key blocks kb = 0, 1):

    A  S'[q][key]      = Q K^T - L/scale            (chains start from the row constants)      2 x KS MFMAs
    B  dP'[q][key]     = dO V^T - D                                                             2 x KS MFMAs
    C  dV^T[col][key] += dO^T P,   P = exp2(c S')                                               4 x DT MFMAs
    D  dK^T[col][key] += Q^T dS,   dS = P o dP'                                                 4 x DT MFMAs

as ONE asm statement in which every instruction has a place: each MFMA is followed by the fillers that issue in
its shadow -- LDS fragment reads several MFMAs ahead of their use behind COUNTED lgkmcnt waits, the exponentials,
the dS products, the bf16 packs -- so the matrix pipe waits neither for an LDS round trip nor for the VALU.  A wave
alone on its SIMD hides about 24 clocks of other issue per 32-clock MFMA (MI355X_MICROARCH.md, row 'one wave per
SIMD'); the stage order is chosen so that no stage needs more:

    A0 (S' of kb 0) | A1 (S' of kb 1, exp of kb 0 beside it) | B (the other exps) | C (sp-major, dS beside it) | D

The schedule is CYCLIC over sub-tiles: the reads and row-constant loads a sub-tile needs first are issued in the
tail of the previous body with the NEXT sub-tile's addresses (a prologue statement issues them for the very first
one), so a body starts with its operands already in flight.  Bodies are instantiated per (LDS buffer, sub-tile);
the one for sub-tile 1 carries `s_waitcnt vmcnt(0); s_barrier` ("the next tile has landed") in front of its first
read of the other buffer.  A MASKED variant (sequence tail, causal diagonal) zeroes P element-wise behind each exp.

Registers (the kernel is compiled with amdgpu_num_vgpr(60): hipcc allocates v0..v59 only; everything below is
named here and nowhere else):
    a[0 : 16 NT)       dK^T tiles (kb, dt) at a[16 (kb DT + dt)],  NT = 2 DT;      a[16 NT : 32 NT)  dV^T tiles
    v[60 ...)          K fragments kf[kb][s] (B operands of S'), then SACC, DPACC (S'/P and dP'/dS, 16 registers per
                       key block), PF, DSF (packed P and dS, [kb][sp] x 4), seven 4-register fragment SLOTs, ROFFV
                       (KS per-lane addresses of the wave's V rows, set once by the kernel)
Operands: %[r0..] row-read addresses, %[t0..] transposed-read addresses, %[rc] row-constant address, %[c2], and
for the masked variant %[hi], %[lo0], %[lo1].  All are LDS byte addresses with the buffer / sub-tile parts folded
into immediates.

`python tools/gen_dkdv_body.py --check` prints the per-gap issue load of the schedule.
"""
import argparse
import os
import re

GAP_BUDGET = 20          # clocks of other issue hidden per MFMA (one wave per SIMD)
COST = {"lds": 4, "valu": 4, "exp": 8, "cvt": 4, "mask": 16}
READ_AHEAD = 7           # issue a fragment read this many MFMAs before its consumer ...
READ_LATEST = 4          # ... and not later than this many
NSLOT = 7
WAIT_AGE = int(os.environ.get("FA2_GEN_WAIT_AGE", "3"))
WAIT_LOOK = int(os.environ.get("FA2_GEN_WAIT_LOOK", "0"))       # see render_lines; 2 was MEASURED 2-3 % slower (fused backward)


class Regs:
    def __init__(self, D):
        self.D, self.KS, self.DT = D, D // 16, D // 32
        KS, DT = self.KS, self.DT
        self.NT = 2 * DT
        v = 60
        self.KF = v; v += 8 * KS
        self.SACC = v; v += 32
        self.DPACC = v; v += 32
        self.PF = v; v += 16
        self.DSF = v; v += 16
        self.SLOT = v; v += 4 * NSLOT
        self.ROFFV = v; v += KS
        self.VEND = v
        assert v <= 256, v
        self.A_DK = 0
        self.A_DV = 16 * self.NT

    def kf(self, kb, s): b = self.KF + 4 * (kb * self.KS + s); return f"v[{b}:{b + 3}]"
    def sacc(self, kb): b = self.SACC + 16 * kb; return f"v[{b}:{b + 15}]"
    def sreg(self, kb, r): return f"v{self.SACC + 16 * kb + r}"
    def dpacc(self, kb): b = self.DPACC + 16 * kb; return f"v[{b}:{b + 15}]"
    def dreg(self, kb, r): return f"v{self.DPACC + 16 * kb + r}"
    def pf(self, kb, sp): b = self.PF + 4 * (2 * kb + sp); return f"v[{b}:{b + 3}]"
    def pfw(self, kb, sp, j): return f"v{self.PF + 4 * (2 * kb + sp) + j}"
    def dsf(self, kb, sp): b = self.DSF + 4 * (2 * kb + sp); return f"v[{b}:{b + 3}]"
    def dsfw(self, kb, sp, j): return f"v{self.DSF + 4 * (2 * kb + sp) + j}"
    def slot(self, i): b = self.SLOT + 4 * i; return f"v[{b}:{b + 3}]"
    def slot_lo(self, i): b = self.SLOT + 4 * i; return f"v[{b}:{b + 1}]"
    def slot_hi(self, i): b = self.SLOT + 4 * i + 2; return f"v[{b}:{b + 1}]"
    def dk(self, kb, dt): b = self.A_DK + 16 * (kb * self.DT + dt); return f"a[{b}:{b + 15}]"
    def dv(self, kb, dt): b = self.A_DV + 16 * (kb * self.DT + dt); return f"a[{b}:{b + 15}]"
    def roffv(self, s): return f"v{self.ROFFV + s}"


class Task:
    __slots__ = ("text", "cost", "release", "deadline", "kind", "key", "gap", "seq", "after")

    def __init__(self, text, cost, release, deadline, kind, key=None, after=None):
        self.text, self.cost, self.release, self.deadline, self.kind, self.key = text, cost, release, deadline, kind, key
        self.after = after or []       # tasks that must be placed (strictly earlier in issue order) before this one
        self.gap = None


def build(D, masked):
    """The generic sub-tile: MFMA list (text, keys it consumes) and the filler tasks, in gap units 0..NS-1 (negative
    release = 'in the previous body', with the @Q/@G/@RC placeholders resolved to the NEXT sub-tile there)."""
    R = Regs(D)
    KS, DT = R.KS, R.DT
    ROWB = 2 * D
    SPB = 16 * ROWB
    NS = 4 * KS + 8 * DT
    gA0, gA1, gB, gC, gD = 0, KS + 1, 2 * KS, 4 * KS, 4 * KS + 4 * DT     # A0 holds KS + 1 MFMAs, A1 KS - 1
    mfma = [None] * NS
    tasks = []

    slot_ctr = [0]
    slot_busy_until = [-(10 ** 6)] * NSLOT       # gap of the slot's last consumer (modulo the cyclic schedule: previous body)

    def take_slot(last_consumer_gap):
        i = slot_ctr[0] % NSLOT
        slot_ctr[0] += 1
        free_after = slot_busy_until[i]
        slot_busy_until[i] = last_consumer_gap
        return i, free_after

    def add_read(text, key, consume, free_after):
        rel = max(consume - READ_AHEAD, free_after + 1)
        dl = max(consume - READ_LATEST, rel)
        t = Task(text, COST["lds"], rel, dl, "lds", key)
        tasks.append(t)
        return t

    # Two passes over the slot rotation so that 'free_after' of the first users refers to the previous body's consumers
    # (the schedule is cyclic): run the allocation once to learn the end state, then for real.
    def allocate(record):
        # A0: k-step 0 of BOTH key blocks (S'[1] takes its start values -- the row constants -- from S'[0]'s registers as
        # its C operand, before S'[0] is accumulated in place: one set of constant loads serves both), then S'[0] k-steps
        # 1..KS-1; A1: S'[1] k-steps 1..KS-1 with the Q fragments read again.
        slot, free = take_slot(1)
        if record:
            key = ("Q", 0, 0)
            add_read(f"ds_read_b128 {R.slot(slot)}, %[r0] offset:@Q+0", key, 0, free)
            mfma[0] = (f"v_mfma_f32_32x32x16_bf16 {R.sacc(1)}, {R.slot(slot)}, {R.kf(1, 0)}, {R.sacc(0)}", [key, ("RCS", 0)])
            mfma[1] = (f"v_mfma_f32_32x32x16_bf16 {R.sacc(0)}, {R.slot(slot)}, {R.kf(0, 0)}, {R.sacc(0)}", [key])
        for kb in (0, 1):
            for s in range(1, KS):
                g = (1 + s) if kb == 0 else (gA1 + s - 1)
                slot, free = take_slot(g)
                if record:
                    key = ("Q", kb, s)
                    add_read(f"ds_read_b128 {R.slot(slot)}, %[r{s}] offset:@Q+0", key, g, free)
                    mfma[g] = (f"v_mfma_f32_32x32x16_bf16 {R.sacc(kb)}, {R.slot(slot)}, {R.kf(kb, s)}, {R.sacc(kb)}", [key])
        take_slot(gA1)                              # one slot number skipped: the rotation closes over a sub-tile
        for s in range(KS):
            g = gB + 2 * s
            sg, fg = take_slot(g + 1)
            if s == 0:                              # dP'[1] first: it starts from the constants in dP'[0]'s registers
                s1, f1 = take_slot(g)
                s0, f0 = take_slot(g + 1)
            else:
                s0, f0 = take_slot(g)
                s1, f1 = take_slot(g + 1)
            if record:
                kG, k0, k1 = ("G", s), ("V0", s), ("V1", s)
                add_read(f"ds_read_b128 {R.slot(sg)}, %[r{s}] offset:@G+0", kG, g, fg)
                if s == 0:
                    add_read(f"ds_read_b128 {R.slot(s1)}, {R.roffv(s)} offset:{32 * ROWB}", k1, g, f1)
                    add_read(f"ds_read_b128 {R.slot(s0)}, {R.roffv(s)}", k0, g + 1, f0)
                    mfma[g] = (f"v_mfma_f32_32x32x16_bf16 {R.dpacc(1)}, {R.slot(sg)}, {R.slot(s1)}, {R.dpacc(0)}", [kG, k1, ("RCD", 0)])
                    mfma[g + 1] = (f"v_mfma_f32_32x32x16_bf16 {R.dpacc(0)}, {R.slot(sg)}, {R.slot(s0)}, {R.dpacc(0)}", [kG, k0])
                else:
                    add_read(f"ds_read_b128 {R.slot(s0)}, {R.roffv(s)}", k0, g, f0)
                    add_read(f"ds_read_b128 {R.slot(s1)}, {R.roffv(s)} offset:{32 * ROWB}", k1, g + 1, f1)
                    mfma[g] = (f"v_mfma_f32_32x32x16_bf16 {R.dpacc(0)}, {R.slot(sg)}, {R.slot(s0)}, {R.dpacc(0)}", [kG, k0])
                    mfma[g + 1] = (f"v_mfma_f32_32x32x16_bf16 {R.dpacc(1)}, {R.slot(sg)}, {R.slot(s1)}, {R.dpacc(1)}", [kG, k1])
        for nm, base, gs, acc, frag in (("GT", "@G", gC, R.dv, R.pf), ("QT", "@Q", gD, R.dk, R.dsf)):
            for sp in (0, 1):
                for dt in range(DT):
                    g = gs + 2 * (sp * DT + dt)
                    slot, free = take_slot(g + 1)
                    if record:
                        ka, kb_ = (nm, sp, dt, 0), (nm, sp, dt, 1)
                        add_read(f"ds_read_b64_tr_b16 {R.slot_lo(slot)}, %[t{2 * dt}] offset:{base}+{sp * SPB}", ka, g, free)
                        add_read(f"ds_read_b64_tr_b16 {R.slot_hi(slot)}, %[t{2 * dt + 1}] offset:{base}+{sp * SPB}", kb_, g, free)
                        mfma[g] = (f"v_mfma_f32_32x32x16_bf16 {acc(0, dt)}, {R.slot(slot)}, {frag(0, sp)}, {acc(0, dt)}", [ka, kb_])
                        mfma[g + 1] = (f"v_mfma_f32_32x32x16_bf16 {acc(1, dt)}, {R.slot(slot)}, {frag(1, sp)}, {acc(1, dt)}", [ka, kb_])

    allocate(False)
    assert slot_ctr[0] % NSLOT == 0, "slot rotation must close over one sub-tile"
    for i in range(NSLOT):
        slot_busy_until[i] -= NS                 # as seen from the next body
    slot_ctr[0] = 0
    allocate(True)

    # ---- VALU work
    def valu(text, kind, rel, dl, after=None):
        t = Task(text, COST[kind], rel, dl, kind, after=after)
        tasks.append(t)
        return t

    last_p_read = {}        # (kb) -> last task reading the S'/P registers of kb      (the next row-constant load must follow it)
    last_d_read = {}
    for kb in (0, 1):
        chain_end = KS if kb == 0 else 2 * KS - 1
        rel_exp = chain_end + 3                  # the chain's last product has left the matrix pipe (> 12 wait states)
        exps = {}
        for sp in (0, 1):
            use_pf = gC + 2 * sp * DT + kb       # first MFMA reading pf[kb][sp]
            for j in range(4):
                pair = []
                for r in (8 * sp + 2 * j, 8 * sp + 2 * j + 1):
                    m = valu(f"v_mul_f32 {R.sreg(kb, r)}, %[c2], {R.sreg(kb, r)}", "valu", rel_exp, use_pf - 5)
                    e = valu(f"v_exp_f32 {R.sreg(kb, r)}, {R.sreg(kb, r)}", "exp", rel_exp, use_pf - 4, after=[m])
                    last = e
                    if masked:
                        rr = (r & 3) + 8 * (r >> 2)
                        # keep iff rr + 32 sh < hi and rr + 32 sh >= lo_kb; @RR+n resolves to n + 32 sh
                        last = valu(f"v_cmp_gt_i32 vcc, %[hi], @RR+{rr}\n\tv_cmp_le_i32 s[10:11], %[lo{kb}], @RR+{rr}\n\t"
                                    f"s_and_b64 vcc, vcc, s[10:11]\n\tv_cndmask_b32 {R.sreg(kb, r)}, 0, {R.sreg(kb, r)}, vcc",
                                    "mask", rel_exp, use_pf - 3, after=[e])
                    exps[r] = last
                    pair.append(last)
                valu(f"v_cvt_pk_bf16_f32 {R.pfw(kb, sp, j)}, {R.sreg(kb, 8 * sp + 2 * j)}, {R.sreg(kb, 8 * sp + 2 * j + 1)}", "cvt",
                     rel_exp, use_pf - 2, after=pair)
        dp_end = gB + 2 * (KS - 1) + kb
        rel_ds = dp_end + 3
        mul = "v_mul_legacy_f32" if masked else "v_mul_f32"      # masked rows: 0 x (whatever the missing row constant was) = 0
        for sp in (0, 1):
            use_ds = gD + 2 * sp * DT + kb
            for j in range(4):
                pair = []
                for r in (8 * sp + 2 * j, 8 * sp + 2 * j + 1):
                    t = valu(f"{mul} {R.dreg(kb, r)}, {R.sreg(kb, r)}, {R.dreg(kb, r)}", "valu", rel_ds, use_ds - 3, after=[exps[r]])
                    pair.append(t)
                    last_p_read[kb] = t
                c = valu(f"v_cvt_pk_bf16_f32 {R.dsfw(kb, sp, j)}, {R.dreg(kb, 8 * sp + 2 * j)}, {R.dreg(kb, 8 * sp + 2 * j + 1)}", "cvt",
                         rel_ds, use_ds - 2, after=pair)
                last_d_read[kb] = c

    # ---- row constants straight into the accumulator registers of the NEXT use (4 x b128 per tile: registers 4g..4g+3 are
    #      rows 8g + 4h ..).  They follow the last reader of those registers in issue order (`after`), in the previous body.
    # The constants land in the kb = 0 tiles only (the kb = 1 chains take them from there as their C operand), but both tiles
    # of a kind must have been read for the last time before: the first MFMA of the new chains overwrites the kb = 1 tile too.
    for g4 in range(4):
        d0 = R.SACC + 4 * g4
        tasks.append(Task(f"ds_read_b128 v[{d0}:{d0 + 3}], %[rc] offset:@RC+{32 * g4}", COST["lds"], -READ_AHEAD - 4, -READ_LATEST, "lds",
                          ("RCS", 0) if g4 == 3 else ("rcs", g4)))
        tasks[-1].release = max(tasks[-1].release, max(last_p_read[0].deadline, last_p_read[1].deadline) - NS + 1)
        tasks[-1].deadline = max(tasks[-1].deadline, tasks[-1].release)
    for g4 in range(4):
        d0 = R.DPACC + 4 * g4
        tasks.append(Task(f"ds_read_b128 v[{d0}:{d0 + 3}], %[rc] offset:@RC+{256 + 32 * g4}", COST["lds"], gB - READ_AHEAD - 4,
                          gB - READ_LATEST, "lds", ("RCD", 0) if g4 == 3 else ("rcd", g4)))
        tasks[-1].release = max(tasks[-1].release, max(last_d_read[0].deadline, last_d_read[1].deadline) - NS + 1)
        tasks[-1].deadline = max(tasks[-1].deadline, tasks[-1].release)
    # the first MFMAs of the next chains write the kb = 1 tiles: their last readers must precede them (they do by a wide
    # margin -- checked here rather than assumed)
    guard_after = [(0, last_p_read[1]), (gB, last_d_read[1])]
    return R, mfma, tasks, NS, guard_after


def place(tasks, NS, budget=None):
    """Places every task into a gap (possibly negative = previous body), earliest deadline first, respecting the
    dependencies in `after`.  Returns (per-gap lists in issue order, per-gap load)."""
    budget = GAP_BUDGET if budget is None else budget
    load = {}
    for i, t in enumerate(tasks):
        t.seq = i
    # LDS reads are placed FIRST, on their own: which of them a body leaves in flight for the next one (the tasks with
    # negative gaps) must not depend on the VALU load, because plain and masked bodies follow each other in any order.
    for phase in (0, 1):
        pending = sorted((t for t in tasks if (t.kind == "lds") == (phase == 0)), key=lambda t: (t.deadline, t.seq))
        guard = 0
        while pending:
            guard += 1
            assert guard < 100000
            progressed = False
            for t in list(pending):
                lo = t.release
                ok = True
                for dep in t.after:
                    if isinstance(dep, tuple):           # ("prev", task): the dependency sits in the PREVIOUS body
                        d = dep[1]
                        if d.gap is None:
                            ok = False
                            break
                        lo = max(lo, d.gap - NS + 1)
                    else:
                        if dep.gap is None:
                            ok = False
                            break
                        lo = max(lo, dep.gap + (1 if dep.kind == "exp" else 0))    # a trans result is not read in the same gap
                if not ok:
                    continue
                g = lo
                while load.get(g % NS, 0) + t.cost > budget and g < t.deadline:
                    g += 1
                assert g <= t.deadline, (t.text, g, t.deadline)
                t.gap = g
                load[g % NS] = load.get(g % NS, 0) + t.cost
                pending.remove(t)
                progressed = True
            assert progressed, "dependency cycle (an LDS task may not depend on a VALU task)"
    per_gap = {}
    for t in tasks:
        per_gap.setdefault(t.gap, []).append(t)
    for g in per_gap:
        per_gap[g].sort(key=lambda t: (0 if t.kind == "lds" else 1, t.seq))
    return per_gap, [load.get(g, 0) for g in range(NS)]


def schedule(D, masked):
    R, mfma, tasks, NS, guard_after = build(D, masked)
    per_gap, load = place(tasks, NS)
    for g_first, t in guard_after:
        assert t.gap - NS < g_first - 2, (t.text, t.gap)
    assert all(t.kind == "lds" for t in tasks if t.gap < 0)
    return R, mfma, tasks, per_gap, load, NS


def render_lines(mfma, per_gap, NS):
    """(body lines with @placeholders, prologue lines) of a cyclic schedule.  Lines tagged '@N ' belong to the NEXT
    body's early work (they use the next unit's bases); the counted lgkmcnt in front of each MFMA is derived from the
    steady-state issue order of the LDS operations (two periods are simulated, the second one is emitted)."""
    gmin = min(per_gap)
    assert gmin >= -NS, gmin

    def gap_items(g):
        own = per_gap.get(g, []) if g >= 0 else []
        nxt = per_gap.get(g - NS, []) if g - NS < 0 else []
        return own, nxt

    issued = []         # keys in issue order; entries are (period, key)
    issue_gap = []      # absolute gap (period * NS + g) at which each was issued
    waited_upto = [-1]  # index into `issued` up to which completion is known
    lines = []
    for period in (0, 1):
        for g in range(NS):
            text, needs = mfma[g]
            pos = -1
            for k in needs:
                idx = max(i for i, (p, kk) in enumerate(issued) if kk == k and p == period) if any(kk == k and p == period for p, kk in issued) else None
                assert idx is not None or period == 0, (k, g)
                if idx is not None:
                    pos = max(pos, idx)
            cnt = min(len(issued) - 1 - pos, 15) if pos >= 0 else None
            # a wait is needed only if it asks for something an earlier wait has not already covered (LDS returns in order)
            if cnt is not None and len(issued) - 1 - cnt <= waited_upto[0]:
                cnt = None
            if cnt is not None and WAIT_LOOK:
                # fewer s_waitcnt: one wait may also cover what the next WAIT_LOOK MFMAs need, as far as those reads have been
                # in flight for WAIT_AGE MFMAs.  Without the age limit it halves the waits and is 2-3 % SLOWER (the merged
                # wait stalls on reads issued a moment ago); with it, about 1 % faster in the fused backward, which sets
                # its own values.  Off (0) for the dQ and dK/dV bodies.
                for g2 in range(g + 1, g + 1 + WAIT_LOOK):
                    p2 = period + g2 // NS
                    for k in mfma[g2 % NS][1]:
                        hits = [i for i, (p, kk) in enumerate(issued) if kk == k and p == p2]
                        # only reads that have been in flight for WAIT_AGE MFMAs or more: younger ones may not have landed
                        if hits and issue_gap[hits[-1]] <= period * NS + g - WAIT_AGE:
                            pos = max(pos, hits[-1])
                cnt = min(len(issued) - 1 - pos, 15)
            if cnt is not None:
                waited_upto[0] = len(issued) - 1 - cnt
            if period == 1:
                if cnt is not None:
                    lines.append(f"s_waitcnt lgkmcnt({cnt})")
                lines.append(text)
            own, nxt = gap_items(g)
            for t in own:
                if t.kind == "lds":
                    issued.append((period, t.key))
                    issue_gap.append(period * NS + g)
                if period == 1:
                    lines.append(t.text)
            for t in nxt:
                if t.kind == "lds":
                    issued.append((period + 1, t.key))
                    issue_gap.append(period * NS + g)
                if period == 1:
                    lines.append("@N " + t.text)
    # prologue = the wrapped tasks alone, in the same order
    pro = []
    for g in range(NS):
        for t in per_gap.get(g - NS, []):
            pro.append("@N " + t.text)
    return lines, pro


def render(D, masked):
    R, mfma, tasks, per_gap, load, NS = schedule(D, masked)
    lines, pro = render_lines(mfma, per_gap, NS)
    return R, lines, pro, load, NS


def resolve(lines, D, buf, sh, first_next_barrier):
    """Substitutes the placeholders for sub-tile (buf, sh); lines tagged '@N ' get the next sub-tile's bases.  In the body of
    sub-tile 1 the first '@N' line is preceded by the wait + barrier that make the other buffer readable."""
    ROWB = 2 * D
    TILEB, HALFB = 64 * ROWB, 32 * ROWB
    BUFB = 2 * TILEB + 512
    out = []
    barrier_done = not first_next_barrier

    def bases(b, s):
        return {"Q": b * BUFB + s * HALFB, "G": b * BUFB + TILEB + s * HALFB, "RC": b * BUFB + 128 * s, "RR": 32 * s}
    cur = bases(buf, sh)
    nxt = bases(buf, 1) if sh == 0 else bases(buf ^ 1, 0)
    for l in lines:
        b = cur
        if l.startswith("@N "):
            l = l[3:]
            b = nxt
            if not barrier_done:
                out.append("s_waitcnt vmcnt(0)")
                out.append("s_barrier")
                barrier_done = True
        l = re.sub(r"@(Q|G|RC|RR)\+(\d+)", lambda m: str(b[m.group(1)] + int(m.group(2))), l)
        out.append(l)
    return out


def c_string(lines):
    return " \\\n".join('    "' + l.replace("\n\t", "\\n\\t") + '\\n\\t"' for l in lines)


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--check", action="store_true")
    ap.add_argument("--out", default=os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", "cuda_flashattention_amd", "csrc",
                                                  "fa2_bwd_dkdv_body.inc"))
    args = ap.parse_args()
    chunks = ["// GENERATED by tools/gen_dkdv_body.py -- do not edit.  Hand-placed main-loop bodies of fa2_bwd_dkdv_kernel:\n"
              "// FA2_DKDV_BODY_D<d>_B<buffer>_S<sub-tile>_M<masked> and the prologues FA2_DKDV_PRO_D<d>_M<masked> (the early\n"
              "// reads of the very first sub-tile).  Register map and schedule: tools/gen_dkdv_body.py.\n"]
    for D in (128, 64):
        R0 = Regs(D)
        assert sorted(render(D, False)[2]) == sorted(render(D, True)[2]), "plain and masked bodies must leave the same reads in flight"
        chunks.append(f"#define FA2_DKDV_D{D}_KF {R0.KF}\n#define FA2_DKDV_D{D}_ROFFV {R0.ROFFV}\n#define FA2_DKDV_D{D}_VEND {R0.VEND}\n"
                      f"#define FA2_DKDV_D{D}_A_DK {R0.A_DK}\n#define FA2_DKDV_D{D}_A_DV {R0.A_DV}\n")
        for masked in (0, 1):
            R, lines, pro, load, NS = render(D, bool(masked))
            if args.check:
                nm = sum("v_mfma" in l for l in lines)
                print(f"D={D} masked={masked}: {len(lines)} lines, {nm} MFMAs, {len(pro)} early reads, max gap load {max(load)}, "
                      f"{sum(l > GAP_BUDGET for l in load)} of {NS} gaps over {GAP_BUDGET}")
                print("   load:", " ".join(f"{l}" for l in load))
            p = resolve(pro, D, 1, 1, False)          # 'next' of (buffer 1, sub-tile 1) is (buffer 0, sub-tile 0): the first one
            p.append("s_waitcnt lgkmcnt(0)")          # in steady state the previous body's last waits cover these reads
            chunks.append(f"#define FA2_DKDV_PRO_D{D}_M{masked} \\\n" + c_string(p) + "\n")
            for buf in (0, 1):
                for sh in (0, 1):
                    body = resolve(lines, D, buf, sh, sh == 1)
                    chunks.append(f"#define FA2_DKDV_BODY_D{D}_B{buf}_S{sh}_M{masked} \\\n" + c_string(body) + "\n")
    disp = ["// dispatch: the body of sub-tile (B, S) of a tile, plain or masked\n"
            "#define FA2_DKDV_OPS_128 [r0] \"v\"(roff[0]), [r1] \"v\"(roff[1]), [r2] \"v\"(roff[2]), [r3] \"v\"(roff[3]), [r4] \"v\"(roff[4]), "
            "[r5] \"v\"(roff[5]), [r6] \"v\"(roff[6]), [r7] \"v\"(roff[7]), [t0] \"v\"(toff[0]), [t1] \"v\"(toff[1]), [t2] \"v\"(toff[2]), "
            "[t3] \"v\"(toff[3]), [t4] \"v\"(toff[4]), [t5] \"v\"(toff[5]), [t6] \"v\"(toff[6]), [t7] \"v\"(toff[7])\n"
            "#define FA2_DKDV_OPS_64 [r0] \"v\"(roff[0]), [r1] \"v\"(roff[1]), [r2] \"v\"(roff[2]), [r3] \"v\"(roff[3]), [t0] \"v\"(toff[0]), "
            "[t1] \"v\"(toff[1]), [t2] \"v\"(toff[2]), [t3] \"v\"(toff[3])\n"]
    chunks.append("".join(disp))
    if not args.check:
        with open(args.out, "w") as f:
            f.write("\n".join(chunks))
        print("wrote", args.out)


if __name__ == "__main__":
    main()
