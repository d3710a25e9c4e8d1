"""GPU: the single-kernel (five-product) backward of csrc/fa2_bwd_fused.hip, through the C ABI.

fa2_backward takes it for bf16, head_dim 128, seq_len % 256 == 0 (causal or not); everything else stays on the dQ and dK/dV
kernels.  Checked here:
  * against the CPU oracle (rel-L2 <= 5e-3 per tensor, the bf16 bar of test_gpu_parity.py), for both ways of summing dQ
    over the key blocks (mode 1: ordered hand-off -- what fa2_backward uses; mode 0: fp32 atomics);
  * dK and dV BIT-EQUAL to the two-kernel backward (same products in the same order), dQ within bf16 rounding of it
    (a different, but fixed, summation order);
  * mode 1 is bit-reproducible run to run (no floating-point atomics; the order of the additions is fixed);
  * chains longer than an XCD has CUs (seq_len 16384: 64 key blocks), head counts that are not a multiple of the XCD count,
    more heads than the grid has workgroups;
  * the causal form (reversed sub-tile order, sums handed down to key block 0, masked bodies around the diagonal) against
    the oracle, against the two-kernel form, and run to run;
  * the dispatch of fa2_backward / fa2_backward_phases and the status codes of the explicit entry point."""
import ctypes

import numpy as np
import pytest

pytestmark = pytest.mark.gpu

torch = pytest.importorskip("torch")

BF16_REL = 5e-3


def _fa():
    import cuda_flashattention_amd as fa
    return fa


def rel(a, b):
    return float(np.linalg.norm(a.astype(np.float64) - b) / max(np.linalg.norm(b), 1e-30))


def f32(t):
    return t.float().cpu().numpy()


def make(B, H, N, d, seed, scale=1.0):
    g = torch.Generator().manual_seed(seed)
    return ((torch.rand(B, H, N, d, generator=g) - 0.5) * scale).to(torch.bfloat16)


P = lambda t: ctypes.c_void_p(t.data_ptr())

_hooks = None


def hooks_lib():
    """tests/loopback/libfa2_mi355x_hooks.so: the core library with fa2_bwd_fused.hip compiled -DFA2_TEST_HOOKS (csrc/Makefile,
    target `hooks`).  The PRODUCT library has no fault-injection or grid switch at all (test_capi_symbols.py); this second
    build exports fa2_test_set_fused_hooks(fault, grid) and is bound here with the product's own signature table."""
    global _hooks
    if _hooks is None:
        import os
        from cuda_flashattention_amd import _capi
        path = os.path.join(os.path.dirname(os.path.abspath(__file__)), "loopback", "libfa2_mi355x_hooks.so")
        h = ctypes.CDLL(path)
        for name, (res, args) in _capi.SIGNATURES.items():
            fn = getattr(h, name)
            fn.restype, fn.argtypes = res, args
        h.fa2_test_set_fused_hooks.restype = None
        h.fa2_test_set_fused_hooks.argtypes = [ctypes.c_int, ctypes.c_int]
        h.fa2_test_last_fused_grid.restype = ctypes.c_int
        _hooks = h
    return _hooks


def backward_via(lib, Q, K, V, O, L, dO, scale, causal, ws, out):
    B, H, N, d = Q.shape
    st = lib.fa2_backward(P(Q), P(K), P(V), P(O), P(L), P(dO), P(out[0]), P(out[1]), P(out[2]), B, H, N, d, scale, 0, 1 if causal else 0,
                          P(ws), ws.numel(), ctypes.c_void_p(torch.cuda.current_stream().cuda_stream))
    assert st == 0, st


def fused(Q, K, V, O, L, dO, scale, mode, ws=None):
    lib = _fa()._capi.lib()
    B, H, N, d = Q.shape
    dQ, dK, dV = torch.empty_like(Q), torch.empty_like(Q), torch.empty_like(Q)
    nb = lib.fa2_backward_fused_workspace_bytes(B, H, N, d)
    assert nb == lib.fa2_backward_workspace_bytes(B, H, N, d, 0) > 0
    ws = torch.empty(nb, dtype=torch.uint8, device="cuda") if ws is None else ws
    st = lib.fa2_backward_fused(P(Q), P(K), P(V), P(O), P(L), P(dO), P(dQ), P(dK), P(dV), B, H, N, d, scale, mode, P(ws), nb,
                                ctypes.c_void_p(torch.cuda.current_stream().cuda_stream))
    assert st == 0, st
    return dQ, dK, dV


def case(B, H, N, seed=0):
    d = 128
    Q, K, V, dO = make(B, H, N, d, seed), make(B, H, N, d, seed + 1), make(B, H, N, d, seed + 2), make(B, H, N, d, seed + 3, 0.4)
    scale = 1.0 / d ** 0.5
    dev = [t.cuda() for t in (Q, K, V, dO)]
    O, L = _fa().flash_attention_2_forward(dev[0], dev[1], dev[2], scale)
    return (Q, K, V, dO), dev, O, L, scale


@pytest.mark.parametrize("mode", [1, 0])
@pytest.mark.parametrize("B,H,N", [
    (1, 1, 256),       # one key block: no hand-off at all
    (1, 2, 512),
    (2, 8, 1024),
    (1, 3, 768),       # 3 heads: not a multiple of the XCD count
    (1, 9, 2048),
    (1, 70, 256),      # more heads than one round of the per-XCD queues
    (3, 100, 512),     # 300 heads: ~37 chains per queue, a head count that is no multiple of anything
])
def test_fused_backward_vs_oracle(B, H, N, mode):
    import oracle
    host, dev, O, L, scale = case(B, H, N, seed=11 * N + H)
    got = fused(dev[0], dev[1], dev[2], O, L, dev[3], scale, mode)
    torch.cuda.synchronize()
    want = oracle.attention_backward(*[f32(t) for t in host], scale)
    for name, g, w in zip(("dQ", "dK", "dV"), got, want):
        assert np.isfinite(f32(g)).all(), name
        assert rel(f32(g), w) <= BF16_REL, (name, rel(f32(g), w))


@pytest.mark.parametrize("B,H,N,causal", [
    (1, 2, 1000, False),     # padded to 1024: the last key block holds 24 keys that do not exist, the last sub-tile 24 such rows
    (1, 3, 2049, False),     # one key and one row past a multiple of 256
    (2, 2, 897, False),      # the smallest N above 640 the rule admits (5 x 1024 <= 7 x 960 = 6720)
    (1, 2, 1000, True),
    (1, 2, 1279, True),
])
def test_ragged_lengths_run_the_single_kernel(B, H, N, causal):
    """seq_len not a multiple of 256 (round 3): fa2_backward pads the loops to the next multiple -- keys past the end masked by
    the masked body variant, rows past the end given row constants that make P vanish, nothing past the end stored -- and stays on
    the five-product kernel whenever that is cheaper than the two extra products of the two-kernel form.  Against the oracle, bit-
    reproducible with stale workspace contents, NaN sentinels around every output intact, and finite at the extremes that would
    turn an unmasked padding key into inf x 0 (scores around -100: L < -88)."""
    import oracle
    fa = _fa()
    lib = fa._capi.lib()
    d = 128
    why = ctypes.c_char_p()
    assert lib.fa2_backward_plan(B, H, N, d, 0, 1 if causal else 0, ctypes.byref(why)) == 1, why.value
    host = [make(B, H, N, d, 5 * N + i, 0.4 if i == 3 else 1.0) for i in range(4)]
    dev = [t.cuda() for t in host]
    scale = d ** -0.5
    O, L = fa.flash_attention_2_forward(dev[0], dev[1], dev[2], scale, causal=causal)
    ws = torch.empty(lib.fa2_backward_workspace_bytes(B, H, N, d, 0), dtype=torch.uint8, device="cuda")
    pad = 64                                                    # NaN sentinels in front of and behind every gradient tensor
    bufs = [torch.full((B * H * N * d + 2 * pad,), float("nan"), dtype=torch.bfloat16, device="cuda") for _ in range(3)]
    out = [b[pad:-pad].view(B, H, N, d) for b in bufs]
    fa.flash_attention_2_backward(dev[0], dev[1], dev[2], O, L, dev[3], scale, causal=causal, dQ=out[0], dK=out[1], dV=out[2], workspace=ws)
    first = [t.clone() for t in out]
    ws.fill_(0x5a)
    fa.flash_attention_2_backward(dev[0], dev[1], dev[2], O, L, dev[3], scale, causal=causal, dQ=out[0], dK=out[1], dV=out[2], workspace=ws)
    torch.cuda.synchronize()
    want = oracle.attention_backward(*[f32(t) for t in host], scale, causal=causal)
    for name, g, g1, b, w in zip(("dQ", "dK", "dV"), out, first, bufs, want):
        assert torch.equal(g, g1), name
        assert np.isfinite(f32(g)).all(), name
        assert rel(f32(g), w) <= BF16_REL, (name, rel(f32(g), w))
        assert bool(torch.isnan(b[:pad].float()).all()) and bool(torch.isnan(b[-pad:].float()).all()), name
    # strongly negative scores: L < -88, so exp(-L) overflows -- what an unmasked key past the end would feed into dQ
    Qn = (3.0 + 0.25 * dev[0].float()).bfloat16()               # scores = -(115 +- 0.5): a soft softmax around a very negative level
    Kn = (-(0.3 + 0.05 * dev[1].float())).bfloat16()
    O2, L2 = fa.flash_attention_2_forward(Qn, Kn, dev[2], 1.0, causal=causal)
    assert float(L2.max()) < -88.0
    g2 = fa.flash_attention_2_backward(Qn, Kn, dev[2], O2, L2, dev[3], 1.0, causal=causal, workspace=ws)
    torch.cuda.synchronize()
    assert all(bool(torch.isfinite(t.float()).all()) for t in g2)
    w2 = oracle.attention_backward(f32(Qn), f32(Kn), f32(dev[2]), f32(dev[3]), 1.0, causal=causal)
    # (dQ = dS K with rows of dS summing to zero and K nearly constant: the product cancels to ~1e-4 of its terms, so bf16
    # dS leaves a few per cent -- the same in the two-kernel form; this part of the test is about inf x 0, not about digits)
    for name, g, w, gate in zip(("dQ", "dK", "dV"), g2, w2, (0.15, 4 * BF16_REL, 4 * BF16_REL)):
        assert rel(f32(g), w) <= gate, (name, rel(f32(g), w))


def test_fa2_backward_takes_the_fused_kernel_and_matches_the_two_kernel_form():
    """fa2_backward (phases 7) on an eligible shape == fa2_backward_fused(mode 1) bit for bit; its dK, dV == the two-kernel
    form (phases 6 after phase 1) bit for bit, its dQ within bf16 rounding of that form's."""
    fa = _fa()
    host, dev, O, L, scale = case(2, 5, 1024, seed=5)
    a = fa.flash_attention_2_backward(dev[0], dev[1], dev[2], O, L, dev[3], scale)
    b = fused(dev[0], dev[1], dev[2], O, L, dev[3], scale, 1)
    c = [torch.empty_like(dev[0]) for _ in range(3)]
    ws = torch.empty(fa._capi.lib().fa2_backward_workspace_bytes(2, 5, 1024, 128, 0), dtype=torch.uint8, device="cuda")
    fa.flash_attention_2_backward(dev[0], dev[1], dev[2], O, L, dev[3], scale, dQ=c[0], dK=c[1], dV=c[2], workspace=ws, phases=1)
    fa.flash_attention_2_backward(dev[0], dev[1], dev[2], O, L, dev[3], scale, dQ=c[0], dK=c[1], dV=c[2], workspace=ws, phases=6)
    torch.cuda.synchronize()
    for x, y in zip(a, b):
        assert torch.equal(x, y)
    assert torch.equal(a[1], c[1]) and torch.equal(a[2], c[2])
    assert rel(f32(a[0]), f32(c[0]).astype(np.float64)) <= 1e-3


def test_ordered_handoff_is_bit_reproducible_and_long_chains_work():
    """seq_len 16384 = 64 key blocks per head, twice the CUs of an XCD: workgroups take a second unit of the same head, whose
    first sub-tiles the chain's tail is still waiting for.  Checked against the ORACLE: one whole head (dQ, dK, dV; the
    oracle's head-parallel fp64 backward, ~0.35 TFLOP of CPU work), plus bit-reproducibility with stale workspace contents
    and bit-equality of dK / dV with the two-kernel form."""
    import oracle
    fa = _fa()
    B, H, N, d = 1, 2, 16384, 128
    g = torch.Generator(device="cuda").manual_seed(3)
    mk = lambda s: ((torch.rand(B, H, N, d, device="cuda", generator=g) - 0.5) * s).bfloat16()
    Q, K, V, dO = mk(1), mk(1), mk(1), mk(0.4)
    scale = d ** -0.5
    O, L = fa.flash_attention_2_forward(Q, K, V, scale)
    ws = torch.empty(fa._capi.lib().fa2_backward_workspace_bytes(B, H, N, d, 0), dtype=torch.uint8, device="cuda")
    r1 = fused(Q, K, V, O, L, dO, scale, 1, ws)
    ws.fill_(0x5a)                                        # stale running sums / flags from an earlier launch must not matter
    r2 = fused(Q, K, V, O, L, dO, scale, 1, ws)
    two = [torch.empty_like(Q) for _ in range(3)]
    fa.flash_attention_2_backward(Q, K, V, O, L, dO, scale, dQ=two[0], dK=two[1], dV=two[2], workspace=ws, phases=1)
    fa.flash_attention_2_backward(Q, K, V, O, L, dO, scale, dQ=two[0], dK=two[1], dV=two[2], workspace=ws, phases=6)
    torch.cuda.synchronize()
    for x, y in zip(r1, r2):
        assert torch.equal(x, y)
    assert torch.equal(r1[1], two[1]) and torch.equal(r1[2], two[2])
    h = 1
    want = oracle.attention_backward_head(f32(Q[0, h]), f32(K[0, h]), f32(V[0, h]), f32(dO[0, h]), scale)
    for name, got, two_k, w in zip(("dQ", "dK", "dV"), r1, two, want):
        assert rel(f32(got[0, h]), w) <= BF16_REL, (name, rel(f32(got[0, h]), w))
        assert rel(f32(two_k[0, h]), w) <= BF16_REL, ("two-kernel " + name, rel(f32(two_k[0, h]), w))
        assert np.abs(f32(got[0, h]) - w).max() < 5e-3, name          # 02_backward/main.cu:292-298


def test_bench_shape_matches_the_two_kernel_form():
    """(4, 16, 8192, 128), the headline shape: 2048 units over a persistent grid of one workgroup per CU."""
    fa = _fa()
    B, H, N, d = 4, 16, 8192, 128
    g = torch.Generator(device="cuda").manual_seed(8)
    mk = lambda s: ((torch.rand(B, H, N, d, device="cuda", generator=g) - 0.5) * s).bfloat16()
    Q, K, V, dO = mk(1), mk(1), mk(1), mk(0.4)
    O, L = fa.flash_attention_2_forward(Q, K, V)
    ws = torch.empty(fa._capi.lib().fa2_backward_workspace_bytes(B, H, N, d, 0), dtype=torch.uint8, device="cuda")
    a = [torch.empty_like(Q) for _ in range(3)]
    b = [torch.empty_like(Q) for _ in range(3)]
    fa.flash_attention_2_backward(Q, K, V, O, L, dO, dQ=a[0], dK=a[1], dV=a[2], workspace=ws)
    fa.flash_attention_2_backward(Q, K, V, O, L, dO, dQ=b[0], dK=b[1], dV=b[2], workspace=ws, phases=1)
    fa.flash_attention_2_backward(Q, K, V, O, L, dO, dQ=b[0], dK=b[1], dV=b[2], workspace=ws, phases=6)
    torch.cuda.synchronize()
    assert torch.equal(a[1], b[1]) and torch.equal(a[2], b[2])
    assert float((a[0].float() - b[0].float()).norm() / b[0].float().norm()) <= 1e-3


@pytest.mark.parametrize("B,H,N", [
    (1, 1, 256),       # one key block: every body masked, no hand-off
    (1, 2, 512),       # two key blocks: a chain of two for the last eight sub-tiles
    (2, 8, 1024),
    (1, 3, 768),       # unit lengths 8, 16, 24 sub-tiles: masked region not aligned to the six-body loop
    (1, 9, 2048),
    (1, 17, 1024),     # 17 heads: the queues take their units in groups of two heads (round 4), and one group has a single head
    (2, 16, 768),      # 32 heads: two full groups per queue
    (3, 100, 512),     # 300 heads: ~19 groups per queue, the last one of some queue partial
    (1, 2, 16384),     # 64 key blocks per head
])
def test_causal_fused_backward(B, H, N):
    """Causal, through fa2_backward: vs the oracle (rel-L2 <= 5e-3; every head up to N = 2048, one whole head at N = 16384),
    bit-reproducible, and within bf16 rounding of the two-kernel form (whose sums over the query tiles run in the opposite
    order: no bit equality here)."""
    import oracle
    fa = _fa()
    d = 128
    small = N <= 2048
    if small:
        host = [make(B, H, N, d, 7 * N + i, 0.4 if i == 3 else 1.0) for i in range(4)]
        dev = [t.cuda() for t in host]
    else:
        g = torch.Generator(device="cuda").manual_seed(N)
        dev = [((torch.rand(B, H, N, d, device="cuda", generator=g) - 0.5) * (0.4 if i == 3 else 1.0)).bfloat16() for i in range(4)]
    scale = d ** -0.5
    O, L = fa.flash_attention_2_forward(dev[0], dev[1], dev[2], scale, causal=True)
    ws = torch.empty(fa._capi.lib().fa2_backward_workspace_bytes(B, H, N, d, 0), dtype=torch.uint8, device="cuda")
    a = [torch.empty_like(dev[0]) for _ in range(3)]
    b = [torch.empty_like(dev[0]) for _ in range(3)]
    c = [torch.empty_like(dev[0]) for _ in range(3)]
    fa.flash_attention_2_backward(dev[0], dev[1], dev[2], O, L, dev[3], scale, causal=True, dQ=a[0], dK=a[1], dV=a[2], workspace=ws)
    ws.fill_(0xa5)
    fa.flash_attention_2_backward(dev[0], dev[1], dev[2], O, L, dev[3], scale, causal=True, dQ=b[0], dK=b[1], dV=b[2], workspace=ws)
    for ph in (1, 6):
        fa.flash_attention_2_backward(dev[0], dev[1], dev[2], O, L, dev[3], scale, causal=True, dQ=c[0], dK=c[1], dV=c[2], workspace=ws,
                                      phases=ph)
    torch.cuda.synchronize()
    for x, y in zip(a, b):
        assert torch.equal(x, y)
    for name, x, y in zip(("dQ", "dK", "dV"), a, c):
        assert np.isfinite(f32(x)).all(), name
        assert rel(f32(x), f32(y).astype(np.float64)) <= 1.5e-3, (name, rel(f32(x), f32(y).astype(np.float64)))
    if small:
        want = oracle.attention_backward(*[f32(t) for t in host], scale, causal=True)
        for name, x, w in zip(("dQ", "dK", "dV"), a, want):
            assert rel(f32(x), w) <= BF16_REL, (name, rel(f32(x), w))
    else:                 # long chains: one whole head against the oracle's head-parallel causal backward
        h = H - 1
        want = oracle.attention_backward_head(*[f32(t[0, h]) for t in dev], scale, causal=True)
        for name, x, y, w in zip(("dQ", "dK", "dV"), a, c, want):
            assert rel(f32(x[0, h]), w) <= BF16_REL, (name, rel(f32(x[0, h]), w))
            assert rel(f32(y[0, h]), w) <= BF16_REL, ("two-kernel " + name, rel(f32(y[0, h]), w))


@pytest.mark.parametrize("causal", [False, True])
def test_any_number_of_resident_workgroups(causal):
    """The hand-off never needs co-residency: a unit only waits for units taken from its queue before it.  With the grid
    forced down to 1, 2, 3, 5 and 11 workgroups (test build's fa2_test_set_fused_hooks; 24 units here, up to three per queue in
    flight) the kernel must finish and produce the SAME BITS as the product library with one workgroup per CU -- a single
    workgroup walks every unit of every XCD's queue in order, so any unit that waited for a later one would spin into its
    bounded-poll error (NaNs)."""
    fa = _fa()
    hl = hooks_lib()
    B, H, N, d = 1, 3, 2048, 128
    host, dev, O, L, scale = case(B, H, N, seed=17)
    if causal:
        O, L = fa.flash_attention_2_forward(dev[0], dev[1], dev[2], scale, causal=True)
    ws = torch.empty(fa._capi.lib().fa2_backward_workspace_bytes(B, H, N, d, 0), dtype=torch.uint8, device="cuda")

    def run(lib):
        out = [torch.empty_like(dev[0]) for _ in range(3)]
        backward_via(lib, dev[0], dev[1], dev[2], O, L, dev[3], scale, causal, ws, out)
        torch.cuda.synchronize()
        return out

    ref = run(fa._capi.lib())
    assert all(bool(torch.isfinite(t.float()).all()) for t in ref)
    try:
        for wgs in (1, 2, 3, 5, 11):
            hl.fa2_test_set_fused_hooks(0, wgs)
            got = run(hl)
            assert hl.fa2_test_last_fused_grid() == wgs          # the test build's own launcher and kernels ran, at this grid
            for a, b in zip(got, ref):
                assert torch.equal(a, b), wgs
    finally:
        hl.fa2_test_set_fused_hooks(0, 0)


def test_causal_unit_groups_with_few_workgroups():
    """Round 4's causal unit order: a queue takes its units in groups of two heads, key-block-major inside a group; heads go to
    a queue's chains in chain order; a ticket whose chain got no head (the odd head's partner in the last group) is skipped and
    the first chain of a group without a head ends the queue.  17 heads x 4 key blocks with the grid forced down to 1, 2, 5
    and 11 workgroups (test build): few queues then take MANY groups each, the partial group lands in a different place every
    time, and the bits must equal the full grid's -- which test_causal_fused_backward checks against the oracle."""
    fa = _fa()
    hl = hooks_lib()
    B, H, N, d = 1, 17, 1024, 128
    host, dev, O, L, scale = case(B, H, N, seed=53)
    O, L = fa.flash_attention_2_forward(dev[0], dev[1], dev[2], scale, causal=True)
    ws = torch.empty(fa._capi.lib().fa2_backward_workspace_bytes(B, H, N, d, 0), dtype=torch.uint8, device="cuda")

    def run(lib):
        out = [torch.full_like(dev[0], float("nan")) for _ in range(3)]
        backward_via(lib, dev[0], dev[1], dev[2], O, L, dev[3], scale, True, ws, out)
        torch.cuda.synchronize()
        assert lib.fa2_backward_status(P(ws), ws.numel(), B, H, N, d, 0, ctypes.c_void_p(torch.cuda.current_stream().cuda_stream)) == 0
        return out

    ref = run(fa._capi.lib())
    assert all(bool(torch.isfinite(t.float()).all()) for t in ref)
    try:
        for wgs in (1, 2, 5, 11):
            hl.fa2_test_set_fused_hooks(0, wgs)
            got = run(hl)
            assert hl.fa2_test_last_fused_grid() == wgs
            for a, b in zip(got, ref):
                assert torch.equal(a, b), wgs
    finally:
        hl.fa2_test_set_fused_hooks(0, 0)


@pytest.mark.parametrize("causal", [False, True])
@pytest.mark.parametrize("B,H,N", [
    (1, 1, 256),       # one key block: no hand-off
    (1, 2, 512),
    (2, 8, 1024),
    (1, 3, 768),
    (1, 9, 2048),
    (1, 17, 1024),     # causal: a partial group of heads in one queue
    (1, 3, 1000),      # ragged: padded to 1024 (the last key block's bodies masked, rows past the end given vanishing constants)
    (2, 2, 3400),      # ragged, 14 key blocks
    (1, 2, 16384),     # 64 key blocks per head
])
def test_fused_backward_head_dim_64(B, H, N, causal):
    """Round 4: head_dim 64 at aligned lengths runs the single five-product kernel too (its own generated bodies: 40 MFMAs per
    sub-tile; the dQ tile's two key halves are two running sums, added by the output pass).  Through fa2_backward, which must
    say so (fa2_backward_plan = 1): against the oracle (whole tensors up to N = 2048, one whole head at N = 16384), bit-
    reproducible with a dirty workspace, and within bf16 rounding of the two-kernel form."""
    import oracle
    fa = _fa()
    lib = fa._capi.lib()
    d = 64
    why = ctypes.c_char_p()
    assert lib.fa2_backward_plan(B, H, N, d, 0, 1 if causal else 0, ctypes.byref(why)) == 1, why.value
    g = torch.Generator(device="cuda").manual_seed(64 * N + H)
    dev = [((torch.rand(B, H, N, d, device="cuda", generator=g) - 0.5) * (0.4 if i == 3 else 1.0)).bfloat16() for i in range(4)]
    scale = d ** -0.5
    O, L = fa.flash_attention_2_forward(dev[0], dev[1], dev[2], scale, causal=causal)
    ws = torch.empty(lib.fa2_backward_workspace_bytes(B, H, N, d, 0), dtype=torch.uint8, device="cuda")
    a = [torch.full_like(dev[0], float("nan")) for _ in range(3)]
    b = [torch.full_like(dev[0], float("nan")) for _ in range(3)]
    c = [torch.empty_like(dev[0]) for _ in range(3)]
    fa.flash_attention_2_backward(dev[0], dev[1], dev[2], O, L, dev[3], scale, causal=causal, dQ=a[0], dK=a[1], dV=a[2], workspace=ws)
    ws.fill_(0xa5)
    fa.flash_attention_2_backward(dev[0], dev[1], dev[2], O, L, dev[3], scale, causal=causal, dQ=b[0], dK=b[1], dV=b[2], workspace=ws)
    assert lib.fa2_backward_status(P(ws), ws.numel(), B, H, N, d, 0, ctypes.c_void_p(torch.cuda.current_stream().cuda_stream)) == 0
    for ph in (1, 6):
        fa.flash_attention_2_backward(dev[0], dev[1], dev[2], O, L, dev[3], scale, causal=causal, dQ=c[0], dK=c[1], dV=c[2], workspace=ws,
                                      phases=ph)
    torch.cuda.synchronize()
    for x, y in zip(a, b):
        assert torch.equal(x, y)
    for name, x, y in zip(("dQ", "dK", "dV"), a, c):
        assert np.isfinite(f32(x)).all(), name
        assert rel(f32(x), f32(y).astype(np.float64)) <= 1.5e-3, (name, rel(f32(x), f32(y).astype(np.float64)))
    if N <= 2048:
        want = oracle.attention_backward(*[f32(t) for t in dev], scale, causal=causal)
        for name, x, w in zip(("dQ", "dK", "dV"), a, want):
            assert rel(f32(x), w) <= BF16_REL, (name, rel(f32(x), w))
    else:
        h = H - 1
        want = oracle.attention_backward_head(*[f32(t[0, h]) for t in dev], scale, causal=causal)
        for name, x, w in zip(("dQ", "dK", "dV"), a, want):
            assert rel(f32(x[0, h]), w) <= BF16_REL, (name, rel(f32(x[0, h]), w))


def test_head_dim_64_with_few_workgroups():
    """The head_dim-64 kernel's hand-off under forced grids of 1, 3 and 11 workgroups (test build): same bits as the full grid."""
    fa = _fa()
    hl = hooks_lib()
    B, H, N, d = 1, 5, 1024, 64
    g = torch.Generator(device="cuda").manual_seed(6464)
    dev = [((torch.rand(B, H, N, d, device="cuda", generator=g) - 0.5) * (0.4 if i == 3 else 1.0)).bfloat16() for i in range(4)]
    scale = d ** -0.5
    ws = torch.empty(fa._capi.lib().fa2_backward_workspace_bytes(B, H, N, d, 0), dtype=torch.uint8, device="cuda")
    for causal in (False, True):
        O, L = fa.flash_attention_2_forward(dev[0], dev[1], dev[2], scale, causal=causal)

        def run(lib):
            out = [torch.full_like(dev[0], float("nan")) for _ in range(3)]
            backward_via(lib, dev[0], dev[1], dev[2], O, L, dev[3], scale, causal, ws, out)
            torch.cuda.synchronize()
            return out
        ref = run(fa._capi.lib())
        assert all(bool(torch.isfinite(t.float()).all()) for t in ref)
        try:
            for wgs in (1, 3, 11):
                hl.fa2_test_set_fused_hooks(0, wgs)
                got = run(hl)
                assert hl.fa2_test_last_fused_grid() == wgs
                for x, y in zip(got, ref):
                    assert torch.equal(x, y), (causal, wgs)
        finally:
            hl.fa2_test_set_fused_hooks(0, 0)


def test_two_launches_on_two_streams_share_the_gpu():
    """Two persistent grids in flight at once (separate workspaces, separate streams), each wanting every CU: whatever the
    dispatcher gives each of them, both must finish and produce the bits of a launch that had the GPU to itself."""
    fa = _fa()
    B, H, N, d = 2, 8, 2048, 128
    a_host, a_dev, aO, aL, scale = case(B, H, N, seed=31)
    b_host, b_dev, bO, bL, _ = case(B, H, N, seed=41)
    nb = fa._capi.lib().fa2_backward_workspace_bytes(B, H, N, d, 0)
    wsa, wsb = (torch.empty(nb, dtype=torch.uint8, device="cuda") for _ in range(2))
    alone_a = fa.flash_attention_2_backward(a_dev[0], a_dev[1], a_dev[2], aO, aL, a_dev[3], scale, workspace=wsa)
    alone_b = fa.flash_attention_2_backward(b_dev[0], b_dev[1], b_dev[2], bO, bL, b_dev[3], scale, causal=False, workspace=wsb)
    torch.cuda.synchronize()
    sa, sb = torch.cuda.Stream(), torch.cuda.Stream()
    outs = []
    for rep in range(4):
        oa = [torch.empty_like(a_dev[0]) for _ in range(3)]
        ob = [torch.empty_like(a_dev[0]) for _ in range(3)]
        with torch.cuda.stream(sa):
            fa.flash_attention_2_backward(a_dev[0], a_dev[1], a_dev[2], aO, aL, a_dev[3], scale, dQ=oa[0], dK=oa[1], dV=oa[2],
                                          workspace=wsa, stream=sa)
        with torch.cuda.stream(sb):
            fa.flash_attention_2_backward(b_dev[0], b_dev[1], b_dev[2], bO, bL, b_dev[3], scale, dQ=ob[0], dK=ob[1], dV=ob[2],
                                          workspace=wsb, stream=sb)
        outs.append((oa, ob))
    torch.cuda.synchronize()
    for oa, ob in outs:
        for x, y in zip(oa, alone_a):
            assert torch.equal(x, y)
        for x, y in zip(ob, alone_b):
            assert torch.equal(x, y)


def test_a_lost_progress_word_ends_in_nans_not_in_a_hang():
    """Fault injection (test build only): key block 1 of every head never publishes its progress.  Key block 2 waits for it
    with a BOUNDED poll (FA2_FUSED_SPIN_LIMIT = 2^22 loads, one constant for the generated bodies and the unit queue), gives
    up, raises the error word -- which ends everybody else's waiting -- and the output pass poisons dQ: the call returns
    within seconds, dQ is all NaN, dK / dV are intact, and fa2_backward_status reports FA2_ERR_HANDOFF_TIMEOUT (-7).
    Without the injection the same call is clean again (the control block is reset by every launch)."""
    import time
    fa = _fa()
    hl = hooks_lib()
    B, H, N, d = 1, 2, 1024, 128                      # four key blocks per head: blocks 2 and 3 depend on block 1
    host, dev, O, L, scale = case(B, H, N, seed=23)
    ws = torch.empty(fa._capi.lib().fa2_backward_workspace_bytes(B, H, N, d, 0), dtype=torch.uint8, device="cuda")
    out = [torch.empty_like(dev[0]) for _ in range(3)]
    stream = ctypes.c_void_p(torch.cuda.current_stream().cuda_stream)
    status = lambda lib: lib.fa2_backward_status(P(ws), ws.numel(), B, H, N, d, 0, stream)
    try:
        hl.fa2_test_set_fused_hooks(1, 0)
        t0 = time.perf_counter()
        backward_via(hl, dev[0], dev[1], dev[2], O, L, dev[3], scale, False, ws, out)       # the launch itself returns FA2_OK
        assert status(hl) == -7                                                              # ... the status call does not
        assert time.perf_counter() - t0 < 60.0
        assert bool(torch.isnan(out[0].float()).all())                   # dQ poisoned ...
        assert bool(torch.isfinite(out[1].float()).all()) and bool(torch.isfinite(out[2].float()).all())   # ... dK, dV never depended on it
    finally:
        hl.fa2_test_set_fused_hooks(0, 0)
    backward_via(hl, dev[0], dev[1], dev[2], O, L, dev[3], scale, False, ws, out)
    assert status(hl) == 0
    assert bool(torch.isfinite(out[0].float()).all())
    ref = [torch.empty_like(dev[0]) for _ in range(3)]
    backward_via(fa._capi.lib(), dev[0], dev[1], dev[2], O, L, dev[3], scale, False, ws, ref)   # the product build: same bits
    assert status(fa._capi.lib()) == 0
    for a, b in zip(out, ref):
        assert torch.equal(a, b)


def test_status_describes_the_last_backward_on_the_workspace():
    """fa2_backward_status reads an error word in the workspace: every backward on that workspace must leave it describing
    ITSELF.  A stale 1 (a fresh torch.empty, or an earlier timed-out launch) followed by a run of the two kernels
    (fa2_backward_phases(..., 6)) or of the atomics form (fa2_backward_fused mode 0) -- neither has a hand-off -- must read OK,
    not FA2_ERR_HANDOFF_TIMEOUT."""
    fa = _fa()
    lib = fa._capi.lib()
    B, H, N, d = 1, 2, 1024, 128
    host, dev, O, L, scale = case(B, H, N, seed=29)
    nb = lib.fa2_backward_workspace_bytes(B, H, N, d, 0)
    out = [torch.empty_like(dev[0]) for _ in range(3)]
    stream = ctypes.c_void_p(torch.cuda.current_stream().cuda_stream)
    for run in ("two_kernel", "atomics"):
        ws = torch.full((nb,), 0x01, dtype=torch.uint8, device="cuda")        # every int of the control block reads 0x01010101
        if run == "two_kernel":
            for ph in (1, 6):
                assert lib.fa2_backward_phases(P(dev[0]), P(dev[1]), P(dev[2]), P(O), P(L), P(dev[3]), P(out[0]), P(out[1]), P(out[2]),
                                               B, H, N, d, scale, 0, 0, P(ws), nb, stream, ph) == 0
        else:
            assert lib.fa2_backward_fused(P(dev[0]), P(dev[1]), P(dev[2]), P(O), P(L), P(dev[3]), P(out[0]), P(out[1]), P(out[2]),
                                          B, H, N, d, scale, 0, P(ws), nb, stream) == 0
        assert lib.fa2_backward_status(P(ws), nb, B, H, N, d, 0, stream) == 0, run
        assert all(bool(torch.isfinite(t.float()).all()) for t in out)


def test_environment_cannot_inject_faults_or_shrink_the_grid(monkeypatch):
    """Round 2's FA2_FUSED_FAULT / FA2_FUSED_GRID environment switches are gone from the product library: setting them
    changes nothing."""
    fa = _fa()
    B, H, N, d = 1, 2, 1024, 128
    host, dev, O, L, scale = case(B, H, N, seed=29)
    a = fa.flash_attention_2_backward(dev[0], dev[1], dev[2], O, L, dev[3], scale)
    monkeypatch.setenv("FA2_FUSED_FAULT", "1")
    monkeypatch.setenv("FA2_FUSED_GRID", "1")
    b = fa.flash_attention_2_backward(dev[0], dev[1], dev[2], O, L, dev[3], scale)
    torch.cuda.synchronize()
    for x, y in zip(a, b):
        assert bool(torch.isfinite(x.float()).all()) and torch.equal(x, y)


def test_backward_plan_names_the_implementation():
    """fa2_backward_plan: 1 = the single kernel on this device (gfx950, 256 CUs) for eligible shapes, 2 elsewhere, with a reason."""
    lib = _fa()._capi.lib()
    why = ctypes.c_char_p()
    assert lib.fa2_backward_plan(4, 16, 8192, 128, 0, 0, ctypes.byref(why)) == 1 and b"single" in why.value
    assert lib.fa2_backward_plan(4, 16, 8192, 64, 0, 0, ctypes.byref(why)) == 1 and b"single" in why.value      # round 4: head_dim 64, aligned
    assert lib.fa2_backward_plan(4, 16, 8200, 64, 0, 0, ctypes.byref(why)) == 1          # padding to 8448 costs 2 %
    assert lib.fa2_backward_plan(4, 16, 800, 64, 0, 0, ctypes.byref(why)) == 2 and b"two kernels" in why.value and b"head_dim 64" in why.value
    assert lib.fa2_backward_plan(1, 1, 300, 128, 0, 0, ctypes.byref(why)) == 2
    assert lib.fa2_backward_plan(1, 1, 256, 128, 1, 0, ctypes.byref(why)) == 2 and b"fp32" in why.value
    assert lib.fa2_backward_plan(1, 1, 256, 128, 2, 0, None) < 0


def test_status_codes():
    lib = _fa()._capi.lib()
    x = torch.zeros(1, 1, 256, 128, dtype=torch.bfloat16, device="cuda")
    l = torch.zeros(1, 1, 256, device="cuda")
    ws = torch.empty(lib.fa2_backward_workspace_bytes(1, 1, 256, 128, 0), dtype=torch.uint8, device="cuda")
    call = lambda N, d, mode, nb: lib.fa2_backward_fused(P(x), P(x), P(x), P(x), P(l), P(x), P(x), P(x), P(x), 1, 1, N, d, 0.1, mode,
                                                         P(ws), nb, None)
    assert lib.fa2_backward_fused_workspace_bytes(1, 1, 300, 128) == 0
    assert lib.fa2_backward_fused_workspace_bytes(1, 1, 300, 64) == 0          # 512 against 320: two kernels
    assert lib.fa2_backward_fused_workspace_bytes(1, 1, 256, 64) == lib.fa2_backward_fused_workspace_bytes(1, 1, 256, 128)
    assert call(300, 128, 1, ws.numel()) != 0          # not a multiple of 256
    assert call(300, 64, 1, ws.numel()) != 0           # head_dim 64: aligned lengths only
    assert call(256, 64, 0, ws.numel()) != 0           # head_dim 64 has no atomics form
    assert call(256, 128, 2, ws.numel()) != 0          # no such mode
    assert call(256, 128, 1, 1024) != 0                # workspace too small
    # bit 3 of fa2_backward_phases on a shape the single kernel does not take
    x64 = torch.zeros(1, 1, 320, 64, dtype=torch.bfloat16, device="cuda")
    l64 = torch.zeros(1, 1, 320, device="cuda")
    st = lib.fa2_backward_phases(P(x64), P(x64), P(x64), P(x64), P(l64), P(x64), P(x64), P(x64), P(x64), 1, 1, 320, 64, 0.1, 0, 0, P(ws),
                                 ws.numel(), None, 8)
    assert st != 0                                     # head_dim 64, seq_len not a multiple of 256
    for ph in (15, 8 | 2, 8 | 4):                      # bit 3 does not combine with the two-kernel bits
        st = lib.fa2_backward_phases(P(x), P(x), P(x), P(x), P(l), P(x), P(x), P(x), P(x), 1, 1, 256, 128, 0.1, 0, 0, P(ws), ws.numel(), None, ph)
        assert st == -6, (ph, st)
