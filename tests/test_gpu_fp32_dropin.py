"""GPU: the reference-signature drop-ins flash_attention_2_forward / flash_attention_2_backward
(fp32, single head, argument-for-argument the reference's wrappers) on the reference's own
test cases with the reference's own gates, plus the extended fp32 entry points on odd shapes.
Reads like 02_flash_attention_v2_forward/main.cu and 02_flash_attention_v2_backward/main.cu:
CPU result first, device call, max-abs-diff gate."""
import numpy as np
import pytest

pytestmark = pytest.mark.gpu
torch = pytest.importorskip("torch")


def _lib():
    from cuda_flashattention_amd import _capi
    return _capi.lib()


def dev(a):
    return torch.from_numpy(np.ascontiguousarray(a, dtype=np.float32)).cuda()


def ref_forward(Q, K, V, scale):
    """flash_attention_2_forward(d_Q, d_K, d_V, d_O, d_L, seq_len, head_dim, scale) as
    02_forward/main.cu:56-67 calls it: device pointers in, synchronize, copy back."""
    N, d = Q.shape
    dQ_, dK_, dV_ = dev(Q), dev(K), dev(V)
    O = torch.empty(N, d, device="cuda")
    L = torch.empty(N, device="cuda")
    st = _lib().flash_attention_2_forward(dQ_.data_ptr(), dK_.data_ptr(), dV_.data_ptr(),
                                          O.data_ptr(), L.data_ptr(), N, d, float(scale))
    assert st == 0
    torch.cuda.synchronize()
    return O.cpu().numpy(), L.cpu().numpy()


def ref_backward(Q, K, V, O, L, dO, scale):
    N, d = Q.shape
    t = [dev(x) for x in (Q, K, V, O, L, dO)]
    out = [torch.full((N, d), float("nan"), device="cuda") for _ in range(3)]   # must be overwritten
    st = _lib().flash_attention_2_backward(*[x.data_ptr() for x in t], *[x.data_ptr() for x in out],
                                           N, d, float(scale))
    assert st == 0
    torch.cuda.synchronize()
    return [x.cpu().numpy() for x in out]


def test_k0_naive00(golden):
    """00_naive_attention/main.cpp:45-65 through the GPU path (scale 1/sqrt(2)), tol 1e-4."""
    g = golden("k0_naive00.npz")
    O, _ = ref_forward(g["Q"], g["K"], g["V"], 1.0 / np.sqrt(2.0))
    assert np.abs(O - g["expected"]).max() < 1e-4


def test_simple_forward_k1(golden):
    """"Simple test PASSED": 02_forward/main.cu:115-262, N=d=4, scale 1, gate 1e-4 (:247)."""
    g = golden("k1_fwd_simple.npz")
    O, L = ref_forward(g["Q"], g["K"], g["V"], 1.0)
    assert np.abs(O - g["O"]).max() < 1e-4
    assert np.abs(L - g["L"]).max() < 1e-5


def test_random_forward_k3(golden):
    """"Test PASSED": 02_forward/main.cu:12-112, N=512, d=64, gate 5e-3 (:89)."""
    g = golden("k3_fwd_rand.npz")
    O, L = ref_forward(g["Q"], g["K"], g["V"], float(g["scale"]))
    diff = np.abs(O - g["O"])
    assert diff.max() < 5e-3 and np.count_nonzero(diff > 1e-3) == 0    # :75, :89
    assert diff.max() < 2e-6                                            # what the exact-f32 path actually achieves
    assert np.abs(L - g["L"]).max() < 1e-5


def test_simple_backward_k2(golden):
    """"Test Case 1: PASSED": 02_backward/main.cu:51-189, gate 1e-3 each (:172-178)."""
    g = golden("k2_bwd_simple.npz")
    dQ, dK, dV = ref_backward(g["Q"], g["K"], g["V"], g["O"], g["L"], g["dO"], 1.0)
    for got, k in ((dQ, "dQ"), (dK, "dK"), (dV, "dV")):
        assert np.abs(got - g[k]).max() < 1e-3, k
        assert np.abs(got - g[k]).max() < 2e-6, k
    # figures recorded in the reference's notes (IMPLEMENTATION_SUMMARY.md:22-24): ~9e-8


def test_complex_backward_k4(golden):
    """"Test Case 2: PASSED": 02_backward/main.cu:195-309, N=128, d=64, gate 5e-3 (:292-298);
    O and L fed to the GPU come from the CPU forward, as in the reference (:239)."""
    g = golden("k4_bwd_rand.npz")
    dQ, dK, dV = ref_backward(g["Q"], g["K"], g["V"], g["O"], g["L"], g["dO"], float(g["scale"]))
    for got, k in ((dQ, "dQ"), (dK, "dK"), (dV, "dV")):
        assert np.abs(got - g[k]).max() < 5e-3, k
        assert np.abs(got - g[k]).max() < 1e-6, k


def test_ring_pattern_single_gpu_k5(golden):
    """03_attention_1GPU.cu / 04_ring_attention.cu data (N=5096, d=64, scale 1) on one GPU, with
    the reference's criterion compare_outputs(rtol=5e-3, atol=1.0) (attention_helper.h:184)."""
    from oracle import recipes
    g = golden("k5_ring_pattern_rows.npz")
    N, d = int(g["N"]), int(g["d"])
    Q, K, V = recipes.ring_pattern(N, d)
    O, _ = ref_forward(Q, K, V, 1.0)
    assert recipes.compare_outputs(g["O_rows"], O[g["rows"]], rtol=5e-3, atol=1.0) == 0
    assert recipes.compare_outputs(recipes.ring_pattern_expected(N, d), O, rtol=5e-3, atol=1.0) == 0
    # the reference's own fp32 CPU sum over 5096 keys of magnitude 1e4 is off by ~0.6 here;
    # against the closed form the exact-f32 MFMA path is an order of magnitude closer
    exp = recipes.ring_pattern_expected(N, d)
    assert np.abs(O - exp).max() / np.abs(exp).max() < 1e-4      # fp32 accumulation over 5096 keys


@pytest.mark.parametrize("B,H,N,d,causal", [
    (1, 2, 128, 64, False), (2, 3, 200, 20, False), (1, 2, 129, 100, False), (1, 1, 257, 128, False),
    (1, 1, 1, 7, False), (2, 2, 160, 48, True), (1, 1, 333, 128, True),
])
def test_fp32_extended_vs_oracle(B, H, N, d, causal):
    import cuda_flashattention_amd as fa
    import oracle
    rng = np.random.default_rng(N + d)
    Q, K, V = (rng.uniform(-0.5, 0.5, (B, H, N, d)).astype(np.float32) for _ in range(3))
    dO = rng.uniform(-0.2, 0.2, (B, H, N, d)).astype(np.float32)
    s = 1.0 / np.sqrt(d)
    t = [dev(x) for x in (Q, K, V, dO)]
    O, L = fa.flash_attention_2_forward(t[0], t[1], t[2], s, causal=causal)
    dQ, dK, dV = fa.flash_attention_2_backward(t[0], t[1], t[2], O, L, t[3], s, causal=causal)
    torch.cuda.synchronize()
    Or, Lr = oracle.attention_forward(Q, K, V, s, causal=causal)
    gr = oracle.attention_backward(Q, K, V, dO, s, causal=causal)
    assert np.abs(O.cpu().numpy() - Or).max() < 2e-6
    assert np.abs(L.cpu().numpy() - Lr).max() < 5e-6
    for got, want, k in zip((dQ, dK, dV), gr, ("dQ", "dK", "dV")):
        assert np.abs(got.cpu().numpy() - want).max() < 2e-6, k


def test_fa1_literal_cases_through_the_dropin(golden):
    """The FA1 step's literal forward cases (01_flash_attention_v1/main.cu:195-345, d in {1, 2, 4, 32},
    N from 1 to 64) through the reference-signature fp32 forward, with that main's own gate |diff| <= 1e-3
    (:157-163) against the reference's naive_attention outputs (tests/golden/k6_fa1_cases.npz)."""
    import cuda_flashattention_amd as fa
    g = golden("k6_fa1_cases.npz")
    names = sorted({k[:-2] for k in g.files})
    assert len(names) == 8
    for name in names:
        Q, K, V, Oref = (g[name + s] for s in ("_Q", "_K", "_V", "_O"))
        N, d = Q.shape
        O, L = fa.flash_attention_2_forward(torch.from_numpy(Q).cuda(), torch.from_numpy(K).cuda(),
                                            torch.from_numpy(V).cuda(), 1.0 / np.sqrt(d))
        torch.cuda.synchronize()
        assert np.abs(O.cpu().numpy() - Oref).max() <= 1e-3, name


def test_fa1_baseline_reference_cases(golden):
    """flash_attention (the FlashAttention-1 step, 01_flash_attention_v1/main.cu:7-20): the reference's own literal test
    cases (main.cu:195-345, tests/golden/k6_fa1_cases.npz: inputs and the outputs of the reference's naive check) with its
    gate |d| < 1e-4, and l, m consistent with the oracle's log-sum-exp."""
    import ctypes
    import cuda_flashattention_amd as fa
    import oracle
    lib = fa._capi.lib()
    g = golden("k6_fa1_cases.npz")
    names = sorted({k[:-2] for k in g.keys() if k.endswith("_Q")})
    assert len(names) >= 5
    for nm in names:
        Q, K, V, Oref = (np.ascontiguousarray(g[f"{nm}_{t}"], dtype=np.float32) for t in "QKVO")
        N, d = Q.shape
        dev = [torch.from_numpy(a).cuda() for a in (Q, K, V)]
        O = torch.full((N, d), float("nan"), device="cuda")
        l = torch.empty(N, device="cuda")
        m = torch.empty(N, device="cuda")
        _, Lr = oracle.naive_forward_pass(Q, K, V, float(1.0 / np.sqrt(d)))
        # Bc is a test dimension in the reference (its test 8 runs one case at Bc = 1, 2, 4, main.cu:342-344; the others at
        # Bc = 2 .. 16): every case here at Bc = 1, 2, 4, 16, 32 -- the tile length changes the order of the running (l, m, O)
        # updates, not the result -- and one value above the clamp (1000 -> 64).
        for Bc in (1, 2, 4, 16, 32, 1000):
            O.fill_(float("nan"))
            st = lib.flash_attention(dev[0].data_ptr(), dev[1].data_ptr(), dev[2].data_ptr(), O.data_ptr(), l.data_ptr(), m.data_ptr(),
                                     N, d, Bc, 1024)
            assert st == 0
            torch.cuda.synchronize()
            assert np.abs(O.cpu().numpy() - Oref).max() < 1e-4, (nm, Bc)
            assert np.abs((m + torch.log(l)).cpu().numpy() - Lr).max() < 1e-4, (nm, Bc)


def test_fa1_baseline_matches_oracle_medium():
    import cuda_flashattention_amd as fa
    import oracle
    lib = fa._capi.lib()
    rng = np.random.default_rng(3)
    for N, d in ((300, 64), (1000, 128), (77, 5)):
        Q, K, V = (rng.uniform(-0.5, 0.5, (N, d)).astype(np.float32) for _ in range(3))
        dev = [torch.from_numpy(a).cuda() for a in (Q, K, V)]
        O = torch.empty(N, d, device="cuda")
        l = torch.empty(N, device="cuda")
        m = torch.empty(N, device="cuda")
        assert lib.flash_attention(dev[0].data_ptr(), dev[1].data_ptr(), dev[2].data_ptr(), O.data_ptr(), l.data_ptr(), m.data_ptr(),
                                   N, d, 32, 1024) == 0
        torch.cuda.synchronize()
        Or, Lr = oracle.naive_forward_pass(Q, K, V, float(1.0 / np.sqrt(d)))
        assert np.abs(O.cpu().numpy() - Or).max() < 1e-5
        assert np.abs((m + torch.log(l)).cpu().numpy() - Lr).max() < 1e-4
    one = ctypes_void = None
    assert lib.flash_attention(None, None, None, None, None, None, 4, 4, 32, 1024) == -1
    assert lib.flash_attention(dev[0].data_ptr(), dev[1].data_ptr(), dev[2].data_ptr(), O.data_ptr(), l.data_ptr(), m.data_ptr(),
                               77, 5, 0, 1024) == -2          # Bc = 0


def test_read_clocks_brackets_a_stretch_of_work():
    """fa2_read_clocks: per XCC two 64-bit device counters (shader-clock ticks -- the XCC's own counter -- and 100 MHz
    reference ticks); two calls around some work give a plausible mean shader clock on every XCC (what bench.py's
    `sustained` object reports).  The XCCs need not agree exactly: each holds its own clock (measured on a lightly loaded
    stretch: 1106 - 1174 MHz across the eight) -- one more reason never to difference counters of two XCCs."""
    import cuda_flashattention_amd as fa
    lib = fa._capi.lib()
    x = torch.rand(4096, 4096, device="cuda")
    s = torch.cuda.current_stream().cuda_stream
    a = fa.ops.read_clocks()
    for _ in range(200):
        x = x @ x * 1e-3
    b = fa.ops.read_clocks()
    torch.cuda.synchronize()
    ca, cb = a.cpu().tolist(), b.cpu().tolist()
    present = [i for i in range(16) if ca[i][1] and cb[i][1]]
    assert len(present) == 8, present                       # SPX mode: all eight XCCs got a workgroup in both samples
    per = [(cb[i][0] - ca[i][0]) / (cb[i][1] - ca[i][1]) * 100.0 for i in present]
    assert all(cb[i][1] > ca[i][1] for i in present)
    assert all(100.0 < m < 2600.0 for m in per), per
    assert max(per) - min(per) < 0.25 * max(per), per       # eight clock domains, one power budget
    assert abs(fa.ops.mean_shader_clock_mhz(a, b) - sum(per) / 8) < 1e-6
    assert lib.fa2_read_clocks(None, s) == -1


def test_bare_mfma_probe_reports_a_plausible_ceiling():
    """fa2_mfma_probe (measurement aid): every SIMD on nothing but bf16 MFMAs with random operands; what bench.py reports as
    this device's own ceiling.  Between the attention kernels' 1.2 - 1.4 PFLOP/s and the nominal 2.52."""
    import cuda_flashattention_amd as fa
    tf, mhz = fa.ops.bare_mfma_tflops(seconds=0.05)
    assert 1300.0 < tf < 2520.0, tf
    assert 1000.0 < mhz < 2600.0, mhz
    assert 0.85 < tf / (mhz * 1e6 * 256 * 4 * 1024 / 1e12) <= 1.02      # (nearly) back-to-back issue: 1024 flop per clock per SIMD
    lib = fa._capi.lib()
    assert lib.fa2_mfma_probe(None, None, 1, 1, None) == -1
