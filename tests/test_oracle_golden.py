"""CPU: the oracle (oracle/naive_attention.c) against the committed golden vectors, which
were produced by the REFERENCE's own code (oracle/gen_golden.py), and against the literal
known answers quoted in the reference's mains / SURVEY 8c."""
import numpy as np
import pytest

import oracle
from oracle import recipes


def test_k0_naive00_literal(golden):
    """00_naive_attention/main.cpp:45-65: 2x2, four literal outputs, tol 1e-4."""
    g = golden("k0_naive00.npz")
    O = oracle.naive_attention(g["Q"], g["K"], g["V"])
    assert np.abs(O - g["expected"]).max() < 1e-4
    assert np.array_equal(O, g["O"])          # bit-identical to the reference's output
    Q, K, V, exp = recipes.naive00()
    assert np.array_equal(Q, g["Q"]) and np.array_equal(V, g["V"])


def test_k1_fwd_simple(golden):
    """02_forward/main.cu:134-155, N=d=4, scale 1; rows quoted in SURVEY 8c."""
    g = golden("k1_fwd_simple.npz")
    O, L = oracle.naive_forward_pass(g["Q"], g["K"], g["V"], 1.0)
    assert np.array_equal(O, g["O"]) and np.array_equal(L, g["L"])
    np.testing.assert_allclose(O[0], [7, 8, 9, 10], atol=1e-4)
    np.testing.assert_allclose(O[1], [7.9242344, 8.9242344, 9.9242344, 10.9242344], atol=1e-4)
    np.testing.assert_allclose(O[3], [8.7784433, 9.7784433, 10.7784433, 11.7784433], atol=1e-4)
    np.testing.assert_allclose(L, [2.0064087, 2.0064087, 1.7436684, 2.6265235], atol=1e-5)


def test_k2_bwd_simple(golden):
    """02_backward/main.cu:78-107, N=d=4, scale 1."""
    g = golden("k2_bwd_simple.npz")
    dQ, dK, dV = oracle.naive_attention_backward(g["Q"], g["K"], g["V"], g["dO"], 1.0)
    for a, k in ((dQ, "dQ"), (dK, "dK"), (dV, "dV")):
        assert np.array_equal(a, g[k]), k
    np.testing.assert_allclose(dQ[0], [-0.4987866, -0.0086156, 0.1662621, 0.3411399], atol=1e-6)
    np.testing.assert_allclose(dK, dQ.T, atol=1e-6)
    np.testing.assert_allclose(np.diag(dV), 0.4753669, atol=1e-6)
    # the O(N^2 d) form used at scale agrees with the reference's O(N^3) Jacobian form
    f = oracle.attention_backward(g["Q"], g["K"], g["V"], g["dO"], 1.0)
    for a, k in zip(f, ("dQ", "dK", "dV")):
        assert np.abs(a - g[k]).max() < 1e-6, k


def test_k3_fwd_rand(golden):
    """02_forward/main.cu:14-33: N=512, d=64, srand(42)."""
    g = golden("k3_fwd_rand.npz")
    Q, K, V = recipes.fwd_rand(512, 64, 42)
    assert np.array_equal(Q, g["Q"]) and np.array_equal(K, g["K"]) and np.array_equal(V, g["V"])
    np.testing.assert_allclose(Q.ravel()[:3], [-0.334, -0.259, -0.479], atol=1e-6)
    O, L = oracle.naive_forward_pass(Q, K, V, float(g["scale"]))
    assert np.array_equal(O, g["O"]) and np.array_equal(L, g["L"])
    np.testing.assert_allclose(O.ravel()[:4], [-0.0011205, -0.0117837, 0.0200985, 0.0203981], atol=1e-6)
    np.testing.assert_allclose(L[:2], [6.2395129, 6.2424507], atol=1e-5)
    assert abs(float(O.sum(dtype=np.float64)) - 40.619786) < 1e-3
    O64, L64 = oracle.attention_forward(Q, K, V, float(g["scale"]))
    assert np.abs(O64 - g["O"]).max() < 1e-6 and np.abs(L64 - g["L"]).max() < 2e-6


def test_k4_bwd_rand(golden):
    """02_backward/main.cu:200-227: N=128, d=64, srand(42)."""
    g = golden("k4_bwd_rand.npz")
    Q, K, V, dO = recipes.bwd_rand(128, 64, 42)
    for a, k in ((Q, "Q"), (K, "K"), (V, "V"), (dO, "dO")):
        assert np.array_equal(a, g[k]), k
    assert abs(Q.ravel()[0] - (-0.417)) < 1e-6
    s = float(g["scale"])
    O, L = oracle.naive_forward_pass(Q, K, V, s)
    assert np.array_equal(O, g["O"]) and np.array_equal(L, g["L"])
    dQ, dK, dV = oracle.naive_attention_backward(Q, K, V, dO, s)
    for a, k in ((dQ, "dQ"), (dK, "dK"), (dV, "dV")):
        assert np.array_equal(a, g[k]), k
    assert abs(dQ.ravel()[0] - (-1.41408134e-04)) < 1e-9
    assert abs(dK.ravel()[0] - 2.19425987e-04) < 1e-9
    assert abs(dV.ravel()[0] - (-4.07864386e-03)) < 1e-8
    f = oracle.attention_backward(Q, K, V, dO, s)
    for a, k in zip(f, ("dQ", "dK", "dV")):
        assert np.abs(a - g[k]).max() < 1e-6, k


def test_cfg1(golden):
    """BASELINE config 1 (B=1,H=2,N=128,d=64) through the [B][H][N][d] forms."""
    g = golden("cfg1_b1h2n128d64.npz")
    s = float(g["scale"])
    O, L = oracle.attention_forward(g["Q"], g["K"], g["V"], s)
    assert np.abs(O - g["O"]).max() < 1e-6 and np.abs(L - g["L"]).max() < 2e-6
    grads = oracle.attention_backward(g["Q"], g["K"], g["V"], g["dO"], s)
    for a, k in zip(grads, ("dQ", "dK", "dV")):
        assert np.abs(a - g[k]).max() < 1e-6, k


def test_k5_ring_pattern_rows(golden):
    """04_ring_attention.cu:19-21 + create_simple_test_data: stored reference rows vs the
    closed form, under the reference's own criterion (rtol 5e-3 AND atol 1.0)."""
    g = golden("k5_ring_pattern_rows.npz")
    N, d = int(g["N"]), int(g["d"])
    expected = recipes.ring_pattern_expected(N, d)[g["rows"]]
    assert recipes.compare_outputs(g["O_rows"], expected, rtol=5e-3, atol=1.0) == 0
    assert np.abs(g["O_rows"] - expected).max() / np.abs(expected).max() < 1e-4
    # oracle on a strided subset of rows (full N^2 d would take seconds; rows are independent)
    Q, K, V = recipes.ring_pattern(N, d)
    O, _ = oracle.attention_forward(Q, K, V, 1.0, rows=(0, 97))
    sel = np.arange(0, N, 97)
    assert recipes.compare_outputs(recipes.ring_pattern_expected(N, d)[sel], O[sel], rtol=5e-3, atol=1.0) == 0


def test_causal_oracle_matches_masked_dense():
    """Causal masking has no counterpart in the reference (parity unpinned by it): pin the
    oracle's causal path against an explicit masked-softmax in numpy float64."""
    rng = np.random.default_rng(3)
    N, d = 48, 16
    Q, K, V, dO = (rng.standard_normal((N, d)).astype(np.float32) * 0.5 for _ in range(4))
    s = 0.25
    S = (Q.astype(np.float64) @ K.astype(np.float64).T) * s
    S[np.triu_indices(N, 1)] = -np.inf
    P = np.exp(S - S.max(1, keepdims=True))
    P /= P.sum(1, keepdims=True)
    O, L = oracle.attention_forward(Q, K, V, s, causal=True)
    np.testing.assert_allclose(O, P @ V, atol=1e-6)
    dP = dO.astype(np.float64) @ V.astype(np.float64).T
    D = (dO * (P @ V)).sum(1, keepdims=True)
    dS = P * (dP - D)
    dQ, dK, dV = oracle.attention_backward(Q, K, V, dO, s, causal=True)
    np.testing.assert_allclose(dQ, dS @ K * s, atol=1e-6)
    np.testing.assert_allclose(dK, dS.T @ Q * s, atol=1e-6)
    np.testing.assert_allclose(dV, P.T @ dO, atol=1e-6)


def test_ring_step_composes_to_full_attention():
    """oracle_ring_step (ring_attention_kernel.cu:67-137): folding P shards one by one equals
    one-shot attention, for any shard order (the ring visits shards in rank-dependent order)."""
    rng = np.random.default_rng(5)
    N, d, P = 96, 32, 4
    Q, K, V = (rng.standard_normal((N, d)).astype(np.float32) * 0.5 for _ in range(3))
    s = 1.0 / np.sqrt(d)
    Oref, Lref = oracle.naive_forward_pass(Q, K, V, float(s))
    for order in ([0, 1, 2, 3], [2, 1, 0, 3], [3, 2, 1, 0]):
        O = np.zeros_like(Q); L = np.zeros(N, np.float32); M = np.full(N, -np.inf, np.float32)
        for i, blk in enumerate(order):
            sl = slice(blk * N // P, (blk + 1) * N // P)
            oracle.ring_step(Q, np.ascontiguousarray(K[sl]), np.ascontiguousarray(V[sl]), O, L, M,
                             float(s), last=(i == P - 1))
        assert np.abs(O - Oref).max() < 2e-6 and np.abs(L - Lref).max() < 2e-6


@pytest.mark.skipif(not oracle.have_ref(), reason="oracle/_ref not built (needs /root/reference; dev container only)")
def test_oracle_bit_identical_to_reference_build():
    """The restatement vs the reference's own code compiled from /root/reference, on fresh
    random inputs (beyond the committed vectors): bit-identical."""
    R = oracle.ref()
    assert oracle.ref_self_test() == 0
    rng = np.random.default_rng(11)
    for N, d, s in ((37, 20, 0.3), (64, 64, 0.0), (130, 8, 1.0)):
        Q, K, V, dO = (rng.uniform(-1, 1, (N, d)).astype(np.float32) for _ in range(4))
        a = oracle.naive_forward_pass(Q, K, V, s)
        b = oracle.naive_forward_pass(Q, K, V, s, lib=R)
        assert np.array_equal(a[0], b[0]) and np.array_equal(a[1], b[1])
        sc = s if s else 1.0 / np.sqrt(d)
        ga = oracle.naive_attention_backward(Q, K, V, dO, float(sc))
        gb = oracle.naive_attention_backward(Q, K, V, dO, float(sc), lib=R)
        assert all(np.array_equal(x, y) for x, y in zip(ga, gb))
        assert np.array_equal(oracle.naive_attention(Q, K, V), oracle.naive_attention(Q, K, V, lib=R))


def test_k6_fa1_literal_cases(golden):
    """01_flash_attention_v1/main.cu:195-345: the superseded FA1 step's literal forward cases, judged there
    against naive_attention with |diff| <= 1e-3 (extra forward known-answer tests).  The
    oracle reproduces the reference's outputs bit for bit; the recipes reproduce its inputs."""
    g = golden("k6_fa1_cases.npz")
    names = [n for n, *_ in recipes.fa1_cases()]
    assert len(names) == 8 and {k[:-2] for k in g.files} == set(names)
    for name, Q, K, V in recipes.fa1_cases():
        assert np.array_equal(Q, g[name + "_Q"]) and np.array_equal(K, g[name + "_K"]) and np.array_equal(V, g[name + "_V"])
        assert np.array_equal(oracle.naive_attention(Q, K, V), g[name + "_O"])
    # closed-form spot checks of the reference outputs themselves
    assert np.allclose(g["single_element_O"], [[42.0]])
    assert np.allclose(g["uniform_3x2_O"], np.tile([[3.0, 4.0]], (3, 1)), atol=1e-6)      # uniform weights: the mean of V


@pytest.mark.parametrize("causal", [False, True])
def test_head_parallel_backward_equals_the_slab_form(causal):
    """oracle.attention_backward_head (query rows split over threads, private partial sums) against attention_backward
    (one thread per slab): same arithmetic, a different summation order for dK / dV only -- equal to fp64 rounding."""
    rng = np.random.default_rng(21)
    N, d = 203, 24
    Q, K, V, dO = (rng.uniform(-0.5, 0.5, (N, d)).astype(np.float32) for _ in range(4))
    a = oracle.attention_backward(Q, K, V, dO, 0.3, causal=causal)
    b = oracle.attention_backward_head(Q, K, V, dO, 0.3, causal=causal)
    assert np.array_equal(a[0], b[0])                       # dQ: a row is one thread's work either way
    for x, y in zip(a[1:], b[1:]):
        np.testing.assert_allclose(x, y, rtol=0, atol=1e-7)


def test_cpu_baseline_loops_match_the_reference_order_oracle():
    """oracle.fwdbwd_heads is what bench.py times as cpu_baseline: it must BE the naive forward + backward.  O rows equal
    naive_forward_pass bit for bit (same operation order); dQ rows and each thread's dK / dV share equal the O(N^2 d)
    backward restricted to the thread's rows (numpy float64) to fp32 rounding."""
    rng = np.random.default_rng(8)
    BH, N, d, nblk, threads = 2, 96, 16, 2, 3
    Q, K, V, dO = (rng.uniform(-0.5, 0.5, (BH, N, d)).astype(np.float32) for _ in range(4))
    s = 0.25
    rows, O, dQ, dK, dV = oracle.fwdbwd_heads(Q, K, V, dO, s, nblk=nblk, threads=threads)
    sel = oracle.fwdbwd_heads_rows(N, nblk)
    assert rows == threads * nblk * oracle.RB and len(sel) == nblk * oracle.RB
    for t in range(threads):
        h = t % BH
        Oref, _ = oracle.naive_forward_pass(Q[h], K[h], V[h], s)
        assert np.array_equal(O[t], Oref[sel])
        q, k, v, g = (a[h].astype(np.float64) for a in (Q, K, V, dO))
        S = q[sel] @ k.T * s
        P = np.exp(S - S.max(1, keepdims=True)); P /= P.sum(1, keepdims=True)
        dS = P * (g[sel] @ v.T - (g[sel] * (P @ v)).sum(1, keepdims=True))
        np.testing.assert_allclose(dQ[t], dS @ k * s, atol=2e-6)
        np.testing.assert_allclose(dK[t], dS.T @ q[sel] * s, atol=2e-6)
        np.testing.assert_allclose(dV[t], P.T @ g[sel], atol=2e-6)
