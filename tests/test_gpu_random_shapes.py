"""GPU: a seeded sweep over odd shapes -- sequence lengths that are not multiples of any tile size, head counts
that defeat the XCD-aware block order, causal on and off, both head sizes -- for the bf16 forward/backward and
the fp8 forward, against the oracle.  Catches tail-tile, masking and padding mistakes the fixed cases miss."""
import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu


def _rel(a, b):
    return float(np.linalg.norm(a.astype(np.float64) - b) / max(np.linalg.norm(b), 1e-30))


def _shapes(seed, count, dims):
    rng = np.random.default_rng(seed)
    out = []
    for _ in range(count):
        out.append((int(rng.integers(1, 3)), int(rng.integers(1, 6)), int(rng.integers(1, 700)), int(rng.choice(dims)),
                    bool(rng.integers(0, 2))))
    return out


def test_bf16_forward_backward_random_shapes():
    import cuda_flashattention_amd as fa
    import oracle
    f = lambda t: t.float().cpu().numpy()
    for i, (B, H, N, d, causal) in enumerate(_shapes(2024, 24, (64, 128))):
        g = torch.Generator().manual_seed(100 + i)
        mk = lambda s: ((torch.rand(B, H, N, d, generator=g) - 0.5) * s).bfloat16()
        Q, K, V, dO = mk(1.0), mk(1.0), mk(1.0), mk(0.4)
        s = 1.0 / d ** 0.5
        O, L = fa.flash_attention_2_forward(Q.cuda(), K.cuda(), V.cuda(), s, causal=causal)
        dQ, dK, dV = fa.flash_attention_2_backward(Q.cuda(), K.cuda(), V.cuda(), O, L, dO.cuda(), s, causal=causal)
        torch.cuda.synchronize()
        Or, Lr = oracle.attention_forward(f(Q), f(K), f(V), s, causal=causal)
        gr = oracle.attention_backward(f(Q), f(K), f(V), f(dO), s, causal=causal)
        tag = f"case {i}: B{B} H{H} N{N} d{d} causal={causal}"
        assert np.isfinite(f(O)).all() and np.isfinite(f(dQ)).all() and np.isfinite(f(dK)).all() and np.isfinite(f(dV)).all(), tag
        assert _rel(f(O), Or) <= 5e-3, tag
        assert np.abs(L.cpu().numpy() - Lr).max() <= 1e-4, tag
        for name, got, ref in (("dQ", dQ, gr[0]), ("dK", dK, gr[1]), ("dV", dV, gr[2])):
            assert _rel(f(got), ref) <= 5e-3, f"{tag} {name}"


def test_bf16_backward_random_shapes_of_the_single_kernel():
    """The same sweep over the shapes fa2_backward gives to its single five-product kernel: d = 128, seq_len a multiple of
    256 (1 .. 12 key blocks per head) or -- every other case -- a ragged length the padding rule admits (897 .. 3072, checked
    with fa2_backward_plan), any batch and head count, causal on and off, value scales that move the softmax."""
    import ctypes
    import cuda_flashattention_amd as fa
    import oracle
    f = lambda t: t.float().cpu().numpy()
    rng = np.random.default_rng(909)
    for i in range(14):
        B, H, N = int(rng.integers(1, 3)), int(rng.integers(1, 12)), 256 * int(rng.integers(1, 13))
        if i % 2:
            N = int(rng.integers(897, 3073))
        causal, amp = bool(rng.integers(0, 2)), float(rng.choice([0.5, 1.0, 3.0]))
        d = 128
        assert fa._capi.lib().fa2_backward_plan(B, H, N, d, 0, int(causal), ctypes.POINTER(ctypes.c_char_p)()) == 1, N
        g = torch.Generator().manual_seed(500 + i)
        mk = lambda s: ((torch.rand(B, H, N, d, generator=g) - 0.5) * s).bfloat16()
        Q, K, V, dO = mk(amp), mk(amp), mk(1.0), mk(0.4)
        s = 1.0 / d ** 0.5
        O, L = fa.flash_attention_2_forward(Q.cuda(), K.cuda(), V.cuda(), s, causal=causal)
        dQ, dK, dV = fa.flash_attention_2_backward(Q.cuda(), K.cuda(), V.cuda(), O, L, dO.cuda(), s, causal=causal)
        torch.cuda.synchronize()
        gr = oracle.attention_backward(f(Q), f(K), f(V), f(dO), s, causal=causal)
        tag = f"case {i}: B{B} H{H} N{N} causal={causal} amp={amp}"
        for name, got, ref in (("dQ", dQ, gr[0]), ("dK", dK, gr[1]), ("dV", dV, gr[2])):
            assert np.isfinite(f(got)).all(), f"{tag} {name}"
            assert _rel(f(got), ref) <= 5e-3, f"{tag} {name} {_rel(f(got), ref)}"


def test_fp8_forward_random_shapes():
    import cuda_flashattention_amd as fa
    import oracle
    f = lambda t: t.float().cpu().numpy()
    for i, (B, H, N, d, causal) in enumerate(_shapes(77, 16, (128,))):
        g = torch.Generator().manual_seed(300 + i)
        mk = lambda: (torch.rand(B, H, N, d, generator=g) - 0.5).to(torch.float8_e4m3fn)
        Q, K, V = mk(), mk(), mk()
        s = 1.0 / d ** 0.5
        O, L = fa.flash_attention_2_forward(Q.cuda(), K.cuda(), V.cuda(), s, causal=causal)
        torch.cuda.synchronize()
        Or, Lr = oracle.attention_forward(f(Q), f(K), f(V), s, causal=causal)
        tag = f"case {i}: B{B} H{H} N{N} causal={causal}"
        assert np.isfinite(f(O)).all(), tag
        assert _rel(f(O), Or) <= 5e-2, tag
        assert np.abs(L.cpu().numpy() - Lr).max() <= 1e-4, tag


def test_forward_random_long_shapes_and_amplitudes():
    """Forward only, at lengths where the rounds without lane maxima run (bf16: every tile after a workgroup's first; fp8:
    the rounds the key-norm bound clears), with input amplitudes that move the softmax from flat to peaked (scores up to
    ~+-40 natural units at amplitude 6), ragged lengths, causal on and off -- bf16 at both head sizes and fp8, against the
    oracle.  L keeps the 1e-4 gate at every amplitude for bf16; fp8 at amplitude <= 1 (its inputs are e4m3: larger logits
    carry larger absolute error)."""
    import cuda_flashattention_amd as fa
    import oracle
    f = lambda t: t.float().cpu().numpy()
    rng = np.random.default_rng(4242)
    for i in range(14):
        B, H, N = 1, int(rng.integers(1, 4)), int(rng.integers(700, 5000))
        kind = ("bf16_128", "bf16_64", "fp8")[i % 3]
        d = 64 if kind == "bf16_64" else 128
        dt = torch.float8_e4m3fn if kind == "fp8" else torch.bfloat16
        causal = bool(rng.integers(0, 2))
        amp = float(rng.choice([0.5, 1.0]) if kind == "fp8" else rng.choice([0.5, 1.0, 3.0, 6.0]))
        g = torch.Generator().manual_seed(500 + i)
        mk = lambda s: ((torch.rand(B, H, N, d, generator=g) - 0.5) * s).to(dt)
        Q, K, V = mk(amp), mk(amp), mk(1.0)
        s = 1.0 / d ** 0.5
        O, L = fa.flash_attention_2_forward(Q.cuda(), K.cuda(), V.cuda(), s, causal=causal)
        torch.cuda.synchronize()
        Or, Lr = oracle.attention_forward(f(Q), f(K), f(V), s, causal=causal)
        tag = f"case {i}: {kind} H{H} N{N} causal={causal} amp={amp}"
        assert np.isfinite(f(O)).all(), tag
        assert _rel(f(O), Or) <= (5e-2 if kind == "fp8" else 5e-3), tag
        assert np.abs(L.cpu().numpy() - Lr).max() <= 1e-4, tag

