"""Dev aid: where is dQ wrong?  Error map per (32-row block, 32-column block) for a few tiny shapes."""
import sys
import numpy as np, torch
sys.path.insert(0, ".")
import cuda_flashattention_amd as fa
import oracle

def run(N, d, causal=False):
    g = torch.Generator().manual_seed(1)
    mk = lambda s: ((torch.rand(1, 1, N, d, generator=g) - 0.5) * s).bfloat16()
    Q, K, V, dO = mk(1), mk(1), mk(1), mk(0.4)
    scale = 1.0 / d ** 0.5
    O, L = fa.flash_attention_2_forward(Q.cuda(), K.cuda(), V.cuda(), scale, causal=causal)
    dQ, dK, dV = fa.flash_attention_2_backward(Q.cuda(), K.cuda(), V.cuda(), O, L, dO.cuda(), scale, causal=causal)
    torch.cuda.synchronize()
    f = lambda t: t.float().cpu().numpy()[0, 0]
    rQ = oracle.attention_backward(f(Q)[None, None], f(K)[None, None], f(V)[None, None], f(dO)[None, None], scale, causal=causal)[0][0, 0]
    e = np.abs(f(dQ) - rQ)
    print(f"N={N} d={d} causal={causal}: max err {e.max():.3e}, ref max {np.abs(rQ).max():.3e}")
    nb = (N + 31) // 32
    for rb in range(nb):
        print("   rows %4d.. :" % (32 * rb), " ".join("%8.1e" % e[32 * rb:32 * rb + 32, 32 * c:32 * c + 32].max() for c in range(d // 32)))
    # per-key-block contribution check: recompute dQ restricted to key blocks to see which block is missing/garbled
    return e.max()

for N in (32, 64, 128, 192, 256):
    run(N, 128)
run(128, 64)
