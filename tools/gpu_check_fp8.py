"""Dev aid: fp8 (e4m3) forward against the oracle fed the same rounded inputs."""
import sys, os
import numpy as np, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import cuda_flashattention_amd as fa
import oracle

def run(B, H, N, causal, seed=0, amp=1.0):
    d = 128
    g = torch.Generator().manual_seed(seed)
    mk = lambda: ((torch.rand(B, H, N, d, generator=g) - 0.5) * amp).to(torch.float8_e4m3fn)
    Q, K, V = mk(), mk(), mk()
    s = 1.0 / d ** 0.5
    O, L = fa.flash_attention_2_forward(Q.cuda(), K.cuda(), V.cuda(), s, causal=causal)
    torch.cuda.synchronize()
    f = lambda t: t.float().numpy()
    Or, Lr = oracle.attention_forward(f(Q), f(K), f(V), s, causal=causal)
    Og, Lg = O.float().cpu().numpy(), L.cpu().numpy()
    rel = np.linalg.norm(Og - Or) / np.linalg.norm(Or)
    print(f"fp8 B{B} H{H} N{N} causal={causal}: relL2(O)={rel:.3e} max|dO|={np.abs(Og - Or).max():.3e} max|dL|={np.abs(Lg - Lr).max():.3e}"
          f" nan={int(np.isnan(Og).sum())}", flush=True)
    return rel

if __name__ == "__main__":
    bad = 0
    for cfg in [(1, 1, 64, False), (1, 1, 256, False), (1, 2, 320, False), (1, 3, 333, False), (2, 8, 1024, False), (1, 2, 256, True), (1, 3, 777, True), (1, 8, 2048, True)]:
        bad += run(*cfg) > 5e-2
    B, H, N, d = 1, 16, 32768, 128
    mk = lambda: (torch.rand(B, H, N, d, device="cuda") - 0.5).to(torch.float8_e4m3fn)
    Q, K, V = mk(), mk(), mk()
    O = torch.empty(B, H, N, d, dtype=torch.bfloat16, device="cuda"); L = torch.empty(B, H, N, device="cuda")
    for causal in (True, False):
        for _ in range(2): fa.flash_attention_2_forward(Q, K, V, None, causal=causal, O=O, L=L)
        torch.cuda.synchronize()
        e0 = torch.cuda.Event(enable_timing=True); e1 = torch.cuda.Event(enable_timing=True); e0.record()
        for _ in range(5): fa.flash_attention_2_forward(Q, K, V, None, causal=causal, O=O, L=L)
        e1.record(); torch.cuda.synchronize(); ms = e0.elapsed_time(e1) / 5
        fl = 4.0 * B * H * N * N * d * (0.5 if causal else 1.0)
        print(f"fp8 fwd (1,16,32768,128) causal={causal}: {ms:.3f} ms  {fl / ms / 1e9:.0f} TFLOP/s")
    sys.exit(1 if bad else 0)
