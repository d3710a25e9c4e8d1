"""Dev aid: the two side configurations' forwards -- bf16 (4,16,4096,64) and fp8-e4m3 causal (1,16,32768,128) -- timed as
bench.py times them (50 ramp launches, median of 3 x 20), after a parity spot check against the oracle at small sizes.
A/B: FA2_LIB_PATH=var/<name>.so python tools/gpu_fwd_side.py"""
import sys, os
import numpy as np, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import cuda_flashattention_amd as fa
import oracle

f = lambda t: t.float().cpu().numpy()
rel = lambda a, b: float(np.linalg.norm(a - b) / max(np.linalg.norm(b), 1e-30))


def spot(B, H, N, d, causal, dt=torch.bfloat16, gate=5e-3):
    g = torch.Generator().manual_seed(N + d)
    Q, K, V = (((torch.rand(B, H, N, d, generator=g) - 0.5)).to(dt) for _ in range(3))
    s = 1.0 / d ** 0.5
    O, L = fa.flash_attention_2_forward(Q.cuda(), K.cuda(), V.cuda(), s, causal=causal)
    torch.cuda.synchronize()
    Or, Lr = oracle.attention_forward(f(Q), f(K), f(V), s, causal=causal)
    e, el = rel(f(O), Or), float(np.abs(L.cpu().numpy() - Lr).max())
    ok = e < gate and el < 1e-4
    print(f"spot {dt} B{B} H{H} N{N} d{d} c={int(causal)}: O {e:.2e} |dL| {el:.1e} {'ok' if ok else 'BAD'}", flush=True)
    return ok


def med(fn):
    for _ in range(50): fn()
    torch.cuda.synchronize()
    v = []
    for _ in range(3):
        e0 = torch.cuda.Event(enable_timing=True); e1 = torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(20): fn()
        e1.record(); torch.cuda.synchronize()
        v.append(e0.elapsed_time(e1) / 20)
    return sorted(v)[1]


if __name__ == "__main__":
    ok = all([spot(1, 8, 320, 64, False), spot(1, 2, 300, 64, True), spot(2, 3, 1000, 64, False), spot(1, 2, 777, 64, True),
              spot(1, 4, 129, 64, False), spot(1, 8, 640, 128, False),
              spot(1, 2, 777, 128, True, torch.float8_e4m3fn, 5e-2), spot(1, 2, 1024, 128, False, torch.float8_e4m3fn, 5e-2)])
    for (B, H, N, d, dt, causal) in ((4, 16, 4096, 64, torch.bfloat16, False), (1, 16, 32768, 128, torch.float8_e4m3fn, True),
                                     (4, 16, 8192, 128, torch.bfloat16, False), (4, 16, 4096, 64, torch.bfloat16, True)):
        Q, K, V = ((torch.rand(B, H, N, d, device="cuda") - 0.5).to(dt) for _ in range(3))
        O = torch.empty(B, H, N, d, dtype=torch.bfloat16, device="cuda"); L = torch.empty(B, H, N, device="cuda")
        ms = med(lambda: fa.flash_attention_2_forward(Q, K, V, None, causal=causal, O=O, L=L))
        fl = 4.0 * B * H * N * N * d * (0.5 if causal else 1.0)
        print(f"fwd {dt} ({B},{H},{N},{d}) causal={causal}: {ms:.4f} ms  {fl / ms / 1e9:.0f} TFLOP/s", flush=True)
    sys.exit(0 if ok else 1)
