"""Dev aid: the bf16 forward at BASELINE configs[1] (4,16,4096,64) and at the north-star shape (4,16,8192,128), timed as
bench.py times its side figures (50 ramp launches, median of 3 x 20), for A/B runs of library variants on ONE device:
FA2_LIB_PATH=var/<name>.so python tools/gpu_ab_fwd2.py"""
import sys, os, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import cuda_flashattention_amd as fa


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


out = []
for (B, H, N, d, causal) in ((4, 16, 4096, 64, False), (4, 16, 8192, 128, False), (4, 16, 4096, 64, True)):
    Q, K, V = ((torch.rand(B, H, N, d, device="cuda") - 0.5).bfloat16() for _ in range(3))
    O = torch.empty_like(Q); L = torch.empty(B, H, N, device="cuda")
    ms = med(lambda: fa.flash_attention_2_forward(Q, K, V, None, causal=causal, O=O, L=L))
    fl = 4.0 * B * H * N * N * d * (0.5 if causal else 1.0)
    out.append(f"N={N} d={d}{'c' if causal else ' '} {ms:.4f} ms {fl / ms / 1e9:.0f} TF")
print(f"{os.environ.get('FA2_LIB_PATH', 'product'):22s} " + " | ".join(out), flush=True)
