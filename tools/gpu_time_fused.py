"""A/B timing of the two-kernel backward against the fused five-product prototype at (4,16,8192,128) (dev aid).
    python tools/gpu_time_fused.py [mode]      alternating blocks of 10 launches, medians over the blocks"""
import sys, statistics, torch
MODE = int(sys.argv[1]) if len(sys.argv) > 1 else 1
sys.path.insert(0, "."); sys.path.insert(0, "tools")
import cuda_flashattention_amd as fa
from gpu_check_fused import fused, lib
B, H, N, d = 4, 16, 8192, 128
mk = lambda: (torch.rand(B, H, N, d, device="cuda") - 0.5).bfloat16()
Q, K, V, dO = mk(), mk(), mk(), mk()
O, L = fa.flash_attention_2_forward(Q, K, V)
ws = torch.empty(lib.fa2_backward_fused_workspace_bytes(B, H, N, d), dtype=torch.uint8, device="cuda")
ws2 = torch.empty(lib.fa2_backward_workspace_bytes(B, H, N, d, 0), dtype=torch.uint8, device="cuda")
dQ, dK, dV = torch.empty_like(Q), torch.empty_like(Q), torch.empty_like(Q)
def two():
    for ph in (1, 6): fa.flash_attention_2_backward(Q, K, V, O, L, dO, dQ=dQ, dK=dK, dV=dV, workspace=ws2, phases=ph)
fus = lambda: fused(Q, K, V, O, L, dO, d**-0.5, ws, MODE)
def block(f, n=10):
    e0 = torch.cuda.Event(enable_timing=True); e1 = torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(n): f()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / n
for _ in range(20): two(); fus()
ta, tb = [], []
for _ in range(5):
    ta.append(block(two)); tb.append(block(fus))
print(f"two-kernel {statistics.median(ta):.3f} ms   fused mode {MODE} {statistics.median(tb):.3f} ms   (blocks: "
      + " ".join(f"{a:.2f}/{b:.2f}" for a, b in zip(ta, tb)) + ")")
