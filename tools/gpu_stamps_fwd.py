import sys, os, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import cuda_flashattention_amd as fa
B, H, N, d = 4, 16, 8192, 128
mk = lambda: (torch.rand(B, H, N, d, device="cuda") - 0.5).bfloat16()
Q, K, V = mk(), mk(), mk()
O = torch.empty_like(Q); L = torch.zeros(B, H, N, device="cuda")
for _ in range(3): fa.flash_attention_2_forward(Q, K, V, None, O=O, L=L)
torch.cuda.synchronize()
e0 = torch.cuda.Event(enable_timing=True); e1 = torch.cuda.Event(enable_timing=True); e0.record()
for _ in range(5): fa.flash_attention_2_forward(Q, K, V, None, O=O, L=L)
e1.record(); torch.cuda.synchronize(); ms = e0.elapsed_time(e1) / 5
st = L.view(-1).view(torch.int64)[: 2048 * 4 * 4].view(-1, 4).double().cpu()
a, xb, sy, tot = [st[:, i].mean().item() for i in range(4)]
steps = 2 * 129
print("kernel %.3f ms; per wave loop ticks %.0f (x8 waves/SIMD = %.0f -> %.1f MHz if ticks are clocks)" % (ms, tot, 8 * tot, 8 * tot / ms / 1e3))
print("per step: A %.0f  XB %.0f  sync+dma(per tile) %.0f ; share A %.1f%% XB %.1f%% sync %.1f%%" % (
    a / steps, xb / steps, sy / 129, 100 * a / tot, 100 * xb / tot, 100 * sy / tot))
