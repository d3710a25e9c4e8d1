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
nw = 2048 * 8
st = L.view(-1).view(torch.int64)[: nw * 8].view(-1, 8).double().cpu()
a, x, b, r, sy, tot, vm, bar = [st[:, i].mean().item() for i in range(8)]
steps = 2 * 129
print("kernel %.3f ms; per wave loop ticks %.0f (8 blocks per CU in turn -> %.1f MHz if ticks are clocks)" % (ms, tot, 8 * tot / ms / 1e3))
print("per half-tile step (16 MFMAs per wave): A %.0f  X %.0f  B %.0f  rotate %.0f  barrier+dma(per tile) %.0f" % (
    a / steps, x / steps, b / steps, r / steps, sy / 129))
print("per tile: vmcnt wait %.0f, barrier wait %.0f, dma issue %.0f" % (vm / 129, bar / 129, sy / 129))
print("share: A %.1f%% X %.1f%% B %.1f%% rot %.1f%% sync %.1f%%" % tuple(100 * v / tot for v in (a, x, b, r, sy)))
