import sys, os, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import cuda_flashattention_amd as fa
B, H, N, d = (int(x) for x in sys.argv[1:5]) if len(sys.argv) > 4 else (4, 16, 8192, 128)
mk = lambda: (torch.rand(B, H, N, d, device="cuda") - 0.5).bfloat16()
Q, K, V = mk(), mk(), mk()
O = torch.empty_like(Q); L = torch.zeros(B, H, N, device="cuda")
for _ in range(3): fa.flash_attention_2_forward(Q, K, V, None, O=O, L=L)
torch.cuda.synchronize()
e0 = torch.cuda.Event(enable_timing=True); e1 = torch.cuda.Event(enable_timing=True); e0.record()
for _ in range(5): fa.flash_attention_2_forward(Q, K, V, None, O=O, L=L)
e1.record(); torch.cuda.synchronize(); ms = e0.elapsed_time(e1) / 5
nw = B * H * (N // 256) * 8
st = L.view(-1).view(torch.int64)[: nw * 8].view(-1, 8).double().cpu()
a, x, b, r, sy, tot, vm, bar = [st[:, i].mean().item() for i in range(8)]
ntile = (N // 64 + 1 + 2) // 3 * 3
steps = 2 * ntile
print("kernel %.3f ms; per wave loop ticks %.0f (%d blocks per CU in turn -> %.1f MHz if ticks are clocks)" % (ms, tot, nw // 8 // 256, (nw // 8 // 256) * tot / ms / 1e3))
print("per half-tile step (%d MFMAs per wave): A %.0f  X %.0f  B %.0f  rotate %.0f  barrier+dma(per tile) %.0f" % (
    d // 4, a / steps, x / steps, b / steps, r / steps, sy / ntile))
print("per tile: vmcnt wait %.0f, barrier wait %.0f, dma issue %.0f" % (vm / ntile, bar / ntile, sy / ntile))
print("share: A %.1f%% X %.1f%% B %.1f%% rot %.1f%% sync %.1f%%" % tuple(100 * v / tot for v in (a, x, b, r, sy)))
