import sys, os, torch, numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import cuda_flashattention_amd as fa
B, H, N, d = 4, 16, 8192, 128
mk = lambda: (torch.rand(B, H, N, d, device="cuda") - 0.5).bfloat16()
Q, K, V = mk(), mk(), mk()
O = torch.empty_like(Q); L = torch.zeros(B, H, N, device="cuda")
for _ in range(3): fa.flash_attention_2_forward(Q, K, V, None, O=O, L=L)
torch.cuda.synchronize()
nb = 2048
st = L.view(-1).view(torch.int64)[-nb * 8:].view(nb, 8).cpu().numpy()
t0, t1, t2, t3, hw = (st[:, i] for i in range(5))
base = t0.min()
us = lambda x: (x - base) / 100.0
print("kernel span %.1f us; block duration mean %.1f us (prologue %.1f, loop %.1f, epilogue %.1f)" % (
    us(t3.max()), (t3 - t0).mean() / 100, (t1 - t0).mean() / 100, (t2 - t1).mean() / 100, (t3 - t2).mean() / 100))
# HW_ID: wave_id[3:0] simd[5:4] pipe[7:6] cu[11:8] sh[12] se[15:13](gfx9: se_id 14:13?) ... use (xcc? not in HW_ID) -> group by (se, sh, cu) bits 8..15
cu = (hw >> 8) & 0xff
xcc = (hw >> 16) & 0xff     # whatever is up there
key = cu + 256 * xcc
gaps = []
per = {}
for k in np.unique(key):
    idx = np.where(key == k)[0]
    o = idx[np.argsort(t0[idx])]
    per[k] = len(o)
    for a, b in zip(o[:-1], o[1:]):
        gaps.append((t0[b] - t3[a]) / 100.0)
gaps = np.array(gaps)
print("distinct CU keys %d; blocks per key min %d max %d" % (len(per), min(per.values()), max(per.values())))
print("gap between a block's end and the next block's start on the same CU key: mean %.2f us, median %.2f, max %.2f" % (gaps.mean(), np.median(gaps), gaps.max()))
print("first block start spread %.2f us; last end - earliest last end %.2f us" % (us(np.sort(t0)[255]), (t3.max() - np.sort(t3)[-256]) / 100.0))
