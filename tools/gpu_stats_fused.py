"""Wait statistics of the chained fused backward (needs a library built with -DFA2_FUSED_STATS; dev aid)."""
import sys, torch
sys.path.insert(0, "."); sys.path.insert(0, "tools")
import cuda_flashattention_amd as fa
from gpu_check_fused import fused, lib
B, H, N, d = 4, 16, 8192, 128
mk = lambda: (torch.rand(B, H, N, d, device="cuda") - 0.5).bfloat16()
Q, K, V, dO = mk(), mk(), mk(), mk()
O, L = fa.flash_attention_2_forward(Q, K, V)
nb = lib.fa2_backward_fused_workspace_bytes(B, H, N, d)
ws = torch.zeros(nb, dtype=torch.uint8, device="cuda")
for _ in range(2): fused(Q, K, V, O, L, dO, d**-0.5, ws, 1)
torch.cuda.synchronize()
al = lambda x: (x + 255) & ~255
base = 3 * al(B*H*N*4)
ctl = ws[base + al(B*H*N*d*4):].view(torch.int32).cpu()
e = 32 * 17
units = B * H * (N // 256)
steps, polls = int(ctl[e+1]), int(ctl[e+2])
cyc = lambda o: int(ctl[e+o]) & 0xffffffff | (int(ctl[e+o+1]) << 32)
print(f"error={int(ctl[e])} units={units} steps that waited={steps} ({steps/units:.1f}/unit of {N//32}) polls={polls} ({polls/max(steps,1):.2f}/wait)")
print(f"cycles waited/unit={cyc(4)/units:.0f} total/unit={cyc(6)/units:.0f}")
import numpy as np
ncb = N // 256
rec = ctl[32*18 + 16*(B*H+16) + B*H*ncb:][:8*units].numpy().reshape(B*H, ncb, 8)
t0 = rec[..., 4:6].copy().view(np.int64)[..., 0]; t1 = rec[..., 6:8].copy().view(np.int64)[..., 0]
base = t0.min()
dur = (t1 - t0)
print("kernel span (cycles):", int(t1.max() - base))
for hd in np.argsort(t0[:, 0])[[0, 1, 8, 16, 63]]:
    print(f"head {hd:2d} xcc {rec[hd,0,0]}: start cb0 {int(t0[hd,0]-base):9d} cb31 {int(t0[hd,-1]-base):9d} | end cb0 {int(t1[hd,0]-base):9d} cb31 {int(t1[hd,-1]-base):9d}"
          f" | dur cb0 {int(dur[hd,0])} cb1 {int(dur[hd,1])} cb16 {int(dur[hd,16])} cb31 {int(dur[hd,-1])} | waited cb1 {rec[hd,1,1]} cb16 {rec[hd,16,1]} cb31 {rec[hd,-1,1]}")
print("mean dur by cb:", " ".join(f"{int(x)}" for x in dur.mean(0)[::4]))
print("mean waited by cb:", " ".join(f"{int(x)}" for x in rec[..., 1].mean(0)[::4]))
print("mean cycles from ticket to first body:", int(rec[..., 2].mean()), " epilogue:", int(rec[..., 3].mean()), " (by cb:",
      " ".join(str(int(x)) for x in rec[..., 2].mean(0)[::8]), ")")
