"""Ad-hoc first-light check of the bf16 forward against the oracle (dev aid; the real
parity tests live in tests/)."""
import sys, time
import numpy as np, torch
sys.path.insert(0, ".")
import cuda_flashattention_amd as fa
import oracle
from oracle import recipes

def run(B, H, N, d, causal=False, seed=0):
    g = torch.Generator(device="cpu").manual_seed(seed)
    Q = (torch.rand(B, H, N, d, generator=g) - 0.5).bfloat16()
    K = (torch.rand(B, H, N, d, generator=g) - 0.5).bfloat16()
    V = (torch.rand(B, H, N, d, generator=g) - 0.5).bfloat16()
    scale = 1.0 / d ** 0.5
    O, L = fa.flash_attention_2_forward(Q.cuda(), K.cuda(), V.cuda(), scale, causal=causal)
    torch.cuda.synchronize()
    Oref, Lref = oracle.attention_forward(Q.float().numpy(), K.float().numpy(), V.float().numpy(), scale, causal=causal)
    Og = O.float().cpu().numpy(); Lg = L.cpu().numpy()
    rel = np.linalg.norm(Og - Oref) / np.linalg.norm(Oref)
    print(f"B{B} H{H} N{N} d{d} causal={causal}: relL2(O)={rel:.3e} max|dO|={np.abs(Og-Oref).max():.3e} max|dL|={np.abs(Lg-Lref).max():.3e}", flush=True)
    return rel

if __name__ == "__main__":
    print(torch.cuda.get_device_name(0))
    bad = 0
    for cfg in [(1,1,256,128), (1,2,128,64), (2,8,512,128), (1,8,1024,64), (1,3,333,128), (1,1,77,64), (2,8,512,128,True), (1,2,300,64,True), (1,8,2048,128,True)]:
        r = run(*cfg)
        bad += r > 5e-3
    # timing at cfg3
    B,H,N,d = 4,16,8192,128
    Q = (torch.rand(B,H,N,d, device="cuda")-0.5).bfloat16(); K=(torch.rand(B,H,N,d, device="cuda")-0.5).bfloat16(); V=(torch.rand(B,H,N,d, device="cuda")-0.5).bfloat16()
    O = torch.empty_like(Q); L = torch.empty(B,H,N, device="cuda")
    for _ in range(3): fa.flash_attention_2_forward(Q,K,V,None,O=O,L=L)
    torch.cuda.synchronize()
    e0 = torch.cuda.Event(enable_timing=True); e1 = torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(10): fa.flash_attention_2_forward(Q,K,V,None,O=O,L=L)
    e1.record(); torch.cuda.synchronize()
    ms = e0.elapsed_time(e1)/10
    print(f"fwd cfg3 {ms:.3f} ms  {4*B*H*N*N*d/ms/1e9:.1f} TFLOP/s", flush=True)
    sys.exit(1 if bad else 0)
