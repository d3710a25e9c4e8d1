import sys, torch
sys.path.insert(0, ".")
import cuda_flashattention_amd as fa
for (B,H,N,d) in ((4,16,8192,64),(4,16,4096,64),(4,16,8192,128)):
    mk = lambda s=1.0: ((torch.rand(B,H,N,d,device="cuda")-0.5)*s).bfloat16()
    Q,K,V,dO = mk(),mk(),mk(),mk(0.4)
    O,L = fa.flash_attention_2_forward(Q,K,V)
    ws = torch.empty(fa._capi.lib().fa2_backward_workspace_bytes(B,H,N,d,0), dtype=torch.uint8, device="cuda")
    dQ,dK,dV = (torch.empty_like(Q) for _ in range(3))
    def t(ph, n=10):
        f = lambda: fa.flash_attention_2_backward(Q,K,V,O,L,dO,dQ=dQ,dK=dK,dV=dV,workspace=ws,phases=ph)
        for _ in range(3): f()
        torch.cuda.synchronize()
        e0=torch.cuda.Event(enable_timing=True); e1=torch.cuda.Event(enable_timing=True); e0.record()
        for _ in range(n): f()
        e1.record(); torch.cuda.synchronize(); return e0.elapsed_time(e1)/n
    t(1)
    dq, dkdv = t(2), t(4)
    fl = 2.0*B*H*N*N*d
    print(f"({B},{H},{N},{d}): dq {dq:.3f} ms ({3*fl/dq/1e9:.0f} TF exec)  dkdv {dkdv:.3f} ms ({4*fl/dkdv/1e9:.0f} TF exec)  sum {dq+dkdv:.3f} ms = {5*fl/(dq+dkdv)/1e9:.0f} TF algorithmic", flush=True)
    t7 = t(8) if fa._capi.lib().fa2_backward_plan(B, H, N, d, 0, 0, None) == 1 else float("nan")
    print(f"      single five-product kernel (+ output pass): {t7:.3f} ms = {5*fl/t7/1e9:.0f} TF algorithmic", flush=True)
