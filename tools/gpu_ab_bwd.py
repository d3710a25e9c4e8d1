import os, subprocess, sys
code = r'''
import sys, os, torch
sys.path.insert(0, ".")
import cuda_flashattention_amd as fa
B,H,N,d = 4,16,8192,128
mk = lambda: (torch.rand(B,H,N,d,device="cuda")-0.5).bfloat16()
Q,K,V,dO = mk(),mk(),mk(),mk()
O = torch.empty_like(Q); L = torch.empty(B,H,N,device="cuda")
dQ,dK,dV = torch.empty_like(Q),torch.empty_like(Q),torch.empty_like(Q)
ws = torch.empty(fa._capi.lib().fa2_backward_workspace_bytes(B,H,N,d,0), dtype=torch.uint8, device="cuda")
fa.flash_attention_2_forward(Q,K,V,None,O=O,L=L)
def t(ph):
    f = lambda: fa.flash_attention_2_backward(Q,K,V,O,L,dO,None,dQ=dQ,dK=dK,dV=dV,workspace=ws,phases=ph)
    f(); f(); torch.cuda.synchronize()
    e0=torch.cuda.Event(enable_timing=True); e1=torch.cuda.Event(enable_timing=True); e0.record()
    for _ in range(10): f()
    e1.record(); torch.cuda.synchronize(); return e0.elapsed_time(e1)/10
t(1)
print("dq %.3f ms | dkdv %.3f ms" % (t(2), t(4)))
'''
for lib in sys.argv[1:]:
    env = dict(os.environ)
    if lib != "default": env["FA2_LIB_PATH"] = os.path.abspath(lib)
    print("==", lib, flush=True)
    subprocess.run([sys.executable, "-c", code], env=env)
