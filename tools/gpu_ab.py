"""Dev aid: run gpu_check_fwd-style d=64/d=128 spot checks + timing for several library builds
(FA2_LIB_PATH variants) in separate subprocesses."""
import os, subprocess, sys
libs = sys.argv[1:]
code = r'''
import sys, numpy as np, torch
sys.path.insert(0, ".")
import cuda_flashattention_amd as fa, oracle
f = lambda t: t.float().cpu().numpy()
def spot(B,H,N,d,c):
    g = torch.Generator().manual_seed(0)
    Q,K,V = ((torch.rand(B,H,N,d,generator=g)-0.5).bfloat16() for _ in range(3))
    s = 1.0/d**0.5
    O,L = fa.flash_attention_2_forward(Q.cuda(),K.cuda(),V.cuda(),s,causal=c); torch.cuda.synchronize()
    Or,Lr = oracle.attention_forward(f(Q),f(K),f(V),s,causal=c)
    return float(np.linalg.norm(f(O)-Or)/np.linalg.norm(Or))
print("relL2 d64 %.3e | d64 causal %.3e | d128 %.3e" % (spot(1,2,128,64,False), spot(1,2,300,64,True), spot(1,2,256,128,False)))
B,H,N,d = 4,16,8192,128
mk = lambda: (torch.rand(B,H,N,d,device="cuda")-0.5).bfloat16()
Q,K,V = mk(),mk(),mk(); O = torch.empty_like(Q); L = torch.empty(B,H,N,device="cuda")
for _ in range(3): fa.flash_attention_2_forward(Q,K,V,None,O=O,L=L)
torch.cuda.synchronize(); e0=torch.cuda.Event(enable_timing=True); e1=torch.cuda.Event(enable_timing=True); e0.record()
for _ in range(10): fa.flash_attention_2_forward(Q,K,V,None,O=O,L=L)
e1.record(); torch.cuda.synchronize(); ms=e0.elapsed_time(e1)/10
print("fwd cfg3 %.3f ms %.0f TF" % (ms, 4*B*H*N*N*d/ms/1e9))
'''
for lib in libs:
    env = dict(os.environ)
    if lib != "default": env["FA2_LIB_PATH"] = os.path.abspath(lib)
    print("==", lib, flush=True)
    subprocess.run([sys.executable, "-c", code], env=env)
