import os, subprocess, sys
code = r'''
import sys, torch
sys.path.insert(0, ".")
import cuda_flashattention_amd as fa
B,H,N,d = 4,16,8192,128
mk = lambda: (torch.rand(B,H,N,d,device="cuda")-0.5).bfloat16()
Q,K,V = mk(),mk(),mk(); O = torch.empty_like(Q); L = torch.empty(B,H,N,device="cuda")
for _ in range(3): fa.flash_attention_2_forward(Q,K,V,None,O=O,L=L)
torch.cuda.synchronize(); e0=torch.cuda.Event(enable_timing=True); e1=torch.cuda.Event(enable_timing=True); e0.record()
for _ in range(10): fa.flash_attention_2_forward(Q,K,V,None,O=O,L=L)
e1.record(); torch.cuda.synchronize(); ms=e0.elapsed_time(e1)/10
print("fwd cfg3 %.3f ms %.0f TF" % (ms, 4*B*H*N*N*d/ms/1e9))
'''
for lib in sys.argv[1:]:
    env = dict(os.environ)
    if lib != "default": env["FA2_LIB_PATH"] = os.path.abspath(lib)
    print("==", lib, flush=True)
    subprocess.run([sys.executable, "-c", code], env=env)
