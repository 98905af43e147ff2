import os, sys
sys.path.insert(0, "/root/repo/csm-train-pytorch_amd")
import torch
from csm.hip import ops
dev="cuda"; g=torch.Generator(device=dev).manual_seed(0)
def rnd(*s): return (torch.randn(*s, device=dev, generator=g)*0.5).to(torch.bfloat16)
cases=[("dec_w13_fwd","nt",16384,16384,1024),("dec_w2_dx","nn",16384,8192,1024),("dec_qkv_fwd","nt",16384,1536,1024),("dec_o_fwd","nt",16384,1024,1024),("dec_w13_dx","nn",16384,1024,16384),("qkv_fwd","nt",8192,3072,2048)]
for name,mode,m,n,k in cases:
    if mode=="nt": A,B,tA,tB=rnd(m,k),rnd(n,k),False,False
    else: A,B,tA,tB=rnd(m,k),rnd(k,n),False,True
    C=torch.empty(m,n,dtype=torch.bfloat16,device=dev)
    for v in (1,4):
        ops.lib.csm_set_gemm_variant(v)
        for _ in range(3): ops.gemm(A,B,C,None,tA,tB)
        torch.cuda.synchronize()
        e0,e1=torch.cuda.Event(enable_timing=True),torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(10): ops.gemm(A,B,C,None,tA,tB)
        e1.record(); torch.cuda.synchronize()
        t=e0.elapsed_time(e1)/10*1e-3
        print(f"v{v} {name:12s} {mode} {m}x{n}x{k}: {t*1e6:8.1f} us {2.0*m*n*k/t/1e12:7.1f} TF/s")
