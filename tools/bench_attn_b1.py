#!/usr/bin/env python3
"""The attention backward at one sample per GPU (BASELINE configs 3 / 4): dK / dV passes as they are against the passes split by
query head (attention.hip SPLIT).  MOLLY_ATTN_SPLIT_MAX=<blocks> moves the bound below which the split applies and benches the
sizes around it.    python tools/bench_attn_b1.py"""
import torch, sys
sys.path.insert(0, '/root/repo')
from molly_amd import ops
BF=torch.bfloat16
def bench(nh,nkv,T,B=1,hd=128):
    M=B*T
    g=torch.Generator(device='cuda').manual_seed(0)
    qkv=(torch.rand(M,(nh+2*nkv)*hd,device='cuda',generator=g)-0.5).to(BF)
    q,k,v=qkv[:,:nh*hd],qkv[:,nh*hd:(nh+nkv)*hd],qkv[:,(nh+nkv)*hd:]
    o,lse=ops.attn_fwd(q,k,v,B,T,nh,nkv,hd,hd**-0.5,True,None,None)
    do=(torch.rand(M,nh*hd,device='cuda',generator=g)-0.5).to(BF)
    dqkv=torch.zeros_like(qkv)
    dq,dk,dv=dqkv[:,:nh*hd],dqkv[:,nh*hd:(nh+nkv)*hd],dqkv[:,(nh+nkv)*hd:]
    n=ops.attn_bwd_workspace(B,T,nh,nkv,hd)
    ws=torch.empty(n,dtype=torch.float32,device='cuda') if n else None
    delta=torch.empty(B,nh,T,dtype=torch.float32,device='cuda')
    res={}
    for tag,w in (('unsplit',None),('split',ws)):
        if tag=='split' and ws is None: continue
        for _ in range(3): ops.attn_bwd(q,k,v,o,do,lse,B,T,nh,nkv,hd,hd**-0.5,True,dq,dk,dv,None,None,delta_ws=delta,ws=w)
        torch.cuda.synchronize()
        e0,e1=torch.cuda.Event(enable_timing=True),torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(20): ops.attn_bwd(q,k,v,o,do,lse,B,T,nh,nkv,hd,hd**-0.5,True,dq,dk,dv,None,None,delta_ws=delta,ws=w)
        e1.record(); e1.synchronize()
        res[tag]=e0.elapsed_time(e1)/20*1e3
    print(f"B={B} nh={nh} nkv={nkv} T={T}: " + "  ".join(f"{k} {v:7.1f} us" for k,v in res.items()))
import os
if os.environ.get("MOLLY_ATTN_SPLIT_MAX"):
    bench(32, 8, 4096, B=2); bench(16, 8, 2048, B=4); bench(32, 8, 3072, B=4); bench(16, 8, 2048, B=8); bench(32, 8, 2048, B=8)
else:
    bench(32, 8, 3072); bench(32, 8, 4096); bench(32, 8, 5120); bench(16, 8, 2048); bench(32, 8, 1024); bench(16, 8, 2048, B=2); bench(32, 8, 3072, B=2)
