import os, sys, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from molly_amd import ops
from molly_amd._lib import lib
def t(fn, n=3):
    fn(); best=1e9
    for _ in range(3):
        e0,e1=torch.cuda.Event(enable_timing=True),torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(n): fn()
        e1.record(); torch.cuda.synchronize(); best=min(best,e0.elapsed_time(e1)/n)
    return best
g=torch.Generator(device="cuda").manual_seed(0)
rnd=lambda *s:(torch.rand(*s,device="cuda",generator=g)*2-1).bfloat16()
lib().call("molly_gemm_force_tile",512)
M=16384
x,w=rnd(M,2048),rnd(12288,2048); o=torch.empty(M,12288,dtype=torch.bfloat16,device="cuda")
dy,w2=rnd(M,12288),rnd(12288,2048); o2=torch.empty(M,2048,dtype=torch.bfloat16,device="cuda")
a,b=rnd(M,12288),rnd(M,2048); o3=torch.empty(12288,2048,dtype=torch.bfloat16,device="cuda")
fl=2.0*M*12288*2048
for gm in (1,2,4,8,16,32):
    lib().call("molly_gemm_set_group_m", gm)
    r1=fl/t(lambda:ops.gemm(x,w,out=o))/1e9
    r2=fl/t(lambda:ops.gemm(dy,w2,out=o2,b_kmajor=True))/1e9
    r3=fl/t(lambda:ops.gemm(a,b,out=o3,a_kmajor=True,b_kmajor=True))/1e9
    print(f"GROUP_M {gm:3d}: NT {r1:7.0f}  NN {r2:7.0f}  TN {r3:7.0f} TF/s")
