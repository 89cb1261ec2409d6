import os, sys, torch
sys.path.insert(0, "/root/repo")
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
M,N,K=12288,2048,16384   # wgrad gate|up: out [M,N], contraction K tokens
a,b=rnd(K,M),rnd(K,N); out=torch.empty(M,N,dtype=torch.bfloat16,device="cuda")
fl=2.0*M*N*K
print("TN normal        %.0f TF/s"%(fl/t(lambda:ops.gemm(a,b,out=out,a_kmajor=True,b_kmajor=True))/1e9))
a0=rnd(1,M).expand(K,M); b0=rnd(1,N).expand(K,N)
print("TN A rows cached %.0f TF/s"%(fl/t(lambda:ops.gemm(a0,b,out=out,a_kmajor=True,b_kmajor=True))/1e9))
print("TN A,B cached    %.0f TF/s"%(fl/t(lambda:ops.gemm(a0,b0,out=out,a_kmajor=True,b_kmajor=True))/1e9))
# NT reference same flops: out[M2,N2] K2
x,w=rnd(16384,2048),rnd(12288,2048); o2=torch.empty(16384,12288,dtype=torch.bfloat16,device="cuda")
print("NT normal        %.0f TF/s"%(2.0*16384*12288*2048/t(lambda:ops.gemm(x,w,out=o2))/1e9))
x0=rnd(1,2048).expand(16384,2048); w0=rnd(1,2048).expand(12288,2048)
print("NT A,B cached    %.0f TF/s"%(2.0*16384*12288*2048/t(lambda:ops.gemm(x0,w0,out=o2))/1e9))
# big TN (lm_head)
M,N,K=151936,2048,16384
a,b=rnd(K,M),rnd(K,N); out=torch.empty(M,N,dtype=torch.bfloat16,device="cuda"); fl=2.0*M*N*K
print("TN lm_head normal %.0f TF/s"%(fl/t(lambda:ops.gemm(a,b,out=out,a_kmajor=True,b_kmajor=True),1)/1e9))
a0=rnd(1,M).expand(K,M)
print("TN lm_head A cached %.0f TF/s"%(fl/t(lambda:ops.gemm(a0,b,out=out,a_kmajor=True,b_kmajor=True),1)/1e9))
