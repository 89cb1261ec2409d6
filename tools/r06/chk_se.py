import torch, sys
sys.path.insert(0,'/root/repo')
from molly_amd import ops
from molly_amd._lib import lib
dev='cuda'
g=torch.Generator(device=dev).manual_seed(0)
rnd=lambda *s:(torch.rand(*s,device=dev,generator=g)*2-1).bfloat16()
M=32768
for n,k in ((4096,2048),(12288,2048)):
    a,b=rnd(M,k),rnd(n,k); c=torch.empty(M,n,dtype=torch.bfloat16,device=dev)
    for tag,ctx in (("default-thread-ctx",None),("fresh ctx",ops.GemmContext())):
        def run():
            if ctx is None: ops.gemm_nt(a,b,out=c)
            else:
                with ops.use_gemm_context(ctx): ops.gemm_nt(a,b,out=c)
        if ctx is not None: ctx.ensure_workspace(64<<20)
        run(); 
        cfg = lib().query("molly_gemm_last_config") if ctx is None else ctx.get("last_config")
        best=1e9
        for _ in range(5):
            e0,e1=torch.cuda.Event(enable_timing=True),torch.cuda.Event(enable_timing=True)
            e0.record()
            for _ in range(5): run()
            e1.record(); torch.cuda.synchronize()
            best=min(best,e0.elapsed_time(e1)/5)
        print(n,k,tag,cfg,round(best*1e3,1),'us',round(2*M*n*k/best/1e9),'TF')
