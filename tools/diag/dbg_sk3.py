import sys, torch
sys.path.insert(0, '/root/repo')
from molly_amd import ops
g = torch.Generator(device="cuda").manual_seed(1)
rnd = lambda *s: (torch.rand(*s, device="cuda", generator=g) * 2 - 1).bfloat16()
M,N,K = 32768,1280,5120
a,w,bias,res = rnd(M,K), rnd(N,K), rnd(N), rnd(M,N)
on = ops.GemmContext(); on.ensure_workspace(1<<30); on.set("streamk",2)
with ops.use_gemm_context(on): first = ops.gemm_nt(a,w,bias=bias,res=res).clone()
hdr = (64 + 8192*64)//4
for mode in ("none","fill_slabs","read_slabs","fill+read"):
    bad = 0
    for it in range(10):
        if "fill" in mode: on.ws[hdr:].uniform_(-1000,1000)       # another kernel dirties the slab region from every XCD
        if "read" in mode: s_ = on.ws[hdr:hdr+(256*2*65536)].sum()   # ... and leaves clean copies of it in every L2
        with ops.use_gemm_context(on): got = ops.gemm_nt(a,w,bias=bias,res=res)
        torch.cuda.synchronize()
        if not torch.equal(got, first):
            bad += 1
            if bad == 1: 
                d=(got.float()-first.float()).abs(); print("   first bad: max", d.max().item(), "frac", (d>0).float().mean().item())
    print(mode, "wrong launches", bad, "of 10, timeouts", on.streamk_timeouts())
