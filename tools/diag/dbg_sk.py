import sys, torch
sys.path.insert(0, '/root/repo')
from molly_amd import ops
def run(M,N,K,blocks=None, label=""):
    g = torch.Generator(device="cuda").manual_seed(1)
    a = torch.randint(-3,4,(M,K),device="cuda",generator=g).bfloat16()
    b = torch.randint(-3,4,(N,K),device="cuda",generator=g).bfloat16()
    ref = a.float() @ b.float().t()
    c = ops.GemmContext(); c.ensure_workspace(0)
    if blocks is not None: c.set("persistent_blocks", blocks)
    with ops.use_gemm_context(c):
        out = ops.gemm_nt(a, b, out_dtype=torch.float32)
    torch.cuda.synchronize()
    err = (out-ref).abs()
    tm, tn = (M+255)//256, (N+255)//256
    print(label, M,N,K,"blocks",blocks,"cfg",c.get("last_config"),"timeouts",c.streamk_timeouts(),"max err",err.max().item())
    mp = torch.zeros(tm,tn)
    for i in range(tm):
        for j in range(tn):
            mp[i,j] = err[i*256:(i+1)*256, j*256:(j+1)*256].max()
    print((mp>0).int())
    # within a bad tile: which rows/cols
    bad = (mp>0).nonzero()
    if len(bad):
        i,j = bad[0].tolist()
        e = err[i*256:(i+1)*256, j*256:(j+1)*256]
        rows = (e.max(1).values>0).nonzero().flatten(); cols=(e.max(0).values>0).nonzero().flatten()
        print(" tile",i,j,"bad rows",rows[:8].tolist(),"..",len(rows),"bad cols",cols[:8].tolist(),"..",len(cols))
        o = out[i*256:(i+1)*256, j*256:(j+1)*256]; r = ref[i*256:(i+1)*256, j*256:(j+1)*256]
        # is out == partial sum over some k range?
        for k0,k1 in ((0,K//2),(K//2,K),(0,K)):
            pr = a[i*256:(i+1)*256,k0:k1].float() @ b[j*256:(j+1)*256,k0:k1].float().t()
            print("   equals k[%d:%d]?"%(k0,k1), torch.equal(o,pr), (o-pr).abs().max().item())
run(512,512,512, label="A")          # 4 tiles, nk 8, upt 4, U 16 -> grid 8: 2 units each: every tile 2 pieces
run(2048,1024,512, label="B")
run(2048,1024,512, blocks=8, label="C")   # 32 tiles over 8 blocks: 4 whole tiles each -> no pieces
run(2048,1024,512, blocks=16, label="D")
run(2048,1024,512, blocks=24, label="E")  # 32*4/24 = 5.33 units
run(4096,1280,1280, label="F")
