import os, sys, torch
ROOT = "/root/repo"
sys.path.insert(0, ROOT)
from molly_amd._lib import MollyLib, lib
A, B = lib(), MollyLib(os.path.join(ROOT, "tools/variants/libmolly_r2start.so"))
dev = "cuda"
g = torch.Generator(device=dev).manual_seed(0)
st = torch.cuda.current_stream().cuda_stream
for n in (4096 + 8, 262144, 951552, 951552 - 262144 * 3, 1903104, 12, 516, 1028, 2052):
    for off in (0, 4, 12):
        p0 = torch.randn(n + 64, device=dev, generator=g)
        gr = (torch.randn(n + 64, device=dev, generator=g) * 1e-3).bfloat16()
        coef = torch.tensor([0.379], device=dev)
        outs = []
        for L in (A, B):
            master, m, v = p0.clone(), torch.zeros_like(p0), torch.zeros_like(p0)
            pout = torch.zeros(n + 64, dtype=torch.bfloat16, device=dev)
            for step in (1, 2):
                L.call("molly_adamw_step", st, master[off:], m[off:], v[off:], gr[off:], pout[off:], n, 1e-3, 0.9, 0.999, 1e-8, 1e-2, step, coef, None)
            torch.cuda.synchronize()
            outs.append((master.clone(), m.clone(), v.clone(), pout.clone()))
        eq = [torch.equal(a, b) for a, b in zip(*outs)]
        if not all(eq):
            d = (outs[0][0] - outs[1][0]).abs()
            print("n", n, "off", off, "equal(master,m,v,p)", eq, "max diff", d.max().item(), "first idx", d.nonzero().flatten()[:5].tolist())
print("done")
