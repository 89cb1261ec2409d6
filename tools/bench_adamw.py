import sys, torch
sys.path.insert(0, '.')
from molly_amd import ops
n = 1 << 28
dev = "cuda"
mst = torch.randn(n, device=dev); m = torch.zeros(n, device=dev); v = torch.zeros(n, device=dev)
g = torch.randn(n, device=dev).bfloat16(); out = torch.empty(n, dtype=torch.bfloat16, device=dev)
sc = torch.ones(1, device=dev)
for _ in range(3): ops.adamw_step(mst, m, v, g, out, 3e-5, 0.9, 0.999, 1e-8, 0.01, 1, sc)
best = 1e9
for r in range(5):
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(5): ops.adamw_step(mst, m, v, g, out, 3e-5, 0.9, 0.999, 1e-8, 0.01, 2, sc)
    e1.record(); torch.cuda.synchronize()
    best = min(best, e0.elapsed_time(e1) / 5)
print(f"adamw {n/1e6:.0f}M elems: {best:.3f} ms  {28.0*n/(best*1e-3)/1e12:.2f} TB/s")
