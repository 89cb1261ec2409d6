import os, sys
sys.path.insert(0, "/root/repo/tests"); sys.path.insert(0, "/root/repo")
import torch, torch.multiprocessing as mp
import json
from test_gpu_two_ranks import _worker
from conftest import GOLD
def main():
    meta = json.load(open(os.path.join(GOLD, "tiny_meta.json")))
    mgr = mp.Manager()
    res = {}
    for stage, port in ((2, 29675), (0, 29677), (2, 29679), (0, 29681)):
        ret = mgr.dict()
        mp.spawn(_worker, args=(2, port, meta, ret, stage), nprocs=2, join=True)
        print("stage", stage, "norms", ["%.9g" % x for x in ret[0][1]], "equal ranks", torch.equal(ret[0][0], ret[1][0]))
        if stage in res:
            print("   same stage repeated: params identical", torch.equal(res[stage], ret[0][0]))
        res[stage] = ret[0][0]
    a, b = res[0].float(), res[2].float()
    d = (a - b).abs()
    bad = d > 2 ** -6 * b.abs() + 1e-30
    print("stage0 vs stage2: differ", int((a != b).sum()), "of", a.numel(), " beyond 2^-6:", int(bad.sum()), " max abs diff", d.max().item(),
          " worst rel", (d / (b.abs() + 1e-30))[bad].max().item() if bad.any() else 0.0)
    idx = bad.nonzero().flatten()[:10].tolist()
    print("   first bad idx", idx, [(a[i].item(), b[i].item()) for i in idx])
if __name__ == "__main__":
    main()
