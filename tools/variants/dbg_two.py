import os, sys
sys.path.insert(0, "/root/repo/tests"); sys.path.insert(0, "/root/repo")
import torch, torch.multiprocessing as mp
import json
from test_gpu_two_ranks import _worker
from conftest import GOLD
def main():
    meta = json.load(open(os.path.join(GOLD, "tiny_meta.json")))
    mgr = mp.Manager()
    for stage, port in ((2, 29675), (0, 29677)):
        ret = mgr.dict()
        mp.spawn(_worker, args=(2, port, meta, ret, stage), nprocs=2, join=True)
        print("stage", stage, "norms", ["%.9g" % x for x in ret[0][1]], "equal ranks", torch.equal(ret[0][0], ret[1][0]))
if __name__ == "__main__":
    main()
