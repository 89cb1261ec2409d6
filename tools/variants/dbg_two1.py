import os, sys
sys.path.insert(0, "/root/repo/tests"); sys.path.insert(0, "/root/repo")
import torch, torch.multiprocessing as mp, torch.distributed as dist
import json
from test_gpu_two_ranks import _model, _batches, _args
from conftest import GOLD
def worker(rank, world, port, meta, ret, stage, steps):
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), RANK=str(rank), WORLD_SIZE=str(world))
    torch.cuda.set_device(0)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    from molly_amd.trainer import Zero2Optimizer
    m = _model(meta)
    opt = Zero2Optimizer(m._rt.P.flat, m._rt.G.flat, m.n_decay, lr=1e-3, weight_decay=1e-2, max_grad_norm=1.0, chunk_elems=1 << 18, stage=stage)
    m.attach_optimizer(opt)
    b = _batches(meta)[rank]
    norms = []
    for _ in range(steps):
        m.forward_backward(*_args(b))
        norms.append(float(opt.step(lr=1e-3).item()))
    opt.wait_all_params(); torch.cuda.synchronize()
    ret[rank] = (m._rt.P.flat.cpu().clone(), norms, m._rt.G.flat.cpu().clone(), opt.master.cpu().clone(), [(int(s), int(p)) for s, p in opt.buckets], int(m.n_decay))
    dist.barrier(); dist.destroy_process_group()
def main():
    meta = json.load(open(os.path.join(GOLD, "tiny_meta.json")))
    mgr = mp.Manager(); res = {}
    for stage, port in ((2, 29775), (0, 29777)):
        ret = mgr.dict()
        mp.spawn(worker, args=(2, port, meta, ret, stage, 1), nprocs=2, join=True)
        res[stage] = ret[0]
        print("stage", stage, "norms", ret[0][1], "buckets", ret[0][4][:4], "...", len(ret[0][4]), "n_decay", ret[0][5])
    a, b = res[0][0].float(), res[2][0].float()
    d = (a - b).abs()
    print("after ONE step: params differ at", int((a != b).sum()), "max abs", d.max().item(), "grads equal", torch.equal(res[0][2], res[2][2]))
    idx = (a != b).nonzero().flatten()
    print("   idx", idx[:20].tolist(), "...", idx[-5:].tolist())
    print("   vals", [(a[i].item(), b[i].item()) for i in idx[:6].tolist()])
if __name__ == "__main__":
    main()
