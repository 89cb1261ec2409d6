#!/usr/bin/env python3
"""Are whole optimizer steps (forward, backward, side-stream AdamW under the next forward) reproducible?  Two models built from
the same seed take the same four steps on the same batches; parameters and losses must agree bit for bit.
    python tools/r04/check_step_determinism.py [batch] [MOLLY env knobs apply]"""
import os, sys
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import molly_amd
from molly_amd import config as C
from molly_amd.synth import synth_batch
from molly_amd.trainer import Zero2Optimizer

B = int(sys.argv[1]) if len(sys.argv) > 1 else 8
T = 2048
batches = [synth_batch(B, T, [("protein", 512)], seed=42 + 1000 * i) for i in range(3)]


def run(tag):
    torch.manual_seed(1234)                      # (the embedding shell is materialised from the global generator at construction)
    cfg = C.molly("1.7b", k_tokens=512)
    m = molly_amd.OmicsOne(cfg)
    m.model = molly_amd.Qwen3ForCausalLM(cfg.text_config)
    m.dna_rna_model = molly_amd.EsmForMaskedLM(cfg.dna_rna_config)
    m.protein_model = molly_amd.EsmForMaskedLM(cfg.protein_config)
    torch.manual_seed(1234)
    m.prepare(torch.device("cuda", 0), random_init_seed=1234)
    rt = m._rt
    opt = Zero2Optimizer(rt.P.flat, rt.G.flat, m.n_decay, lr=3e-5, weight_decay=1e-2, max_grad_norm=1.0)
    m.attach_optimizer(opt)
    p0 = rt.P.flat.clone()
    losses, snaps = [], []
    for i in range(5):
        b = batches[i % len(batches)]
        loss = m.forward_backward(b["input_ids"], b["attention_mask"], b["omic_ids"], b["omic_info_list"], b["labels"])
        opt.step(lr=3e-5)
        losses.append(loss.clone())
        if i in (0, 1, 4):
            opt.wait_all_params(); torch.cuda.synchronize()
            snaps.append(rt.P.flat.clone())
    torch.cuda.synchronize()
    out = (p0, [float(x) for x in losses], snaps)
    del m, opt, rt
    torch.cuda.empty_cache()
    return out


a = run("a")
b = run("b")
print("initial parameters equal:", bool(torch.equal(a[0], b[0])))
print("losses a:", a[1])
print("losses b:", b[1])
for k, (x, y) in enumerate(zip(a[2], b[2])):
    ne = int((x != y).sum())
    print(f"snapshot {k}: {ne} of {x.numel()} parameters differ" + (f", max |d| {float((x.float() - y.float()).abs().max()):.3e}" if ne else ""))
