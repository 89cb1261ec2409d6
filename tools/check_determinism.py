#!/usr/bin/env python3
"""Run the full-size Molly-1.7B step repeatedly and list every gradient tensor that differs between runs (expected: none)."""
import sys, torch
sys.path.insert(0, '.')
import molly_amd
from molly_amd import config as C
from molly_amd.synth import synth_batch
from molly_amd._lib import lib
cfg = C.molly("1.7b", k_tokens=512)
m = molly_amd.OmicsOne(cfg)
m.model = molly_amd.Qwen3ForCausalLM(cfg.text_config)
m.dna_rna_model = molly_amd.EsmForMaskedLM(cfg.dna_rna_config)
m.protein_model = molly_amd.EsmForMaskedLM(cfg.protein_config)
m.prepare("cuda", random_init_seed=1234)
b = synth_batch(8, 2048, [("protein", 512)], seed=42)
args = [b[k] for k in ("input_ids", "attention_mask", "omic_ids", "omic_info_list", "labels")]
if len(sys.argv) > 1:
    lib().call("molly_gemm_set_persistent_blocks", int(sys.argv[1]))
m.forward_backward(*args)
ref = m._rt.G.flat.clone()
for it in range(int(sys.argv[2]) if len(sys.argv) > 2 else 6):
    m.forward_backward(*args)
    torch.cuda.synchronize()
    bad = []
    for n, v in m._rt.G.views.items():
        o = m._rt.G.offsets[n]
        r = ref[o:o + v.numel()].view(v.shape)
        if not torch.equal(r, v):
            d = (r.float() - v.float()).abs()
            bad.append((n, int((r != v).sum()), d.max().item(), v.float().abs().max().item()))
    print("iter", it, "tensors differing:", len(bad))
    for x in bad[:12]:
        print("   ", x)
