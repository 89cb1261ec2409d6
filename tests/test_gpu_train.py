"""GPU: the step loop (GA window, clip, AdamW, LR schedule) against the oracle's restatement of the same semantics, plus the
checkpoint layout round trip."""
import json
import os

import numpy as np
import pytest
import torch

from conftest import tiny_batch, tiny_state_dict
from test_gpu_model import build_tiny

pytestmark = pytest.mark.gpu


def test_one_clipped_adamw_step_matches_reference_golden(tiny_meta, tiny_gold):
    """forward/backward + ZeRO-2(dp1) step on the HIP path vs the reference's torch.optim.AdamW step (fixture G5)."""
    from molly_amd.trainer import Zero2Optimizer
    m = build_tiny(tiny_meta)
    batch = tiny_batch(tiny_gold, tiny_meta)
    m.forward_backward(batch["input_ids"], batch["attention_mask"], batch["omic_ids"], batch["omic_info_list"], batch["labels"])
    opt = Zero2Optimizer(m._rt.P.flat, m._rt.G.flat, m.n_decay, lr=float(tiny_gold["opt/lr"]), weight_decay=1e-2,
                         max_grad_norm=1.0)
    gn = opt.step()
    torch.cuda.synchronize()
    ref_gn = float(tiny_gold["opt/grad_norm"])
    assert abs(gn.item() - ref_gn) <= 2e-2 * ref_gn            # bf16 gradients
    # fp32 master weights after the step vs the reference's parameters (lr 3e-5: the update is ~1e-5 per element)
    names = [k[len("opt/phead/"):] for k in tiny_gold if k.startswith("opt/phead/")]
    P = m._rt.P
    bad = tot = 0
    err_sum = 0.0
    for n in names:
        if n == "model.lm_head.weight":
            continue
        off = P.offsets[n]
        got = opt.master[off:off + 256].cpu()             # world=1: the owned-chunk layout is the identity
        ref = torch.from_numpy(tiny_gold["opt/phead/" + n]).flatten()[:256]
        n_el = min(len(ref), P.views[n].numel())
        start = tiny_state_dict(tiny_meta)[n].flatten()[:n_el]
        upd_ref = ref[:n_el] - start                      # compare the UPDATE (the master starts from bf16-rounded weights)
        upd_got = got[:n_el] - start.bfloat16().float()
        scale = upd_ref.abs().max().item() + 1e-12
        e = (upd_got - upd_ref).abs() / scale
        bad += int((e > 0.5).sum())
        tot += n_el
        err_sum += float(e.sum())
    # Adam's first step is ~lr*sign(g): bf16 gradient noise can flip the sign of near-zero entries only
    assert bad / tot < 0.02, (bad, tot)
    assert err_sum / tot < 0.08, err_sum / tot
    with torch.no_grad():
        out = m(input_ids=batch["input_ids"], attention_mask=batch["attention_mask"], omic_ids=batch["omic_ids"],
                omic_info_list=batch["omic_info_list"], labels=batch["labels"])
    # loss after one step: the fp32 reference drops 6.9816 -> 6.8912.  The forward here runs on the bf16 copy of the fp32
    # master (as DeepSpeed bf16 does): a 3e-5 update is below bf16 resolution for most weights, so only part of the drop is
    # visible after ONE step — it must be a drop, and not larger than the fp32 one.
    d_ref = float(tiny_gold["fwd/loss"]) - float(tiny_gold["opt/loss_after"])
    d_got = float(tiny_gold["fwd/loss"]) - out.loss.item()
    assert 0.0 < d_got < 1.5 * d_ref, (d_got, d_ref)


def test_trainer_ga_semantics_and_loss_decreases(tiny_meta, tmp_path):
    from molly_amd.data import DatasetConfig, OmicsDataset, ToyOmicTokenizer, ToyTextTokenizer, qwen_omics_collate_fn
    from molly_amd.trainer import TrainArgs, Trainer
    rows = [dict(task="Solubility-Solubility", input=f"Is <protein>{'MKTAYIAKQR' * (1 + i % 3)}</protein> soluble?", think="",
                 output="Yes." if i % 2 else "No.", label=str(i % 2), kind="protein", task_num=i) for i in range(16)]
    ds = OmicsDataset(rows, ToyTextTokenizer(), DatasetConfig(max_len=192, cal_metric_pos=None, dna_rna_k_tokens=64,
                                                               protein_k_tokens=64),
                      dna_rna_tokenizer=ToyOmicTokenizer("dna"), protein_tokenizer=ToyOmicTokenizer("protein"))
    m = build_tiny(tiny_meta)
    args = TrainArgs(output_dir=str(tmp_path), per_device_train_batch_size=4, gradient_accumulation_steps=2,
                     num_train_epochs=6, learning_rate=2e-3, logging_steps=1, warmup_ratio=0.1, save_steps=6,
                     save_total_limit=1)
    logs = []
    hist = Trainer(m, ds, qwen_omics_collate_fn, args, log_fn=logs.append).train()
    assert len(hist) == 12                                        # 16 samples / (4 x GA 2) = 2 steps/epoch x 6 epochs
    # logged loss = SUM of the GA micro losses (reference quirk): starts near 2 x ln(vocab-ish), and decreases
    assert hist[0]["loss"] > hist[-1]["loss"] * 1.5, (hist[0], hist[-1])
    assert hist[0]["loss"] > 8.0                                  # ~2 x 6.9
    assert hist[0]["learning_rate"] == 0.0 and hist[1]["learning_rate"] > 0   # HF linear warmup: first step at lr 0
    ck = os.path.join(str(tmp_path), "checkpoint-12")
    assert os.path.exists(os.path.join(ck, "pytorch_model.bin")) and not os.path.exists(os.path.join(str(tmp_path), "checkpoint-6"))


def test_checkpoint_roundtrip_reference_layout(tiny_meta, tiny_gold, tmp_path):
    from molly_amd.trainer import save_model
    m = build_tiny(tiny_meta)
    batch = tiny_batch(tiny_gold, tiny_meta)
    kw = dict(input_ids=batch["input_ids"], attention_mask=batch["attention_mask"], omic_ids=batch["omic_ids"],
              omic_info_list=batch["omic_info_list"])
    with torch.no_grad():
        ref = m(**kw).logits.clone()
    save_model(m, str(tmp_path))
    sd = torch.load(os.path.join(str(tmp_path), "pytorch_model.bin"))
    shapes = tiny_meta["state_dict_shapes"]
    assert set(sd.keys()) <= set(shapes.keys())                   # the reference's keys (App. C), minus unused heads
    assert all(list(sd[k].shape) == shapes[k] for k in sd)
    proj = torch.load(os.path.join(str(tmp_path), "protein_projector.bin"))
    assert set(proj.keys()) == {"weight", "bias"}
    # fresh model with different weights, then load (reference: src/inference_lora.py:243)
    m2 = build_tiny({**tiny_meta, "config": {**tiny_meta["config"], "seed_w": 99}})
    with torch.no_grad():
        other = m2(**kw).logits
    assert (other.float() - ref.float()).abs().max() > 0.05
    missing, unexpected = m2.load_state_dict(sd, strict=False)
    assert not unexpected
    with torch.no_grad():
        again = m2(**kw).logits
    assert torch.equal(again, ref)


class _LoopbackComm:
    """World-1 stand-in that makes ORDERING observable on one GPU: the 'reduce-scatter' round-trips the bucket through a
    temporary (launched too early it would resurrect stale gradients), the 'all-gather' is the only thing that publishes
    the AdamW output (a forward that does not wait for it reads stale parameters)."""

    def reduce_scatter(self, out_chunk, region):
        tmp = region.clone()
        region.copy_(tmp)

    def all_gather(self, region, chunk):
        region.copy_(chunk)

    def all_reduce(self, t):
        pass


def test_overlapped_comm_schedule_equals_synchronous(tiny_meta, tiny_gold):
    from molly_amd.trainer import Zero2Optimizer
    batch = tiny_batch(tiny_gold, tiny_meta)
    args = (batch["input_ids"], batch["attention_mask"], batch["omic_ids"], batch["omic_info_list"], batch["labels"])

    def run(overlap):
        m = build_tiny(tiny_meta)
        rt = m._rt
        kw = dict(lr=1e-3, chunk_elems=32768)
        if overlap:
            opt = Zero2Optimizer(rt.P.flat, rt.G.flat, m.n_decay, overlap=True, comm=_LoopbackComm(), **kw)
            opt.P_out = rt.P.flat.clone()                      # AdamW output is invisible until the 'all-gather' runs
            assert len(opt.buckets) > 8
        else:
            opt = Zero2Optimizer(rt.P.flat, rt.G.flat, m.n_decay, overlap=False, **kw)
        m.attach_optimizer(opt)
        losses = []
        for step in range(4):
            for micro in range(2):                             # GA = 2: hooks may only fire on the last micro-step
                loss = m.forward_backward(*args, accumulate=micro > 0, final_micro=micro == 1)
            losses.append(loss.clone())                 # the returned loss is a view of a reused device scalar
            opt.step()
        torch.cuda.synchronize()
        return torch.stack(losses).cpu(), rt.P.flat.clone().cpu()

    l0, p0 = run(False)
    l1, p1 = run(True)
    assert torch.equal(l0, l1), (l0, l1)
    assert torch.equal(p0, p1)
    assert l0[-1] < l0[0]


def test_trainer_evaluation_eval_loss_and_early_stopping(tiny_meta, tmp_path):
    """HF eval semantics: eval_loss = sample-weighted mean of the per-batch token-mean losses (tail batch smaller); records
    every eval_steps; early stopping (with load_best_model_at_end, as HF requires) stops after `patience` non-improving
    evaluations and leaves the best parameters under <output_dir>/best."""
    from molly_amd.data import DatasetConfig, OmicsDataset, ToyOmicTokenizer, ToyTextTokenizer, qwen_omics_collate_fn
    from molly_amd.trainer import TrainArgs, Trainer
    mk = lambda n, off: [dict(task="Solubility-Solubility", input=f"Is <protein>{'MKTAYIAKQR' * (1 + (i + off) % 3)}</protein> soluble?",
                              think="", output="Yes." if (i + off) % 2 else "No.", label=str(i % 2), kind="protein", task_num=i)
                         for i in range(n)]
    cfg = DatasetConfig(max_len=192, cal_metric_pos=None, dna_rna_k_tokens=64, protein_k_tokens=64)
    tok = dict(dna_rna_tokenizer=ToyOmicTokenizer("dna"), protein_tokenizer=ToyOmicTokenizer("protein"))
    train_ds = OmicsDataset(mk(16, 0), ToyTextTokenizer(), cfg, **tok)
    eval_ds = OmicsDataset(mk(7, 1), ToyTextTokenizer(), cfg, **tok)           # 7 samples, batch 3 -> 3 + 3 + 1
    m = build_tiny(tiny_meta)
    args = TrainArgs(output_dir=str(tmp_path), per_device_train_batch_size=4, num_train_epochs=3, learning_rate=2e-3,
                     logging_steps=100, per_device_eval_batch_size=3, eval_steps=2)
    tr = Trainer(m, train_ds, qwen_omics_collate_fn, args, log_fn=lambda s: None, eval_dataset=eval_ds)
    v = tr.evaluate()
    with torch.no_grad():
        per = []
        for i0 in range(0, 7, 3):
            b = qwen_omics_collate_fn([eval_ds[i] for i in range(i0, min(7, i0 + 3))])
            per.append((m(input_ids=b["input_ids"], attention_mask=b["attention_mask"], omic_ids=b["omic_ids"],
                          omic_info_list=b["omic_info_list"], labels=b["labels"]).loss.item(), b["input_ids"].shape[0]))
    want = sum(l * k for l, k in per) / 7
    assert abs(v - want) < 1e-5 and abs(v - sum(l for l, _ in per) / 3) > 1e-6      # weighted by samples, not by batches
    hist = tr.train()
    ev = [h for h in hist if "eval_loss" in h]
    assert [h["step"] for h in ev] == [2, 4, 6, 8, 10, 12]
    assert ev[-1]["eval_loss"] < ev[0]["eval_loss"] and tr.best_step is not None
    # early stopping: a learning rate of 0 never improves after the first evaluation -> stops at the 1 + patience-th one
    m2 = build_tiny(tiny_meta)
    args2 = TrainArgs(output_dir=str(tmp_path / "es"), per_device_train_batch_size=4, num_train_epochs=10, learning_rate=0.0,
                      weight_decay=0.0, logging_steps=100, per_device_eval_batch_size=4, eval_steps=1,
                      early_stopping_patience=2, load_best_model_at_end=True)
    tr2 = Trainer(m2, train_ds, qwen_omics_collate_fn, args2, log_fn=lambda s: None, eval_dataset=eval_ds)
    h2 = tr2.train()
    assert [h["step"] for h in h2 if "eval_loss" in h] == [1, 2, 3]
    assert os.path.exists(os.path.join(str(tmp_path / "es"), "best", "pytorch_model.bin"))


def test_step_loop_trajectory_vs_reference_pieces(tiny_meta):
    """G5: the HIP step — forward_backward over a GA = 2 window (gradients summed), global-norm clip, AdamW,
    HF linear warmup — against the trajectory the reference's OmicsOne walks under torch AdamW / clip_grad_norm_ / HF's
    scheduler (tests/golden/gen_golden_steps.py): every micro loss, every step's gradient norm and learning rate, and the
    parameters after four steps (judged on the UPDATE they received, at bf16 resolution of the parameters)."""
    import math
    from conftest import steps_fixture, tiny_state_dict
    from molly_amd.trainer import Zero2Optimizer
    from molly_amd.trainer.zero2 import linear_warmup_lr
    g, batch = steps_fixture()
    GA, STEPS, TOTAL = (int(x) for x in g["meta"])
    base = float(g["base_lr"])
    m = build_tiny(tiny_meta)
    opt = Zero2Optimizer(m._rt.P.flat, m._rt.G.flat, m.n_decay, lr=base, weight_decay=1e-2, max_grad_norm=1.0)
    warm = math.ceil(0.1 * TOTAL)
    for s in range(STEPS):
        for k in range(GA):
            b = batch(s, k)
            loss = m.forward_backward(b["input_ids"], b["attention_mask"], b["omic_ids"], b["omic_info_list"], b["labels"],
                                      accumulate=k > 0, final_micro=k == GA - 1)
            assert abs(loss.item() - g["loss"][s, k]) <= 3e-3, (s, k, loss.item(), g["loss"][s, k])
        lr = linear_warmup_lr(s, base, warm, TOTAL)
        assert abs(lr - g["lr"][s]) < 1e-12
        norm = opt.step(lr=lr)
        assert abs(norm.item() - g["grad_norm"][s]) <= 1.5e-2 * g["grad_norm"][s], (s, norm.item(), g["grad_norm"][s])
    torch.cuda.synchronize()
    p0 = tiny_state_dict(tiny_meta)
    W = m._rt.P.views
    worst = 0.0
    for n in (k[len("pnorm/"):] for k in g if k.startswith("pnorm/")):
        ref = torch.from_numpy(g["phead/" + n])
        got = W[n].float().cpu().flatten()[:256]
        start = p0[n].flatten()[:256]
        upd_ref, upd_got = ref - start, got - start.bfloat16().float()
        # three effective steps (the first runs at lr 0) of ~lr each: the update is ~7e-4 per entry; the bf16 parameter can
        # hold it to half a step of its own magnitude
        tol = 0.35 * upd_ref.abs().max().item() + 2 ** -8 * ref.abs().max().item()
        err = (upd_got - upd_ref).abs().max().item()
        worst = max(worst, err / tol)
        assert err <= tol, (n, err, tol)
        cos = torch.nn.functional.cosine_similarity(upd_got, upd_ref, dim=0).item()
        assert cos > 0.9 or upd_ref.abs().max().item() < 2 ** -8 * ref.abs().max().item(), (n, cos)
    print(f"worst update error / tolerance {worst:.3f}")


def test_non_finite_gradient_skips_the_step_like_deepspeed():
    """An inf/NaN gradient must not poison the fp32 master, the moments or the bf16 parameters (DeepSpeed's ZeRO step skips the
    update on overflow and does not advance Adam): the clip kernel turns a non-finite norm into a NaN coefficient, AdamW leaves
    everything untouched for it and counts the skip; the next finite step uses the bias corrections of step 1."""
    from molly_amd.trainer import Zero2Optimizer
    n, n_decay = 1 << 16, 1 << 15
    g = torch.Generator(device="cuda").manual_seed(0)
    P = torch.randn(n, device="cuda", generator=g).bfloat16()
    G = (torch.randn(n, device="cuda", generator=g) * 0.1).bfloat16()
    good = G.clone()
    opt = Zero2Optimizer(P, G, n_decay, lr=1e-2, max_grad_norm=1.0)
    ref = Zero2Optimizer(P.clone(), good.clone(), n_decay, lr=1e-2, max_grad_norm=1.0)
    p0 = P.clone()
    G[123] = float("inf")
    norm = opt.step()
    torch.cuda.synchronize()
    assert not torch.isfinite(norm).item()
    assert torch.equal(P, p0) and opt.skipped_steps() == 1
    assert float(opt.m.abs().max()) == 0.0 and float(opt.v.abs().max()) == 0.0
    G.copy_(good)
    opt.step()                                     # host step counter is 2, one skipped: Adam's t = 1
    ref.step()
    torch.cuda.synchronize()
    assert torch.equal(P, ref.P) and torch.equal(opt.master, ref.master) and torch.equal(opt.m, ref.m)
    G[7] = float("nan")
    before = P.clone()
    opt.step()
    torch.cuda.synchronize()
    assert torch.equal(P, before) and opt.skipped_steps() == 2


def test_roctx_ranges_do_not_change_the_step(tiny_meta, tiny_gold, monkeypatch):
    """MOLLY_ROCTX=1 brackets the step's phases with roctx ranges (molly_amd/tracing.py): same loss and gradient norm, bit for bit,
    and every range opened is closed."""
    from molly_amd import tracing
    from molly_amd.trainer import Zero2Optimizer
    depth = []
    push, pop = torch.cuda.nvtx.range_push, torch.cuda.nvtx.range_pop
    monkeypatch.setattr(torch.cuda.nvtx, "range_push", lambda name: (depth.append(name), push(name))[1])
    monkeypatch.setattr(torch.cuda.nvtx, "range_pop", lambda: (depth.append(None), pop())[1])
    got = []
    for on in (False, True):
        monkeypatch.setattr(tracing.roctx, "ON", on)
        m = build_tiny(tiny_meta)
        batch = tiny_batch(tiny_gold, tiny_meta)
        loss = m.forward_backward(batch["input_ids"], batch["attention_mask"], batch["omic_ids"], batch["omic_info_list"],
                                  batch["labels"])
        opt = Zero2Optimizer(m._rt.P.flat, m._rt.G.flat, m.n_decay, lr=3e-5, weight_decay=1e-2, max_grad_norm=1.0)
        gn = opt.step()
        torch.cuda.synchronize()
        got.append((float(loss), float(gn)))
        if not on:
            assert depth == []
    assert got[0] == got[1], got
    names = [d for d in depth if d is not None]
    assert len(names) == 6 and len(depth) == 12, depth
    assert names[1].endswith("decoder forward") and names[-1].endswith("AdamW + all-gather")
