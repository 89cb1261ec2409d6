"""CPU: host logic of the training loop (gradient-accumulation windows exactly as the pasted HF loop cuts them, reference
src/trainer/domain_loss.py:584-608) and of the checkpoint readers (HF directory layouts)."""
import json
import math
import os
import types

import pytest
import torch

from molly_amd.trainer import TrainArgs, Trainer
from molly_amd.trainer.zero2 import Zero2Optimizer


class _Kernels:                                   # torch stand-in for the shard arithmetic (test-only, like test_zero2_gloo)
    def sqnorm(self, g, out, accumulate):
        s = g.float().pow(2).sum()
        out[0] = out[0] + s if accumulate else s

    def clip_coef(self, norm_sq, max_norm, pre_scale, norm_out, coef_out):
        n = norm_sq[0].sqrt() * pre_scale
        norm_out[0] = n
        coef_out[0] = torch.clamp(max_norm / (n + 1e-6), max=1.0) * pre_scale

    def adamw(self, master, m, v, grad, param_out, lr, b1, b2, eps, wd, step, gscale):
        master -= lr * grad.float() * gscale[0]
        param_out.copy_(master.to(param_out.dtype))


class _FakeModel:
    """Records what the loop asks of the model: (accumulate, final_micro) per micro-batch."""

    def __init__(self, n=64):
        self.P = torch.zeros(n, dtype=torch.bfloat16)
        self.G = torch.zeros(n, dtype=torch.bfloat16)
        self.n_decay = 32
        self.calls, self.steps_seen = [], []
        self._rt = types.SimpleNamespace(dev=torch.device("cpu"), P=types.SimpleNamespace(flat=self.P),
                                         G=types.SimpleNamespace(flat=self.G), llm=types.SimpleNamespace(lora=None))

    def _runtime(self):
        return self._rt

    def attach_optimizer(self, opt):
        self._rt.opt = opt

    def forward_backward(self, input_ids, attention_mask, omic_ids, omic_info_list, labels, accumulate=False, final_micro=True):
        self.calls.append((bool(accumulate), bool(final_micro)))
        self.G += 1 if accumulate else 0
        if not accumulate:
            self.G.fill_(1)
        return torch.tensor(float(input_ids[0]))


def _collate(items):
    return {"input_ids": torch.tensor(items), "attention_mask": None, "omic_ids": None, "omic_info_list": None, "labels": None}


def _run(n_samples, B, GA, epochs=1.0, max_steps=-1, logging_steps=1):
    m = _FakeModel()
    opt = Zero2Optimizer(m.P, m.G, m.n_decay, kernels=_Kernels())
    steps = []
    real_step = opt.step
    opt.step = lambda lr=None: (steps.append(len(m.calls)), real_step(lr=lr))[1]
    logs = []
    tr = Trainer(m, list(range(n_samples)), _collate, TrainArgs(per_device_train_batch_size=B, gradient_accumulation_steps=GA,
                                                               num_train_epochs=epochs, max_steps=max_steps,
                                                               logging_steps=logging_steps), log_fn=logs.append, optimizer=opt)
    hist = tr.train()
    return m, steps, hist


def test_last_partial_window_of_an_epoch_steps():
    # 7 micro-batches, GA 3 -> windows of 3, 3, 1: three optimizer steps per epoch (HF: ceil(7/3)), none dropped
    m, steps, hist = _run(n_samples=7, B=1, GA=3, epochs=2)
    assert steps == [3, 6, 7, 10, 13, 14]
    assert [c for c in m.calls[:7]] == [(False, False), (True, False), (True, True), (False, False), (True, False), (True, True),
                                        (False, True)]
    assert len(hist) == 6 and hist[-1]["step"] == 6
    assert hist[2]["epoch"] == 1.0 and abs(hist[0]["epoch"] - 3 / 7) < 1e-3          # fractional epochs, HF's formula


def test_fewer_micro_batches_than_ga_still_steps_and_terminates():
    m, steps, hist = _run(n_samples=2, B=1, GA=8, epochs=3)
    assert steps == [2, 4, 6] and all(c[1] == (i % 2 == 1) for i, c in enumerate(m.calls))


def test_total_steps_and_schedule_follow_hf_counting():
    m, steps, hist = _run(n_samples=10, B=2, GA=2, epochs=1)          # 5 micro-batches -> ceil(5/2) = 3 steps
    assert len(steps) == 3
    # linear warmup over ceil(3 * 0.1) = 1 step: the first step runs at lr 0, like get_linear_schedule_with_warmup
    assert hist[0]["learning_rate"] == 0.0 and hist[1]["learning_rate"] > hist[2]["learning_rate"] > 0


def test_logged_loss_is_the_window_sum_averaged_over_steps():
    m, steps, hist = _run(n_samples=4, B=1, GA=2, epochs=1, logging_steps=2)
    # micro losses are the sample ids of a seeded permutation of 0..3: their sum is 6 whatever the order; 2 steps
    assert len(hist) == 1 and hist[0]["loss"] == 3.0


def test_reads_every_hf_weight_layout(tmp_path):
    from safetensors.torch import save_file
    from molly_amd.loaders import load_pretrained, read_checkpoint_dir
    import molly_amd
    from molly_amd.config import EncConfig
    cfg = EncConfig(vocab_size=33, hidden_size=32, intermediate_size=64, num_hidden_layers=2, num_attention_heads=2)
    ref = molly_amd.EsmForMaskedLM.from_config(cfg, seed=5)
    sd = {k: v.detach().clone() for k, v in ref.state_dict().items()}
    sd["lm_head.bias"] = torch.zeros(33)                                  # a head Molly never runs: accepted, not an error
    # (a) single safetensors, (b) sharded safetensors + index, (c) pytorch_model.bin
    a, b, c = tmp_path / "a", tmp_path / "b", tmp_path / "c"
    for d in (a, b, c):
        d.mkdir()
    save_file(sd, str(a / "model.safetensors"))
    keys = sorted(sd)
    half = len(keys) // 2
    save_file({k: sd[k] for k in keys[:half]}, str(b / "model-00001-of-00002.safetensors"))
    save_file({k: sd[k] for k in keys[half:]}, str(b / "model-00002-of-00002.safetensors"))
    (b / "model.safetensors.index.json").write_text(json.dumps({"weight_map": {
        **{k: "model-00001-of-00002.safetensors" for k in keys[:half]}, **{k: "model-00002-of-00002.safetensors" for k in keys[half:]}}}))
    torch.save(sd, str(c / "pytorch_model.bin"))
    for d in (a, b, c):
        got = read_checkpoint_dir(str(d))
        assert sorted(got) == keys and all(torch.equal(got[k], sd[k]) for k in keys)
        shell = molly_amd.EsmForMaskedLM(cfg)
        load_pretrained(shell, str(d), "encoder", log=lambda *_: None)
        assert all(torch.equal(v, sd[k]) for k, v in torch.nn.Module.state_dict(shell).items())
    # a checkpoint that lacks a tensor the forward reads is an error, never a silent random init
    part = tmp_path / "part"
    part.mkdir()
    save_file({k: v for k, v in sd.items() if "layer.1.output" not in k}, str(part / "model.safetensors"))
    with pytest.raises(RuntimeError, match="lacks 2 tensors"):
        load_pretrained(molly_amd.EsmForMaskedLM(cfg), str(part), "encoder", log=lambda *_: None)
    with pytest.raises(FileNotFoundError):
        read_checkpoint_dir(str(tmp_path))


def test_stand_in_tokenizers_are_refused_for_pretrained_weights(tmp_path):
    from molly_amd.loaders import setup_tokenizers
    *_, real = setup_tokenizers("qwen3-0.6b", "nt-500m", "esm2-650m", log=lambda *_: None)
    assert real is False


def test_get_omics_one_config_reads_three_hf_config_dirs(tmp_path):
    """reference: src/model/config.py:49-86 — three model paths -> one OmicsModalConfig (the reference goes through
    transformers.AutoConfig; the fields the arithmetic reads must come out the same from the plain JSON)."""
    from molly_amd.config import get_omics_one_config
    qwen = dict(model_type="qwen3", vocab_size=151936, hidden_size=2048, intermediate_size=6144, num_hidden_layers=28,
                num_attention_heads=16, num_key_value_heads=8, head_dim=128, rms_norm_eps=1e-6, tie_word_embeddings=True,
                rope_parameters={"rope_theta": 1000000.0, "rope_type": "default"}, max_position_embeddings=40960,
                eos_token_id=151645, attention_bias=False, torch_dtype="bfloat16")
    nt = dict(model_type="esm", vocab_size=4105, hidden_size=1280, intermediate_size=5120, num_hidden_layers=24,
              num_attention_heads=20, max_position_embeddings=1002, position_embedding_type="absolute", token_dropout=False,
              pad_token_id=1, mask_token_id=2, layer_norm_eps=1e-12, emb_layer_norm_before=False)
    esm = dict(model_type="esm", vocab_size=33, hidden_size=1280, intermediate_size=5120, num_hidden_layers=33,
               num_attention_heads=20, max_position_embeddings=1026, position_embedding_type="rotary", token_dropout=True,
               pad_token_id=1, mask_token_id=32, layer_norm_eps=1e-5)
    dirs = []
    for name, c in (("qwen", qwen), ("nt", nt), ("esm", esm)):
        d = tmp_path / name
        d.mkdir()
        (d / "config.json").write_text(json.dumps(c))
        dirs.append(str(d))
    cfg = get_omics_one_config(*dirs)
    assert cfg.dna_rna_project_token_num == 64 and cfg.protein_project_token_num == 64          # reference defaults
    t = cfg.text_config
    assert (t.hidden_size, t.num_hidden_layers, t.num_key_value_heads, t.head_dim, t.rope_theta, t.tie_word_embeddings) == \
        (2048, 28, 8, 128, 1e6, True)
    assert cfg.dna_rna_config.position_embedding_type == "absolute" and cfg.dna_rna_config.max_position_embeddings == 1002
    assert cfg.dna_rna_config.layer_norm_eps == 1e-12 and cfg.protein_config.token_dropout is True
    assert cfg.text_config.use_cache is False and cfg.protein_config.gradient_checkpointing is False   # reference :80-84
    # the presets are these same shapes
    from molly_amd import config as C
    p = C.molly("1.7b")
    for k in ("hidden_size", "intermediate_size", "num_hidden_layers", "num_attention_heads", "num_key_value_heads", "head_dim",
              "vocab_size", "tie_word_embeddings", "rope_theta"):
        assert getattr(p.text_config, k) == getattr(t, k), k
    for k in ("hidden_size", "intermediate_size", "num_hidden_layers", "num_attention_heads", "vocab_size",
              "position_embedding_type", "max_position_embeddings", "token_dropout", "mask_token_id"):
        assert getattr(p.dna_rna_config, k) == getattr(cfg.dna_rna_config, k), k
        assert getattr(p.protein_config, k) == getattr(cfg.protein_config, k), k
    with pytest.raises(NotImplementedError):
        C.LlmConfig.from_dict({**qwen, "attention_bias": True})
