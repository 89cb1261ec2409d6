#!/usr/bin/env python3
"""Launcher with the reference's flag names (reference: src/train.py:237-594, the subset its example scripts pass:
scripts/train/examples/run_train_1B_z2_b1.sh).  One process per GPU:
    python -m torch.distributed.run --nproc-per-node N -m molly_amd.train <flags>
Model paths may be HF model directories (config.json; weights from model.safetensors / pytorch_model.bin, single file or
sharded with an index; tokenizers through transformers.AutoTokenizer when the directory holds tokenizer files) or the built-in shape presets `qwen3-{0.6b,1.7b,4b,8b}`, `esm2-650m`, `nt-500m`, `tiny` with --no-load-pretrained.
Cosmetic flags of the reference (swanlab, report_to, enable-list, attn_impl, use_liger, ...) are accepted and ignored.
"""
import argparse
import json
import os
import re
import sys

import torch


def _preset(path, kind):
    from . import config as C
    p = str(path).lower()
    if kind == "text" and p.startswith("qwen3-"):
        return C.qwen3(p.split("-", 1)[1])
    if p == "esm2-650m":
        return C.esm2_650m()
    if p == "nt-500m":
        return C.nt_500m_human_ref()
    if p == "tiny":
        meta = json.load(open(os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tests", "golden",
                                           "tiny_meta.json")))["config"]
        return {"text": C.LlmConfig.from_dict(meta["text"]), "dna": C.EncConfig.from_dict(meta["dna_rna"]),
                "protein": C.EncConfig.from_dict(meta["protein"])}[kind]
    cfg = C._load_json_config(path)
    return C.LlmConfig.from_dict(cfg) if kind == "text" else C.EncConfig.from_dict(cfg)


def zero_stage_of(deepspeed_config):
    """ZeRO stage named by the reference's `--deepspeed_config src/configs/ds_z{0,1,2}_config.json` (the JSON's
    zero_optimization.stage when the file exists, else the z<N> in its name; no flag = the ZeRO-2 default of the examples).
    Stage 1 (sharded optimizer state, all-reduced gradients) steps to the same parameters as stage 2 and runs as stage 2;
    stage 3 (sharded parameters) is outside the hot path."""
    if not deepspeed_config:
        return 2
    stage = None
    if os.path.exists(deepspeed_config):
        stage = json.load(open(deepspeed_config)).get("zero_optimization", {}).get("stage")
    if stage is None:
        m = re.search(r"z(\d)", os.path.basename(deepspeed_config))
        stage = int(m.group(1)) if m else 2
    if stage not in (0, 1, 2):
        raise ValueError(f"--deepspeed_config {deepspeed_config}: ZeRO stage {stage} is not supported (0, 1, 2 are)")
    return 0 if stage == 0 else 2


def main(argv=None):
    ap = argparse.ArgumentParser()
    for f, kw in [("--experiment-name", {}), ("--output_dir", dict(default="out")), ("--text-model-path", dict(required=True)),
                  ("--dna-rna-model-path", dict(required=True)), ("--protein-model-path", dict(required=True)),
                  ("--train-dataset-path", dict(required=True)), ("--eval-dataset-path", {}), ("--mode", dict(default="sft")),
                  ("--device", dict(default="cuda")), ("--save_strategy", {}), ("--eval_strategy", {}), ("--logging_strategy", {}),
                  ("--enable-list", dict(nargs="*")), ("--attn_impl", {}), ("--use_liger", {}), ("--save_trainable", {}),
                  ("--swanlab-mode", {}), ("--swanlab-team", {}), ("--swanlab-project", {}), ("--report_to", dict(nargs="*")),
                  ("--deepspeed_config", {})]:
        ap.add_argument(f, **kw)
    for f, d in [("--dna-rna-k-tokens", 64), ("--protein-k-tokens", 64), ("--max-len", 1024), ("--max-src-len", 1024),
                 ("--eval-max-len", 1024), ("--eval-max-src-len", 1024), ("--per_device_train_batch_size", 1),
                 ("--per_device_eval_batch_size", 1), ("--read-nums", 0), ("--eval-read-nums", 0), ("--save_steps", 0),
                 ("--eval_steps", 0), ("--logging_steps", 20), ("--gradient-accumulation-steps", 1), ("--save-total-limit", 0),
                 ("--early-stopping-patience", 0), ("--seed", 42), ("--train-iters", -1), ("--local_rank", 0),
                 ("--lora_r", 64)]:
        ap.add_argument(f, type=int, default=d)
    for f, d in [("--num_train_epochs", 1.0), ("--learning_rate", 3e-5), ("--warmup_ratio", 0.1), ("--weight-decay", 1e-2),
                 ("--eps", 1e-8)]:
        ap.add_argument(f, type=float, default=d)
    ap.add_argument("--metric-for-best-model", default="eval_loss")
    for f in ("--train-mlp", "--train-llm", "--train-bio", "--bf16", "--no-load-pretrained", "--swanlab", "--save_only_model",
              "--skip-eval", "--use-lora", "--load_best_model_at_end", "--greater_is_better"):
        ap.add_argument(f, action="store_true")
    a = ap.parse_args(argv)
    lora = None
    if a.use_lora:
        # reference src/train.py:654-657 -> pre_train_lora (src/utils/tools.py:345-396): base + encoders frozen, adapters on
        # every LLM Linear but lm_head, projectors trainable whatever --train-mlp says
        from .lora import LoraConfig
        lora = LoraConfig(r=a.lora_r, lora_alpha=64.0, lora_dropout=0.05, seed=a.seed)

    import torch.distributed as dist
    world = int(os.environ.get("WORLD_SIZE", "1"))
    local = int(os.environ.get("LOCAL_RANK", "0"))
    torch.cuda.set_device(local)
    if world > 1:
        import datetime
        dist.init_process_group(os.environ.get("MOLLY_DIST_BACKEND", "nccl"), timeout=datetime.timedelta(minutes=30),
                                **({"device_id": torch.device("cuda", local)}
                                   if os.environ.get("MOLLY_DIST_BACKEND", "nccl") == "nccl" else {}))   # reference :606-610
    torch.manual_seed(a.seed)

    import molly_amd
    from .config import OmicsModalConfig
    from .data import DatasetConfig, OmicsDataset, qwen_omics_collate_fn
    from .trainer import TrainArgs, Trainer, save_model
    cfg = OmicsModalConfig(text_config=_preset(a.text_model_path, "text"), dna_rna_config=_preset(a.dna_rna_model_path, "dna"),
                           protein_config=_preset(a.protein_model_path, "protein"))
    cfg.dna_rna_project_token_num, cfg.protein_project_token_num = a.dna_rna_k_tokens, a.protein_k_tokens
    m = molly_amd.OmicsOne(cfg)
    m.model = molly_amd.Qwen3ForCausalLM(cfg.text_config)
    m.dna_rna_model = molly_amd.EsmForMaskedLM(cfg.dna_rna_config)
    m.protein_model = molly_amd.EsmForMaskedLM(cfg.protein_config)
    from .loaders import load_pretrained, setup_tokenizers
    text_tok, dna_tok, prot_tok, real_tok = setup_tokenizers(a.text_model_path, a.dna_rna_model_path, a.protein_model_path)
    if not a.no_load_pretrained:
        if not real_tok:
            raise RuntimeError("pretrained weights need their tokenizers: the model directories hold no tokenizer files "
                               "(stand-in tokenizers are for --no-load-pretrained smoke runs only)")
        for sub, path, what in ((m.model, a.text_model_path, "LLM"), (m.dna_rna_model, a.dna_rna_model_path, "dna/rna encoder"),
                                (m.protein_model, a.protein_model_path, "protein encoder")):
            load_pretrained(sub, path, what)
    m.set_special_tokens(text_tok)
    # random init ONLY for an explicit --no-load-pretrained run: a checkpoint that lacks a tensor must fail, not be made up
    m.prepare(torch.device("cuda", local), train_llm=a.train_llm and lora is None, train_mlp=a.train_mlp or lora is not None,
              random_init_seed=1234 if a.no_load_pretrained else None, lora=lora, train_bio=a.train_bio)
    dcfg = DatasetConfig(max_len=a.max_len, max_src_len=a.max_src_len, mode=a.mode, cal_metric_pos=None,
                         dna_rna_k_tokens=a.dna_rna_k_tokens, protein_k_tokens=a.protein_k_tokens)
    ds = OmicsDataset(a.train_dataset_path, text_tok, dcfg, dna_rna_tokenizer=dna_tok, protein_tokenizer=prot_tok,
                      read_nums=a.read_nums or None, shuffle=True, seed=a.seed)
    targs = TrainArgs(output_dir=a.output_dir, per_device_train_batch_size=a.per_device_train_batch_size,
                      gradient_accumulation_steps=a.gradient_accumulation_steps, num_train_epochs=a.num_train_epochs,
                      max_steps=a.train_iters, learning_rate=a.learning_rate, weight_decay=a.weight_decay,
                      warmup_ratio=a.warmup_ratio, adam_epsilon=a.eps, logging_steps=a.logging_steps, save_steps=a.save_steps,
                      save_total_limit=a.save_total_limit or None, seed=a.seed,
                      per_device_eval_batch_size=a.per_device_eval_batch_size, eval_steps=a.eval_steps,
                      early_stopping_patience=a.early_stopping_patience, load_best_model_at_end=a.load_best_model_at_end,
                      zero_stage=zero_stage_of(a.deepspeed_config))
    eval_ds = None
    if not a.skip_eval and a.eval_dataset_path:                  # reference: src/train.py:206-231
        if a.metric_for_best_model != "eval_loss" or a.greater_is_better:
            raise NotImplementedError("only eval_loss (lower is better) is tracked — the reference's default")
        ecfg = DatasetConfig(max_len=a.eval_max_len, max_src_len=a.eval_max_src_len, mode=a.mode, cal_metric_pos=None,
                             dna_rna_k_tokens=a.dna_rna_k_tokens, protein_k_tokens=a.protein_k_tokens)
        eval_ds = OmicsDataset(a.eval_dataset_path, text_tok, ecfg, dna_rna_tokenizer=dna_tok, protein_tokenizer=prot_tok,
                               read_nums=a.eval_read_nums or None)
    tr = Trainer(m, ds, qwen_omics_collate_fn, targs, eval_dataset=eval_ds)
    tr.train()
    if (not dist.is_initialized()) or dist.get_rank() == 0:
        save_model(m, a.output_dir)
    if dist.is_initialized():
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
