#!/usr/bin/env python3
"""Batch inference launcher with the reference's flag names and output format (reference: src/inference_lora.py:20-106 flags,
:161-250 `_load_model`, :252-339 `run_dataset`; script scripts/infer/inference_nt_lora.sh).

    python -m molly_amd.inference --text-model-path P --dna-rna-model-path P --protein-model-path P \\
        --trained-model-path CKPT [--use-lora] --dataset-path test.parquet|.jsonl --json-file out.jsonl ...

`--trained-model-path` holds either `pytorch_model.bin` (full fine-tune, reference :236-246) or, with `--use-lora`, a PEFT
adapter (adapter_config.json + adapter_model.safetensors|.bin) plus `dna_rna_projector.bin` / `protein_projector.bin`
(reference :208-234).  The adapter is merged into the LLM weights at load (molly_amd.lora.merge_lora_adapter): the decode
loop then runs the base kernels at no per-token adapter cost; `--lora-live` keeps it un-merged (PEFT's default behaviour).
One JSON line per sample with the reference's keys: decoded_output, input, gt_output, gt_label, task, kind (:316-323).
Model paths are HF directories (config.json + pytorch_model.bin) or the shape presets of molly_amd.train with
--no-load-pretrained.  Tokenizers: the deterministic stand-ins of molly_amd.data (no vocab files exist offline).
"""
import argparse
import json
import os
import sys

import torch


def parse_args(argv=None):
    ap = argparse.ArgumentParser()
    ap.add_argument("--local-rank", type=int, default=0)
    ap.add_argument("--text-model-path", required=True)
    ap.add_argument("--dna-rna-model-path", required=True)
    ap.add_argument("--dna-rna-k-tokens", type=int, default=64)
    ap.add_argument("--protein-model-path", default=None)
    ap.add_argument("--protein-k-tokens", type=int, default=64)
    ap.add_argument("--trained-model-path", required=True)
    ap.add_argument("--use-lora", action="store_true")
    ap.add_argument("--lora-live", action="store_true", help="keep the adapter un-merged (extension)")
    ap.add_argument("--dataset-path", required=True)
    ap.add_argument("--max-length", type=int, default=1024)
    ap.add_argument("--batch-size", type=int, default=8)
    ap.add_argument("--temperature", type=float, default=0.8)
    ap.add_argument("--top-p", type=float, default=0.95)
    ap.add_argument("--top-k", type=int, default=20)
    ap.add_argument("--device", default="cuda", choices=["cpu", "cuda"])
    ap.add_argument("--seed", type=int, default=42)
    ap.add_argument("--max-samples", type=int, default=None)
    ap.add_argument("--repetition-penalty", type=float, default=1.0)
    ap.add_argument("--json-file", required=True)
    ap.add_argument("--attn_impl", default="flash_attention_2")          # accepted, ignored: one attention kernel exists
    ap.add_argument("--use_liger", default=False)
    ap.add_argument("--no-load-pretrained", action="store_true", help="random-init sub-models (offline smoke runs)")
    ap.add_argument("--max-new-tokens", type=int, default=3072, help="reference hard-codes 3072 (src/model/omics_one.py:223)")
    ap.add_argument("--greedy", action="store_true", help="do_sample=False (the reference always samples)")
    return ap.parse_args(argv)


class MultiModalInfer:
    """reference: src/inference_lora.py:108-339."""

    def __init__(self, args):
        self.args = args
        torch.manual_seed(args.seed)
        if args.device != "cuda" or not torch.cuda.is_available():
            raise Exception("cuda not available, please check env.")          # reference :119-120 (GPU only there too)
        self.device = torch.device("cuda", 0)
        from .loaders import setup_tokenizers
        self.text_tokenizer, self.dna_rna_tokenizer, self.protein_tokenizer, self.real_tokenizers = setup_tokenizers(
            args.text_model_path, args.dna_rna_model_path, args.protein_model_path)
        self._load_model()

    def _load_model(self):
        import molly_amd
        from .config import OmicsModalConfig
        from .train import _preset
        a = self.args
        cfg = OmicsModalConfig(text_config=_preset(a.text_model_path, "text"), dna_rna_config=_preset(a.dna_rna_model_path, "dna"),
                               protein_config=_preset(a.protein_model_path, "protein"))
        cfg.dna_rna_project_token_num, cfg.protein_project_token_num = a.dna_rna_k_tokens, a.protein_k_tokens
        m = molly_amd.OmicsOne(cfg)
        m.set_special_tokens(self.text_tokenizer)
        m.model = molly_amd.Qwen3ForCausalLM(cfg.text_config)
        m.dna_rna_model = molly_amd.EsmForMaskedLM(cfg.dna_rna_config)
        m.protein_model = molly_amd.EsmForMaskedLM(cfg.protein_config)
        if not a.no_load_pretrained:
            from .loaders import load_pretrained
            if not self.real_tokenizers:
                raise RuntimeError("pretrained weights need their tokenizers: the model directories hold no tokenizer files")
            for sub, path, what in ((m.model, a.text_model_path, "LLM"), (m.dna_rna_model, a.dna_rna_model_path, "dna/rna encoder"),
                                    (m.protein_model, a.protein_model_path, "protein encoder")):
                load_pretrained(sub, path, what)
        full = os.path.join(a.trained_model_path, "pytorch_model.bin")
        if not a.use_lora and os.path.exists(full):
            # reference :236-246: the whole OmicsOne state dict, strict like the reference's plain load_state_dict
            sd = torch.load(full, map_location="cpu")
            m.load_state_dict(sd)
            print(f"Multimodal loaded ({len(sd)} tensors).")
        lora_cfg = None
        if a.use_lora and a.lora_live:
            from .lora import LoraConfig
            with open(os.path.join(a.trained_model_path, "adapter_config.json")) as f:
                c = json.load(f)
            lora_cfg = LoraConfig(r=int(c["r"]), lora_alpha=float(c["lora_alpha"]), lora_dropout=0.0)
        m.prepare(self.device, train_llm=False, train_mlp=False, random_init_seed=1234 if a.no_load_pretrained else None,
                  lora=lora_cfg)
        if a.use_lora:
            from .lora import load_live_adapter, merge_lora_adapter
            n = (load_live_adapter if a.lora_live else merge_lora_adapter)(m, a.trained_model_path)
            print(f"LoRA mode enabled: {n} target matrices {'attached' if a.lora_live else 'merged'}.")
        self.model = m

    def run_dataset(self):
        from .data import DatasetConfig, OmicsDataset, qwen_omics_collate_fn_inference
        a = self.args
        d = os.path.dirname(a.json_file)
        if d:
            os.makedirs(d, exist_ok=True)
        cfg = DatasetConfig(max_len=a.max_length, max_src_len=a.max_length, mode="sft", padding=True, input_field="input",
                            output_field="output", dna_rna_k_tokens=a.dna_rna_k_tokens, protein_k_tokens=a.protein_k_tokens,
                            type="Test")
        ds = OmicsDataset(a.dataset_path, self.text_tokenizer, dataset_config=cfg, dna_rna_tokenizer=self.dna_rna_tokenizer,
                          protein_tokenizer=self.protein_tokenizer, read_nums=a.max_samples, type="Test")
        gen = torch.Generator(device=self.device).manual_seed(a.seed)
        n = 0
        with open(a.json_file, "a", encoding="utf-8") as jf:
            for i0 in range(0, len(ds), a.batch_size):
                batch = qwen_omics_collate_fn_inference([ds[i] for i in range(i0, min(len(ds), i0 + a.batch_size))])
                with torch.no_grad():
                    outputs = self.model.generate(
                        input_ids=batch["input_ids"], attention_mask=batch["attention_mask"], omic_ids=batch["omic_ids"],
                        omic_info_list=batch["omic_info_list"], do_sample=not a.greedy, max_length=a.max_length,
                        temperature=a.temperature, top_p=a.top_p, top_k=a.top_k, repetition_penalty=a.repetition_penalty,
                        max_new_tokens=a.max_new_tokens, generator=gen)
                decoded = self.text_tokenizer.batch_decode(outputs.cpu(), skip_special_tokens=True)
                for i, value in enumerate(decoded):
                    json.dump({"decoded_output": value, "input": batch["input"][i], "gt_output": batch["raw_output"][i],
                               "gt_label": batch["raw_label"][i], "task": batch["raw_task"][i], "kind": batch["raw_kind"][i]},
                              jf, ensure_ascii=False)
                    jf.write("\n")
                    jf.flush()
                    n += 1
        return n


def main(argv=None):
    inferer = MultiModalInfer(parse_args(argv))
    n = inferer.run_dataset()
    print(f"wrote {n} samples to {inferer.args.json_file}")


if __name__ == "__main__":
    main()
