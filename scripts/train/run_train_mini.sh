#!/usr/bin/env bash
# The "mini" smoke run the reference's README advertises (README.md:60-61) but does not ship: tiny Qwen3-shape LLM + tiny
# ESM-shape encoders, a handful of protein/DNA-text samples, a few optimizer steps on one MI355X.
set -euo pipefail
cd "$(dirname "$0")/../.."
python - <<'PY'
import json
rows = [dict(task="Solubility-Solubility", input=f"Is <protein>{'MKTAYIAKQR' * (1 + i % 3)}</protein> soluble?", think="",
             output="Yes." if i % 2 else "No.", label=str(i % 2), kind="protein", task_num=i) for i in range(16)]
rows += [dict(task="tf-h", input=f"Does <dna>{'ACGTTGCA' * (2 + i % 4)}</dna> bind?", think="", output="It does.", label="1",
              kind="dna", task_num=i) for i in range(16)]
open("/tmp/molly_mini.jsonl", "w").write("\n".join(json.dumps(r) for r in rows))
PY
python -m molly_amd.train --experiment-name mini --output_dir /tmp/molly_mini_out \
  --text-model-path tiny --dna-rna-model-path tiny --protein-model-path tiny --no-load-pretrained \
  --dna-rna-k-tokens 64 --protein-k-tokens 64 --train-mlp --train-llm \
  --train-dataset-path /tmp/molly_mini.jsonl --max-len 256 --mode sft \
  --per_device_train_batch_size 4 --gradient-accumulation-steps 2 --num_train_epochs 2 --learning_rate 1e-3 \
  --logging_steps 1 --warmup_ratio 0.1 --bf16 --seed 42
