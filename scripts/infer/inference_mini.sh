#!/usr/bin/env bash
# Counterpart of the reference's scripts/infer/inference_nt_lora.sh on the mini model: batch-generate over the mini dataset
# with the LoRA adapter that scripts/train/run_train_mini.sh --use-lora (or any molly_amd.train run) left in $1.
# usage: scripts/infer/inference_mini.sh <checkpoint-dir> <output.jsonl> [--use-lora [--lora-live]]
set -euo pipefail
CKPT="${1:?Usage: $0 <checkpoint-dir> <output.jsonl> [extra flags]}"
OUT="${2:?Usage: $0 <checkpoint-dir> <output.jsonl> [extra flags]}"
shift 2
cd "$(dirname "$0")/../.."
python -m molly_amd.inference \
  --text-model-path tiny --dna-rna-model-path tiny --protein-model-path tiny --no-load-pretrained \
  --dna-rna-k-tokens 64 --protein-k-tokens 64 \
  --trained-model-path "$CKPT" --dataset-path /tmp/molly_mini.jsonl \
  --max-length 256 --batch-size 8 --temperature 0.8 --top-p 0.95 --repetition-penalty 1.1 --seed 42 \
  --max-new-tokens 64 --json-file "$OUT" "$@"
