#!/usr/bin/env python3
"""Headline benchmark: training tokens/s of Molly-1.7B (Qwen3-1.7B + NT-500M + ESM2-650M), bf16, on N MI355X.

One step = one pass of the hot path over one synthetic mixed protein/text batch: ESM-2 encoder forward -> projector ->
injection -> Qwen3 forward -> fused lm_head+CE -> full backward -> ZeRO-2 step (reduce-scatter, clip, AdamW on the fp32
shard, all-gather).  Nothing is skipped or cached inside the timed region.  Workload = BASELINE.json configs[1]
(SURVEY.md §8d C2): seq_len 2048 text + one 512-residue protein span per sample, B samples per GPU, GA=1, LLM +
projectors trainable, encoders frozen.  Weak scaling: per-GPU work is fixed as N grows.

    python bench.py [--gpus N --steps K --warmup W]        # N>1: launched by torch.distributed.run, one rank per GPU
Prints ONE JSON line on rank 0 (contract in the task statement), with `roofline` for the dominant kernel (the bf16 MFMA
GEMM) measured with HIP events inside the timed region, and `cpu_baseline` = the CPU oracle timed on the host cores.
"""
import argparse
import json
import math
import os
import statistics
import sys
import time

import torch

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

MFMA_BF16_PEAK_TFLOPS = 2500.0        # dense bf16 MFMA, /opt/skills/guides/MI355X_MICROARCH.md "Chip-level parameters"


def algorithmic_flops_per_token(cfg, T):
    """F_llm = 2*P_mm + 4*L*nh*hd*(T/2) per token forward; training = 3x."""
    h, hd, nh, nkv, ff, L, V = (cfg.hidden_size, cfg.head_dim, cfg.num_attention_heads, cfg.num_key_value_heads,
                                cfg.intermediate_size, cfg.num_hidden_layers, cfg.vocab_size)
    p_mm = L * (h * nh * hd + 2 * h * nkv * hd + nh * hd * h + 3 * h * ff) + h * V
    return 3 * (2 * p_mm + 4 * L * nh * hd * (T / 2))


def enc_flops_per_token(cfg, K):
    he, ffe, Le = cfg.hidden_size, cfg.intermediate_size, cfg.num_hidden_layers
    return 2 * Le * (4 * he * he + 2 * he * ffe) + 4 * Le * he * K


def _cpu_step_worker(llm_layers: int, enc_layers: int, threads: int, T: int = 256, K: int = 64):
    """One fwd + bwd + clipped AdamW step of the CPU oracle (oracle/molly_ref.py) at Molly-1.7B widths; prints JSON."""
    from oracle import molly_ref as R
    from molly_amd import config as C
    from molly_amd.params import enc_param_specs, llm_norm_specs, llm_param_specs
    from molly_amd.synth import synth_batch
    torch.manual_seed(0)
    torch.set_num_threads(threads)
    llm_c, prot_c = C.qwen3("1.7b"), C.esm2_650m()
    llm_c.num_hidden_layers, prot_c.num_hidden_layers = llm_layers, enc_layers
    llm = R.LlmCfg(**{k: getattr(llm_c, k) for k in R.LlmCfg.__dataclass_fields__})
    prot = R.EncCfg(**{k: getattr(prot_c, k) for k in R.EncCfg.__dataclass_fields__})
    sd = {}
    for n, shp in llm_param_specs(llm_c) + llm_norm_specs(llm_c):
        t = torch.empty(shp)
        t.fill_(1.0) if "norm" in n else t.normal_(0, 0.02)
        sd[n] = t.requires_grad_(True)
    for n, shp in enc_param_specs(prot_c, "protein_model."):
        t = torch.empty(shp)
        (t.fill_(1.0) if ("LayerNorm" in n or "layer_norm" in n) and n.endswith("weight") else
         (t.zero_() if n.endswith("bias") else t.normal_(0, 0.02)))
        sd[n] = t
    sd["protein_projector.weight"] = torch.empty(llm_c.hidden_size, prot_c.hidden_size).normal_(0, 0.02).requires_grad_(True)
    sd["protein_projector.bias"] = torch.zeros(llm_c.hidden_size, requires_grad=True)
    batch = synth_batch(1, T, [("protein", K)], seed=42)
    params = {n: p for n, p in sd.items() if p.requires_grad}
    state = {n: (torch.zeros_like(p), torch.zeros_like(p)) for n, p in params.items()}
    secs = []
    for step in (1, 2):                      # step 1 faults the memory in; step 2 is the timed one
        t0 = time.time()
        loss, _ = R.omics_forward(sd, llm, None, prot, batch, {"dna_rna": K, "protein": K})
        loss.backward()
        with torch.no_grad():
            total = math.sqrt(sum(float(p.grad.pow(2).sum(dtype=torch.float64)) for p in params.values()))
            coef = min(1.0, 1.0 / (total + 1e-6))
            for n, p in params.items():
                R.adamw_step(p, p.grad * coef, state[n][0], state[n][1], step, 3e-5, 0.0 if R.is_no_decay(n) else 1e-2)
                p.grad = None
        secs.append(time.time() - t0)
    flops = T * algorithmic_flops_per_token(llm_c, T) + K * (enc_flops_per_token(prot_c, K) + 6 * prot_c.hidden_size * llm_c.hidden_size)
    print(json.dumps({"seconds": secs[-1], "first_step_seconds": secs[0], "tokens": T, "llm_layers": llm_layers,
                      "enc_layers": enc_layers, "threads": threads, "K": K, "flops": flops}), flush=True)


def cpu_baseline():
    """The CPU oracle (`kind: port`, oracle/molly_ref.py) timed on the host cores on a BOUNDED sample of the same workload,
    in a child process with a hard timeout so the default bench always finishes in minutes.  A whole fp32 Molly-1.7B step
    (28 Qwen3 + 33 ESM-2 layers, 30 GB of state) needs minutes on a big host just to fault its memory in, so the sample
    is a reduced-DEPTH model at the full widths and vocabulary (4 of 28 Qwen3 layers, 4 of 33 ESM-2 layers; embedding,
    lm_head+CE and optimizer complete), B=1, T=256, one 64-residue protein span, second of two fwd+bwd+clipped-AdamW steps.
    `value` is scaled to the metric's unit through the algorithmic FLOP count: the CPU's sustained FLOP/s on the sample
    divided by the FLOPs per token of the full BASELINE workload (SURVEY.md §8d formula, the same one the GPU line uses)."""
    import subprocess
    from molly_amd import config as C
    threads = max(1, min(os.cpu_count() or 1, 32))            # >32 intra-op threads only add barrier cost at these sizes
    cmd = [sys.executable, os.path.abspath(__file__), "--cpu-baseline-worker", "4", "4", str(threads)]
    try:
        r = subprocess.run(cmd, capture_output=True, text=True, timeout=240, env={**os.environ, "HIP_VISIBLE_DEVICES": ""})
    except subprocess.TimeoutExpired:
        return {"value": None, "unit": "tokens/s", "cores": threads, "kind": "port", "sample": "CPU oracle exceeded its 240 s box"}
    line = [l for l in r.stdout.splitlines() if l.startswith("{")]
    if r.returncode != 0 or not line:
        return {"value": None, "unit": "tokens/s", "cores": threads, "kind": "port", "sample": f"CPU oracle failed: {r.stderr[-200:]}"}
    d = json.loads(line[-1])
    cpu_flops = d["flops"] / d["seconds"]
    T, K = 2048, 512
    full_per_token = (T * algorithmic_flops_per_token(C.qwen3("1.7b"), T) +
                      K * (enc_flops_per_token(C.esm2_650m(), K) + 6 * 1280 * 2048)) / T
    return {"value": round(cpu_flops / full_per_token, 2), "unit": "tokens/s", "cores": threads, "kind": "port",
            "sample": (f"Molly-1.7B widths fp32, reduced depth (4/28 Qwen3 + 4/33 ESM-2 layers, full vocab head), B=1 T={d['tokens']} "
                       f"protein K={d['K']}: 2nd fwd+bwd+clipped-AdamW step {d['seconds']:.1f} s = {cpu_flops / 1e9:.0f} GFLOP/s "
                       f"algorithmic; value = that rate / {full_per_token / 1e9:.2f} GFLOP per token of the full T=2048 workload")}


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=8)
    ap.add_argument("--warmup", type=int, default=2)
    ap.add_argument("--batch", type=int, default=8, help="samples per GPU per step")
    ap.add_argument("--seq", type=int, default=2048)
    ap.add_argument("--k-protein", type=int, default=512)
    ap.add_argument("--model", default="1.7b")
    ap.add_argument("--train-mode", choices=("full", "lora", "mlp", "bio"), default="full",
                    help="full = headline (--train-llm --train-mlp); lora = --use-lora r=64; mlp = projectors only (side figures)")
    ap.add_argument("--zero-stage", type=int, default=2, choices=(0, 2),
                    help="2 = ZeRO-2 (reduce-scatter / sharded AdamW / all-gather); 0 = the reference's ds_z0 fallback (all-reduce)")
    ap.add_argument("--event-stride", type=int, default=7,
                    help="HIP events around every n-th GEMM launch of the timed region (1 = all: 2-3 %% slower steps)")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--cpu-baseline-worker", nargs=3, type=int, metavar=("LLM_LAYERS", "ENC_LAYERS", "THREADS"))
    args = ap.parse_args()
    if args.cpu_baseline_worker:
        _cpu_step_worker(*args.cpu_baseline_worker)
        return

    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local = int(os.environ.get("LOCAL_RANK", "0"))
    # test hooks (a one-GPU box can still run the multi-process path end to end): MOLLY_BENCH_DEVICE pins every rank to
    # one device, MOLLY_DIST_BACKEND=gloo replaces RCCL (which refuses two ranks on one GPU)
    if "MOLLY_BENCH_DEVICE" in os.environ:
        local = int(os.environ["MOLLY_BENCH_DEVICE"])
    torch.cuda.set_device(local)
    dev = torch.device("cuda", local)
    import torch.distributed as dist
    if world > 1:
        backend = os.environ.get("MOLLY_DIST_BACKEND", "nccl")
        dist.init_process_group(backend, **({"device_id": dev} if backend == "nccl" else {}))
    assert world == args.gpus, f"--gpus {args.gpus} but WORLD_SIZE={world}: launch with torch.distributed.run"

    import __graft_entry__ as ge
    if rank == 0:
        ge.build()
    if world > 1:
        dist.barrier()
    import molly_amd
    from molly_amd import config as C, ops
    from molly_amd.synth import synth_batch
    from molly_amd.trainer import Zero2Optimizer

    cfg = C.molly(args.model, k_tokens=args.k_protein)
    m = molly_amd.OmicsOne(cfg)
    m.model = molly_amd.Qwen3ForCausalLM(cfg.text_config)
    m.dna_rna_model = molly_amd.EsmForMaskedLM(cfg.dna_rna_config)
    m.protein_model = molly_amd.EsmForMaskedLM(cfg.protein_config)
    torch.manual_seed(1234)
    if args.train_mode in ("full", "bio"):
        m.prepare(dev, random_init_seed=1234, train_bio=args.train_mode == "bio")   # same seed on every rank: replicas start identical
    else:
        from molly_amd.lora import LoraConfig
        m.prepare(dev, random_init_seed=1234, train_llm=False, train_mlp=True,
                  lora=LoraConfig(r=64, lora_alpha=64, lora_dropout=0.05, seed=42) if args.train_mode == "lora" else None)
    rt = m._rt
    opt = Zero2Optimizer(rt.P.flat, rt.G.flat, m.n_decay, lr=3e-5, weight_decay=1e-2, max_grad_norm=1.0,
                         stage=args.zero_stage)
    m.attach_optimizer(opt)

    B, T, K = args.batch, args.seq, args.k_protein
    batches = [synth_batch(B, T, [("protein", K)], seed=42 + rank + 1000 * i) for i in range(4)]

    def step(i):
        b = batches[i % len(batches)]
        loss = m.forward_backward(b["input_ids"], b["attention_mask"], b["omic_ids"], b["omic_info_list"], b["labels"])
        opt.step(lr=3e-5)
        return loss

    for i in range(args.warmup):
        step(i)
    torch.cuda.synchronize()
    if world > 1:
        dist.barrier()
    torch.cuda.synchronize()
    ops.GEMM_PROFILE = []
    ops.GEMM_PROFILE_STRIDE = args.event_stride
    step_ev = []
    t0 = time.perf_counter()
    for i in range(args.steps):
        e0 = torch.cuda.Event(enable_timing=True); e0.record()
        loss = step(args.warmup + i)
        e1 = torch.cuda.Event(enable_timing=True); e1.record()
        step_ev.append((e0, e1))
    torch.cuda.synchronize()
    if world > 1:
        dist.barrier()
    torch.cuda.synchronize()
    dt = time.perf_counter() - t0
    prof, ops.GEMM_PROFILE = ops.GEMM_PROFILE, None
    if world > 1:
        tmax = torch.tensor([dt], device=dev, dtype=torch.float64)
        dist.all_reduce(tmax, op=dist.ReduceOp.MAX)
        dt = float(tmax.item())
    loss_v = float(loss.item())

    if rank == 0:
        tokens = world * B * T * args.steps
        # dominant kernel = gemm256_kernel (the 256x256 ping-pong MFMA GEMM); its three operand-layout instantiations
        # are separate rows in rocprofv3 --stats.  Launches that went to the 128x128 kernel (small grids) are listed apart.
        # every GEMM launch of the timed region is listed; HIP events bracket every `stride`-th one (7: coprime with the 4- and
        # 9-GEMM patterns of the decoder / encoder layers, so every shape is sampled).  Per class, the launches that were not
        # timed are priced at the class's sampled rate.
        n_launch = len(prof)
        timed = [(a.elapsed_time(b), f, kcfg, lay) for a, b, f, kcfg, lay in prof if a is not None]
        achieved = sum(f for _, f, *_ in timed) / (sum(t for t, *_ in timed) * 1e-3) / 1e12
        by_kernel = {}
        names = {(False, False): "NT gemm256_kernel<false,false>", (False, True): "NN gemm256_kernel<false,true>",
                 (True, True): "TN gemm256_kernel<true,true>"}
        def klass(kcfg, lay):
            key = names[lay] if kcfg % 1000 == 512 else "gemm_kernel<...,128,2,64>"
            if kcfg >= 100000:                    # grouped launch: cfg = 512 + 1000 + 100000 * problems
                key += f" grouped x{kcfg // 100000} (a layer's weight gradients in one launch)"
            elif kcfg // 1000 > 1:
                key += " + splitk_reduce"
            return key
        for a, b, f, kcfg, lay in prof:
            d = by_kernel.setdefault(klass(kcfg, lay), [0, 0, 0.0, 0.0, 0.0])     # launches, timed, ms timed, flops timed, flops all
            d[0] += 1; d[4] += f
            if a is not None:
                d[1] += 1; d[2] += a.elapsed_time(b); d[3] += f
        gemm_ms = sum(v[2] * v[4] / v[3] for v in by_kernel.values() if v[3] > 0)  # all launches, at each class's sampled rate
        by_kernel = {k: {"launches": v[0], "timed": v[1], "avg_launch_us": round(v[2] * 1e3 / max(v[1], 1), 2),
                         "achieved_tflops": round(v[3] / (v[2] * 1e-3) / 1e12, 1) if v[2] > 0 else None}
                     for k, v in by_kernel.items()}
        step_ms = [a.elapsed_time(b) for a, b in step_ev]
        # HBM-side traffic of the dominant kernel cannot be counted live (PMC needs rocprofv3): quote the committed PMC
        # passes of this same workload (profiles/r01b_pmc_hbm_traffic.csv, tools/pmc_hbm_traffic.py; corrected as MI355X_MICROARCH.md §HBM prescribes)
        traffic, traffic_src = None, None
        tj = os.path.join(os.path.dirname(os.path.abspath(__file__)), "profiles", "r01b_hbm_traffic.json")
        if args.train_mode == "full" and (B, T, K, args.model) == (8, 2048, 512, "1.7b") and os.path.exists(tj):
            with open(tj) as f:
                tr = json.load(f)
            traffic, traffic_src = tr["gemm_hbm_bytes_per_launch"], tr["source"]
        flops_step = B * (T * algorithmic_flops_per_token(cfg.text_config, T) +
                          K * (enc_flops_per_token(cfg.protein_config, K) + 3 * 2 * cfg.protein_config.hidden_size *
                               cfg.text_config.hidden_size))
        out = {
            "metric": f"training tokens/sec Molly-{args.model.upper()} bf16", "value": round(tokens / dt, 1), "unit": "tokens/s",
            "n_gpus": world, "steps": args.steps, "warmup": args.warmup, "ms_per_step": round(dt / args.steps * 1e3, 2),
            "higher_is_better": True, "scaling": "weak", "vs_baseline": None, "dtype": "bf16", "data": "synthetic",
            "config": {"workload": f"Molly-{args.model.upper()} (Qwen3-{args.model.upper()} + NT-500M + ESM2-650M) train step, "
                                   f"seq_len {T} text + {K}-residue protein span per sample, {B} samples/GPU, GA=1, "
                                   + {"full": "LLM+projectors trainable", "bio": "LLM+projectors+encoders trainable (--train-bio)", "lora": "LoRA r=64 adapters+projectors trainable, base frozen",
                                      "mlp": "projectors trainable, LLM frozen"}[args.train_mode] +
                                   ("" if args.train_mode == "bio" else ", encoders frozen") + f", ZeRO-{args.zero_stage} dp{world}",
                       "global_batch": world * B, "seq_len": T, "parallelism": f"dp{world}",
                       "scored_token_fraction": 0.25,
                       "note": "prompt = 75% of each sample with labels -100 (SURVEY 8d); lm_head+CE run on the scored rows only "
                               "(identical loss/gradients); model_tflops_per_gpu uses the full algorithmic FLOP count"},
            "step_ms_p50": round(statistics.median(step_ms), 2),
            "model_tflops_per_gpu": round(flops_step / (dt / args.steps) / 1e12, 1),
            "mfma_roofline_frac_step": round(flops_step / (dt / args.steps) / 1e12 / MFMA_BF16_PEAK_TFLOPS, 4),
            "loss": round(loss_v, 4),
            "roofline": {"bound": "mfma", "kernel": "bf16 MFMA GEMM (gemm256_kernel / gemm_kernel, all launches of the timed region)",
                         "achieved": round(achieved, 1), "peak": MFMA_BF16_PEAK_TFLOPS, "unit": "TFLOP/s",
                         "frac": round(achieved / MFMA_BF16_PEAK_TFLOPS, 4), "traffic": traffic, "traffic_unit": "bytes/launch (L2 fabric-side: "
                         "Infinity Cache + HBM)", "traffic_source": traffic_src, "launches": n_launch,
                         "avg_launch_us": round(gemm_ms * 1e3 / n_launch, 2),
                         "gemm_share_of_step": round(gemm_ms / sum(step_ms), 3), "by_kernel": by_kernel},
        }
        if not args.no_cpu_baseline:
            out["cpu_baseline"] = cpu_baseline()
        print(json.dumps(out), flush=True)
    if world > 1:
        dist.barrier()
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
