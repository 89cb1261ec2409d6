#!/usr/bin/env python3
"""Headline benchmark: training tokens/s of Molly-1.7B (Qwen3-1.7B + NT-500M + ESM2-650M), bf16, on N MI355X.

One step = one pass of the hot path over one synthetic mixed protein/text batch: ESM-2 encoder forward -> projector ->
injection -> Qwen3 forward -> fused lm_head+CE -> full backward -> ZeRO-2 step (reduce-scatter, clip, AdamW on the fp32
shard, all-gather).  Nothing is skipped or cached inside the timed region.  Workload = BASELINE.json configs[1]
(SURVEY.md §8d C2): seq_len 2048 text + one 512-residue protein span per sample, B samples per GPU, GA=1, LLM +
projectors trainable, encoders frozen.  Weak scaling: per-GPU work is fixed as N grows.

    python bench.py [--gpus N --steps K --warmup W]        # N>1: one rank per GPU — under torch.distributed.run when the
                                                           # caller launched it so (WORLD_SIZE set), else bench.py starts
                                                           # the N ranks itself (fresh children, before any GPU call)
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


DEFAULT_BATCH = 16          # samples per GPU of the headline workload (tests/test_gpu_fullsize.py checks the kernels at THIS batch's shapes)


def algorithmic_flops_per_token(cfg, T):
    """F_llm = 2*P_mm + 4*L*nh*hd*(T/2) per token forward; training = 3x."""
    h, hd, nh, nkv, ff, L, V = (cfg.hidden_size, cfg.head_dim, cfg.num_attention_heads, cfg.num_key_value_heads,
                                cfg.intermediate_size, cfg.num_hidden_layers, cfg.vocab_size)
    p_mm = L * (h * nh * hd + 2 * h * nkv * hd + nh * hd * h + 3 * h * ff) + h * V
    return 3 * (2 * p_mm + 4 * L * nh * hd * (T / 2))


def enc_flops_per_token(cfg, K):
    he, ffe, Le = cfg.hidden_size, cfg.intermediate_size, cfg.num_hidden_layers
    return 2 * Le * (4 * he * he + 2 * he * ffe) + 4 * Le * he * K


def _host_cpu():
    """(physical cores, CPU model string) from lscpu; falls back to os.cpu_count()."""
    import subprocess
    cores, model = None, "unknown"
    try:
        txt = subprocess.run(["lscpu"], capture_output=True, text=True, timeout=10).stdout
        kv = {l.split(":", 1)[0].strip(): l.split(":", 1)[1].strip() for l in txt.splitlines() if ":" in l}
        model = kv.get("Model name", model)
        cores = int(kv["Core(s) per socket"]) * int(kv["Socket(s)"])
    except Exception:
        pass
    avail = len(os.sched_getaffinity(0)) if hasattr(os, "sched_getaffinity") else (os.cpu_count() or 1)
    cores = min(cores or avail, avail)                      # never more threads than this process may run on
    return max(1, cores), model


# Intra-op threads of the CPU oracle.  SURVEY 8d says "physical cores"; measured on the GPU host of this pool (2 x EPYC 9575F,
# 128 physical cores, gpurun_out/r02_t2_cpu*.log): the down-scaled C2 step takes 31.5 s with 128 threads, 17.3 s with 64 and
# 12.0 s with 32 — at 512 tokens per step the matmuls are too small to feed two sockets, barriers and NUMA traffic dominate.
# The baseline therefore uses min(physical cores, 32) threads (the CPU's best case of the three), and reports both numbers.
CPU_THREAD_CAP = 32


# the two CPU-baseline configurations of SURVEY.md §8d / BASELINE.md §3
CPU_CONFIGS = {
    # C1 exactly: BASELINE configs[0] — Qwen3-0.6B + ESM2-t6-8M, fp32, B=2, T=256, one 64-residue protein span
    "c1": dict(llm="0.6b", enc="esm2_8m", B=2, T=256, K=64),
    # C2 down-scaled: the headline model (Qwen3-1.7B + ESM2-650M, full depth, full vocabulary), B=1, T=512, K=128
    "c2s": dict(llm="1.7b", enc="esm2_650m", B=1, T=512, K=128),
}


def _cpu_step_worker(which: str, threads: int, budget_s: float, warmups: int = 2, timed: int = 5):
    """fwd + bwd + clipped AdamW steps of the CPU oracle (oracle/molly_ref.py, fp32) on one of CPU_CONFIGS; prints JSON.
    Protocol: `warmups` un-timed steps, then up to `timed` timed ones, median reported; the loop stops
    early when `budget_s` of wall time is used up (the steps done so far are reported, at least one timed step)."""
    from oracle import molly_ref as R
    from molly_amd import config as C
    from molly_amd.params import enc_param_specs, llm_norm_specs, llm_param_specs
    from molly_amd.synth import synth_batch
    t_start = time.time()
    torch.manual_seed(0)
    torch.set_num_threads(threads)
    cc = CPU_CONFIGS[which]
    llm_c = C.qwen3(cc["llm"])
    prot_c = C.esm2_650m() if cc["enc"] == "esm2_650m" else C.esm2_t6_8m()
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
    B, T, K = cc["B"], cc["T"], cc["K"]
    batch = synth_batch(B, T, [("protein", K)], seed=42)         # same generator and seed as the GPU run's rank 0
    params = {n: p for n, p in sd.items() if p.requires_grad}
    state = {n: (torch.zeros_like(p), torch.zeros_like(p)) for n, p in params.items()}
    secs, step = [], 0
    while len(secs) < warmups + timed:
        if len(secs) > warmups and time.time() - t_start > budget_s:
            break
        step += 1
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
        if len(secs) <= warmups and time.time() - t_start > budget_s and len(secs) >= 1:
            warmups = len(secs)                                  # out of time inside the warm-ups: one timed step still follows
    timed_s = secs[warmups:] or secs[-1:]
    print(json.dumps({"config": which, "B": B, "T": T, "K": K, "threads": threads, "warmups": min(warmups, len(secs) - 1) if secs[warmups:] else len(secs) - 1,
                      "step_seconds": [round(x, 3) for x in timed_s], "median_seconds": statistics.median(timed_s),
                      "tokens_per_step": B * T, "loss": float(loss.detach())}), flush=True)


def _run_cpu_config(which: str, threads: int, budget_s: float, warmups: int = 2, timed: int = 5):
    import subprocess
    cmd = [sys.executable, os.path.abspath(__file__), "--cpu-baseline-worker", which, str(threads), str(budget_s), str(warmups), str(timed)]
    try:
        r = subprocess.run(cmd, capture_output=True, text=True, timeout=budget_s * 3 + 120,
                           env={**os.environ, "HIP_VISIBLE_DEVICES": "", "WORLD_SIZE": "1"})
    except subprocess.TimeoutExpired:
        return None, f"CPU oracle ({which}) exceeded its box"
    line = [l for l in r.stdout.splitlines() if l.startswith("{")]
    if r.returncode != 0 or not line:
        return None, f"CPU oracle ({which}) failed: {r.stderr[-200:]}"
    return json.loads(line[-1]), None


def cpu_baseline(budget_s: float = 150.0):
    """The CPU oracle (`kind: port`, oracle/molly_ref.py, parity-pinned to the reference through tests/golden) timed on the
    host's physical cores, protocol of SURVEY.md §8d / BASELINE.md §3: the down-scaled C2 — the headline MODEL at full depth
    and vocabulary (Qwen3-1.7B + ESM2-650M, fp32), B=1, T=512, one 128-residue protein span — and C1 exactly (Qwen3-0.6B +
    ESM2-t6-8M, fp32, B=2, T=256, K=64); fwd + bwd + clipped AdamW; 2 warm-up steps then the median of up to 5 timed steps,
    each configuration inside a wall-time box so that the default bench finishes in minutes.  `value` = measured tokens/s
    of the down-scaled C2 (tokens of the step / median step time — no FLOP extrapolation); C1 rides along in `c1`.
    Each runs in a child process that never touches the GPU."""
    physical, model = _host_cpu()
    cores = min(physical, CPU_THREAD_CAP)
    out = {"value": None, "unit": "tokens/s", "cores": cores, "physical_cores": physical, "cpu_model": model, "kind": "port"}
    d2, err2 = _run_cpu_config("c2s", cores, budget_s)
    d1, err1 = _run_cpu_config("c1", cores, budget_s / 2)
    if d2 is not None:
        out["value"] = round(d2["tokens_per_step"] / d2["median_seconds"], 2)
        out["timed_steps"] = len(d2["step_seconds"])
        out["warmup_steps"] = d2["warmups"]
        out["median_step_seconds"] = round(d2["median_seconds"], 3)
        out["sample"] = (f"C2 down-scaled (SURVEY 8d): Molly-1.7B full depth fp32 (Qwen3-1.7B + ESM2-650M), B={d2['B']} T={d2['T']} "
                         f"protein K={d2['K']}, fwd+bwd+clipped AdamW, {d2['warmups']} warm-up + {len(d2['step_seconds'])} timed "
                         f"steps {d2['step_seconds']} s, median {d2['median_seconds']:.2f} s, {cores} threads on {physical} physical cores of {model} "
                         f"(more threads are slower at this size: 31.5 / 17.3 / 12.0 s per step at 128 / 64 / 32 threads, measured)")
    else:
        out["sample"] = err2
    if d1 is not None:
        out["c1"] = {"value": round(d1["tokens_per_step"] / d1["median_seconds"], 2), "unit": "tokens/s",
                     "timed_steps": len(d1["step_seconds"]), "median_step_seconds": round(d1["median_seconds"], 3),
                     "sample": (f"C1 exactly: Qwen3-0.6B + ESM2-t6-8M fp32, B={d1['B']} T={d1['T']} K={d1['K']}, {d1['warmups']} warm-up + "
                                f"{len(d1['step_seconds'])} timed steps {d1['step_seconds']} s, median {d1['median_seconds']:.2f} s")}
    else:
        out["c1"] = {"value": None, "sample": err1}
    # SURVEY 8d prescribes the PHYSICAL core count; `value` uses 32 threads because that is this host's best case (above).  The prescribed
    # figure is measured too, in a smaller box (1 warm-up + up to 2 timed steps), so that both are on record in every line
    if physical > cores and d2 is not None:
        dp, errp = _run_cpu_config("c2s", physical, min(budget_s, 100.0), warmups=1, timed=2)
        out["physical_cores_run"] = ({"threads": physical, "value": round(dp["tokens_per_step"] / dp["median_seconds"], 2), "unit": "tokens/s",
                                      "median_step_seconds": round(dp["median_seconds"], 3), "timed_steps": len(dp["step_seconds"]),
                                      "sample": "the same down-scaled C2 step on every physical core (SURVEY 8d's prescription), 1 warm-up step"}
                                     if dp is not None else {"threads": physical, "value": None, "sample": errp})
    return out


HBM_PEAK_TBPS = 8.0                   # MI355X_MICROARCH.md "Chip-level parameters" (6.3 TB/s is what a float4 copy reaches)


def _c5_worker():
    """BASELINE configs[4] ('prefill + decode tokens/s for C5'; reference scripts/infer/inference_nt_lora.sh:18-33,
    src/model/omics_one.py:220-232): Molly-8B with the LoRA adapter merged at load (= the base model's kernels), batch 32,
    3072-token left-aligned prompts with one 1024-residue protein span, greedy decode over the KV cache.  Prefill is priced
    against the MFMA roof, the decode step against HBM (it streams every weight and the whole KV cache once per token)."""
    import molly_amd
    from molly_amd import config as C, ops
    from molly_amd.generate import GenerationSession
    from molly_amd.synth import synth_batch
    dev = torch.device("cuda", 0)
    B, T, K, NEW = 32, 3072, 1024, 48
    cfg = C.molly("8b", k_tokens=K)
    m = molly_amd.OmicsOne(cfg)
    m.model = molly_amd.Qwen3ForCausalLM(cfg.text_config)
    m.dna_rna_model = molly_amd.EsmForMaskedLM(cfg.dna_rna_config)
    m.protein_model = molly_amd.EsmForMaskedLM(cfg.protein_config)
    m.prepare(dev, train_llm=False, train_mlp=False, random_init_seed=1234)
    b = synth_batch(B, T, [("protein", K)], seed=1)
    res = None
    pre_runs = []
    for rep in range(3):                                   # first pass warms allocations and kernel attributes; the third is a second prefill only
        sess = GenerationSession(m, 2 * NEW)              # room for the greedy steps and the sampled ones behind them
        e0, e1, e2, e3 = (torch.cuda.Event(enable_timing=True) for _ in range(4))
        e0.record()
        logits = sess.prefill(b["input_ids"], b["attention_mask"], b["omic_ids"], b["omic_info_list"])
        e1.record()
        if rep == 2:                                       # (the prefill holds host work — mask to the CPU, cache allocation — and was timed ONCE: one line of the
            torch.cuda.synchronize()                       # round read 1,886 ms beside 1,214 / 1,236 on other boxes; the smaller of two runs is reported, both kept)
            pre_runs.append(e0.elapsed_time(e1))
            del sess
            break
        for _ in range(8):                                 # eager step + graph capture + first replays: not timed
            logits = sess.step(ops.argmax(logits))
        e2.record()
        for _ in range(NEW - 8):
            logits = sess.step(ops.argmax(logits))
        e3.record()
        torch.cuda.synchronize()
        res = (e0.elapsed_time(e1), e2.elapsed_time(e3) / (NEW - 8))
        if rep == 1:
            pre_runs.append(res[0])
        # the same steps with the REFERENCE's inference settings (src/inference_lora.py:293-298: do_sample, temperature 0.8, top-p 0.95,
        # top-k 20, repetition penalty 1.1): one sampling launch per step on the growing history instead of the argmax
        if rep == 1:
            hist = torch.empty(B, 0, dtype=torch.int64, device=dev)
            s0, s1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            for it in range(NEW - 8):
                if it == 4:
                    s0.record()
                nxt = ops.sample_logits(logits, hist if hist.shape[1] else None, 1.1, 0.8, 20, 0.95, 1234, it)
                hist = torch.cat([hist, nxt[:, None]], 1)
                logits = sess.step(nxt)
            s1.record()
            torch.cuda.synchronize()
            n_s = max(hist.shape[1] - 4, 1)
            samp_ms = s0.elapsed_time(s1) / n_s
        del sess
    pre_ms, dec_ms = min(pre_runs), res[1]
    t, pc = cfg.text_config, cfg.protein_config
    h, hd, nh, nkv, ff, L, V = (t.hidden_size, t.head_dim, t.num_attention_heads, t.num_key_value_heads, t.intermediate_size,
                                t.num_hidden_layers, t.vocab_size)
    p_layers = L * (h * nh * hd + 2 * h * nkv * hd + nh * hd * h + 3 * h * ff)
    pre_flops = B * (T * (2 * p_layers + 4 * L * nh * hd * (T / 2)) + 2 * h * V +
                     K * (enc_flops_per_token(pc, K) + 2 * pc.hidden_size * h))
    t_mid = T + 8 + (NEW - 8) / 2                                   # average cache length over the timed steps
    w_bytes = 2 * (p_layers + h * V)                                # every layer matrix + the (untied) lm_head, bf16
    kv_bytes = 2 * L * B * t_mid * nkv * hd * 2
    t_mid_s = T + NEW + 4 + n_s / 2                                 # the sampled steps: average cache length over THEIR timed region
    kv_bytes_s = 2 * L * B * t_mid_s * nkv * hd * 2
    out = {"workload": f"Molly-8B, LoRA merged at load, batch {B}, prompt {T} (one {K}-residue protein span), greedy, {NEW - 8} timed "
                       "decode steps through the captured hipGraph",
           "prefill": {"ms": round(pre_ms, 1), "ms_runs": [round(x, 1) for x in pre_runs], "tokens_per_s": round(B * T / pre_ms * 1e3, 1),
                       "achieved_tflops": round(pre_flops / pre_ms / 1e9, 1), "bound": "mfma",
                       "frac": round(pre_flops / pre_ms / 1e9 / MFMA_BF16_PEAK_TFLOPS, 4)},
           "decode": {"ms_per_step": round(dec_ms, 3), "tokens_per_s": round(B / dec_ms * 1e3, 1), "bound": "hbm",
                      "bytes_per_step": int(w_bytes + kv_bytes), "weights_bytes": int(w_bytes), "kv_cache_bytes": int(kv_bytes),
                      "achieved_TBps": round((w_bytes + kv_bytes) / dec_ms / 1e9, 3), "peak_TBps": HBM_PEAK_TBPS,
                      "frac": round((w_bytes + kv_bytes) / dec_ms / 1e9 / HBM_PEAK_TBPS, 4)},
           # same session, the reference's own sampling settings (temperature 0.8, top-p 0.95, top-k 20, repetition penalty 1.1:
           # src/inference_lora.py:293-298) — the fused sampling launch (csrc/sampling.hip) replaces the argmax of every step
           "decode_sampled": {"ms_per_step": round(samp_ms, 3), "tokens_per_s": round(B / samp_ms * 1e3, 1),
                              "settings": "do_sample, T 0.8, top_p 0.95, top_k 20, repetition_penalty 1.1", "steps_timed": int(n_s),
                              # (these steps run BEHIND the greedy ones: the cache is NEW tokens longer at their start, + 4 untimed sampled steps)
                              "kv_cache_bytes": int(kv_bytes_s),
                              "frac": round((w_bytes + kv_bytes_s) / samp_ms / 1e9 / HBM_PEAK_TBPS, 4)}}
    print(json.dumps(out), flush=True)
    return 0


SECONDARY = {
    # BASELINE configs[2] / [3] at their DEFINING per-GPU batch (B = 1: scripts/train/examples/run_train_4B_z2_b1.sh:29,47,
    # run_train_8B_z0_b1.sh:29,47), GA = 2
    "c3": ["--model", "4b", "--batch", "1", "--seq", "3072", "--micro", "dna:512,rna:512,protein:512;dna:512,rna:512,protein:512"],
    "c4": ["--model", "8b", "--batch", "1", "--seq", "4096", "--micro", "protein:1024;dna:1000"],
    "c5": [],
    # the headline workload under --use-lora (reference src/utils/tools.py:378-389: r 64, alpha 64, dropout 0.05 on every Linear of the
    # LLM; base frozen, adapters + projectors trainable) — VERDICT r04 item 7
    "lora": ["--train-mode", "lora"],
}


def secondary_block(budget_s: float):
    """BASELINE configs 3 / 4 / 5 beside the headline line: each in a child process of its own (fresh GPU memory,
    a failure or a timeout becomes a string in the record, never an exception), all inside one wall-time box."""
    import subprocess
    t_end = time.time() + budget_s
    out = {}
    for name in ("c5", "c3", "c4", "lora"):
        left = t_end - time.time()
        if left < 45:
            out[name] = "skipped: the secondary block's time box was used up"
            continue
        cmd = [sys.executable, os.path.abspath(__file__), "--secondary-worker", name, "--no-cpu-baseline", "--no-secondary",
               "--steps", "4", "--warmup", "2", *SECONDARY[name]]
        try:
            r = subprocess.run(cmd, capture_output=True, text=True, timeout=left, env={**os.environ, "WORLD_SIZE": "1", "RANK": "0"})
        except subprocess.TimeoutExpired:
            out[name] = f"timed out after {left:.0f} s"
            continue
        line = [l for l in r.stdout.splitlines() if l.startswith("{")]
        if r.returncode != 0 or not line:
            out[name] = f"failed (rc {r.returncode}): {r.stderr[-300:]}"
            continue
        d = json.loads(line[-1])
        if name == "c5":
            out[name] = d
        else:
            rf = d["roofline"]
            out[name] = {"workload": d["config"]["workload"], "ms_per_step": d["ms_per_step"], "tokens_per_s": d["value"],
                         "executed_tflops_per_gpu": d["executed_tflops_per_gpu"],
                         "mfma_roofline_frac_step_executed": d["mfma_roofline_frac_step_executed"],
                         "gemm": {"bound": "mfma", "achieved": rf["achieved"], "peak": rf["peak"], "frac": rf["frac"],
                                  "launches": rf["launches"], "gemm_share_of_step": rf["gemm_share_of_step"],
                                  "by_kernel": {k: v["achieved_tflops"] for k, v in rf["by_kernel"].items()}}}
    return out


def _self_launch(args, argv):
    """`python bench.py --gpus N` with no launcher around it: start the N ranks ourselves (one process per GPU, reference:
    scripts/train/examples/run_train_1B_z2_b1.sh:60-64 `deepspeed --include localhost:0..7`).  Runs BEFORE anything touches
    the GPU in this process; the ranks are fresh children (never an exec of a GPU-initialised process); returns their rc."""
    import socket
    import subprocess
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        port = s.getsockname()[1]
    env = dict(os.environ)
    env.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")          # dmabuf IPC: RCCL needs it on this pool
    env.setdefault("OMP_NUM_THREADS", "4")
    if not args.dry_run_launch:
        import __graft_entry__ as ge
        ge.build()                                             # build (or verify the digests) ONCE, before any rank exists
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", f"--nproc-per-node={args.gpus}",
           "--master-addr", "127.0.0.1", "--master-port", str(port), os.path.abspath(__file__),
           *[a for a in argv if a != "--dry-run-launch"]]
    if args.dry_run_launch:
        print(json.dumps({"launch": cmd}), flush=True)
        return 0
    return subprocess.run(cmd, env=env).returncode


def parse_micro(spec: str):
    """"protein:1024;dna:1000" -> [[("protein", 1024)], [("dna", 1000)]]: the micro-batches of one optimizer step (GA = their
    number), each a list of omic spans every sample of that micro-batch carries."""
    out = []
    for mb in spec.split(";"):
        spans = []
        for sp in mb.split(","):
            typ, k = sp.strip().split(":")
            assert typ in ("dna", "rna", "protein"), typ
            spans.append((typ, int(k)))
        out.append(spans)
    return out


def attention_flops_per_step(cfg, enc_cfgs, B, T, micro, train_bio=False):
    """Executed attention matmul FLOPs counted the way SURVEY §8d counts them (causal half for the decoder, fwd + 2x bwd,
    no recompute credit; full K x K forward only for a frozen encoder).  enc_cfgs = {"dna_rna": cfg, "protein": cfg};
    micro = parse_micro(...)."""
    llm = 3 * 4 * cfg.num_hidden_layers * cfg.num_attention_heads * cfg.head_dim * (T / 2) * T * B * len(micro)
    enc = 0.0
    for spans in micro:
        for typ, K in spans:
            ec = enc_cfgs["protein" if typ == "protein" else "dna_rna"]
            enc += (3 if train_bio else 1) * 4 * ec.num_hidden_layers * ec.hidden_size * K * K * B
    return llm + enc


def vendor_gemm_yardstick(dev, M):
    """torch.matmul (the vendor library: hipBLASLt / rocBLAS) beside molly's own kernel on the forward and dgrad shapes of the
    decoder layer at this batch, same process, same box, OUTSIDE the timed region: a same-node yardstick next to `vs_baseline:
    null`.  Random data, best of 5 interleaved rounds of 5 launches each, HIP events.  TFLOP/s."""
    import torch
    from molly_amd import ops
    g = torch.Generator(device=dev).manual_seed(0)
    rnd = lambda *s: (torch.rand(*s, device=dev, generator=g) * 2 - 1).bfloat16()
    shapes = [("qkv fwd", "nt", 4096, 2048), ("o fwd", "nt", 2048, 2048), ("gate|up fwd", "nt", 12288, 2048), ("down fwd", "nt", 2048, 6144),
              ("qkv dgrad", "nn", 2048, 4096), ("down dgrad", "nn", 6144, 2048)]

    def t_pair(f, g_):
        """best of 5 INTERLEAVED rounds of 5 launches each (guide 5.4 rule 24: both arms see the same clock and cache state; measured one after
        the other, the arm that ran first read 3-6 % low — tools/r06/chk_se.py)"""
        f(); g_()
        best = [1e9, 1e9]
        for _ in range(5):
            for i, fn in enumerate((f, g_)):
                e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
                e0.record()
                for _ in range(5):
                    fn()
                e1.record()
                torch.cuda.synchronize()
                best[i] = min(best[i], e0.elapsed_time(e1) / 5)
        return best
    out = {}
    for name, form, n, k in shapes:
        a = rnd(M, k)
        b = rnd(n, k) if form == "nt" else rnd(k, n)
        c = torch.empty(M, n, dtype=torch.bfloat16, device=dev)
        fl = 2.0 * M * n * k
        if form == "nt":
            ours, vend = (lambda: ops.gemm_nt(a, b, out=c)), (lambda: torch.matmul(a, b.t(), out=c))
        else:
            ours, vend = (lambda: ops.gemm(a, b, out=c, b_kmajor=True)), (lambda: torch.matmul(a, b, out=c))
        t_ours, t_vend = t_pair(ours, vend)
        out[name] = {"M": M, "N": n, "K": k, "molly": round(fl / t_ours / 1e9, 1), "torch_matmul": round(fl / t_vend / 1e9, 1)}
    return out


def copy_bandwidth(dev, nbytes: int = 1 << 30):
    """TB/s (read + written) of a 1 GiB device-to-device copy — what HBM-bound kernels see on THIS box (best of 5, HIP events)."""
    import torch
    a = torch.empty(nbytes // 4, dtype=torch.float32, device=dev).normal_()
    b = torch.empty_like(a)
    b.copy_(a)
    best = 1e9
    for _ in range(5):
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        b.copy_(a)
        e1.record()
        torch.cuda.synchronize()
        best = min(best, e0.elapsed_time(e1))
    return round(2.0 * nbytes / best / 1e9, 3)


def clock_under_load(dev, M):
    """Shader clock and socket power while the step's dominant kernel runs (`rocm-smi`, read-only, polled beside ~2.5 s of queued gate|up forward
    GEMMs, OUTSIDE the timed region).  The part runs every heavy kernel at its power cap, so the clock — not 2.4 GHz — sets the matrix pipes' peak
    (LOG.md round 5 item 15); `roofline.frac` keeps the guide's 2.5 PFLOP/s, this says what that peak is at the clock the kernel really gets."""
    import re
    import subprocess
    import torch
    from molly_amd import ops
    g = torch.Generator(device=dev).manual_seed(0)
    a = (torch.rand(M, 2048, device=dev, generator=g) * 2 - 1).bfloat16()
    b = (torch.rand(12288, 2048, device=dev, generator=g) * 2 - 1).bfloat16()
    c = torch.empty(M, 12288, dtype=torch.bfloat16, device=dev)
    for _ in range(5):
        ops.gemm_nt(a, b, out=c)
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    n = max(50, int(2.5e-3 * 1.3e15 / (2.0 * M * 12288 * 2048) * 1000))          # ~2.5 s of launches at 1.3 PFLOP/s
    e0.record()
    for _ in range(n):
        ops.gemm_nt(a, b, out=c)
    e1.record()
    clocks, watts = [], []
    for _ in range(3):
        r = subprocess.run(["rocm-smi", "--showclocks", "--showpower"], capture_output=True, text=True, timeout=20)
        clocks += [int(x) for x in re.findall(r"sclk clock level: \S+ \((\d+)Mhz\)", r.stdout)]
        watts += [float(x) for x in re.findall(r"Power \(W\): ([0-9.]+)", r.stdout)]
        if e1.query():
            break
    torch.cuda.synchronize()
    if not clocks:
        return None
    mhz = sorted(clocks)[len(clocks) // 2]
    return {"kernel": "gemm256_kernel, gate|up forward shape, sustained", "sclk_mhz": mhz, "socket_power_w": max(watts) if watts else None,
            "sustained_tflops": round(2.0 * M * 12288 * 2048 * n / (e0.elapsed_time(e1) * 1e-3) / 1e12, 1),
            "mfma_peak_tflops_at_sclk": round(MFMA_BF16_PEAK_TFLOPS * mhz / 2400.0, 1),
            "note": "rocm-smi beside ~2.5 s of queued launches; the 2.5 PFLOP/s of roofline.peak is the 2.4 GHz figure"}


def main(argv=None):
    argv = list(sys.argv[1:] if argv is None else argv)
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=8)
    ap.add_argument("--warmup", type=int, default=2)
    ap.add_argument("--batch", type=int, default=DEFAULT_BATCH,
                    help="samples per GPU per step.  BASELINE configs[1] fixes the model, the sequence and the span, not the per-GPU batch; "
                         "16 x 2048 tokens use 100 of the 288 GB and amortise the once-per-step costs (optimizer pass, launch ramps) over "
                         "twice the tokens of rounds 1-3's 8 (same-box sweep, profiles/r04_logs/batch_sweep.log: 8 / 12 / 16 / 24 -> 97.3 k / "
                         "101.0 k / 102.2 k / 104.3 k tokens/s).  The line carries a batch-8 measurement of the same process for continuity")
    ap.add_argument("--no-batch8-reference", action="store_true", help="skip the batch-8 continuity measurement of the default workload")
    ap.add_argument("--seq", type=int, default=2048)
    ap.add_argument("--k-protein", type=int, default=512)
    ap.add_argument("--model", default="1.7b")
    ap.add_argument("--train-mode", choices=("full", "lora", "mlp", "bio"), default="full",
                    help="full = headline (--train-llm --train-mlp); lora = --use-lora r=64; mlp = projectors only (side figures)")
    ap.add_argument("--zero-stage", type=int, default=2, choices=(0, 2),
                    help="2 = ZeRO-2 (reduce-scatter / sharded AdamW / all-gather); 0 = the reference's ds_z0 fallback (all-reduce)")
    ap.add_argument("--bucket-mib", type=float, default=0.0,
                    help="ZeRO bucket size in MiB of bf16 (one reduce-scatter / all-gather each; every xGMI link carries bucket/world "
                         "of it); 0 = the default of 32 MiB per link (256 MiB at 8 GPUs).  SURVEY §5 sweep: 64 ... 1024")
    ap.add_argument("--rs-algo", choices=("rccl", "a2a", "p2p"), default=None,
                    help="gradient reduce-scatter: the library's collective (default) or all_to_all_single + a local fp32 reduction "
                         "in rank order (SURVEY 5 option 2: every xGMI link carries one chunk at once, whatever RCCL would pick)")
    ap.add_argument("--bucket-ab-steps", type=int, default=0,
                    help="N > 1, neither --bucket-mib nor --rs-algo pinned: timed steps per (bucket size, reduce-scatter algorithm) candidate "
                         "of the sweep before the warm-up (untimed region; the fastest is kept, the table lands in comm.bucket_ab); 0 = off.  "
                         "OFF by default since round 6: the sweep rebuilds the optimizer per candidate, and on the one-GPU rehearsal four gloo ranks "
                         "hung in a rebuilt layout's first reduce-scatter (profiles/r06_logs/four_ranks_hang.log) — a first run on real links "
                         "should bring a number home on the default layout (32 MiB per link, the library's reduce-scatter); pass 2 to tune")
    ap.add_argument("--bucket-ab-mib", default="64,128,256,512,1024", help="bucket sizes (MiB of bf16) the sweep tries")
    ap.add_argument("--tune-budget-s", type=float, default=120.0,
                    help="N > 1: wall budget in seconds of the pre-warm-up tuning (bucket sweep, then the GEMM launch-shape A/B).  When a rank "
                         "finds it exceeded before the next candidate, every rank stops together, the best so far is kept and "
                         "comm.bucket_ab.truncated / not_run say so; 0 = no budget")
    ap.add_argument("--bucket-ab-algos", default="rccl,a2a",
                    help="reduce-scatter transports the sweep tries: rccl (the library's reduce-scatter), a2a (all_to_all + local fp32 sum), "
                         "p2p (direct reads / writes of mapped peer buffers, trainer/p2p.py: validated on one GPU only, so opt-in here)")
    ap.add_argument("--exposed-comm-steps", type=int, default=4,
                    help="N>1: extra steps after the timed region with the exchange NOT overlapped, to report the exposed "
                         "communication time (0 = skip)")
    ap.add_argument("--gemm-mode-ab-steps", type=int, default=2,
                    help="N>1: steps per GEMM launch shape (-3 | dyn | 0) measured before the warm-up; the fastest is kept (0 = skip)")
    ap.add_argument("--event-stride", type=int, default=7,
                    help="HIP events around every n-th GEMM launch of the timed region (1 = all: 2-3 %% slower steps)")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-vendor-gemm", action="store_true", help="skip the torch.matmul yardstick (roofline.vendor_gemm_tflops)")
    ap.add_argument("--cpu-budget", type=float, default=150.0,
                    help="wall-time box (s) of the down-scaled-C2 CPU run (2 warm-ups + 5 timed steps of ~12 s fit); C1 gets half")
    ap.add_argument("--micro", default=None,
                    help='micro-batches of one optimizer step, ";"-separated, each a ","-list of type:K omic spans per sample '
                         '(GA = their number).  Default "protein:<k-protein>".  C3: "dna:512,rna:512,protein:512;dna:512,rna:512,'
                         'protein:512"   C4: "protein:1024;dna:1000"')
    ap.add_argument("--no-secondary", action="store_true", help="skip the BASELINE configs 3 / 4 / 5 side measurements")
    ap.add_argument("--secondary-budget", type=float, default=420.0, help="wall-time box (s) of the whole secondary block")
    ap.add_argument("--secondary-worker", default=None, help="(internal) c3 | c4 | c5: run that side measurement, print its JSON")
    ap.add_argument("--cpu-baseline-worker", nargs="+", metavar="CONFIG THREADS BUDGET_S [WARMUPS TIMED]")
    ap.add_argument("--dry-run-launch", action="store_true", help="print the rank launch command instead of running it")
    args = ap.parse_args(argv)
    if args.cpu_baseline_worker:
        w = args.cpu_baseline_worker
        _cpu_step_worker(w[0], int(w[1]), float(w[2]), *(int(x) for x in w[3:5]))
        return 0
    if args.secondary_worker == "c5":
        return _c5_worker()

    # ---- N > 1 without a launcher: start the ranks ourselves (before any GPU call in this process) -----------------
    if args.gpus > 1 and "WORLD_SIZE" not in os.environ:
        return _self_launch(args, argv)

    # MOLLY_BENCH_WATCHDOG_S: after that many seconds every thread's Python stack goes to stderr and the rank exits — a hung collective in a first
    # multi-GPU run then leaves a trace instead of a silent timeout (default: 1500 s for the ranks of an N > 1 run, off at N = 1; 0 = off)
    world = int(os.environ.get("WORLD_SIZE", "1"))
    wd = os.environ.get("MOLLY_BENCH_WATCHDOG_S", "1500" if world > 1 else "")     # (ranks of a multi-GPU run: 25 minutes, under the process group's 30)
    if wd and float(wd) > 0:
        import faulthandler
        faulthandler.dump_traceback_later(float(wd), exit=True)
    rank = int(os.environ.get("RANK", "0"))
    local = int(os.environ.get("LOCAL_RANK", "0"))
    if world != args.gpus:
        raise SystemExit(f"bench.py: --gpus {args.gpus} but WORLD_SIZE={world} (launch N ranks, or none and let bench.py do it)")
    # test hooks (a one-GPU box can still run the multi-process path end to end): MOLLY_BENCH_DEVICE pins every rank to
    # one device, MOLLY_DIST_BACKEND=gloo replaces RCCL (which refuses two ranks on one GPU)
    if "MOLLY_BENCH_DEVICE" in os.environ:
        local = int(os.environ["MOLLY_BENCH_DEVICE"])
    # the library is built (normally: its digests verified, nothing compiled) BEFORE this process touches the GPU or joins a
    # process group — under a file lock, so that under an external launcher every rank may call it: no hipcc child is ever
    # spawned from a GPU-initialised rank while the others sit in a barrier
    import __graft_entry__ as ge
    ge.build()
    torch.cuda.set_device(local)
    dev = torch.device("cuda", local)
    import datetime
    import torch.distributed as dist
    backend = None
    def bring_up_failed(stage, e):
        """A communication bring-up that fails ends the run HERE, from a fresh state: one JSON line with the error (rank 0), exit code 3.
        Nothing is retried from this process — it has initialised the GPU, and re-exec'ing such a process takes the box down."""
        if rank == 0:
            print(json.dumps({"metric": f"training tokens/sec Molly-{args.model.upper()} bf16", "value": None, "unit": "tokens/s",
                              "n_gpus": world, "error": f"{stage}: {type(e).__name__}: {str(e)[:500]}",
                              "env": {k: os.environ[k] for k in ("MASTER_ADDR", "MASTER_PORT", "HSA_ENABLE_IPC_MODE_LEGACY", "NCCL_DEBUG")
                                      if k in os.environ}}), flush=True)
        sys.stderr.write(f"[bench rank {rank}] {stage} failed: {e!r}\n")
        sys.stderr.flush()
        os._exit(3)

    if world > 1:
        backend = os.environ.get("MOLLY_DIST_BACKEND", "nccl")
        # reference: src/train.py:606-610 (nccl, device_id, 30 min timeout)
        try:
            dist.init_process_group(backend, timeout=datetime.timedelta(minutes=30),
                                    **({"device_id": dev} if backend == "nccl" else {}))
        except Exception as e:              # noqa: BLE001
            bring_up_failed("init_process_group", e)

    def barrier():
        if world > 1:
            dist.barrier(**({"device_ids": [local]} if backend == "nccl" else {}))

    import molly_amd
    from molly_amd import config as C, ops
    from molly_amd.synth import synth_batch
    from molly_amd.trainer import Zero2Optimizer
    from molly_amd.trainer.zero2 import preflight_collectives

    comm_check = None
    if world > 1:                            # tiny in-place RS/AG/AR with known answers: fail fast
        try:
            comm_check = preflight_collectives(dev)
        except Exception as e:              # noqa: BLE001
            bring_up_failed("preflight_collectives", e)

    micro = parse_micro(args.micro or f"protein:{args.k_protein}")
    GA = len(micro)
    cfg = C.molly(args.model, k_tokens=max(k for spans in micro for _, k in spans))
    torch.manual_seed(1234)                  # before the shells: whatever they materialise at construction comes from this generator
    m = molly_amd.OmicsOne(cfg)
    m.model = molly_amd.Qwen3ForCausalLM(cfg.text_config)
    m.dna_rna_model = molly_amd.EsmForMaskedLM(cfg.dna_rna_config)
    m.protein_model = molly_amd.EsmForMaskedLM(cfg.protein_config)
    if args.train_mode in ("full", "bio"):
        m.prepare(dev, random_init_seed=1234, train_bio=args.train_mode == "bio")   # same seed on every rank: replicas start identical
    else:
        from molly_amd.lora import LoraConfig
        m.prepare(dev, random_init_seed=1234, train_llm=False, train_mlp=True,
                  lora=LoraConfig(r=64, lora_alpha=64, lora_dropout=0.05, seed=42) if args.train_mode == "lora" else None)
    rt = m._rt
    def make_opt(bucket_mib, rs_algo):
        kw = {}
        if bucket_mib and bucket_mib > 0:
            kw["chunk_elems"] = max(8, int(bucket_mib * (1 << 20) / 2 / world) // 8 * 8)
        o = Zero2Optimizer(rt.P.flat, rt.G.flat, m.n_decay, lr=3e-5, weight_decay=1e-2, max_grad_norm=1.0,
                           stage=args.zero_stage, rs_algo=rs_algo, **kw)
        m.attach_optimizer(o)
        return o
    opt = make_opt(args.bucket_mib, args.rs_algo)

    B, T, K = args.batch, args.seq, args.k_protein
    # 4 different steps' worth of data; a step = GA micro-batches (gradients summed over the window, one optimizer step)
    batches = [[synth_batch(B, T, spans, seed=42 + rank + 1000 * i + 100 * j) for j, spans in enumerate(micro)] for i in range(4)]

    def step(i):
        for j, b in enumerate(batches[i % len(batches)]):
            loss = m.forward_backward(b["input_ids"], b["attention_mask"], b["omic_ids"], b["omic_info_list"], b["labels"],
                                      accumulate=j > 0, final_micro=j == GA - 1)
        opt.step(lr=3e-5)
        return loss

    def timed_steps(n, first):
        """n steps bracketed by barrier + synchronize on both sides -> (wall seconds, [per-step ms], last loss)."""
        torch.cuda.synchronize()
        barrier()
        torch.cuda.synchronize()
        evs = []
        t0 = time.perf_counter()
        for i in range(n):
            e0 = torch.cuda.Event(enable_timing=True); e0.record()
            loss = step(first + i)
            e1 = torch.cuda.Event(enable_timing=True); e1.record()
            evs.append((e0, e1))
        torch.cuda.synchronize()
        barrier()
        torch.cuda.synchronize()
        dt = time.perf_counter() - t0
        return dt, [a.elapsed_time(b) for a, b in evs], loss

    # ---- N > 1: which launch shape of the 256x256 GEMM runs fastest beside this job's own collectives?  Measured on the ranks
    # themselves before the warm-up (untimed region): one settling step + `--gemm-mode-ab-steps` timed steps per shape, max over
    # ranks, the fastest kept for the timed region.  Every shape computes bit-identical results (tests/test_gpu_dynamic_fetch.py),
    # so the choice is speed only.  MOLLY_GEMM_PERSISTENT_MULTI pins one shape and skips this.
    # ---- N > 1: bucket size x reduce-scatter algorithm, measured on the job's own ranks before the warm-up (untimed region).  The
    # reference's ds_z2_config.json:18-27 fixes both bucket sizes at 5e8 elements; xGMI is 7 point-to-point links per GPU, so what a
    # bucket costs depends on how the library's algorithm spreads it over them — tuned HERE rather than guessed: every candidate
    # rebuilds the optimizer with its layout (the fp32 masters restart from the current bf16 parameters — synthetic run, untimed
    # region), runs one settling step + `--bucket-ab-steps` timed steps, the slowest rank's time counts, the fastest layout is kept.
    bucket_ab = None
    if (world > 1 and args.zero_stage == 2 and args.bucket_ab_steps > 0 and args.bucket_mib <= 0 and args.rs_algo is None
            and "MOLLY_RS_ALGO" not in os.environ):
        from molly_amd.trainer.zero2 import sweep_exchange
        total_mib = rt.P.flat.numel() * 2 / (1 << 20)
        sizes = sorted({min(float(x), total_mib) for x in args.bucket_ab_mib.split(",") if x.strip()})

        from molly_amd.trainer.zero2 import dist_agree

        def build(mib, algo):
            # the part that can fail on one rank only (memory for the layout's receive buffers, an IPC mapping): followed by a collective
            # verdict inside sweep_exchange, so the ranks skip a candidate TOGETHER
            nonlocal opt
            if os.environ.get("MOLLY_BENCH_VERBOSE"):
                print(f"[rank {rank}] sweep: building {mib:g} MiB / {algo}", file=sys.stderr, flush=True)
            # the previous layout's last all-gathers may still be in flight on its communication stream (nobody has run the forward that waits for
            # them): drain them on every rank, and meet, BEFORE the optimizer that owns them goes away — a new layout's first collectives
            # issued beside the old one's last ones hung four gloo ranks in reduce_scatter (profiles/r06_logs/four_ranks_hang.log)
            if opt is not None:
                opt.wait_all_params()
            torch.cuda.synchronize()
            dist.barrier()
            opt = None
            m._rt.opt = None
            torch.cuda.empty_cache()
            opt = make_opt(mib, algo)

        def measure(mib, algo):
            if os.environ.get("MOLLY_BENCH_VERBOSE"):
                print(f"[rank {rank}] sweep: measuring {mib:g} MiB / {algo}", file=sys.stderr, flush=True)
            step(0)
            dtm, _, _ = timed_steps(args.bucket_ab_steps, 1)
            tm = torch.tensor([dtm], device=dev, dtype=torch.float64)
            dist.all_reduce(tm, op=dist.ReduceOp.MAX)
            return float(tm.item()) / args.bucket_ab_steps * 1e3

        def bcast(obj):
            box = [obj]
            dist.broadcast_object_list(box, src=0)
            return box[0]
        algos = [a.strip() for a in args.bucket_ab_algos.split(",") if a.strip()]
        bucket_ab = sweep_exchange([(mib, algo) for algo in algos for mib in sizes], measure, args.bucket_ab_steps, build=build,
                                   agree=dist_agree(dev), budget_s=args.tune_budget_s if args.tune_budget_s > 0 else None, broadcast=bcast)
        ch = bucket_ab["chosen"]
        opt.wait_all_params()
        torch.cuda.synchronize()
        dist.barrier()
        opt = None
        m._rt.opt = None
        torch.cuda.empty_cache()
        opt = make_opt(ch["bucket_mib"], ch["rs_algo"])

    gemm_mode_ab = None
    gemm_mode_truncated = False
    if world > 1 and opt.overlap and args.gemm_mode_ab_steps > 0 and "MOLLY_GEMM_PERSISTENT_MULTI" not in os.environ:
        gemm_mode_ab = {}
        _t_ab = time.monotonic()
        _agree = None
        for mi, mode in enumerate((-3, "dyn", 0)):
            if mi > 0 and args.tune_budget_s > 0:
                # the launch-shape A/B shares the pre-warm-up wall budget with the bucket sweep (what is left of it, at least a third): one rank
                # over its budget stops every rank together, the best shape so far is kept
                from molly_amd.trainer.zero2 import dist_agree
                _agree = _agree or dist_agree(dev)
                if not _agree(time.monotonic() - _t_ab <= max(args.tune_budget_s / 3.0, 1.0), None)[0]:
                    gemm_mode_truncated = True
                    break
            opt.set_gemm_blocks_mode(mode, m)
            step(0)
            dtm, _, _ = timed_steps(args.gemm_mode_ab_steps, 1)
            tm = torch.tensor([dtm], device=dev, dtype=torch.float64)
            dist.all_reduce(tm, op=dist.ReduceOp.MAX)
            gemm_mode_ab[str(mode)] = round(float(tm.item()) / args.gemm_mode_ab_steps * 1e3, 2)
        best = min(gemm_mode_ab, key=gemm_mode_ab.get)              # same dict on every rank (all-reduced times)
        opt.set_gemm_blocks_mode("dyn" if best == "dyn" else int(best), m)
        gemm_mode_ab = {"ms_per_step": gemm_mode_ab, "chosen": best, "steps_each": args.gemm_mode_ab_steps}
        if gemm_mode_truncated:
            gemm_mode_ab["truncated"] = True

    for i in range(args.warmup):
        step(i)
    if world > 1:
        opt.comm_events = []                    # per-bucket events of the timed region's collectives (on the stream they ran on)
    ops.GEMM_PROFILE = []
    ops.GEMM_PROFILE_STRIDE = args.event_stride
    dt, step_ms, loss = timed_steps(args.steps, args.warmup)
    prof, ops.GEMM_PROFILE = ops.GEMM_PROFILE, None
    # rounds 1-3 quoted this workload at 8 samples per GPU: the same model in the same process at that batch, 1 settling + 4 timed
    # steps, OUTSIDE the timed region (continuity only; `value` is the default batch's)
    batch8 = None
    if (world == 1 and B != 8 and not args.no_batch8_reference and args.micro is None and args.train_mode == "full"
            and (T, K, args.model) == (2048, 512, "1.7b")):
        saved_batches = batches
        batches = [[synth_batch(8, T, spans, seed=4242 + 1000 * i + 100 * j) for j, spans in enumerate(micro)] for i in range(2)]
        step(0); step(1)                        # both synthetic batches once: activation buffers and the stager's device images at this shape
        dt8, ms8, _ = timed_steps(5, 2)
        p50 = statistics.median(ms8)
        batch8 = {"samples_per_gpu": 8, "step_ms_p50": round(p50, 2), "tokens_per_s_at_p50": round(8 * T * GA / p50 * 1e3, 1),
                  "step_ms": [round(x, 1) for x in ms8]}
        batches = saved_batches
    comm_timings = opt.comm_timings() if world > 1 else None
    opt.comm_events = None
    if world > 1:
        tmax = torch.tensor([dt], device=dev, dtype=torch.float64)
        dist.all_reduce(tmax, op=dist.ReduceOp.MAX)
        dt = float(tmax.item())
    loss_v = float(loss.item())

    # ---- N > 1: how much of the exchange is exposed?  same steps with the collectives on the compute stream ---------
    comm = None
    if world > 1:
        comm = {"backend": backend, "world_size": dist.get_world_size(), "preflight": comm_check,
                "comm_bytes_per_step_per_gpu": opt.comm_bytes_per_step(), "bucket_mib": round(opt.bucket * 2 / (1 << 20), 1),
                "per_link_mib_per_bucket": round(opt.chunk * 2 / (1 << 20), 1), "buckets": len(opt.buckets),
                "overlap": bool(opt.overlap), "rs_algo": opt.rs_algo, "rs_algo_fallback": getattr(opt, "rs_algo_fallback", None),
                # where the gradient sum is rounded: the library's reduce-scatter adds bf16 partial sums hop by hop (DeepSpeed's
                # own behaviour with bf16 gradients); the all-to-all variant sums the `world` copies in fp32 on the owner, once
                "reduce_dtype": ("fp32 on the owning rank, rank order, one rounding (all_to_all + molly_reduce_rows)" if opt.rs_algo == "a2a" else
                                 "fp32 on the owning rank, rank order, one rounding (peer copies read in place: molly_p2p_reduce)" if opt.rs_algo == "p2p"
                                 else "bf16 in the collective (RCCL reduce_scatter: one rounding per hop)"),
                "gemm_blocks_mode": getattr(opt, "gemm_blocks_mode", 256), "gemm_mode_ab": gemm_mode_ab, "bucket_ab": bucket_ab,
                # per bucket, HIP events on the communication stream around each collective of the timed region (overlapped: the
                # time includes waiting for CUs beside the backward); us_per_bucket = the first timed step's buckets in launch order
                "timings_us": comm_timings}
        # the world as the BACKEND counts it: a sum of ones through its own all-reduce (a communicator that silently spans fewer
        # ranks than WORLD_SIZE shows here)
        ones = torch.ones(1, device=dev)
        dist.all_reduce(ones)
        comm["world_size_by_all_reduce"] = int(ones.item())
        if backend == "nccl":
            try:
                comm["rccl_version"] = ".".join(str(x) for x in torch.cuda.nccl.version())
            except Exception:
                pass
            for k in ("NCCL_ALGO", "NCCL_PROTO", "NCCL_MIN_NCHANNELS", "NCCL_MAX_NCHANNELS", "RCCL_MSCCL_ENABLE", "HSA_ENABLE_IPC_MODE_LEGACY"):
                if k in os.environ:
                    comm.setdefault("env", {})[k] = os.environ[k]
        if args.exposed_comm_steps > 0 and opt.overlap:
            opt.set_overlap(False)
            step(0)
            dt2, ms2, _ = timed_steps(args.exposed_comm_steps, 1)
            opt.set_overlap(True)
            t2 = torch.tensor([dt2], device=dev, dtype=torch.float64)
            dist.all_reduce(t2, op=dist.ReduceOp.MAX)
            comm["step_ms_p50_no_overlap"] = round(statistics.median(ms2), 2)
            comm["exposed_comm_ms_removed_by_overlap"] = round(statistics.median(ms2) - statistics.median(step_ms), 2)
            # what the overlap does NOT hide, for the layout the timed region ran: the same steps with every collective a no-op
            # (measurement only, last thing this process does with the optimizer: the replicas diverge from here on)
            from molly_amd.trainer.zero2 import _NullComm
            real_comm, opt.comm = opt.comm, _NullComm()
            step(0)
            dt3, ms3, _ = timed_steps(args.exposed_comm_steps, 1)
            opt.comm = real_comm
            t3 = torch.tensor([statistics.median(ms3)], device=dev, dtype=torch.float64)
            dist.all_reduce(t3, op=dist.ReduceOp.MAX)
            comm["step_ms_p50_no_exchange"] = round(float(t3.item()), 2)
            comm["exposed_comm_ms"] = round(statistics.median(step_ms) - float(t3.item()), 2)

    if rank == 0:
        tokens = world * B * T * GA * args.steps
        # dominant kernel = gemm256_kernel (the 256x256 ping-pong MFMA GEMM); its three operand-layout instantiations
        # are separate rows in rocprofv3 --stats, its stream-K form (encoder projections and other grids that do not fill whole
        # rounds of the chip) another three.  Launches that went to the 128x128 kernel (a side narrower than a tile) are listed apart.
        # every GEMM launch of the timed region is listed; HIP events bracket every `stride`-th one (7: coprime with the 4- and
        # 9-GEMM patterns of the decoder / encoder layers, so every shape is sampled).  Per class, the launches that were not
        # timed are priced at the class's sampled rate.
        n_launch = len(prof)
        timed = [(a.elapsed_time(b), f, kcfg, lay) for a, b, f, kcfg, lay in prof if a is not None]
        achieved = sum(f for _, f, *_ in timed) / (sum(t for t, *_ in timed) * 1e-3) / 1e12
        if os.environ.get("MOLLY_GEMM_TABLE"):
            # (diagnostic: every distinct (2MNK, launch config, layout) of the timed launches — where a step's GEMM time sits, shape by shape)
            tab = {}
            for t_ms, f, kcfg, lay in timed:
                d = tab.setdefault((f, kcfg, lay), [0, 0.0])
                d[0] += 1; d[1] += t_ms
            tot = sum(v[1] for v in tab.values())
            with open(os.environ["MOLLY_GEMM_TABLE"], "w") as fh:
                fh.write(f"# {sum(v[0] for v in tab.values())} timed GEMM launches, {tot:.2f} ms in all; 2MNK GFLOP | cfg | layout (A k-major, B k-major) | launches | avg us | TFLOP/s | share\n")
                for (f, kcfg, lay), v in sorted(tab.items(), key=lambda kv: -kv[1][1]):
                    fh.write(f"{f / 1e9:10.2f} {kcfg:7d} {str(lay):15s} {v[0]:5d} {v[1] / v[0] * 1e3:9.1f} {f * v[0] / (v[1] * 1e-3) / 1e12:8.1f} {v[1] / tot:6.3f}\n")
        by_kernel = {}
        names = {(False, False): "NT gemm256_kernel<false,false>", (False, True): "NN gemm256_kernel<false,true>",
                 (True, True): "TN gemm256_kernel<true,true>"}
        def klass(kcfg, lay):
            # cfg = 128 | 512 | 513 (drawing its tiles) | 514 (streaming epilogue) (+ 1000 * split-K factor, + 50000 stream-K, + 100000 * problems of a grouped launch)
            key = ("NT gemm256_kernel<false,false,...,SE> (streaming epilogue: plain launches of whole tiles)" if kcfg % 1000 == 514 else
                   names[lay] if kcfg % 1000 in (512, 513) else
                   "gemm_rows_kernel (64-row tiles: small grids at <= 1,024 rows)" if kcfg % 1000 == 32 else
                   "gemm_skinny_kernel (decode rows)" if kcfg % 1000 == 16 else "gemm_kernel<...,128,2,64> (a side narrower than a tile)")
            if kcfg % 1000 in (16, 32):
                return key
            if kcfg >= 100000:
                key += f" grouped x{kcfg // 100000} (a layer's weight gradients in one launch)"
            elif kcfg // 1000 >= 50:
                key += " stream-K (grids that do not fill whole rounds: encoder projections, B=1 decoder shapes)"
            elif kcfg // 1000 > 1:
                key += " + splitk_reduce"
            return key
        for a, b, f, kcfg, lay in prof:
            d = by_kernel.setdefault(klass(kcfg, lay), [0, 0, 0.0, 0.0, 0.0])     # launches, timed, ms timed, flops timed, flops all
            d[0] += 1; d[4] += f
            if a is not None:
                d[1] += 1; d[2] += a.elapsed_time(b); d[3] += f
        gemm_ms = sum(v[2] * v[4] / v[3] for v in by_kernel.values() if v[3] > 0)  # all launches, at each class's sampled rate
        gemm_flops_step = sum(v[4] for v in by_kernel.values()) / args.steps       # EXECUTED GEMM flops per step (exact 2MNK)
        by_kernel = {k: {"launches": v[0], "timed": v[1], "avg_launch_us": round(v[2] * 1e3 / max(v[1], 1), 2),
                         "achieved_tflops": round(v[3] / (v[2] * 1e-3) / 1e12, 1) if v[2] > 0 else None}
                     for k, v in by_kernel.items()}
        # HBM-side traffic of the dominant kernel cannot be counted live (PMC needs rocprofv3): quote the committed PMC
        # passes of this same workload (tools/pmc_hbm_traffic.py; corrected as MI355X_MICROARCH.md §HBM prescribes)
        traffic, traffic_src = None, None
        prof_dir = os.path.join(os.path.dirname(os.path.abspath(__file__)), "profiles")
        tj = next((os.path.join(prof_dir, f) for f in ("r06_hbm_traffic.json", "r05_hbm_traffic.json", "r04_hbm_traffic.json", "r03_hbm_traffic.json", "r02b_hbm_traffic.json", "r02_hbm_traffic.json", "r01b_hbm_traffic.json")
                   if os.path.exists(os.path.join(prof_dir, f))), None)
        if args.train_mode == "full" and (B, T, K, args.model) == (16, 2048, 512, "1.7b") and args.micro is None and tj and any(r in tj for r in ("r06", "r05", "r04")):
            with open(tj) as f:
                tr = json.load(f)
            traffic, traffic_src = tr["gemm_hbm_bytes_per_launch"], tr["source"]
        enc_cfgs = {"dna_rna": cfg.dna_rna_config, "protein": cfg.protein_config}
        flops_step = 0.0
        for spans in micro:
            flops_step += B * T * algorithmic_flops_per_token(cfg.text_config, T)
            for typ, k in spans:
                ec = enc_cfgs["protein" if typ == "protein" else "dna_rna"]
                flops_step += B * k * (enc_flops_per_token(ec, k) + 3 * 2 * ec.hidden_size * cfg.text_config.hidden_size)
        # executed = what the kernels really ran: every GEMM launch's 2MNK (lm_head on the scored rows only) + the attention
        # matmuls (causal half, no recompute credit)
        exec_step = gemm_flops_step + attention_flops_per_step(cfg.text_config, enc_cfgs, B, T, micro, args.train_mode == "bio")
        sps = dt / args.steps
        out = {
            "metric": f"training tokens/sec Molly-{args.model.upper()} bf16", "value": round(tokens / dt, 1), "unit": "tokens/s",
            "n_gpus": world, "steps": args.steps, "warmup": args.warmup, "ms_per_step": round(sps * 1e3, 2),
            "higher_is_better": True, "scaling": "weak", "vs_baseline": None, "dtype": "bf16", "data": "synthetic",
            "config": {"workload": f"Molly-{args.model.upper()} (Qwen3-{args.model.upper()} + NT-500M + ESM2-650M) train step, "
                                   f"seq_len {T} text + " + ("; ".join(" + ".join(f"{k}-token {t} span" for t, k in spans) for spans in micro)
                                                             if args.micro else f"{K}-residue protein span") +
                                   f" per sample, {B} samples/GPU, GA={GA}, "
                                   + {"full": "LLM+projectors trainable", "bio": "LLM+projectors+encoders trainable (--train-bio)", "lora": "LoRA r=64 adapters+projectors trainable, base frozen",
                                      "mlp": "projectors trainable, LLM frozen"}[args.train_mode] +
                                   ("" if args.train_mode == "bio" else ", encoders frozen") + f", ZeRO-{args.zero_stage} dp{world}",
                       "global_batch": world * B, "seq_len": T, "parallelism": f"dp{world}",
                       "scored_token_fraction": 0.25, "batch8_reference": batch8,
                       "note": "prompt = 75% of each sample with labels -100 (SURVEY 8d); lm_head+CE run on the scored rows only "
                               "(identical loss/gradients).  executed_* count the FLOPs the kernels ran; model_* credit the full "
                               "algorithmic count of SURVEY 8d (lm_head on every row)"},
            # rounds 1-3 quoted this workload at 8 samples per GPU: the like-for-like figure of THIS process at that batch (also kept in
            # config.batch8_reference), first-class so that round-over-round tracking does not depend on the default batch
            "batch8_reference": batch8,
            "step_ms_p50": round(statistics.median(step_ms), 2),
            "step_ms": [round(x, 1) for x in step_ms],      # per step, HIP events (ms_per_step is the wall clock over all of them / steps)
            # THE step fraction: FLOPs the kernels executed / step time / dense bf16 MFMA peak
            "executed_tflops_per_gpu": round(exec_step / sps / 1e12, 1),
            "mfma_roofline_frac_step_executed": round(exec_step / sps / 1e12 / MFMA_BF16_PEAK_TFLOPS, 4),
            # credit for work the kernels SKIP (lm_head + CE on the 75 % of rows whose label is -100: exact, zero loss and gradient):
            # SURVEY 8d's algorithmic count / step time — not a utilisation figure
            "model_tflops_per_gpu": round(flops_step / sps / 1e12, 1),
            "model_flops_credit_frac_incl_skipped_rows": round(flops_step / sps / 1e12 / MFMA_BF16_PEAK_TFLOPS, 4),
            "loss": round(loss_v, 4),
            "roofline": {"bound": "mfma", "kernel": "bf16 MFMA GEMM (gemm256_kernel / gemm_kernel, all launches of the timed region)",
                         "achieved": round(achieved, 1), "peak": MFMA_BF16_PEAK_TFLOPS, "unit": "TFLOP/s",
                         "frac": round(achieved / MFMA_BF16_PEAK_TFLOPS, 4), "traffic": traffic, "traffic_unit": "bytes/launch (L2 fabric-side: "
                         "Infinity Cache + HBM)", "traffic_source": traffic_src, "launches": n_launch,
                         "avg_launch_us": round(gemm_ms * 1e3 / n_launch, 2),
                         "gemm_share_of_step": round(gemm_ms / sum(step_ms), 3), "by_kernel": by_kernel},
        }
        if comm is not None:
            out["comm"] = comm
        if world == 1 and not args.no_vendor_gemm:
            try:
                out["roofline"]["vendor_gemm_tflops"] = vendor_gemm_yardstick(dev, B * T)
            except Exception as e:          # noqa: BLE001  (a yardstick must never cost the line)
                out["roofline"]["vendor_gemm_tflops"] = {"error": f"{type(e).__name__}: {str(e)[:160]}"}
            try:
                out["roofline"]["clock_under_load"] = clock_under_load(dev, B * T)
            except Exception as e:  # noqa: BLE001  (a yardstick: rocm-smi absent or unreadable must not cost the headline line)
                out["roofline"]["clock_under_load"] = {"error": f"{type(e).__name__}: {str(e)[:160]}"}
        # ---- what kind of box was this?  The devices of the pool differ by +-2.5 % in every MFMA-bound number (a power-capped part: the clock
        # gives), more than a round's step changes.  Three box properties measured OUTSIDE the timed region and the headline divided by the
        # vendor library's throughput on the step's largest forward shape make lines from different boxes comparable.
        if world == 1 and not args.no_vendor_gemm:
            cal = {}
            vg = out["roofline"].get("vendor_gemm_tflops") or {}
            if "gate|up fwd" in vg:
                cal["torch_matmul_tflops"] = vg["gate|up fwd"]["torch_matmul"]
                cal["torch_matmul_shape"] = "gate|up forward: M = %d, N = 12288, K = 2048 (hipBLASLt through torch.matmul, random data)" % (B * T)
            cu = out["roofline"].get("clock_under_load") or {}
            cal["sclk_mhz"] = cu.get("sclk_mhz")
            try:
                cal["copy_TBps"] = copy_bandwidth(dev)
            except Exception as e:  # noqa: BLE001
                cal["copy_TBps"] = None
                cal["copy_error"] = f"{type(e).__name__}: {str(e)[:120]}"
            out["box_calibration"] = cal
            if cal.get("torch_matmul_tflops"):
                # tokens/s per TFLOP/s the vendor's GEMM reaches on this box: moves with the code, not with the device
                out["value_per_vendor_tflop"] = round(out["value"] / cal["torch_matmul_tflops"], 3)
        if world == 1 and not args.no_secondary and args.secondary_worker is None:
            # the other BASELINE configs, each in a child of its own: release this process's HBM first
            del m, opt, rt, batches
            import gc
            gc.collect()
            torch.cuda.empty_cache()
            out["secondary"] = secondary_block(args.secondary_budget)
        if not args.no_cpu_baseline and world == 1:                  # rank 0 at N=1 only (contract)
            out["cpu_baseline"] = cpu_baseline(args.cpu_budget)
        print(json.dumps(out), flush=True)
    if world > 1:
        barrier()
        dist.destroy_process_group()
    return 0


if __name__ == "__main__":
    sys.exit(main())
