"""Training loop of the hot path — the step semantics the reference inherits from HF Trainer 4.53 + DeepSpeed
(spec = the pasted loop, reference src/trainer/domain_loss.py:565-770, 881-1024, 844-873):

  * gradient accumulation windows of `gradient_accumulation_steps` micro-batches; the micro-batch loss is the token mean
    of THAT micro-batch and is NOT divided by the accumulation count (the model's forward has **kwargs, :1011-1013;
    SURVEY.md §0.4-4) -> gradients are SUMMED over the window;
  * per window: reduce-scatter -> global-norm clip (max_grad_norm) -> AdamW -> LR scheduler step -> zero grads (:676-724);
  * linear warmup (warmup_ratio) then linear decay (HF `linear` schedule); lr of optimizer step k uses k scheduler steps;
  * logged loss = sum of the window's micro losses, averaged over the optimizer steps since the last log and over ranks
    (:856-861) — i.e. GA x larger than a per-micro-batch mean, exactly like upstream;
  * non-finite micro losses are replaced by the running average in the LOG only (:655-661);
  * checkpoints: `checkpoint-N/pytorch_model.bin` = torch pickle of the full state dict incl. frozen encoders
    (src/trainer/omics_trainer.py:105), rotated by save_total_limit.
Batches are dealt to ranks round-robin (rank r takes micro-batches r, r+DP, ... of one seeded permutation), the
partitioning accelerate's BatchSamplerShard applies in the reference.
"""
from __future__ import annotations

import json
import math
import os
import shutil
import time
from dataclasses import dataclass, field
from typing import Callable, List, Optional

import torch
import torch.distributed as dist

from .zero2 import Zero2Optimizer, linear_warmup_lr


@dataclass
class TrainArgs:
    output_dir: str = "out"
    per_device_train_batch_size: int = 1
    gradient_accumulation_steps: int = 1
    num_train_epochs: float = 1.0
    max_steps: int = -1
    learning_rate: float = 3e-5
    weight_decay: float = 1e-2
    warmup_ratio: float = 0.1
    max_grad_norm: float = 1.0
    adam_beta1: float = 0.9
    adam_beta2: float = 0.999
    adam_epsilon: float = 1e-8
    logging_steps: int = 20
    save_steps: int = 0
    save_total_limit: Optional[int] = None
    seed: int = 42
    # evaluation (HF `eval_strategy="steps"`; reference example: --eval_steps 25000 --eval_strategy steps)
    per_device_eval_batch_size: int = 1
    eval_steps: int = 0                       # 0 = never
    early_stopping_patience: int = 0          # evaluations without improvement of eval_loss before stopping (0 = off);
    #                                           the reference registers EarlyStoppingCallback(patience), which HF only
    #                                           honours together with load_best_model_at_end (its scripts leave that off)
    load_best_model_at_end: bool = False
    zero_stage: int = 2                       # 2 = ZeRO-2 (ds_z2_config.json), 0 = plain DP all-reduce (ds_z0_config.json)


def save_model(model, output_dir: str):
    """reference: OmicsTrainer.save_model (src/trainer/omics_trainer.py:85-105): under LoRA the PEFT adapter + the two
    projector .bin files; otherwise `pytorch_model.bin` with the reference's keys."""
    os.makedirs(output_dir, exist_ok=True)
    rt = model._runtime()
    if getattr(rt, "opt", None) is not None:
        rt.opt.wait_all_params()              # an overlapped all-gather of the last step may still be publishing parameters
    torch.cuda.synchronize()
    if rt.llm.lora is not None:
        from ..lora import save_adapter
        save_adapter(model, output_dir)
        return
    sd = {k: v.detach().to("cpu").clone() for k, v in model.state_dict().items()}
    torch.save(sd, os.path.join(output_dir, "pytorch_model.bin"))
    # the projectors on their own as well (what the LoRA branch of the reference writes and inference_lora.py reads)
    for name in ("dna_rna_projector", "protein_projector"):
        torch.save({k[len(name) + 1:]: v for k, v in sd.items() if k.startswith(name + ".")},
                   os.path.join(output_dir, f"{name}.bin"))


class Trainer:
    def __init__(self, model, train_dataset, collate_fn: Callable, args: TrainArgs, log_fn: Callable = print,
                 eval_dataset=None, optimizer=None):
        self.model, self.ds, self.collate, self.args, self.log = model, train_dataset, collate_fn, args, log_fn
        self.eval_ds = eval_dataset
        self.best_metric, self.best_step, self._bad_evals = None, None, 0
        self.world = dist.get_world_size() if dist.is_initialized() else 1
        self.rank = dist.get_rank() if dist.is_initialized() else 0
        rt = model._runtime()
        if optimizer is None and self.world > 1:
            # first contact with the communication backend BEFORE the optimizer picks its collective forms: the three
            # collectives of the step on small buffers with known answers; a wrong in-place result selects the staged forms
            # (bench.py and smoke_dist did this; a plain `python -m molly_amd.train` run never had)
            from .zero2 import preflight_collectives
            self.comm_preflight = preflight_collectives(rt.dev)
        self.opt = optimizer if optimizer is not None else Zero2Optimizer(
            rt.P.flat, rt.G.flat, model.n_decay, lr=args.learning_rate, betas=(args.adam_beta1, args.adam_beta2),
            eps=args.adam_epsilon, weight_decay=args.weight_decay, max_grad_norm=args.max_grad_norm, stage=args.zero_stage)
        model.attach_optimizer(self.opt)
        self.history: List[dict] = []

    def _micro_batches(self, epoch: int):
        a = self.args
        g = torch.Generator().manual_seed(a.seed + epoch)
        perm = torch.randperm(len(self.ds), generator=g).tolist()
        B = a.per_device_train_batch_size
        n_micro = len(perm) // B                                   # drop_last=False upstream; the tail batch is dealt too
        if len(perm) % B:
            n_micro += 1
        for mb in range(self.rank, n_micro - (n_micro % self.world if self.world > 1 else 0), self.world):
            idx = perm[mb * B:(mb + 1) * B]
            yield self.collate([self.ds[i] for i in idx])

    @torch.no_grad()
    def evaluate(self) -> float:
        """HF `Trainer.evaluate` semantics for eval_loss (HF:trainer.py evaluation_loop): every batch's loss (its own token
        mean) counts once per SAMPLE of the batch, the mean runs over all samples of all ranks.  Batches are dealt to ranks
        round-robin, tail batch included."""
        a, m = self.args, self.model
        B = a.per_device_eval_batch_size
        n = len(self.eval_ds)
        acc = torch.zeros(2, dtype=torch.float64, device=m._rt.dev)              # [sum loss * samples, samples]
        for mb, i0 in enumerate(range(0, n, B)):
            if mb % self.world != self.rank:
                continue
            batch = self.collate([self.eval_ds[i] for i in range(i0, min(n, i0 + B))])
            out = m(input_ids=batch["input_ids"], attention_mask=batch["attention_mask"], omic_ids=batch["omic_ids"],
                    omic_info_list=batch["omic_info_list"], labels=batch["labels"])
            k = batch["input_ids"].shape[0]
            acc[0] += out.loss.double() * k
            acc[1] += k
        if self.world > 1:
            dist.all_reduce(acc)
        return float((acc[0] / acc[1]).item())

    def _maybe_evaluate(self, step: int) -> bool:
        """-> True when early stopping fires.  Tracks the best eval_loss (lower is better: the reference's
        --metric-for-best-model default, src/train.py:543-548)."""
        a = self.args
        if self.eval_ds is None or not a.eval_steps or step % a.eval_steps:
            return False
        v = self.evaluate()
        rec = {"step": step, "eval_loss": round(v, 4)}
        self.history.append(rec)
        if self.rank == 0:
            self.log(json.dumps(rec))
        if self.best_metric is None or v < self.best_metric:
            self.best_metric, self.best_step, self._bad_evals = v, step, 0
            if a.load_best_model_at_end and self.rank == 0:
                save_model(self.model, os.path.join(a.output_dir, "best"))
        else:
            self._bad_evals += 1
        return bool(a.early_stopping_patience) and a.load_best_model_at_end and self._bad_evals >= a.early_stopping_patience

    def _micro_per_epoch(self) -> int:
        n_micro = math.ceil(len(self.ds) / self.args.per_device_train_batch_size)
        if self.world > 1:
            n_micro -= n_micro % self.world
        return n_micro // self.world

    def _mirror_skipped(self):
        """After optimizer step s: copy the device-side skipped-step counter into pinned slot s % 2 and record an event behind
        the copy.  (Two slots: the copy of step s never overwrites the value step s's reader may still be looking at.)"""
        scal = getattr(self.opt, "scal", None)
        if scal is None or not scal.is_cuda:
            return
        if getattr(self, "_skip_host", None) is None:
            self._skip_host = torch.zeros(2, dtype=torch.float32, pin_memory=True)
            self._skip_ev = [None, None]
            self._skip_n = 0
        k = self._skip_n % 2
        self._skip_host[k:k + 1].copy_(scal[3:4], non_blocking=True)
        ev = torch.cuda.Event()
        ev.record(torch.cuda.current_stream())
        self._skip_ev[k] = ev
        self._skip_n += 1

    def _skipped_seen(self) -> int:
        """Skipped steps among ALL earlier optimizer steps: waits for the copy made after the previous step (its event), so the
        LR sequence after an overflow-skipped step is the same on every run — the tick does not depend on how far the host runs
        ahead of the GPU.  The wait is free in steady state: it is called after this step's forward / backward have
        been queued, so the GPU has a whole step of work while the host waits for the end of the previous one.  (No resume state
        exists to persist it in: like the reference's run, the loop saves model weights only — save_model.)"""
        if getattr(self, "_skip_host", None) is None or self._skip_n == 0:
            return 0
        k = (self._skip_n - 1) % 2
        self._skip_ev[k].synchronize()
        return int(self._skip_host[k])

    def train(self):
        """Window bookkeeping of the pasted HF loop (reference src/trainer/domain_loss.py:584-608): an epoch of
        `steps_in_epoch` micro-batches makes ceil(steps_in_epoch / GA) optimizer steps — the LAST window of an epoch may be
        shorter and still steps (`do_sync_step = (step+1) % GA == 0 or (step+1) == steps_in_epoch`, :608)."""
        a, m = self.args, self.model
        B, GA = a.per_device_train_batch_size, a.gradient_accumulation_steps
        micro_per_epoch = self._micro_per_epoch()
        if micro_per_epoch == 0:
            raise ValueError(f"{len(self.ds)} samples make fewer micro-batches of {B} than there are ranks ({self.world}): "
                             "no rank would ever step")
        steps_per_epoch = max(math.ceil(micro_per_epoch / GA), 1)      # HF: num_update_steps_per_epoch
        total = a.max_steps if a.max_steps > 0 else math.ceil(a.num_train_epochs * steps_per_epoch)
        warmup = math.ceil(total * a.warmup_ratio)
        step, epoch = 0, 0
        window_loss = torch.zeros((), dtype=torch.float32, device=m._rt.dev)
        last_logged, t0 = 0, time.time()
        while step < total:
            in_window = 0
            for mi, batch in enumerate(self._micro_batches(epoch)):
                do_sync = (mi + 1) % GA == 0 or (mi + 1) == micro_per_epoch
                loss = m.forward_backward(batch["input_ids"], batch["attention_mask"], batch["omic_ids"],
                                          batch["omic_info_list"], batch["labels"], accumulate=in_window > 0,
                                          final_micro=do_sync)
                # reference :655-661 — a non-finite micro loss adds the running average of the current logging span instead
                window_loss += torch.where(torch.isfinite(loss), loss, window_loss / (1 + step - last_logged))
                in_window += 1
                if not do_sync:
                    continue
                # DeepSpeed (what the reference runs under) does not step the LR scheduler on an overflow-skipped step: the
                # schedule's tick is the number of steps APPLIED.  The skipped counter lives on the device; it is mirrored into
                # pinned memory behind every step and read back through that copy's event: the tick counts every earlier skip,
                # deterministically, and the host still runs one step ahead of the GPU (_skipped_seen).
                lr = linear_warmup_lr(step - self._skipped_seen(), a.learning_rate, warmup, total)
                gnorm = self.opt.step(lr=lr)
                self._mirror_skipped()
                step += 1
                in_window = 0
                if step % a.logging_steps == 0 or step == total:
                    wl = window_loss.clone()
                    if self.world > 1:
                        dist.all_reduce(wl)
                        wl /= self.world
                    # learning_rate = the rate this step used (the pasted loop reads it before the scheduler steps, :717)
                    rec = {"step": step, "loss": round(float(wl.item()) / (step - last_logged), 4),
                           "grad_norm": float(gnorm.item()), "learning_rate": lr,
                           "epoch": round(epoch + (mi + 1) / micro_per_epoch, 4), "elapsed_s": round(time.time() - t0, 2)}
                    skipped = getattr(self.opt, "skipped_steps", None)
                    n_skipped = int(skipped()) if skipped is not None else 0      # one host read per log line
                    if n_skipped > 0:
                        rec["skipped_steps"] = n_skipped
                    last_logged = step
                    window_loss.zero_()
                    self.history.append(rec)
                    if self.rank == 0:
                        self.log(json.dumps(rec))
                if a.save_steps and step % a.save_steps == 0 and self.rank == 0:
                    self._save_checkpoint(step)
                if self._maybe_evaluate(step):
                    total = step                                   # early stopping: leave both loops
                if step >= total:
                    break
            epoch += 1
        if a.load_best_model_at_end and self.best_step is not None:
            self._load_best()
        return self.history

    def _load_best(self):
        """HF `load_best_model_at_end`: training ends on the weights of the best evaluation, not the last step's.  `best/` was
        written by rank 0 (`_maybe_evaluate`); every rank reads it back into its flat buffers and refreshes the fp32 masters."""
        a, m = self.args, self.model
        if dist.is_initialized():
            dist.barrier()
        d = os.path.join(a.output_dir, "best")
        rt = m._runtime()
        self.opt.wait_all_params()
        torch.cuda.synchronize()
        if rt.llm.lora is not None:
            from ..lora import load_live_adapter
            load_live_adapter(m, d)
        else:
            sd = torch.load(os.path.join(d, "pytorch_model.bin"), map_location="cpu")
            with torch.no_grad():
                for n, v in rt.P.views.items():
                    if n in sd:
                        v.copy_(sd[n].to(v.dtype))
        self.opt.refresh_master()

    def _save_checkpoint(self, step: int):
        a = self.args
        d = os.path.join(a.output_dir, f"checkpoint-{step}")
        save_model(self.model, d)
        if a.save_total_limit:
            ck = sorted((int(x.split("-")[1]), x) for x in os.listdir(a.output_dir) if x.startswith("checkpoint-"))
            for _, old in ck[:-a.save_total_limit]:
                shutil.rmtree(os.path.join(a.output_dir, old), ignore_errors=True)
