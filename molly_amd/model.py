"""`OmicsOne` — drop-in for the reference's multimodal wrapper (reference: src/model/omics_one.py:10-233).

Same constructor, attributes, `forward` / `generate` / `process_omic_sequences` signatures, error behaviour and
state-dict keys; the arithmetic runs in libmolly_hip.so.  Differences that are deliberate (and invisible in results):
  * omic rows are grouped on the host in ONE pass and moved with ONE H2D copy per modality, instead of the per-row
    `.to(device)` + device->host sync assert + per-span slice-copy launches (reference :104-118, :71-72, :93-97);
  * the encoders' MaskedLM head, which the reference computes and discards (:75-91), is never run;
  * lm_head + CE run fused and chunked (what Liger's fused-linear-CE does under --use_liger, reference src/train.py:130-132).

Sub-models are parameter shells with HuggingFace names (`Qwen3ForCausalLM`, `EsmForMaskedLM` below) that callers attach
exactly like the reference does (src/train.py:127,143,152); `prepare()` re-homes every tensor into flat HBM buffers.
"""
from __future__ import annotations

import os

import math
import re
from typing import Dict, List, Optional, Tuple

import torch
import torch.nn as nn

from . import ops
from .config import EncConfig, LlmConfig, OmicsModalConfig
from .esm import EsmEngine
from .params import (FlatBuffer, enc_param_specs, is_no_decay, llm_norm_specs, llm_param_specs, projector_specs,
                     trainable_specs)
from .qwen3 import Qwen3Engine

BF16 = torch.bfloat16


class CausalLMOutputWithPast(dict):
    """Minimal stand-in for HF's ModelOutput: attribute AND key access (HF Trainer reads outputs["loss"])."""

    def __init__(self, loss=None, logits=None, past_key_values=None, hidden_states=None, attentions=None):
        super().__init__()
        for k, v in dict(loss=loss, logits=logits, past_key_values=past_key_values, hidden_states=hidden_states,
                         attentions=attentions).items():
            if v is not None:
                self[k] = v
        self.loss, self.logits, self.past_key_values = loss, logits, past_key_values
        self.hidden_states, self.attentions = hidden_states, attentions


# ---- parameter shells with the HF module tree (names = checkpoint keys; nn.Linear leaves for LoRA discovery,
# reference src/utils/tools.py:354-361) -------------------------------------------------------------------------
def _lin(i, o, bias):
    m = nn.Linear(i, o, bias=bias, device="meta")
    return m


class _MetaSafe(nn.Module):
    """nn.Module whose still-unloaded (meta) tensors survive the calls the reference makes on its sub-models
    (reference: src/inference_lora.py:243-250 — plain `load_state_dict(sd)` then `.to(torch.bfloat16).to(device).eval()`):
      * `load_state_dict` ASSIGNS the checkpoint tensors to parameters that have no storage yet (nn.Module's default would
        "copy" into meta tensors, a silent no-op), and copies as usual into parameters that do;
      * `.to()/.cuda()/.float()/...` move what has storage and leave meta tensors alone (they never had a value to move);
        `prepare()` later refuses any tensor that is still meta unless a random init was explicitly asked for."""

    # checkpoint keys of the HF modules that Molly's forward never reads (the encoders' MaskedLM /
    # contact heads, rotary inv_freq / position_ids buffers).  The reference's `pytorch_model.bin` carries them and its
    # loader is strict (src/inference_lora.py:243), so they are accepted on load, kept on the host, and written back by
    # state_dict() — a checkpoint round-trips through this class with the reference's exact key set.
    _DEAD_KEY = re.compile(r"(^|\.)(esm\.contact_head\.|lm_head\.(bias$|dense\.|layer_norm\.|decoder\.)|"
                           r"(esm\.|self\.)?rotary_embeddings\.inv_freq$|esm\.embeddings\.position_ids$)")

    def state_dict(self, *args, destination=None, prefix="", keep_vars=False):
        sd = super().state_dict(*args, destination=destination, prefix=prefix, keep_vars=keep_vars)
        for k, v in getattr(self, "_passthrough", {}).items():
            sd[prefix + k] = v
        return sd

    def load_state_dict(self, state_dict, strict: bool = True, assign: bool = False):
        own_keys = set(super().state_dict().keys())
        dead = {k: v for k, v in state_dict.items() if k not in own_keys and self._DEAD_KEY.search(k)}
        if dead:
            if not hasattr(self, "_passthrough"):
                object.__setattr__(self, "_passthrough", {})
            self._passthrough.update({k: v.detach().cpu() for k, v in dead.items()})
            state_dict = {k: v for k, v in state_dict.items() if k not in dead}
        if not assign and any(t.is_meta for t in list(self.parameters()) + list(self.buffers())):
            # per tensor: assign where the destination is meta, copy elsewhere
            own = {**dict(self.named_parameters(remove_duplicate=False)), **dict(self.named_buffers(remove_duplicate=False))}
            to_copy = {k: v for k, v in state_dict.items() if k in own and not own[k].is_meta}
            res = super().load_state_dict(state_dict, strict=strict, assign=True)
            with torch.no_grad():
                for k, v in to_copy.items():
                    own[k].copy_(v)
                    _set_tensor(self, k, own[k])
        else:
            res = super().load_state_dict(state_dict, strict=strict, assign=assign)
        for mod in self.modules():                       # assignment unties shared tensors: tie them again (HF tie_weights)
            if hasattr(mod, "_retie_weights"):
                mod._retie_weights()
        return res

    def _apply(self, fn, recurse=True):
        return super()._apply(lambda t: t if t.is_meta else fn(t), recurse)


def _set_tensor(root: nn.Module, name: str, value: torch.Tensor):
    """Put `value` at dotted `name` under `root`, as a Parameter or a buffer — whichever the slot currently is
    (the reference's `freeze_subtree`, src/utils/tools.py:277-311, turns frozen parameters into buffers)."""
    mod = root
    *path, leaf = name.split(".")
    for k in path:
        mod = mod[int(k)] if k.isdigit() else getattr(mod, k)
    if leaf in mod._buffers:
        mod._buffers[leaf] = value.detach()
    else:
        old = mod._parameters.get(leaf)
        rg = value.requires_grad if isinstance(value, nn.Parameter) else (old.requires_grad if old is not None else True)
        setattr(mod, leaf, value if isinstance(value, nn.Parameter) else nn.Parameter(value, requires_grad=rg))


class _Shell(_MetaSafe):
    def materialize(self, std=0.02, seed=0):
        g = torch.Generator().manual_seed(seed)
        for n, p in list(self.named_parameters()):
            shape = p.shape
            if is_no_decay(n) and n.endswith("weight"):
                val = torch.ones(shape)
            elif n.endswith("bias"):
                val = torch.zeros(shape)
            else:
                val = torch.randn(shape, generator=g) * std
            mod, leaf = self, n
            *path, leaf = n.split(".")
            for k in path:
                mod = getattr(mod, k) if not k.isdigit() else mod[int(k)]
            setattr(mod, leaf, nn.Parameter(val))
        return self


class _Norm(nn.Module):
    def __init__(self, n, bias=False):
        super().__init__()
        self.weight = nn.Parameter(torch.empty(n, device="meta"))
        if bias:
            self.bias = nn.Parameter(torch.empty(n, device="meta"))


class Qwen3ForCausalLM(_Shell):
    def __init__(self, config: LlmConfig):
        super().__init__()
        c = self.config = config
        h, hd = c.hidden_size, c.head_dim
        self.model = nn.Module()
        self.model.embed_tokens = nn.Embedding(c.vocab_size, h, device="meta")
        layers = []
        for _ in range(c.num_hidden_layers):
            l = nn.Module()
            l.self_attn = nn.Module()
            l.self_attn.q_proj = _lin(h, c.num_attention_heads * hd, False)
            l.self_attn.k_proj = _lin(h, c.num_key_value_heads * hd, False)
            l.self_attn.v_proj = _lin(h, c.num_key_value_heads * hd, False)
            l.self_attn.o_proj = _lin(c.num_attention_heads * hd, h, False)
            l.self_attn.q_norm, l.self_attn.k_norm = _Norm(hd), _Norm(hd)
            l.mlp = nn.Module()
            l.mlp.gate_proj, l.mlp.up_proj, l.mlp.down_proj = (_lin(h, c.intermediate_size, False),
                                                               _lin(h, c.intermediate_size, False),
                                                               _lin(c.intermediate_size, h, False))
            l.input_layernorm, l.post_attention_layernorm = _Norm(h), _Norm(h)
            layers.append(l)
        self.model.layers = nn.ModuleList(layers)
        self.model.norm = _Norm(h)
        self.lm_head = _lin(h, c.vocab_size, False)
        if c.tie_word_embeddings:
            self.lm_head.weight = self.model.embed_tokens.weight

    def _retie_weights(self):
        if self.config.tie_word_embeddings and "weight" in self.lm_head._parameters and \
                "weight" in self.model.embed_tokens._parameters:
            self.lm_head.weight = self.model.embed_tokens.weight

    def materialize(self, std=0.02, seed=0):
        super().materialize(std, seed)
        if self.config.tie_word_embeddings:
            self.lm_head.weight = self.model.embed_tokens.weight
        else:
            g = torch.Generator().manual_seed(seed + 1)
            self.lm_head.weight = nn.Parameter(torch.randn(self.lm_head.weight.shape, generator=g) * std)
        return self

    @classmethod
    def from_config(cls, config, seed=0):
        return cls(config).materialize(seed=seed)

    def get_input_embeddings(self):
        return self.model.embed_tokens


class EsmForMaskedLM(_Shell):
    def __init__(self, config: EncConfig):
        super().__init__()
        c = self.config = config
        he = c.hidden_size
        self.esm = nn.Module()
        self.esm.embeddings = nn.Module()
        self.esm.embeddings.word_embeddings = nn.Embedding(c.vocab_size, he, device="meta")
        if c.position_embedding_type == "absolute":
            self.esm.embeddings.position_embeddings = nn.Embedding(c.max_position_embeddings, he, device="meta")
        self.esm.encoder = nn.Module()
        layers = []
        for _ in range(c.num_hidden_layers):
            l = nn.Module()
            l.attention = nn.Module()
            l.attention.self = nn.Module()
            l.attention.self.query, l.attention.self.key, l.attention.self.value = (_lin(he, he, True), _lin(he, he, True),
                                                                                    _lin(he, he, True))
            l.attention.output = nn.Module()
            l.attention.output.dense = _lin(he, he, True)
            l.attention.LayerNorm = _Norm(he, bias=True)
            l.intermediate = nn.Module()
            l.intermediate.dense = _lin(he, c.intermediate_size, True)
            l.output = nn.Module()
            l.output.dense = _lin(c.intermediate_size, he, True)
            l.LayerNorm = _Norm(he, bias=True)
            layers.append(l)
        self.esm.encoder.layer = nn.ModuleList(layers)
        self.esm.encoder.emb_layer_norm_after = _Norm(he, bias=True)

    @classmethod
    def from_config(cls, config, seed=0):
        return cls(config).materialize(seed=seed)


# ---- batch assembly: molly_amd/batch.py (one host pass, one pinned H2D copy, the rest in molly_batch_assemble) ----------
# ---- the model ---------------------------------------------------------------------------------------------------
def _in_gemm_ctx(fn):
    """Run a method with the model's own GEMM context current (launch knobs + scratch: ops.GemmContext)."""
    import functools

    @functools.wraps(fn)
    def wrapped(self, *a, **k):
        with ops.use_gemm_context(self._runtime().gemm_ctx):
            return fn(self, *a, **k)
    return wrapped


class OmicsOne(_MetaSafe):
    def __init__(self, config: OmicsModalConfig):
        super().__init__()
        self.text_config = config.text_config
        self.dna_rna_config = config.dna_rna_config
        self.protein_config = config.protein_config
        self.model = None
        self.dna_rna_model = None
        self.protein_model = None
        self.dna_rna_projector = nn.Linear(self.dna_rna_config.hidden_size, self.text_config.hidden_size)
        self.dna_rna_project_token_num = config.dna_rna_project_token_num
        self.protein_projector = nn.Linear(self.protein_config.hidden_size, self.text_config.hidden_size)
        self.protein_project_token_num = config.protein_project_token_num
        self._rt = None

    # reference: src/model/omics_one.py:32-47 (ids are stored, never read)
    def set_special_tokens(self, tokenizer):
        for m in ("dna", "rna", "protein"):
            for part in ("start", "end", "pad"):
                setattr(self, f"{m}_{part}_token_id", tokenizer.convert_tokens_to_ids(f"<|{m}_{part}|>"))

    # ---- runtime --------------------------------------------------------------------------------------------
    def prepare(self, device="cuda", train_llm=True, train_mlp=True, ce_chunk_rows=16384, rope_table_dtype=BF16,
                random_init_seed: Optional[int] = None, lora=None, train_bio=False):
        """Re-home all tensors into flat bf16 HBM buffers and build the HIP engines.

        Trainable group (what `rt.P` / `rt.G` / `n_decay` describe and the ZeRO-2 optimizer steps) — reference
        src/utils/tools.py:313-338 (`set_up_trainable_param`) and :345-396 (`pre_train_lora`):
          train_llm                      -> the whole LLM + the projectors (reference default `--train-llm --train-mlp`);
                                            without train_mlp (`--train-llm` alone, tools.py:313-338 accepts any combination)
                                            the projectors stay OUTSIDE the flat group as frozen buffers: the backward still
                                            runs through the injected rows' text neighbours, nothing is computed for them
          lora=LoraConfig (`--use-lora`) -> rank-r adapters on every LLM Linear except lm_head + the projectors; base frozen
          train_mlp only                 -> the two projectors; the LLM is frozen (backward only propagates through it)
          neither                        -> nothing (inference)
          train_bio (`--train-bio`)      -> adds both encoders (all their parameters) to whichever group the above selects;
                                            otherwise the encoders are frozen buffers, never communicated."""
        assert self.model is not None and self.dna_rna_model is not None and self.protein_model is not None, \
            "attach .model / .dna_rna_model / .protein_model first (reference: src/train.py:127,143,152)"
        if random_init_seed is None:
            # a tensor that never received a value (checkpoint not loaded, or a key it lacked) must not silently become
            # random numbers: only an explicit random_init_seed (== the reference's --no-load-pretrained) initialises
            hollow = [n for n, t in super().state_dict().items() if t.is_meta]
            if hollow:
                raise RuntimeError(f"{len(hollow)} tensors have no value (first: {hollow[:3]}): load a checkpoint "
                                   "(load_state_dict) or pass random_init_seed for a --no-load-pretrained run")
        dev = torch.device(device)
        if dev.type != "cuda":
            raise RuntimeError("molly_amd.OmicsOne runs on the GPU only; the CPU oracle lives in /oracle (tests only)")
        if dev.index is None:
            dev = torch.device("cuda", torch.cuda.current_device())
        if train_llm and lora is not None:
            raise ValueError("lora and train_llm are exclusive: the reference freezes the base under --use-lora")
        sd = self.state_dict()
        pw, pb = projector_specs(self.text_config, self.dna_rna_config, self.protein_config)
        full = bool(train_llm)
        adapters = (lora is not None) or ((train_mlp or train_bio) and not train_llm)
        enc_specs = {pre: enc_param_specs(cfg, pre) for pre, cfg in (("dna_rna_model.", self.dna_rna_config),
                                                                     ("protein_model.", self.protein_config))}
        enc_d, enc_nd = [], []
        if train_bio:
            # encoder tensors join the trainable group, split like everything else into decayed matrices / no-decay
            # gains+biases; q|k|v weights (and biases) stay adjacent inside their part, so the fused views still exist
            for specs in enc_specs.values():
                enc_d += [(n, sh) for n, sh in specs if not is_no_decay(n)]
                enc_nd += [(n, sh) for n, sh in specs if is_no_decay(n)]
        Q = None
        frozen_proj = None
        if full or not adapters:
            decay, no_decay = trainable_specs(self.text_config, self.dna_rna_config, self.protein_config)
            if full and not train_mlp:
                # `--train-llm` alone: the flat (ZeRO) group is the LLM only; the projectors are frozen buffers of their own
                decay, no_decay = llm_param_specs(self.text_config), llm_norm_specs(self.text_config)
                frozen_proj = FlatBuffer(pw + pb, dev)
            base = FlatBuffer(decay + enc_d + no_decay + enc_nd, dev, pad_to=8 * 64)
            n_decay = base.offsets[no_decay[0][0]]
        else:
            # frozen LLM in one buffer, the small trainable group (adapters + projectors [+ encoders]) in another
            base = FlatBuffer(llm_param_specs(self.text_config) + llm_norm_specs(self.text_config) +
                              ([] if (train_mlp or lora is not None) else pw + pb), dev)
            q_decay = pw if (train_mlp or lora is not None) else []
            q_nd = pb if (train_mlp or lora is not None) else []
            if lora is not None:
                from .lora import lora_specs
                q_decay = lora_specs(self.text_config, lora) + q_decay
            Q = FlatBuffer(q_decay + enc_d + q_nd + enc_nd, dev, pad_to=8 * 64)
            n_decay = Q.offsets[(q_nd + enc_nd)[0][0]]
        self.n_decay = n_decay
        enc = {}
        for pre, specs in enc_specs.items():
            # frozen encoders own a buffer each; trained ones live in the trainable group
            enc[pre] = (base if (full or not adapters) else Q) if train_bio else FlatBuffer(specs, dev)
        gen = None
        bufs = [base] + ([Q] if Q is not None else []) + ([frozen_proj] if frozen_proj is not None else []) + \
            ([] if train_bio else list(enc.values()))
        for buf in bufs:
            for n, v in buf.views.items():
                if ".lora_" in n:
                    continue                                  # adapters are initialised below
                src = sd[n]
                if src.is_meta:
                    # un-materialised shell (== the reference's --no-load-pretrained, src/train.py:107-116):
                    # initialise on the device, N(0, 0.02) matrices / unit gains / zero biases
                    if random_init_seed is None:
                        raise RuntimeError(f"parameter {n} has no value: load a checkpoint or pass random_init_seed")
                    if gen is None:
                        gen = torch.Generator(device=dev).manual_seed(random_init_seed)
                    if is_no_decay(n) and n.endswith("weight"):
                        v.fill_(1.0)
                    elif n.endswith("bias"):
                        v.zero_()
                    else:
                        v.normal_(0.0, 0.02, generator=gen)
                else:
                    v.copy_(src.to(dev))
        # re-point module parameters (and the buffers the reference's freeze_subtree made of frozen ones) at the flat views,
        # so state_dict()/save keep working and see updates, and the staging copies are released
        trainable_bufs = [base] if full else ([Q] if Q is not None else [])
        with torch.no_grad():
            named = dict(self.named_parameters(remove_duplicate=False))
            buffers = dict(self.named_buffers(remove_duplicate=False))
            for buf in bufs:
                for n, v in buf.views.items():
                    if n in named:
                        _set_tensor(self, n, nn.Parameter(v, requires_grad=named[n].requires_grad and
                                                          any(buf is tb for tb in trainable_bufs)))
                    elif n in buffers:
                        _set_tensor(self, n, v)
            if self.text_config.tie_word_embeddings:
                emb = self.model.model.embed_tokens
                if "weight" in emb._buffers:
                    self.model.lm_head._buffers["weight"] = emb._buffers["weight"]
                    self.model.lm_head._parameters.pop("weight", None)
                else:
                    self.model.lm_head.weight = emb.weight
        rt = type("Runtime", (), {})()
        rt.dev, rt.base, rt.enc = dev, base, enc
        # the launch state of every GEMM this model issues (knobs + scratch): its own, so that nothing set for it (an optimizer's
        # launch shape beside collectives) leaks into another model of the process, and nothing set elsewhere leaks in
        rt.gemm_ctx = ops.GemmContext()
        rt.gemm_ctx.ensure_workspace(0, dev)
        rt.full, rt.train_llm, rt.train_mlp = full, bool(train_llm), bool(train_mlp or lora is not None)
        rt.train_bio = bool(train_bio)
        if full:
            rt.P, rt.G = base, base.like()
        elif Q is not None:
            rt.P, rt.G = Q, Q.like()
        else:
            rt.P, rt.G = base, None
        rt.W = dict(base.views)                               # name -> weight view, whichever buffer owns it
        if Q is not None:
            rt.W.update(Q.views)
        if frozen_proj is not None:
            rt.W.update(frozen_proj.views)
        lora_rt = None
        if lora is not None:
            from .lora import LoraRuntime
            lora_rt = LoraRuntime(self.text_config, lora, Q, rt.G)
            if gen is None:
                gen = torch.Generator(device=dev).manual_seed(1234 if random_init_seed is None else random_init_seed)
            lora_rt.init_gaussian(gen)
        rt.llm = Qwen3Engine(self.text_config, base, rt.G if full else None, dev, ce_chunk_rows=ce_chunk_rows,
                             rope_table_dtype=rope_table_dtype, lora=lora_rt, backward=rt.G is not None)
        eg = rt.G if train_bio else None
        rt.dna = EsmEngine(self.dna_rna_config, enc["dna_rna_model."], dev, "dna_rna_model.", rope_table_dtype, grads=eg)
        rt.prot = EsmEngine(self.protein_config, enc["protein_model."], dev, "protein_model.", rope_table_dtype, grads=eg)
        self._rt = rt
        return self

    def infer_trainable(self):
        """(train_llm, train_mlp, train_bio) as the module tree says it — what the reference's `set_up_trainable_param`
        (src/utils/tools.py:313-338) leaves behind: a frozen sub-tree has had its Parameters re-registered as buffers
        (`freeze_subtree`, :277-311) or carries requires_grad=False; a trainable one still has Parameters that require grad."""
        def live(*mods):
            return any(p.requires_grad for m in mods if m is not None for p in m.parameters())
        return (live(self.model), live(self.dna_rna_projector, self.protein_projector),
                live(self.dna_rna_model, self.protein_model))

    def prepare_from_module_state(self, device=None, **kw):
        """`prepare()` with the trainable set read off the module tree — the call to make after the reference's own
        `set_up_trainable_param(model, args)`, or after `inference_lora.py`'s `load_state_dict(...)` +
        `.to(torch.bfloat16).to(device).eval()` (nothing requires grad in eval use: pass through `torch.no_grad`)."""
        if device is None:
            devs = {t.device for t in list(self.parameters()) + list(self.buffers()) if t.is_cuda}
            device = devs.pop() if len(devs) == 1 else "cuda"
        llm, mlp, bio = self.infer_trainable() if self.training else (False, False, False)    # .eval(): inference only
        return self.prepare(device, train_llm=llm, train_mlp=mlp, train_bio=bio, **kw)

    def _runtime(self):
        if self._rt is None:
            self.prepare_from_module_state()
        return self._rt

    def _stager(self):
        rt = self._rt
        if getattr(rt, "stager", None) is None:
            from .batch import BatchStager
            rt.stager = BatchStager(rt.dev, self.text_config.vocab_size,
                                    {"dna_rna": self.dna_rna_config.vocab_size, "protein": self.protein_config.vocab_size})
        return rt.stager

    def _stage(self, input_ids, labels, attention_mask, omic_ids, omic_info_list, want_sort):
        """One host pass + one pinned host->device copy + the device-side assembly (molly_amd/batch.py)."""
        B, T = input_ids.shape
        if input_ids.is_cuda:                 # HF-Trainer-style callers move the batch first; the host pass needs it back
            input_ids = input_ids.cpu()
        return self._stager().stage(B, T, input_ids, labels, attention_mask, omic_ids, omic_info_list,
                                    {"dna_rna": self.dna_rna_project_token_num, "protein": self.protein_project_token_num},
                                    want_sort=want_sort)

    @_in_gemm_ctx
    def _embed_and_inject(self, st, keep_for_backward, wait_embed=None, wait_proj=None):
        """reference: src/model/omics_one.py:164-172 — token embeddings, then encoder -> projector -> overwrite.  `st` is the
        staged batch (device-side ids, encoder ids, scatter indices).
        Launch order: the FROZEN encoders run first — they read no trainable parameter, so their ≈8 ms cover the head of the
        previous step's side-stream AdamW / all-gather (embedding, gains, projectors), which `wait_embed` / `wait_proj` then
        find finished; the values written are those of the reference's order (lookup, then overwrite)."""
        rt = self._rt
        B, T = st.B, st.T
        M = B * T
        if not keep_for_backward and getattr(rt, "opt", None) is not None:
            rt.opt.wait_all_params()              # inference entry points: every parameter must have been published
        rt.llm.reserve(M, B, T, training=keep_for_backward)
        hs = rt.llm.A[0]["x"] if keep_for_backward else rt.llm.x_out
        saved = {}
        encoded = []
        if rt.train_bio or os.environ.get("MOLLY_ENC_FIRST", "1") == "0":   # trained encoders read parameters the optimizer is publishing
            for w in (wait_embed, wait_proj):
                if w is not None:
                    w()
            wait_embed = wait_proj = None
        present = [(name, eng, proj) for name, eng, proj in (("dna_rna", rt.dna, "dna_rna_projector"), ("protein", rt.prot, "protein_projector"))
                   if name in st.groups]
        # Both modality groups in one batch (BASELINE configs 3 / 4: DNA / RNA spans through the NT encoder, protein spans through
        # ESM-2) and both encoders frozen: the two stacks share nothing, and at one sample per GPU (512-1,024 rows) neither fills
        # the chip — a projection is 40-160 workgroups for 256 CUs — so the first one runs on a side stream with a GEMM context
        # (scratch) of its own, beside the second.  Same kernels, same values; MOLLY_ENC_STREAMS=0 runs them one after the other.
        side_done = None
        if len(present) == 2 and not rt.train_bio and os.environ.get("MOLLY_ENC_STREAMS", "1") != "0":
            if getattr(rt, "enc_stream", None) is None:
                rt.enc_stream = torch.cuda.Stream(device=rt.dev)
                rt.enc_ctx = ops.GemmContext()
                rt.enc_ctx.ensure_workspace(0, rt.dev)
            ready = torch.cuda.Event()
            ready.record(torch.cuda.current_stream())              # the staged ids (and whatever wrote the engine's buffers last)
            rt.enc_stream.wait_event(ready)
        for k, (name, eng, proj) in enumerate(present):
            ids64, dst, N, K = st.groups[name]
            on_side = len(present) == 2 and k == 0 and getattr(rt, "enc_stream", None) is not None and not rt.train_bio \
                and os.environ.get("MOLLY_ENC_STREAMS", "1") != "0"
            try:
                if on_side:
                    with torch.cuda.stream(rt.enc_stream), ops.use_gemm_context(rt.enc_ctx):
                        enc_out = eng.forward(ids64, training=False)
                        side_done = torch.cuda.Event()
                        side_done.record(rt.enc_stream)
                else:
                    enc_out = eng.forward(ids64, training=keep_for_backward and rt.train_bio)
            except Exception as e:  # reference re-wraps encoder failures (omics_one.py:89-90)
                raise RuntimeError(f"Error processing omic sequences: {e}")
            encoded.append((name, eng, proj, enc_out, dst))
        if side_done is not None:
            torch.cuda.current_stream().wait_event(side_done)
        if wait_embed is not None:
            wait_embed()
        ops.copy_rows(rt.llm.embed, hs, M, src_idx32=st.ids32)
        if wait_proj is not None:
            wait_proj()
        for name, eng, proj, enc_out, dst in encoded:
            emb = ops.gemm_nt(enc_out, rt.W[proj + ".weight"], bias=rt.W[proj + ".bias"])
            ops.copy_rows(emb, hs, emb.shape[0], dst_idx32=dst)
            saved[name] = (enc_out, dst, proj, eng)
        return hs, saved

    @staticmethod
    def _kv_range(attention_mask, B, T, dev):
        """attention_mask [B,T] of 0/1 with contiguous ones (right- or left-padded) -> per-sample [lo, hi) on the device."""
        from .batch import BatchStager
        kv = BatchStager.kv_range(attention_mask, B, T)
        return (None, None) if kv is None else (kv[0].to(dev), kv[1].to(dev))

    def attach_optimizer(self, opt):
        """Wire a Zero2Optimizer's overlap hooks into the engines (no-ops when the optimizer does not overlap)."""
        rt = self._runtime()
        rt.opt = opt
        mode = getattr(opt, "gemm_blocks_mode", None)
        if mode is not None:                                   # this model's GEMMs run beside the optimizer's collectives
            rt.gemm_ctx.set("persistent_blocks", 256 if mode == "dyn" else mode)
            rt.gemm_ctx.set("dynamic", 1 if mode == "dyn" else 0)
        if rt.full:
            # per-layer overlap: flat offsets of the optimizer's group ARE the LLM's parameter offsets
            opt.hooked = True
            rt.llm.grads_final_hook = opt.on_grads_final
            rt.llm.wait_params_hook = opt.wait_params

    @_in_gemm_ctx
    def forward_backward(self, input_ids, attention_mask, omic_ids, omic_info_list, labels, accumulate=False,
                         final_micro=True):
        """One training micro-step on the native path: forward + loss + full backward into the flat grad buffer.
        Returns the loss as a device scalar (no host sync).  Loss semantics = per-micro-batch token mean
        (HF:loss/loss_utils.py:49-71 with num_items_in_batch=None: the reference swallows it)."""
        rt = self._runtime()
        B, T = input_ids.shape
        M = B * T
        opt = getattr(rt, "opt", None)
        if rt.G is None:
            raise RuntimeError("forward_backward: nothing is trainable (prepare(train_llm/train_mlp/lora))")
        if opt is not None and not rt.full:
            opt.wait_all_params()                 # small adapter/projector group: no per-layer overlap
        wait_embed = wait_proj = None
        if opt is not None and rt.full:
            # parameters read before the decoder layers: embedding (= tied head), then gains/biases (tail region) and the
            # projectors — waited for inside _embed_and_inject, behind the frozen encoders' forward
            wait_embed = lambda: opt.wait_params(0, rt.llm.layer_lo[0])
            wait_proj = lambda: (opt.wait_params(self.n_decay, rt.P.numel), opt.wait_params(rt.llm.layers_hi, self.n_decay))
        # everything the step needs from the batch — ids, shifted labels, scored rows, encoder ids, scatter indices, key ranges
        # and the sorted embedding-gradient index — through ONE pinned upload and the device-side assembly
        st = self._stage(input_ids, labels, attention_mask, omic_ids, omic_info_list, want_sort=rt.train_llm)
        with ops.roctx("molly: encoders + projector + inject"):
            hs, saved = self._embed_and_inject(st, True, wait_embed, wait_proj)
        with ops.roctx("molly: decoder forward"):
            rt.llm.forward(hs, B, T, st.kv_lo, st.kv_hi, labels_shifted=st.labels_shifted, training=True,
                           scored_rows=st.scored_rows)
        if opt is not None:
            opt.wait_all_params()
        with ops.roctx("molly: lm_head + CE + decoder backward"):
            d_hs = rt.llm.loss_and_backward(accumulate=accumulate, final_micro=final_micro)
        # ---- gradient of the input embeddings: text rows -> embed_tokens, omic rows -> projector
        if rt.train_llm and not accumulate and not self.text_config.tie_word_embeddings:
            # untied head (Qwen3-8B): nothing else writes the embedding's gradient, and embed_bwd ADDS to it — start from zero
            # (tied: the head's weight gradient has just overwritten the shared tensor)
            rt.llm.d_embed.zero_()
        if st.emb_index is not None and st.n_overwritten < M:
            order, seg, uid, n_unique_dev, bound = st.emb_index
            ops.embed_bwd(d_hs, order, seg, uid, bound, rt.llm.d_embed, n_unique_dev=n_unique_dev)
        for name, (enc_out, dst_dev, proj, eng) in saved.items():
            if not (rt.train_mlp or rt.train_bio):
                break
            N = enc_out.shape[0]
            d_emb = torch.zeros(N, self.text_config.hidden_size, dtype=BF16, device=rt.dev)
            ops.copy_rows(d_hs, d_emb, N, src_idx32=dst_dev)
            if rt.train_mlp:
                ops.gemm(d_emb, enc_out, out=rt.G.views[proj + ".weight"], accumulate=accumulate, a_kmajor=True,
                         b_kmajor=True)
                ops.colsum(d_emb, rt.G.views[proj + ".bias"], accumulate=accumulate)
            if rt.train_bio:
                # d(encoder output) = d_emb W_proj, then the encoder's own backward (reference `--train-bio`)
                eng.backward(ops.gemm(d_emb, rt.W[proj + ".weight"], b_kmajor=True), accumulate=accumulate)
        if not accumulate:
            # parameters of a modality absent from this batch still own grad slots: they must read as zero
            for name, proj, pre in (("dna_rna", "dna_rna_projector", "dna_rna_model."),
                                    ("protein", "protein_projector", "protein_model.")):
                if name in saved:
                    continue
                if rt.train_mlp:
                    rt.G.views[proj + ".weight"].zero_()
                    rt.G.views[proj + ".bias"].zero_()
                if rt.train_bio:
                    for n, v in rt.G.views.items():
                        if n.startswith(pre):
                            v.zero_()
        return rt.llm.scal[2]

    @_in_gemm_ctx
    def process_omic_sequences(self, hidden_states, omic_ids_list, omic_info_list, device=None):
        """reference: src/model/omics_one.py:49-136 — in-place overwrite of `hidden_states` [B,T,h]; returns it."""
        rt = self._runtime()
        B, T, h = hidden_states.shape
        st = self._stager().stage(B, T, None, None, None, omic_ids_list, omic_info_list,
                                  {"dna_rna": self.dna_rna_project_token_num, "protein": self.protein_project_token_num})
        flat = hidden_states.view(B * T, h)
        for name, eng, proj in (("dna_rna", rt.dna, "dna_rna_projector"), ("protein", rt.prot, "protein_projector")):
            if name not in st.groups:
                continue
            ids64, dst, N, K = st.groups[name]
            try:
                enc_out = eng.forward(ids64)
            except Exception as e:
                raise RuntimeError(f"Error processing omic sequences: {e}")
            emb = ops.gemm_nt(enc_out, rt.W[proj + ".weight"], bias=rt.W[proj + ".bias"])
            ops.copy_rows(emb, flat, emb.shape[0], dst_idx32=dst)
        return hidden_states

    @_in_gemm_ctx
    def forward(self, input_ids=None, attention_mask=None, omic_ids=None, omic_info_list=None, labels=None,
                past_key_values=None, use_cache=None, output_attentions=None, output_hidden_states=None,
                return_dict=None, task_label=None, task_num=None, **kwargs):
        """reference: src/model/omics_one.py:138-185.  Inference / evaluation forward: returns logits (and the loss when
        labels are given).  Training goes through `forward_backward` (the native fused path) — under
        torch.is_grad_enabled() with labels this method runs it and returns a loss whose .backward() hands the already
        computed flat gradients to autograd, so HF-Trainer-style loops keep working."""
        rt = self._runtime()
        if output_attentions or output_hidden_states:
            raise NotImplementedError("output_attentions / output_hidden_states are not produced by the fused path")
        B, T = input_ids.shape
        if labels is not None and torch.is_grad_enabled() and rt.G is not None:
            loss = self.forward_backward(input_ids, attention_mask, omic_ids, omic_info_list, labels)
            return CausalLMOutputWithPast(loss=_attach_grads(self, loss))
        st = self._stage(input_ids, labels, attention_mask, omic_ids, omic_info_list, want_sort=False)
        hs, _ = self._embed_and_inject(st, False)
        loss, logits = rt.llm.forward(hs, B, T, st.kv_lo, st.kv_hi, labels_shifted=st.labels_shifted, training=False,
                                      return_logits=True)
        return CausalLMOutputWithPast(loss=loss.clone() if loss is not None else None,
                                      logits=logits.view(B, T, -1))


    @torch.no_grad()
    @_in_gemm_ctx
    def generate(self, input_ids, attention_mask=None, omic_ids=None, omic_info_list=None, max_length=None, min_length=None,
                 do_sample=True, temperature=0.8, top_p=0.95, top_k=None, num_beams=None, no_repeat_ngram_size=None,
                 **generate_kwargs):
        """reference: src/model/omics_one.py:187-233 — same signature; returns the NEW tokens only.  `max_new_tokens`
        defaults to the reference's hard-coded 3072 (:223) and may be overridden through generate_kwargs.

        `num_beams`, `max_length` and `min_length` are named parameters the reference accepts and DROPS: its call of
        `self.model.generate` (:220-232) forwards do_sample, temperature, top_p, top_k, no_repeat_ngram_size, the pad / eos ids and
        **generate_kwargs only, so `num_beams=4` there still samples (or decodes greedily).  The same happens here (one warning).
        Beam search / beam sampling (molly_amd/beam.py, pinned to HuggingFace's `_beam_search`) is an EXTENSION of this build,
        opt-in through `generate_kwargs["molly_num_beams"]`."""
        if num_beams not in (None, 1) and not getattr(OmicsOne, "_warned_num_beams", False):
            import warnings
            warnings.warn("OmicsOne.generate: `num_beams` is accepted and ignored, as in the reference (src/model/omics_one.py:220-232 "
                          "does not forward it); pass molly_num_beams=N for this build's beam search", stacklevel=2)
            OmicsOne._warned_num_beams = True
        if omic_ids is not None:
            for i in range(len(omic_ids)):
                assert len(omic_ids[i]) == len(omic_info_list[i]), f"Mismatch in omic count vs info count at index {i}"
        from .generate import generate as _gen
        cfg = self.text_config
        # ---- **generate_kwargs: the reference forwards every one of them to HuggingFace's generate (:220-232).  What this build honours is
        # taken out below; anything left over is REFUSED (a silently dropped `stopping_criteria` or `bad_words_ids` would change the output
        # without a word — VERDICT r05).  `use_cache`: the reference sets it to False under world > 1 (:202-204: HF then re-runs the whole
        # sequence per token, same tokens); here the KV cache is this rank's own memory and is always used, so both values mean the same.
        if torch.distributed.is_available() and torch.distributed.is_initialized() and torch.distributed.get_world_size() > 1:
            generate_kwargs.setdefault("use_cache", False)
        use_cache = generate_kwargs.pop("use_cache", None)
        if use_cache not in (None, True, False):
            raise TypeError(f"OmicsOne.generate: use_cache={use_cache!r}")
        kw = dict(max_new_tokens=generate_kwargs.pop("max_new_tokens", 3072), repetition_penalty=generate_kwargs.pop("repetition_penalty", None),
                  generator=generate_kwargs.pop("generator", None), num_beams=int(generate_kwargs.pop("molly_num_beams", 1) or 1),
                  length_penalty=generate_kwargs.pop("length_penalty", 1.0), early_stopping=generate_kwargs.pop("early_stopping", False),
                  min_new_tokens=int(generate_kwargs.pop("min_new_tokens", 0) or 0))
        for k in ("pad_token_id", "eos_token_id"):                 # (the reference passes these itself: a second copy in **kwargs is a TypeError there too)
            if k in generate_kwargs:
                raise TypeError(f"OmicsOne.generate() got multiple values for keyword argument '{k}' (the reference passes the model config's)")
        if generate_kwargs:
            raise NotImplementedError(
                "OmicsOne.generate: generate_kwargs not honoured by this build: " + ", ".join(sorted(generate_kwargs)) +
                " (honoured: max_new_tokens, min_new_tokens, repetition_penalty, use_cache, generator, length_penalty / early_stopping with "
                "molly_num_beams); the reference forwards them to HuggingFace's generate (src/model/omics_one.py:220-232)")
        return _gen(self, input_ids, attention_mask, omic_ids, omic_info_list, do_sample=do_sample, temperature=temperature,
                    top_p=top_p, top_k=top_k, pad_token_id=cfg.pad_token_id, eos_token_id=cfg.eos_token_id,
                    no_repeat_ngram_size=no_repeat_ngram_size, **kw)


class _GradHandOff(torch.autograd.Function):
    @staticmethod
    def forward(ctx, loss, model, *params):
        ctx.model = model
        ctx.n = len(params)
        return loss.clone()

    @staticmethod
    def backward(ctx, gout):
        rt = ctx.model._rt
        grads = []
        named = dict(ctx.model.named_parameters())
        for n, p in named.items():
            if p.requires_grad and n in rt.G.views:
                grads.append(rt.G.views[n].to(p.dtype) * gout)
        return (None, None, *grads)


def _attach_grads(model, loss):
    params = [p for n, p in model.named_parameters() if p.requires_grad and n in model._rt.G.views]
    return _GradHandOff.apply(loss, model, *params)
