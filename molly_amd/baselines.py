"""Enc-Head baselines on the encoder kernels: backbone(s) -> [CLS] -> concat -> Linear head -> loss.

Mirror of the reference's `BackboneWithClsHead` (reference: baselines/model.py:33-215) for the model
types built from the ESM-architecture encoders of the hot path — "NT", "ESM", "NT+ESM", "NT+NT", "ESM+ESM" ("EVO" wraps a
third-party package that is not part of the path: NotImplementedError).  Same constructor arguments, attribute names
(`backbone` / `nt`,`esm` / `nt1`,`nt2` / `esm1`,`esm2`, `head`), state-dict keys, `forward(x1, x2, mask1, mask2, labels)`
returning `.loss` / `.logits`, `freeze_backbone()`.

Arithmetic: each backbone runs through `EsmEngine` (esm.py) — its output is `hidden_states[-1]` of the HF model, the tensor
`_cls` indexes at position 0 (baselines/model.py:115-119); the features are gathered side by side into one [B, dim] buffer
(`copy_rows`), the head is one GEMM against a weight padded to 64 label rows (zero rows: their logits are never read and
their gradient is exactly zero), the loss is `molly_cls_loss_fwd_bwd` (F.cross_entropy, or BCE-with-logits under
`multi_answer`, baselines/model.py:196-204).  Backward: head wgrad / bias column sum / dgrad GEMMs, the feature gradient
scattered back to the [CLS] rows of a zero gradient of the encoder output, then the hand-scheduled encoder backward
(`EsmEngine.backward`).  The reference loads the backbones in fp32 (baselines/model.py:83,93); here they run in bf16 like
everywhere on the path (fp32 accumulation), so parity with the fp32 reference is at bf16 resolution.
"""
from __future__ import annotations

from typing import List, Optional, Tuple

import torch
import torch.nn as nn

from . import ops
from .config import EncConfig
from .esm import EsmEngine
from .model import EsmForMaskedLM, _lin
from .params import FlatBuffer, enc_param_specs, is_no_decay

BF16 = torch.bfloat16
HEAD_ROWS = 64          # label rows of the padded head weight: one K-tile of the head's dgrad GEMM

_PARTS = {"NT": (("backbone", "nt"),), "ESM": (("backbone", "esm"),), "NT+ESM": (("nt", "nt"), ("esm", "esm")),
          "NT+NT": (("nt1", "nt"), ("nt2", "nt")), "ESM+ESM": (("esm1", "esm"), ("esm2", "esm"))}


class SequenceClassifierOutput(dict):
    """`.loss` / `.logits` like transformers.modeling_outputs.SequenceClassifierOutput (baselines/model.py:206-209)."""

    def __init__(self, loss=None, logits=None):
        super().__init__(loss=loss, logits=logits)
        self.loss, self.logits = loss, logits


def _enc_config(x) -> EncConfig:
    if isinstance(x, EncConfig):
        return x
    if isinstance(x, dict):
        return EncConfig.from_dict(x)
    from . import config as C
    return EncConfig.from_dict(C._load_json_config(x))


class BackboneWithClsHead(nn.Module):
    def __init__(self, model_type: str, nt_model=None, esm_model=None, num_labels: int = 2, multi_label: bool = False,
                 multi_answer: bool = False):
        """nt_model / esm_model: an `EncConfig`, a config dict, or the directory of a HF checkpoint (its config.json)."""
        super().__init__()
        if model_type == "EVO":
            raise NotImplementedError("model_type 'EVO' wraps the third-party evo2 package (baselines/model.py:10-30)")
        if model_type not in _PARTS:
            raise ValueError(f"Invalid model_type: {model_type}")                  # baselines/model.py:78
        if num_labels > HEAD_ROWS:
            raise NotImplementedError(f"num_labels={num_labels} > {HEAD_ROWS}")
        self.model_type, self.multi_label, self.multi_answer, self.num_labels = model_type, multi_label, multi_answer, num_labels
        src = {"nt": nt_model, "esm": esm_model}
        self.part_names: List[str] = []
        dim = 0
        for attr, kind in _PARTS[model_type]:
            if src[kind] is None:
                raise ValueError(f"model_type {model_type} needs {kind}_model")
            shell = EsmForMaskedLM(_enc_config(src[kind]))
            setattr(self, attr, shell)
            self.part_names.append(attr)
            dim += shell.config.hidden_size
        self.dim = dim
        self.head = _lin(dim, num_labels, True)
        self.backbone_frozen = False
        self._rt = None

    # reference: baselines/model.py:211-225
    def freeze_backbone(self):
        assert self._rt is None, "freeze_backbone() before prepare()"
        self.backbone_frozen = True
        for a in self.part_names:
            for p in getattr(self, a).parameters():
                p.requires_grad = False
        return self

    # ------------------------------------------------------------------------------------------------------------------
    def prepare(self, device="cuda", random_init_seed: Optional[int] = None):
        """Re-home every tensor into flat bf16 buffers (trainable group: head [+ backbones], decayed matrices first, then
        gains/biases — HF's decay split) and build the engines."""
        dev = torch.device(device)
        if dev.type != "cuda":
            raise RuntimeError("molly_amd baselines run on the GPU only; the CPU oracle lives in /oracle (tests only)")
        sd = self.state_dict()
        head_w, head_b = ("head.weight", (HEAD_ROWS, self.dim)), ("head.bias", (HEAD_ROWS,))
        enc_specs = {a: enc_param_specs(getattr(self, a).config, a + ".") for a in self.part_names}
        train_enc = not self.backbone_frozen
        decay, nodecay = [head_w], [head_b]
        if train_enc:
            for specs in enc_specs.values():
                decay += [(n, s) for n, s in specs if not is_no_decay(n)]
                nodecay += [(n, s) for n, s in specs if is_no_decay(n)]
        P = FlatBuffer(decay + nodecay, dev, pad_to=8 * 64)
        self.n_decay = P.offsets[nodecay[0][0]]
        enc_buf = {a: (P if train_enc else FlatBuffer(enc_specs[a], dev)) for a in self.part_names}
        gen = None
        for buf in [P] + ([] if train_enc else list(enc_buf.values())):
            for n, v in buf.views.items():
                src = sd[n]
                if src.is_meta:
                    if random_init_seed is None:
                        raise RuntimeError(f"parameter {n} has no value: load a state dict or pass random_init_seed")
                    if gen is None:
                        gen = torch.Generator(device=dev).manual_seed(random_init_seed)
                    tgt = v[:self.num_labels] if n.startswith("head.") else v
                    if is_no_decay(n) and n.endswith("weight"):
                        tgt.fill_(1.0)
                    elif n.endswith("bias"):
                        tgt.zero_()
                    else:
                        tgt.normal_(0.0, 0.02, generator=gen)
                elif n.startswith("head."):
                    v[:self.num_labels].copy_(src.to(dev))                      # rows num_labels.. stay zero
                else:
                    v.copy_(src.to(dev))
        # module parameters become views of the flat buffers (state_dict() keeps the reference's keys and shapes)
        with torch.no_grad():
            named = dict(self.named_parameters())
            for buf in [P] + ([] if train_enc else list(enc_buf.values())):
                for n, v in buf.views.items():
                    if n not in named:
                        continue
                    mod = self
                    *path, leaf = n.split(".")
                    for k in path:
                        mod = mod[int(k)] if k.isdigit() else getattr(mod, k)
                    view = v[:self.num_labels] if n.startswith("head.") else v
                    setattr(mod, leaf, nn.Parameter(view, requires_grad=buf is P))
        rt = type("Runtime", (), {})()
        rt.dev, rt.P, rt.G = dev, P, P.like()
        rt.eng = [EsmEngine(getattr(self, a).config, enc_buf[a], dev, a + ".", grads=rt.G if train_enc else None)
                  for a in self.part_names]
        rt.W, rt.b = P.views["head.weight"], P.views["head.bias"]
        rt.dW, rt.db = rt.G.views["head.weight"], rt.G.views["head.bias"]
        rt.scal = torch.zeros(4, dtype=torch.float32, device=dev)                 # [0] 1/count  [1] count  [2] loss
        self._rt = rt
        return self

    # ------------------------------------------------------------------------------------------------------------------
    def _features(self, xs, masks, training: bool):
        rt = self._rt
        B = xs[0].shape[0]
        h = torch.empty(B, self.dim, dtype=BF16, device=rt.dev)
        col = 0
        rt.cls_rows = []
        for eng, x, m in zip(rt.eng, xs, masks):
            x = x.to(rt.dev)
            pad = eng.cfg.pad_token_id
            if m is not None and not torch.equal(m.to(rt.dev) != 0, x != pad):
                raise NotImplementedError("attention_mask must be (input_ids != pad_token_id), what the tokenizers produce")
            out = eng.forward(x, training=training)                                # [B*K, he]
            K = x.shape[1]
            rows = torch.arange(B, dtype=torch.int32, device=rt.dev) * K           # the [CLS] row of every sequence
            ops.copy_rows(out, h[:, col:col + eng.he], B, src_idx32=rows)
            rt.cls_rows.append((rows, K, col))
            col += eng.he
        return h

    def _loss(self, logits, labels, write_grad):
        rt = self._rt
        B = logits.shape[0]
        row_loss = torch.empty(B, dtype=torch.float32, device=rt.dev)
        if self.multi_answer:
            rt.scal[0] = 1.0 / (B * self.num_labels)                                # mean over every element
            ops.cls_loss_fwd_bwd(logits, self.num_labels, row_loss, rt.scal[0:1],
                                 targets=labels.to(rt.dev, torch.float32).contiguous(), write_grad=write_grad)
        else:
            lab = labels.to(rt.dev, torch.int64).contiguous()
            ops.count_valid(lab, rt.scal[0:1], rt.scal[1:2])                        # mean over the scored rows
            ops.cls_loss_fwd_bwd(logits, self.num_labels, row_loss, rt.scal[0:1], labels=lab, write_grad=write_grad)
        ops.sum_f32(row_loss, rt.scal[2:3], scale=rt.scal[0:1])
        return rt.scal[2]

    def _inputs(self, x1, x2, mask1, mask2) -> Tuple[list, list]:
        two = len(self.part_names) == 2
        if two and x2 is None:
            raise ValueError(f"model_type {self.model_type} takes two inputs")
        return ([x1, x2], [mask1, mask2]) if two else ([x1], [mask1])

    def forward(self, x1, x2=None, mask1=None, mask2=None, labels=None):
        """reference: baselines/model.py:122-209.  Inference / evaluation entry: no gradient is produced."""
        if self._rt is None:
            raise RuntimeError("call prepare() first")
        xs, masks = self._inputs(x1, x2, mask1, mask2)
        rt = self._rt
        with torch.no_grad():
            h = self._features(xs, masks, training=False)
            logits = ops.gemm_nt(h, rt.W, bias=rt.b)
            loss = None
            if labels is not None:
                loss = self._loss(logits.clone(), labels, write_grad=False).clone()
        return SequenceClassifierOutput(loss=loss, logits=logits[:, :self.num_labels].float())

    def forward_backward(self, x1, x2=None, mask1=None, mask2=None, labels=None, accumulate: bool = False):
        """One training micro-step: loss + gradients of the trainable group into the flat gradient buffer
        (`accumulate`: add to it — gradient-accumulation micro-steps > 0).  Returns the loss (device scalar)."""
        if self._rt is None:
            raise RuntimeError("call prepare() first")
        assert labels is not None
        xs, masks = self._inputs(x1, x2, mask1, mask2)
        rt = self._rt
        train_enc = not self.backbone_frozen
        with torch.no_grad():
            h = self._features(xs, masks, training=train_enc)
            logits = ops.gemm_nt(h, rt.W, bias=rt.b)                                # [B, 64]; columns >= num_labels unused
            loss = self._loss(logits, labels, write_grad=True)                      # logits now hold d(logits)
            dlog = logits
            ops.gemm(dlog, h, out=rt.dW, accumulate=accumulate, a_kmajor=True, b_kmajor=True)   # dW = dlogits^T h
            ops.colsum(dlog, rt.db, accumulate=accumulate)
            if train_enc:
                dh = ops.gemm(dlog, rt.W, b_kmajor=True)                            # [B, dim]
                for eng, (rows, K, col) in zip(rt.eng, rt.cls_rows):
                    B = dh.shape[0]
                    d_out = torch.zeros(B * K, eng.he, dtype=BF16, device=rt.dev)
                    ops.copy_rows(dh[:, col:col + eng.he], d_out, B, dst_idx32=rows)
                    eng.backward(d_out, accumulate=accumulate)
        return loss
