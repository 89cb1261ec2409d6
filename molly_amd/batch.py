"""Batch staging for the hot path: one host pass, ONE pinned host->device copy, the rest on the device.

What the reference does around every model call — reference src/model/omics_one.py:104-118 (a Python loop that moves every
omic row to the device on its own), :69-72 (stack, mask, and an `assert` that synchronises device and host), :93-97 (one
slice-copy launch per span) — and what round 1 of this library still did on the host (a numpy loop per span, an argsort per
step for the embedding gradient, six small pageable uploads) is replaced by:

  host   one pass over `omic_info_list` (a list of dicts cannot be read anywhere else) -> a packed int32 table
         (b, start, group, row); the reference's checks (id range, unknown type, count mismatch) and ours (trailing pads,
         span inside the sequence) vectorised on the CPU tensors — no device sync; everything written into ONE pinned int32
         image [token ids | labels | key ranges | span table | dna/rna ids | protein ids];
  copy   one `copy_(non_blocking=True)` of that image (a ring of pinned images, so the host never waits for the GPU);
  device `molly_batch_assemble`: shifted labels + scored-row list, encoder ids, scatter indices, overwritten mask and the
         sorted embedding-gradient index — all at static addresses (the device buffers are cached per shape).
"""
from __future__ import annotations

from typing import Dict, List, Optional

import torch

from . import ops
from ._lib import lib

GROUPS = ("dna_rna", "protein")
_GROUP_OF = {"dna": 0, "rna": 0, "protein": 1}


class StagedBatch:
    """Device-side views of one staged batch (valid until the stager is used again on the same shape)."""
    __slots__ = ("B", "T", "ids32", "labels_shifted", "scored_rows", "n_scored", "kv_lo", "kv_hi", "groups", "overwritten",
                 "emb_index", "n_overwritten")


class BatchStager:
    def __init__(self, device, text_vocab: int, enc_vocab: Dict[str, int], ring: int = 3):
        self.dev, self.V, self.enc_vocab = device, int(text_vocab), enc_vocab
        self.ring = ring
        self._pinned: Dict[int, list] = {}       # size class -> [(tensor, event | None)] * ring
        self._slot = 0
        self._dev: Dict[tuple, dict] = {}        # shape key -> device buffers

    # ---- host side ---------------------------------------------------------------------------------------------
    @staticmethod
    def _spans(omic_ids, omic_info_list, K_cfg: Dict[str, int], T: int):
        """reference: src/model/omics_one.py:99-118.  -> (table rows [(b, start, group, row)], per group the list of (b, j))."""
        table, members = [], ([], [])
        for b, infos in enumerate(omic_info_list):
            assert len(omic_ids[b]) == len(infos), f"Mismatch in DNA count vs start_pos count at index {b}"
            for j, info in enumerate(infos):
                t = info["type"]
                if t == "pad":
                    continue
                g = _GROUP_OF.get(t)
                if g is None:
                    raise ValueError(f"Unsupported omic type: {t}")
                table.append((b, int(info["start"]), g, len(members[g])))
                members[g].append((b, j))
        return table, members

    @staticmethod
    def _rows(omic_ids, members) -> Optional[torch.Tensor]:
        if not members:
            return None
        if isinstance(omic_ids, torch.Tensor):
            bs, js = zip(*members)
            return omic_ids[list(bs), list(js)].to(torch.int64).cpu()
        return torch.stack([torch.as_tensor(omic_ids[b][j]) for b, j in members], 0).to(torch.int64).cpu()

    @staticmethod
    def _check_rows(ids: torch.Tensor, vocab: int, pad: int = 1):
        # reference: src/model/omics_one.py:71-72 (there a device->host sync; here the ids are still on the host)
        assert bool((ids < vocab).all()), f"out-of-range token: {ids[ids >= vocab]}"
        # the attention kernel masks keys by a per-sequence valid LENGTH; the reference's mask is `ids != 1` (:70).  They
        # coincide when pads are trailing, which is what the reference tokenisation emits (padding='max_length',
        # src/dataset/omics_dataset.py:430-444).  Interior pads are rejected loudly.
        m = ids != pad
        n_valid = m.sum(1)
        last = torch.where(m.any(1), (m.long() * torch.arange(1, ids.shape[1] + 1)).max(1).values, torch.zeros_like(n_valid))
        if not torch.equal(n_valid, last):
            raise NotImplementedError("omic ids with pad tokens (id 1) in the interior of a sequence are not supported")

    @staticmethod
    def kv_range(attention_mask, B: int, T: int):
        """attention_mask [B,T] of 0/1 with contiguous ones (right- or left-padded) -> per-sample [lo, hi) or None."""
        if attention_mask is None:
            return None
        m = attention_mask.cpu().bool()
        if bool(m.all()):
            return None
        idx = torch.arange(T)
        lo = torch.where(m.any(1), (~m).long().cumprod(1).sum(1), torch.zeros(B, dtype=torch.long))
        hi = lo + m.sum(1)
        span = (idx[None, :] >= lo[:, None]) & (idx[None, :] < hi[:, None])
        if not torch.equal(span, m):
            raise NotImplementedError("attention_mask must be one contiguous run of ones per sample (right- or left-padded)")
        return lo.to(torch.int32), hi.to(torch.int32)

    def _pinned_image(self, n: int) -> torch.Tensor:
        cls = 1 << max(12, (n - 1).bit_length())
        slots = self._pinned.setdefault(cls, [[torch.empty(cls, dtype=torch.int32, pin_memory=True), None]
                                              for _ in range(self.ring)])
        self._slot = (self._slot + 1) % self.ring
        slot = slots[self._slot]
        if slot[1] is not None:
            slot[1].synchronize()                # the copy that last used this image (ring steps ago) has long finished
        return slot

    DEV_CACHE = 8          # distinct (B, T, shapes, image size) buffer sets kept; least recently used beyond that are dropped

    def _device_buffers(self, key, n_img, M, shapes, want_sort):
        d = self._dev.pop(key, None)           # re-inserted below: dict order = recency
        if d is None:
            # inference batches vary T with the prompt length: without a bound every distinct shape would keep its buffer set
            # (and, with want_sort, its radix-sort workspace) for the life of the process
            while len(self._dev) >= self.DEV_CACHE:
                self._dev.pop(next(iter(self._dev)))
            e = lambda n, dt: torch.empty(max(int(n), 1), dtype=dt, device=self.dev)
            d = dict(img=e(n_img, torch.int32), labels_shifted=e(M, torch.int64), scored=e(M, torch.int32),
                     counts=torch.zeros(2, dtype=torch.int32, device=self.dev), overwritten=e(M, torch.uint8))
            for g, (N, K) in enumerate(shapes):
                d[f"om{g}"] = e(N * K, torch.int64)
                d[f"dst{g}"] = e(N * K, torch.int32)
        self._dev[key] = d
        if want_sort and "order" not in d:
            e = lambda n, dt: torch.empty(max(int(n), 1), dtype=dt, device=self.dev)
            d.update(keys=e(2 * M, torch.int32), vals=e(M, torch.int32), order=e(M, torch.int32), seg=e(M + 1, torch.int32),
                     uid=e(M, torch.int64), ws_bytes=lib().query("molly_batch_sort_workspace", M))
            d["ws"] = e(d["ws_bytes"], torch.uint8)
        return d

    # ---- the staging call --------------------------------------------------------------------------------------
    def stage(self, B: int, T: int, input_ids=None, labels=None, attention_mask=None, omic_ids=None, omic_info_list=None,
              K_cfg: Optional[Dict[str, int]] = None, want_sort: bool = False) -> StagedBatch:
        M = B * T
        table, members = ([], ([], []))
        if omic_ids is not None and omic_info_list is not None:
            table, members = self._spans(omic_ids, omic_info_list, K_cfg, T)
        rows = [self._rows(omic_ids, members[g]) for g in range(2)]
        shapes, ks = [], []
        for g, r in enumerate(rows):
            if r is None:
                shapes.append((0, 0)); ks.append(0)
                continue
            self._check_rows(r, self.enc_vocab[GROUPS[g]])
            shapes.append(tuple(r.shape))
            ks.append(min(int(K_cfg[GROUPS[g]]), r.shape[1]))
        for b, start, g, _ in table:
            if start != -1 and start + 1 + ks[g] > T:
                raise RuntimeError(
                    f"omic span at start={start} (+{ks[g]} tokens) exceeds the sequence length {T} "
                    "(reference fails here too: src/model/omics_one.py:97 after truncation, SURVEY.md §0.4-6)")
        kv = self.kv_range(attention_mask, B, T)
        have_ids, have_lab = input_ids is not None, labels is not None
        # ---- layout of the packed image (int32 words)
        off, o = {}, 0
        for name, n in (("ids", M if have_ids else 0), ("labels", M if have_lab else 0), ("kv", 2 * B if kv is not None else 0),
                        ("spans", 4 * len(table)), ("om0", shapes[0][0] * shapes[0][1]), ("om1", shapes[1][0] * shapes[1][1])):
            off[name] = (o, n)
            o += (n + 3) // 4 * 4                 # 16-byte aligned sections
        n_img = max(o, 4)
        slot = self._pinned_image(n_img)
        img = slot[0]
        sec = lambda name: img[off[name][0]:off[name][0] + off[name][1]]
        n_scored = 0
        if have_ids:
            sec("ids").copy_(input_ids.reshape(-1))              # int64 -> int32 on the way into the pinned image
        if have_lab:
            lab = labels.cpu() if labels.is_cuda else labels
            sec("labels").copy_(lab.reshape(-1))
            n_scored = int((lab[:, 1:] != -100).sum())           # the COUNT sizes the head GEMMs; the list is built on the device
        if kv is not None:
            sec("kv")[:B].copy_(kv[0]); sec("kv")[B:].copy_(kv[1])
        if table:
            sec("spans").copy_(torch.tensor(table, dtype=torch.int32).reshape(-1))
        for g, r in enumerate(rows):
            if r is not None:
                sec(f"om{g}").copy_(r.reshape(-1))
        key = (B, T, shapes[0], shapes[1], n_img)
        d = self._device_buffers(key, n_img, M, shapes, want_sort)
        d["img"][:n_img].copy_(img[:n_img], non_blocking=True)   # THE host->device copy of the step
        ev = torch.cuda.Event()
        ev.record()
        slot[1] = ev
        dsec = lambda name: d["img"][off[name][0]:off[name][0] + off[name][1]] if off[name][1] else None
        sort = want_sort and have_ids
        lib().call("molly_batch_assemble", ops._stream(), dsec("ids"), dsec("labels"), B, T, self.V, -100, dsec("spans"),
                   len(table), dsec("om0"), shapes[0][0], shapes[0][1], ks[0], dsec("om1"), shapes[1][0], shapes[1][1], ks[1],
                   d["labels_shifted"] if have_lab else None, d["scored"] if have_lab else None,
                   d["counts"][0:1] if have_lab else None,
                   d["om0"] if shapes[0][0] else None, d["dst0"] if shapes[0][0] else None,
                   d["om1"] if shapes[1][0] else None, d["dst1"] if shapes[1][0] else None, d["overwritten"],
                   d["keys"] if sort else None, d["vals"] if sort else None, d["order"] if sort else None,
                   d["seg"] if sort else None, d["uid"] if sort else None, d["counts"][1:2] if sort else None,
                   d["ws"] if sort else None, d["ws_bytes"] if sort else 0)
        s = StagedBatch()
        s.B, s.T = B, T
        s.ids32 = dsec("ids")
        s.labels_shifted = d["labels_shifted"][:M] if have_lab else None
        s.n_scored = n_scored
        s.scored_rows = d["scored"][:n_scored] if have_lab else None
        s.kv_lo, s.kv_hi = (dsec("kv")[:B], dsec("kv")[B:]) if kv is not None else (None, None)
        s.groups = {}
        for g, (N, K) in enumerate(shapes):
            if N:
                s.groups[GROUPS[g]] = (d[f"om{g}"][:N * K].view(N, K), d[f"dst{g}"][:N * K], N, K)
        s.overwritten = d["overwritten"][:M]
        s.n_overwritten = sum(ks[g] for _, start, g, _ in table if start != -1)
        s.emb_index = (d["order"], d["seg"], d["uid"], d["counts"][1:2], M) if sort else None
        return s
