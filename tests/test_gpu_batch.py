"""GPU: batch assembly on the device (molly_amd/batch.py + csrc/batch.hip) — everything the step derives
from the batch, bit-exact (integer work) against a plain numpy restatement of the reference's semantics
(src/model/omics_one.py:93-118 scatter positions, HF:loss/loss_utils.py:60-63 label shift) on seeded ragged batches."""
import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu


def _expected(b, T, K_cfg, V):
    ids = b["input_ids"].numpy()
    B = ids.shape[0]
    M = B * T
    shifted = np.full((B, T), -100, np.int64)
    shifted[:, :-1] = b["labels"].numpy()[:, 1:]
    shifted = shifted.reshape(-1)
    scored = np.nonzero(shifted != -100)[0].astype(np.int32)
    groups = {"dna_rna": [], "protein": []}
    for bi, infos in enumerate(b["omic_info_list"]):
        for j, info in enumerate(infos):
            if info["type"] == "pad":
                continue
            groups["protein" if info["type"] == "protein" else "dna_rna"].append((bi, j, info["start"]))
    overwritten = np.zeros(M, bool)
    out = {}
    for name, mem in groups.items():
        if not mem:
            continue
        rows = np.stack([np.asarray(b["omic_ids"][bi][j]) for bi, j, _ in mem])
        N, K = rows.shape
        k = min(K_cfg[name], K)
        dst = np.full((N, K), -1, np.int32)
        for i, (bi, _, start) in enumerate(mem):
            if start == -1:
                continue
            dst[i, :k] = bi * T + start + 1 + np.arange(k)
            overwritten[dst[i, :k]] = True
        out[name] = (rows.astype(np.int64), dst.reshape(-1))
    rows = np.nonzero(~overwritten)[0].astype(np.int32)
    flat = ids.reshape(-1)
    order = rows[np.argsort(flat[rows], kind="stable")]
    sk = flat[order]
    bounds = np.nonzero(np.diff(sk))[0] + 1
    seg = np.concatenate([[0], bounds, [len(order)]]).astype(np.int32)
    uid = sk[seg[:-1]].astype(np.int64)
    return shifted, scored, out, overwritten, order, seg, uid


@pytest.mark.parametrize("B,T,spans,ragged,seed", [
    (3, 416, [("protein", 64), ("rna", 64)], True, 11),
    (8, 2048, [("protein", 512)], False, 42),                 # the headline batch
    (2, 512, [("dna", 64), ("rna", 64), ("protein", 64)], True, 5),
    (1, 300, [], True, 9),                                    # text only
])
def test_device_assembly_is_bit_exact(B, T, spans, ragged, seed):
    from molly_amd.batch import BatchStager
    from molly_amd.synth import synth_batch
    V = 151936
    b = synth_batch(B, T, spans, seed=seed, ragged=ragged)
    if seed == 11:
        b["input_ids"][0, 100:140] = 77                        # a long segment of one token id: stable order inside it matters
    K_cfg = {"dna_rna": 48 if seed == 5 else 4096, "protein": 4096}   # k = min(config, K): 48 < 64 leaves a tail unwritten
    st = BatchStager(torch.device("cuda"), V, {"dna_rna": 4105, "protein": 33})
    for _ in range(2):                                         # second call reuses the cached device buffers and another pinned image
        s = st.stage(B, T, b["input_ids"], b["labels"], b["attention_mask"], b["omic_ids"] if spans else None,
                     b["omic_info_list"] if spans else None, K_cfg, want_sort=True)
        torch.cuda.synchronize()
        shifted, scored, groups, overwritten, order, seg, uid = _expected(b, T, K_cfg, V)
        assert np.array_equal(s.ids32.cpu().numpy(), b["input_ids"].numpy().reshape(-1))
        assert np.array_equal(s.labels_shifted.cpu().numpy(), shifted)
        assert s.n_scored == len(scored) and np.array_equal(s.scored_rows.cpu().numpy(), scored)
        assert set(s.groups) == set(groups)
        for name, (rows, dst) in groups.items():
            ids64, dst_dev, N, K = s.groups[name]
            assert ids64.dtype == torch.int64 and np.array_equal(ids64.cpu().numpy(), rows)
            assert np.array_equal(dst_dev.cpu().numpy(), dst)
        assert np.array_equal(s.overwritten.cpu().numpy().astype(bool), overwritten)
        assert s.n_overwritten == int(overwritten.sum())
        o_dev, seg_dev, uid_dev, n_dev, bound = s.emb_index
        n = int(n_dev.item())
        assert n == len(uid) and bound >= n
        assert np.array_equal(uid_dev[:n].cpu().numpy(), uid)
        assert np.array_equal(seg_dev[:n + 1].cpu().numpy(), seg)
        assert np.array_equal(o_dev[:len(order)].cpu().numpy(), order)
        if ragged:
            m = b["attention_mask"].numpy().astype(bool)
            assert np.array_equal(s.kv_lo.cpu().numpy(), np.zeros(B, np.int32))
            assert np.array_equal(s.kv_hi.cpu().numpy(), m.sum(1).astype(np.int32))
        else:
            assert s.kv_lo is None and s.kv_hi is None


def test_reference_error_behaviour_on_the_host_pass():
    from molly_amd.batch import BatchStager
    st = BatchStager(torch.device("cuda"), 1000, {"dna_rna": 4105, "protein": 33})
    ids, lab = torch.zeros(1, 64, dtype=torch.int64), torch.full((1, 64), -100)
    row = torch.full((1, 1, 8), 5, dtype=torch.int64)
    K = {"dna_rna": 8, "protein": 8}
    with pytest.raises(ValueError, match="Unsupported omic type"):                       # reference :118
        st.stage(1, 64, ids, lab, None, row, [[{"type": "lipid", "start": 3}]], K)
    with pytest.raises(AssertionError, match="Mismatch"):                                # reference :168-170
        st.stage(1, 64, ids, lab, None, row, [[{"type": "dna", "start": 3}, {"type": "pad", "start": -1}]], K)
    with pytest.raises(AssertionError, match="out-of-range"):                            # reference :71-72
        st.stage(1, 64, ids, lab, None, torch.full((1, 1, 8), 40, dtype=torch.int64), [[{"type": "protein", "start": 3}]], K)
    with pytest.raises(RuntimeError, match="exceeds the sequence length"):
        st.stage(1, 64, ids, lab, None, row, [[{"type": "dna", "start": 60}]], K)
    bad = row.clone(); bad[0, 0, 3] = 1
    with pytest.raises(NotImplementedError, match="interior"):
        st.stage(1, 64, ids, lab, None, bad, [[{"type": "dna", "start": 3}]], K)
    # left-padded prompt (inference collate): key range starts where the mask does
    mask = torch.ones(1, 64, dtype=torch.int64); mask[0, :10] = 0
    s = st.stage(1, 64, ids, None, mask, None, None, K)
    assert int(s.kv_lo[0]) == 10 and int(s.kv_hi[0]) == 64 and s.labels_shifted is None and s.emb_index is None
