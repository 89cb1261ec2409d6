#!/usr/bin/env python3
"""diagnostic: why does hipcub report hipErrorNoDevice inside torchrun children of __graft_entry__ smoke?"""
import os, sys, ctypes
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
order = os.environ.get("DIAG_ORDER", "lib_first")
rank = int(os.environ.get("RANK", "0"))
def maps():
    return sorted({l.split()[-1] for l in open("/proc/self/maps") if "amdhip" in l or "hsa-runtime" in l})
if order == "lib_first":
    from molly_amd import _lib
    L = _lib.lib()
import torch
import torch.distributed as dist
torch.cuda.set_device(0)
if order != "lib_first":
    from molly_amd import _lib
    L = _lib.lib()
if rank == 0:
    print("order", order, "maps", maps(), flush=True)
if "WORLD_SIZE" in os.environ and int(os.environ["WORLD_SIZE"]) > 1:
    dist.init_process_group("gloo")
x = torch.zeros(4, device="cuda")
print(rank, "ws bytes", L.query("molly_batch_sort_workspace", 512), L.last_error(), flush=True)
from molly_amd.batch import BatchStager
from molly_amd.synth import synth_batch
st = BatchStager(torch.device("cuda", 0), 1024, {"dna_rna": 4105, "protein": 33})
b = synth_batch(2, 256, [("protein", 64)], seed=1, text_vocab=1000, special_ids={"dna": (1010, 1011, 1012), "rna": (1013, 1014, 1015), "protein": (1016, 1017, 1018)}, pad_id=1000)
for attempt in range(2):
    try:
        s = st.stage(2, 256, b["input_ids"], b["labels"], b["attention_mask"], b["omic_ids"], b["omic_info_list"], {"dna_rna": 64, "protein": 64}, want_sort=True)
        torch.cuda.synchronize()
        print(rank, "attempt", attempt, "stage ok, n_unique", int(s.emb_index[3].item()), flush=True)
    except Exception as e:
        print(rank, "attempt", attempt, "stage FAILED:", str(e)[:200], flush=True)
if dist.is_initialized():
    dist.barrier(); dist.destroy_process_group()
