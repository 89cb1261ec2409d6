#!/bin/bash
# the round's driver-style check: GPU test suite, smoke, default bench line
mkdir -p gpurun_out/r05
T=${1:-t1}
timeout 1500 python -m pytest tests -x -q -m gpu > gpurun_out/r05/${T}_pytest.log 2>&1; echo "pytest rc=$?" >> gpurun_out/r05/${T}_pytest.log
tail -4 gpurun_out/r05/${T}_pytest.log
timeout 300 python -c "import __graft_entry__ as g; g.smoke()" 2>&1 | tail -2
timeout 1200 python bench.py > gpurun_out/r05/${T}_bench.json 2> gpurun_out/r05/${T}_bench.err; echo "bench rc=$?"
python3 - <<PY
import json
d = json.loads(open("gpurun_out/r05/${T}_bench.json").read().strip().splitlines()[-1])
print("value", d["value"], "ms", d["ms_per_step"], "p50", d["step_ms_p50"], "exec frac", d["mfma_roofline_frac_step_executed"], "gemm", d["roofline"]["achieved"], d["roofline"]["frac"])
print("batch8", d.get("batch8_reference"))
s = d.get("secondary", {})
for k, v in s.items():
    if isinstance(v, dict):
        if k == "c5":
            print(k, v["prefill"]["tokens_per_s"], v["decode"]["ms_per_step"], v["decode"]["frac"], v.get("decode_sampled", {}).get("ms_per_step"))
        else:
            print(k, v["ms_per_step"], v["tokens_per_s"], v["mfma_roofline_frac_step_executed"], v["gemm"]["frac"])
    else:
        print(k, v)
print("vendor", d["roofline"].get("vendor_gemm_tflops"))
print("cpu", d.get("cpu_baseline"))
PY
