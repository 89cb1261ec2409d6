#!/bin/bash
# what clock and power does the chip run the step's dominant kernel at?  rocm-smi polled while a GEMM loop runs (read-only queries)
mkdir -p gpurun_out/r05
L=gpurun_out/r05/clock_probe_${PROBE:-molly}.log
: > $L
python3 - <<'PY' &
import torch, time, sys, os
sys.path.insert(0, os.getcwd())
from molly_amd import ops
a = (torch.rand(32768, 2048, device="cuda") - 0.5).bfloat16(); b = (torch.rand(12288, 2048, device="cuda") - 0.5).bfloat16()
out = torch.empty(32768, 12288, dtype=torch.bfloat16, device="cuda")
mode = os.environ.get("PROBE", "molly")
q = (torch.rand(32768, 2048, device="cuda") - 0.5).bfloat16(); k = (torch.rand(32768, 1024, device="cuda") - 0.5).bfloat16()
v = (torch.rand(32768, 1024, device="cuda") - 0.5).bfloat16(); o = torch.empty_like(q)
bt = b.t()
n, t0 = 0, time.time()
while time.time() - t0 < 14:
    for _ in range(50):
        if mode == "molly": ops.gemm_nt(a, b, out=out)
        elif mode == "torch": torch.matmul(a, bt, out=out)
        elif mode == "attn": ops.attn_fwd(q, k, v, 16, 2048, 16, 8, 128, 128 ** -0.5, True, out=o)
    n += 50
    torch.cuda.synchronize()
dt = time.time() - t0
fl = 2.0 * 32768 * 12288 * 2048 if mode != "attn" else 4.0 * 16 * 16 * 2048 * 2048 * 128 / 2
print(f"PROBE {mode}: {n} launches in {dt:.2f} s = {fl * n / dt / 1e12:.0f} TFLOP/s sustained", flush=True)
PY
PID=$!
sleep 4
for i in 1 2 3 4; do
  echo "--- sample $i (GEMM loop running)" >> $L
  rocm-smi --showclocks --showpower --showtemp 2>&1 | grep -i "sclk\|mclk\|power\|junction\|fclk" | head -8 >> $L
  sleep 1
done
wait $PID
sleep 3
echo "--- idle" >> $L
rocm-smi --showclocks --showpower 2>&1 | grep -i "sclk\|mclk\|power" | head -6 >> $L
cat $L
