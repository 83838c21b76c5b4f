#!/bin/bash
export TMPDIR=/tmp
python bench.py --steps 10 --warmup 2 > gpurun_out/r2g_bench.json 2> gpurun_out/r2g_bench.err
tail -3 gpurun_out/r2g_bench.err
python - <<'PY'
import json
d = json.loads(open("gpurun_out/r2g_bench.json").read().strip().splitlines()[-1])
for k in ("value", "fit_grad_per_s", "predict_pts_per_s", "phase_ms_per_step", "roofline", "fit_roofline", "parity_in_bench", "shard_sweep"):
    print(k, json.dumps(d.get(k))[:900])
for k, v in d["kernels"].items():
    print("  %-14s %6d launches %9.2f ms total  %8.3f ms avg  %8.1f %s" % (k, v["launches"], v["ms_total"], v["avg_ms"], v["achieved"], v["unit"]))
print("cpu_baseline", json.dumps(d.get("cpu_baseline"))[:600])
PY
MOGP_BENCH_BACKEND=gloo python -m torch.distributed.run --nnodes=1 --nproc-per-node 2 --master-addr 127.0.0.1 --master-port 29511 bench.py --gpus 2 --steps 3 --warmup 1 > gpurun_out/r2g_bench2.json 2> gpurun_out/r2g_bench2.err
tail -2 gpurun_out/r2g_bench2.err; tail -1 gpurun_out/r2g_bench2.json | cut -c1-700
