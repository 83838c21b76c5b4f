#!/bin/bash
cd /root/repo; mkdir -p gpurun_out/k; export TMPDIR=/tmp
timeout 900 python -m pytest tests/test_gpu_parity.py -x -q -m gpu -k "starts or fit_GP_MAP or map" > gpurun_out/k/tests.txt 2>&1
tail -n 5 gpurun_out/k/tests.txt
timeout 900 python bench.py --no-cpu-baseline > gpurun_out/k/bench.json 2> gpurun_out/k/bench.err
python - <<'PY'
import json
d=json.loads(open("gpurun_out/k/bench.json").read().strip().splitlines()[-1])
print(d["value"], d["fit_grad_per_s"], d["predict_pts_per_s"])
print(json.dumps(d["fit_GP_MAP_15_starts_64_emulators"]))
for e in d["shard_sweep"]: print(e["emulators"], e["n"], e.get("fit_GP_MAP_s"), e.get("fit_GP_MAP_emulator_fits_per_s"), e.get("fit_GP_MAP_TFLOPs"))
print(d.get("tsunami_benchmark"))
PY
