#!/bin/bash
# round 3, run A: full GPU suite (with durations), default bench, kernel stats of C4 and C5
export TMPDIR=/tmp
R=/root/repo
cd $R
timeout 1700 python -m pytest tests -m gpu -x -q --durations=25 2>&1 | tail -45 > gpurun_out/r3a_tests.txt
tail -8 gpurun_out/r3a_tests.txt
( time python bench.py ) > gpurun_out/r3a_bench.json 2> gpurun_out/r3a_bench.err
tail -5 gpurun_out/r3a_bench.err
python - <<'PY'
import json
try:
    d = json.loads(open("gpurun_out/r3a_bench.json").read().strip().splitlines()[-1])
    for k in ("value", "fit_grad_per_s", "predict_pts_per_s", "phase_ms_per_step", "roofline", "fit_roofline", "parity_in_bench", "nccl_world1", "cpu_baseline", "fit_GP_MAP_15_starts_64_emulators"):
        print(k, json.dumps(d.get(k))[:1200])
    for e in d.get("shard_sweep", []):
        print("shard", json.dumps(e)[:1500])
    for e in d.get("other_configs", []):
        print("other", json.dumps({k: v for k, v in e.items() if k != "kernels_fit"})[:900])
        for k, v in e["kernels_fit"].items():
            print("    %-14s %6d launches %9.2f ms/fit %8.1f %s" % (k, v["launches"], v["ms_per_fit"], v["achieved"], v["unit"]))
except Exception as exc:
    print("bench parse failed", exc)
PY
cd /tmp
for C in C4 C5; do
  rm -rf $R/gpurun_out/r3a_prof_$C
  ONLY=$C timeout 600 rocprofv3 --kernel-trace --stats -d $R/gpurun_out/r3a_prof_$C -- python $R/tools/big_configs.py > $R/gpurun_out/r3a_prof_$C.log 2>&1
  python $R/tools/prof_summary.py $(find $R/gpurun_out/r3a_prof_$C -name "*.db" | head -1) "ONLY=$C rocprofv3 --kernel-trace --stats -- python tools/big_configs.py" > $R/gpurun_out/r3a_${C}_kernel_stats.txt 2>&1
  head -16 $R/gpurun_out/r3a_${C}_kernel_stats.txt | cut -c1-200
  tail -6 $R/gpurun_out/r3a_prof_$C.log
done
