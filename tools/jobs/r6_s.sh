#!/bin/bash
export TMPDIR=/tmp
R=/root/repo; O=$R/gpurun_out/r6_s; rm -rf $O; mkdir -p $O; cd $R
python bench.py --no-cpu-baseline --no-other-configs > $O/bench.json 2> $O/bench.err
python - <<'PY'
import json
b=json.loads([l for l in open("/root/repo/gpurun_out/r6_s/bench.json") if l.startswith("{")][-1])
print("value", b["value"], b["phase_ms_per_step"])
print("64:", {k:v for k,v in b["fit_GP_MAP_15_starts_64_emulators"].items() if k in ("fit_GP_MAP_s","fit_GP_MAP_first_call_s","fit_GP_MAP_TFLOPs")})
for e in b["shard_sweep"]:
    print(e["emulators"], e["n"], {k:v for k,v in e.items() if k in ("fit_GP_MAP_s","fit_GP_MAP_first_call_s","fit_GP_MAP_TFLOPs")})
PY
