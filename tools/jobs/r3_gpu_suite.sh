#!/bin/bash
# the GPU suite + smoke() on the current tree; usage: TAG=<commit> bash tools/jobs/r3_gpu_suite.sh
export TMPDIR=/tmp
cd /root/repo
O=gpurun_out/suite_${TAG:-head}; rm -rf $O; mkdir -p $O
timeout 2400 python -m pytest tests -m gpu -x -q --durations=10 2>&1 | tail -22 > $O/gpu_tests.txt; tail -3 $O/gpu_tests.txt
python -c "import __graft_entry__ as g; g.smoke()" 2>&1 | tail -2 | tee $O/smoke.txt
python bench.py > $O/bench.json 2> $O/bench.err; python - <<PY
import json
l=[x for x in open('$O/bench.json').read().splitlines() if x.startswith('{')][-1]; d=json.loads(l)
print({k:d[k] for k in ('value','ms_per_step')}, d['phase_ms_per_step'], d['roofline']['frac'], [ (s['emulators'],round(s['fit_ms'],3)) for s in d['shard_sweep']], d['parity_in_bench'])
PY
