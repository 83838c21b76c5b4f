#!/bin/bash
# round 4: the GPU suite + smoke() + the bench line on the current tree; usage: TAG=<commit> bash tools/jobs/r4_suite.sh
export TMPDIR=/tmp
cd /root/repo
O=gpurun_out/suite_${TAG:-head}; rm -rf $O; mkdir -p $O
timeout 2700 python -m pytest tests -m gpu -x -q --durations=12 2>&1 | tail -26 > $O/gpu_tests.txt; tail -4 $O/gpu_tests.txt
python -c "import __graft_entry__ as g; g.smoke()" 2>&1 | tail -2 | tee $O/smoke.txt
python bench.py > $O/bench.json 2> $O/bench.err; python - <<PY
import json
l=[x for x in open('$O/bench.json').read().splitlines() if x.startswith('{')][-1]; d=json.loads(l)
print({k:d[k] for k in ('value','ms_per_step')}, d['phase_ms_per_step'], d['roofline']['frac'])
print([(s['emulators'],s['n'],round(s['fit_ms'],3),round(s['fit_grad_ms'],3)) for s in d['shard_sweep']])
print([(c['config'],round(c['fit_ms'],3),round(c['fit_grad_ms'],3),round(c['predict_ms'],3)) for c in d['other_configs']])
print(d['parity_in_bench']['passed'], {k:(round(v['avg_ms'],3),round(v['achieved'],1)) for k,v in d['kernels'].items()})
PY
tail -5 $O/bench.err
