#!/bin/bash
# round 5: the bench line of the current tree (all extras).  usage: TAG=x bash tools/jobs/r5_bench.sh
export TMPDIR=/tmp
cd /root/repo
O=gpurun_out/r5bench_${TAG:-head}; rm -rf $O; mkdir -p $O
cp mogp_emulator_amd/libmogp_hip.build $O/build_commit.txt
python bench.py > $O/bench.json 2> $O/bench.err; echo "rc=$?"
tail -5 $O/bench.err
python - <<PY
import json
l=[x for x in open('$O/bench.json').read().splitlines() if x.startswith('{')][-1]; d=json.loads(l)
print({k:d[k] for k in ('value','ms_per_step')}, d['phase_ms_per_step'], d['roofline']['frac'], d['roofline']['traffic_source'][:40])
print([(s['emulators'],s['n'],round(s['fit_ms'],3),round(s['fit_grad_ms'],3),round(s['predict_ms'],3)) for s in d['shard_sweep']])
print(d.get('projected_scaling'))
for c in d['other_configs']:
    print(c['config'], round(c['fit_ms'],3), round(c['fit_grad_ms'],3), round(c['predict_ms'],3), c.get('cpu_baseline'), c.get('parity'))
print(d['predict_first_call_ms'], d['predict_steady_ms'])
print(d['parity_in_bench']['passed'], {k:(round(v['avg_ms'],3),round(v['achieved'],1)) for k,v in d['kernels'].items()})
print(d['cpu_baseline']['value'], d['cpu_baseline']['cores'])
PY
