#!/bin/bash
export TMPDIR=/tmp
cd /tmp
rm -rf /root/repo/gpurun_out/r2h_prof
env $CFG BS=${PBS:-8,64} REPS=5 timeout 300 rocprofv3 --kernel-trace --stats -d /root/repo/gpurun_out/r2h_prof -- python /root/repo/tools/shard_sweep.py > /root/repo/gpurun_out/r2h_prof.log 2>&1
cd /root/repo
python - <<'PY'
import sqlite3, glob
f = glob.glob('gpurun_out/r2h_prof/**/*.db', recursive=True)[0]
con = sqlite3.connect(f)
for r in con.execute("select name, grid_x, count(*), avg(end-start)/1e3, min(end-start)/1e3 from kernels where name like '%backsolve%' or name like '%logdet%' or name like '%cov_build%' group by name, grid_x order by name, grid_x"):
    print(r[0][:40], r[1:])
PY
