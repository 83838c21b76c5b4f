#!/bin/bash
# kernel timeline of ONE objective-only evaluation of the 8-emulator shard (8 x n=2000): where the 0.2 ms outside the Cholesky go
export TMPDIR=/tmp
R=/root/repo
O=$R/gpurun_out/r3tl; rm -rf $O; mkdir -p $O
sed 's/n, d, B = 2000, 10, 64/n, d, B = 2000, 10, int(os.environ.get("B", "64"))/' $R/tools/fit_only.py > $R/tools/fit_only_b.py
cd /tmp
B=8 timeout 300 rocprofv3 --kernel-trace -d $O/tl -- python $R/tools/fit_only_b.py > $O/run.log 2>&1
cd $R; rm -f tools/fit_only_b.py
grep "ms per" $O/run.log
python tools/timeline.py $(find $O/tl -name "*.db" | head -1) 12 40 | tee $O/timeline_B8.txt
rm -rf $O/tl
