#!/bin/bash
# round 6: placement of the predictive variance's tiles (super-tile shapes) with time, clock, power and L2-miss traffic side by side
export TMPDIR=/tmp
cd /root/repo; O=/root/repo/gpurun_out/r6_y; rm -rf $O; mkdir -p $O
{ for q in 0 1; do for l in 0 1 2 3 4 5 6; do MOGP_PV_Q=$q MOGP_PV_LGC=$l timeout 120 python tools/pv_placement.py 3; done; done; } 2>&1 | grep -v amdgpu > $O/placement.txt; cat $O/placement.txt
for l in 0 2 3 4 6; do bash tools/pmc_fetch.sh $O/traffic_lgc$l.txt MOGP_PV_LGC=$l > /dev/null 2>&1; echo "== MOGP_PV_LGC=$l"; grep "predict_var" $O/traffic_lgc$l.txt | cut -c1-200; done | tee $O/traffic.txt
