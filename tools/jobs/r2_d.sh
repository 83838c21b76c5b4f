#!/bin/bash
export TMPDIR=/tmp
for cfg in "MOGP_CHOL=right MOGP_OUTER=128" "MOGP_CHOL=right MOGP_OUTER=256" "MOGP_CHOL=right MOGP_OUTER=512" "MOGP_CHOL=left" "MOGP_CHOL=leftla"; do
  echo "== $cfg"
  env $cfg python tools/shard_sweep.py 2>&1 | python -c "
import sys, json
for l in sys.stdin:
    try: d = json.loads(l)
    except Exception: print(l.strip()[:200]); continue
    print('B=%d fit %.3f ms  fit+grad %.3f ms' % (d['B'], d['fit_ms'], d['fit_grad_ms']))
"
done
