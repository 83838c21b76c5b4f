#!/bin/bash
# probe of the diagonal-block factorisation (tools/chol128_probe.hip): builds given in $BINS, stamp-free kernel times + checks
for b in ${BINS:-tools/c128_g4.bin tools/c128_r1.bin}; do
  echo "== $b"
  timeout 60 $b 1 8.0 128 | head -${LINES:-2}
  timeout 60 $b 64 8.0 128 | head -1
  timeout 60 $b 64 1e-10 64 | head -2
done
