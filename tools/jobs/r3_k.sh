#!/bin/bash
cd /root/repo; mkdir -p gpurun_out/k; export TMPDIR=/tmp
timeout 1500 python -m pytest tests -x -q -m gpu > gpurun_out/k/tests.txt 2>&1
tail -n 5 gpurun_out/k/tests.txt
