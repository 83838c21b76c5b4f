#!/bin/bash
export TMPDIR=/tmp
cd /root/repo
timeout 900 python -m pytest tests/test_gpu_parity.py -m gpu -q -x -k "independent_of_how" 2>&1 | tail -15 | cut -c1-220
