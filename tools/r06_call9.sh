#!/bin/bash
export TMPDIR=/tmp
mkdir -p gpurun_out
timeout 1500 python -m pytest tests/test_gpu_multi.py -m gpu -x -q > gpurun_out/pytest_multi.log 2>&1; echo "rc $?" >> gpurun_out/pytest_multi.log
tail -25 gpurun_out/pytest_multi.log
