#!/bin/bash
export TMPDIR=/tmp
mkdir -p gpurun_out
PMC_GEN_DUMP=1 timeout 300 python bench.py --steps 1 --warmup 1 --no-cpu-baseline --no-secondary --no-breakdown --no-counters > gpurun_out/gen_dump.json 2> gpurun_out/gen_dump.log
timeout 1200 python -m pytest tests -m gpu -x -q > gpurun_out/pytest_gpu.log 2>&1; echo "rc $?" >> gpurun_out/pytest_gpu.log
tail -4 gpurun_out/pytest_gpu.log
