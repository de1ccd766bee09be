#!/bin/bash
export TMPDIR=/tmp
mkdir -p gpurun_out
timeout 240 python profiles/microbench/xcd_own.py > gpurun_out/xcd_own_sersic.txt 2>&1; echo "rc $?" >> gpurun_out/xcd_own_sersic.txt
timeout 150 python profiles/microbench/xcd_own.py --source uniform > gpurun_out/xcd_own_uniform.txt 2>&1; echo "rc $?" >> gpurun_out/xcd_own_uniform.txt
tools/fetch_calibration.sh > /dev/null 2>&1
timeout 900 python -m pytest tests/test_gpu_two_ranks_one_device.py tests/test_gpu_multi.py -m gpu -x -q > gpurun_out/pytest_multi.log 2>&1; echo "rc $?" >> gpurun_out/pytest_multi.log
timeout 900 python bench.py > gpurun_out/bench_counters.json 2> gpurun_out/bench_counters.err
tail -c 2500 gpurun_out/xcd_own_sersic.txt; tail -5 gpurun_out/pytest_multi.log
