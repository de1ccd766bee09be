#!/bin/bash
export TMPDIR=/tmp
mkdir -p gpurun_out
timeout 600 python profiles/microbench/xcd_own.py > gpurun_out/xcd_own_sersic.txt 2>&1; echo "rc $?" >> gpurun_out/xcd_own_sersic.txt
timeout 400 python profiles/microbench/xcd_own.py --source uniform > gpurun_out/xcd_own_uniform.txt 2>&1; echo "rc $?" >> gpurun_out/xcd_own_uniform.txt
tools/fetch_calibration.sh > /dev/null 2>&1
timeout 600 python bench.py > gpurun_out/bench_base.json 2> gpurun_out/bench_base.err
tail -c 1500 gpurun_out/xcd_own_sersic.txt
