#!/bin/bash
export TMPDIR=/tmp
mkdir -p gpurun_out
timeout 240 python profiles/microbench/xcd_own.py > gpurun_out/xcd_own_sersic.txt 2>&1; echo "rc $?" >> gpurun_out/xcd_own_sersic.txt

cat gpurun_out/xcd_own_sersic.txt
