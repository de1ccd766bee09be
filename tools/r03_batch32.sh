#!/bin/bash
# round 3, batch 32 (GPU box): slots per claim of a wave, steps between round checks, registers of the cycle start kernel
export TMPDIR=/tmp
R=$PWD
O=$R/gpurun_out/r03_batch32; mkdir -p $O
python tools/sweep.py --packets 1e8 default default libpmc_tc256.so libpmc_tc64.so libpmc_ws8.so libpmc_ws8tc256.so libpmc_cw2.so default libpmc_tc256.so libpmc_ws8.so libpmc_ws8tc256.so default > $O/sweep.txt 2>&1
grep -v "amdgpu.ids" $O/sweep.txt | grep "pkt/s" | cut -c1-180
