#!/bin/bash
# round 2, experiment 6: VALU issue rate at 1-8 waves per SIMD; SQ / TCP / TCC counters of the split walk kernels
export TMPDIR=/tmp
O=gpurun_out/r02_exp6; rm -rf $O; mkdir -p $O
timeout 120 ./profiles/microbench/valu_rate2 > $O/valu_rate2.txt 2>&1; cat $O/valu_rate2.txt
PMC_NUM_GROUPS=1 PMC_SERIAL_WALKS=1 bash tools/pmc_passes.sh > $O/pmc.log 2>&1
cp gpurun_out/pmc/summary.txt $O/pmc_summary.txt; cat $O/pmc_summary.txt
