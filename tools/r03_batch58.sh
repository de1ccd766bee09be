#!/bin/bash
# round 3, batch 58 (GPU box): round thresholds and claim size of the walk kernels once more, on the final build
export TMPDIR=/tmp
R=$PWD
O=$R/gpurun_out/r03_batch58; mkdir -p $O
python tools/sweep.py --packets 1e8 default default libpmc_pr32.so libpmc_pr48.so libpmc_qr4.so libpmc_qr16.so libpmc_ch128.so libpmc_ch512.so default libpmc_pr32.so libpmc_pr48.so libpmc_qr4.so libpmc_qr16.so libpmc_ch128.so libpmc_ch512.so default > $O/sweep.txt 2>&1; grep "pkt/s" $O/sweep.txt
