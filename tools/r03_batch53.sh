#!/bin/bash
# round 3, batch 53 (GPU box): would the statistics flush cost less as a kernel of its own next to the walk kernels?  (tuning experiment: the launch kernel
# without its flush + a kernel that issues as many atomics of the same shape on a side stream; wrong statistics)
export TMPDIR=/tmp
R=$PWD
O=$R/gpurun_out/r03_batch53; mkdir -p $O
python tools/sweep.py --packets 1e8 default default libpmc_noflush.so libpmc_dummy.so libpmc_dummy.so,PMC_DUMMY_FLUSH_BLOCKS=64 libpmc_dummy.so,PMC_DUMMY_FLUSH_BLOCKS=1024 libpmc_dummy.so,PMC_DUMMY_FLUSH_INLINE=1 default libpmc_noflush.so libpmc_dummy.so > $O/sweep.txt 2>&1; grep "pkt/s" $O/sweep.txt
