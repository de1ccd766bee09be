#!/bin/bash
# round 2, experiment 16: service thresholds and workgroups per CU of the pipelined walk kernels, two slot groups (the bench's mode)
export TMPDIR=/tmp
O=gpurun_out/r02_exp16; rm -rf $O; mkdir -p $O
E=PMC_PEEL_BLOCKS_PER_CU=1
timeout 1200 python tools/sweep.py --packets 5e7 default,$E libpmc_p24_32.so,$E libpmc_p32_40.so,$E libpmc_p16_32.so,$E libpmc_p16_40.so,$E libpmc_p24_48.so,$E \
  libpmc_pb256.so,PMC_PEEL_BLOCKS_PER_CU=2 libpmc_pb256.so,PMC_PEEL_BLOCKS_PER_CU=1 libpmc_p24_32.so,$E,PMC_WALK_BLOCKS_PER_CU=2 libpmc_p24_32.so,$E,PMC_NUM_GROUPS=3 > $O/sweep.txt 2> $O/sweep.err
cut -c1-175 $O/sweep.txt
