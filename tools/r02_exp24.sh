#!/bin/bash
# round 2, experiment 24: upper bound of what L2 locality can buy -- the same kernels on a grid whose tables fit in every L2
export TMPDIR=/tmp
O=gpurun_out/r02_exp24; rm -rf $O; mkdir -p $O
E=PMC_NUM_GROUPS=1,PMC_SERIAL_WALKS=1,PMC_TIMING_DUMP=1,PMC_PROFILE_DUMP=1
timeout 600 python tools/sweep.py --packets 2e7 --ski tests/ski/cfg2small.ski libpmc_census.so,$E > $O/small.txt 2> $O/small.err
timeout 600 python tools/sweep.py --packets 2e7 libpmc_census.so,$E > $O/full.txt 2> $O/full.err
for f in small full; do echo "== $f"; cut -c1-150 $O/$f.txt; grep "PMC_TIMING" $O/$f.err | tail -1; grep "PMC_PROFILE peel:\|PMC_PROFILE prop:" $O/$f.err; done
