#!/bin/bash
export TMPDIR=/tmp
O=gpurun_out/r02_exp4; rm -rf $O; mkdir -p $O
timeout 900 python tools/sweep.py libpmc_prof.so,PMC_NUM_GROUPS=1,PMC_PROFILE_DUMP=1 default,PMC_NUM_GROUPS=1 default,PMC_NUM_GROUPS=1,PMC_SERIAL_WALKS=1 \
   libpmc_r40.so,PMC_NUM_GROUPS=1 libpmc_c512.so,PMC_NUM_GROUPS=1 libpmc_inl.so,PMC_NUM_GROUPS=1 \
   libpmc_inl.so,PMC_NUM_GROUPS=1,PMC_WALK_BLOCKS_PER_CU=1,PMC_PEEL_BLOCKS_PER_CU=1 libpmc_inl.so,PMC_NUM_GROUPS=1,PMC_WALK_BLOCKS_PER_CU=2,PMC_PEEL_BLOCKS_PER_CU=1 \
   > $O/sweep.txt 2> $O/sweep.err
cat $O/sweep.txt; grep PMC_PROFILE $O/sweep.err
(cd /tmp && PMC_NUM_GROUPS=1 PMC_SERIAL_WALKS=1 timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d $OLDPWD/$O/kt -- python $OLDPWD/bench.py --steps 1 --warmup 0 --packets 2e7 --no-cpu-baseline > $OLDPWD/$O/kt.log 2>&1)
find $O/kt -name "*kernel_stats.csv" -exec cat {} \; | head -8
find $O -name "*kernel_trace.csv" -size +20M -delete
