#!/bin/bash
# round 2, experiment 2: octree walk split into a peel-off kernel (wave-uniform direction) and a propagation kernel
export TMPDIR=/tmp
O=gpurun_out/r02_exp2; rm -rf $O; mkdir -p $O
timeout 600 python -m pytest tests/test_gpu_parity.py tests/test_gpu_full_size.py -m gpu -x -q > $O/pytest.log 2>&1; tail -15 $O/pytest.log
timeout 900 python tools/sweep.py default,PMC_PROFILE_DUMP=1 default,PMC_NUM_GROUPS=1,PMC_PROFILE_DUMP=1 \
   libpmc_p6.so,PMC_NUM_GROUPS=1 libpmc_p6.so libpmc_p4.so,PMC_NUM_GROUPS=1 libpmc_p4.so \
   default,PMC_PEEL_BLOCKS_PER_CU=2 default,PMC_PEEL_BLOCKS_PER_CU=3 default,PMC_WALK_BLOCKS_PER_CU=2 default,PMC_WALK_BLOCKS_PER_CU=3 \
   > $O/sweep.txt 2> $O/sweep.err
cat $O/sweep.txt; grep PMC_PROFILE $O/sweep.err
profiles/microbench/valu_rate2 > $O/valu_rate2.txt 2>&1; cat $O/valu_rate2.txt
