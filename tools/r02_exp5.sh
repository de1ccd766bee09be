#!/bin/bash
export TMPDIR=/tmp
O=gpurun_out/r02_exp5; rm -rf $O; mkdir -p $O
timeout 900 python tools/sweep.py libpmc_prof.so,PMC_NUM_GROUPS=1,PMC_PROFILE_DUMP=1 default,PMC_NUM_GROUPS=1 default \
   libpmc_q4.so,PMC_NUM_GROUPS=1 libpmc_p8.so,PMC_NUM_GROUPS=1 libpmc_r8.so,PMC_NUM_GROUPS=1 libpmc_r32.so,PMC_NUM_GROUPS=1 libpmc_b256.so,PMC_NUM_GROUPS=1 \
   default,PMC_NUM_GROUPS=1,PMC_SERIAL_WALKS=1 default,PMC_NUM_GROUPS=1,PMC_PEEL_BLOCKS_PER_CU=2 default,PMC_NUM_GROUPS=1,PMC_WALK_BLOCKS_PER_CU=2 \
   default,PMC_NUM_GROUPS=3 default,PMC_NUM_SLOTS=4194304 default,PMC_NUM_SLOTS=16777216 \
   > $O/sweep.txt 2> $O/sweep.err
cat $O/sweep.txt; grep PMC_PROFILE $O/sweep.err
timeout 600 python -m pytest tests/test_gpu_parity.py tests/test_gpu_full_size.py -m gpu -x -q > $O/pytest.log 2>&1; tail -3 $O/pytest.log
