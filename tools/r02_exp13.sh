#!/bin/bash
# round 2, experiment 13: pipelined walk step on 32-byte cell records in depth-first order (octet links)
export TMPDIR=/tmp
O=gpurun_out/r02_exp13; rm -rf $O; mkdir -p $O
timeout 900 python -m pytest tests/test_gpu_parity.py tests/test_gpu_full_size.py tests/test_gpu_config4.py -m gpu -x -q > $O/pytest.log 2>&1; tail -5 $O/pytest.log
E=PMC_NUM_GROUPS=1,PMC_SERIAL_WALKS=1,PMC_TIMING_DUMP=1
timeout 900 python tools/sweep.py --packets 5e7 default,$E default,$E,PMC_CELL_ORDER=ref default,$E,PMC_PEEL_BLOCKS_PER_CU=1 default,$E,PMC_PEEL_BLOCKS_PER_CU=2 default,$E,PMC_WALK_BLOCKS_PER_CU=2 \
   default libpmc_census.so,$E,PMC_PROFILE_DUMP=1 > $O/sweep.txt 2> $O/sweep.err
cat $O/sweep.txt | cut -c1-200; grep "PMC_TIMING" $O/sweep.err | awk 'NR%3==0'; grep "PMC_PROFILE" $O/sweep.err | grep -v phases
