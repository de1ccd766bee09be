#!/bin/bash
# round 2, experiment 14: statistics lists deduplicated at insertion (linear flush), transition / launch kernels at two waves per SIMD
export TMPDIR=/tmp
O=gpurun_out/r02_exp14; rm -rf $O; mkdir -p $O
timeout 900 python -m pytest tests/test_gpu_parity.py tests/test_gpu_full_size.py -m gpu -x -q > $O/pytest.log 2>&1; tail -5 $O/pytest.log
E=PMC_NUM_GROUPS=1,PMC_SERIAL_WALKS=1,PMC_TIMING_DUMP=1
timeout 900 python tools/sweep.py --packets 5e7 default,$E default default,PMC_PEEL_BLOCKS_PER_CU=1 default,PMC_PEEL_BLOCKS_PER_CU=2 default,PMC_NUM_GROUPS=3 > $O/sweep.txt 2> $O/sweep.err
cat $O/sweep.txt | cut -c1-200; grep "PMC_TIMING" $O/sweep.err | awk 'NR%3==0'
