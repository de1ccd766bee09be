#!/bin/bash
# round 2, experiment 12: occupancy, cell shuffle and cursor chunk size at 5e7 packets (launch floors amortised), serial walks
export TMPDIR=/tmp
O=gpurun_out/r02_exp12; rm -rf $O; mkdir -p $O
E=PMC_NUM_GROUPS=1,PMC_SERIAL_WALKS=1,PMC_TIMING_DUMP=1
timeout 1200 python tools/sweep.py --packets 5e7 default,$E default,$E,PMC_PEEL_BLOCKS_PER_CU=1 default,$E,PMC_PEEL_BLOCKS_PER_CU=2 default,$E,PMC_WALK_BLOCKS_PER_CU=2 \
  default,$E,PMC_CELL_SHUFFLE=2 default,$E,PMC_CELL_SHUFFLE=3 default,$E,PMC_CELL_SHUFFLE=0 libpmc_c512.so,$E libpmc_c1024.so,$E libpmc_r32.so,$E \
  > $O/sweep.txt 2> $O/sweep.err
paste -d' ' <(cut -c1-60 $O/sweep.txt) <(grep PMC_TIMING $O/sweep.err | awk 'NR%3==0')
