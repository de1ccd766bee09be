#!/bin/bash
# round 3, batch 6 (GPU box): cycle start split off the transition and launch kernels: parity tests, then occupancy variants
export TMPDIR=/tmp
R=$PWD
O=$R/gpurun_out/r03_batch6; mkdir -p $O
timeout 900 python -m pytest tests/test_gpu_parity.py tests/test_gpu_multi.py -m gpu -x -q > $O/gputests.txt 2>&1; echo "gpu tests rc=$?"; tail -8 $O/gputests.txt | cut -c1-300
S="PMC_NUM_GROUPS=1,PMC_SERIAL_WALKS=1,PMC_TIMING_DUMP=1"
python tools/sweep.py --packets 1e8 libpmc_r02.so default libpmc_tw3.so libpmc_tw4.so libpmc_tw3l2.so libpmc_tw2l3.so libpmc_r02.so default \
   libpmc_r02.so,$S,PMC_WALK_BLOCKS_PER_CU=3 default,$S,PMC_WALK_BLOCKS_PER_CU=3 libpmc_tw3.so,$S,PMC_WALK_BLOCKS_PER_CU=3 libpmc_tw4.so,$S,PMC_WALK_BLOCKS_PER_CU=3 > $O/sweep.txt 2>&1
grep -v "amdgpu.ids" $O/sweep.txt | grep "pkt/s\|PMC_TIMING" | awk '/pkt/ {print last} !/PMC_TIMING/ {print} {last=$0}' | cut -c1-200
