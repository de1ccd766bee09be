#!/bin/bash
# round 2, experiment 19: interleaved statistics accumulators
export TMPDIR=/tmp
R=$PWD
O=$R/gpurun_out/r02_exp19; rm -rf $O; mkdir -p $O
timeout 900 python -m pytest tests -m gpu -x -q > $O/pytest.log 2>&1; tail -5 $O/pytest.log
E=PMC_PEEL_BLOCKS_PER_CU=1
timeout 900 python tools/sweep.py --packets 5e7 default,$E default,$E,PMC_NUM_GROUPS=1,PMC_SERIAL_WALKS=1,PMC_TIMING_DUMP=1 > $O/sweep.txt 2> $O/sweep.err
cut -c1-175 $O/sweep.txt; grep PMC_TIMING $O/sweep.err | tail -1
RUN_NAME=r02_exp19 bash tools/r02_kt.sh
