#!/bin/bash
# round 3, batch 3 (GPU box): (1) upper bound for walk lists binned by direction octant per XCD: every propagation walk folded
# into the +++ octant (wrong physics, same access shape per XCD); (2) bench.py through a one-rank RCCL communicator (flow check
# of the N > 1 path); (3) GPU tests
export TMPDIR=/tmp
R=$PWD
O=$R/gpurun_out/r03_batch3; mkdir -p $O
S="PMC_NUM_GROUPS=1,PMC_SERIAL_WALKS=1,PMC_TIMING_DUMP=1"
python tools/sweep.py --packets 5e7 \
  default,$S,PMC_WALK_BLOCKS_PER_CU=3 \
  libpmc_fold.so,$S,PMC_WALK_BLOCKS_PER_CU=3 \
  default \
  libpmc_fold.so \
  > $O/sweep.txt 2>&1
grep -v "amdgpu.ids" $O/sweep.txt | grep "pkt/s\|peel\|PMC_TIMING" | awk '/pkt/ {print last} !/PMC_TIMING/ {print} {last=$0}'
BENCH_FORCE_COMM=1 python bench.py --steps 2 --warmup 1 --packets 2e7 --no-cpu-baseline --no-secondary > $O/bench_forced_comm.json 2> $O/bench_forced_comm.err; echo "bench rc=$?"; tail -c 1500 $O/bench_forced_comm.json; tail -3 $O/bench_forced_comm.err
timeout 900 python -m pytest tests -m gpu -x -q > $O/gputests.txt 2>&1; echo "gpu tests rc=$?"; tail -3 $O/gputests.txt
