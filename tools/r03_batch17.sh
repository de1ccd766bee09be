#!/bin/bash
# round 3, batch 17 (GPU box): statistics log (partition + LDS sums instead of sector-wise atomics): parity, A/B, kernel trace
export TMPDIR=/tmp
R=$PWD
O=$R/gpurun_out/r03_batch17; mkdir -p $O
timeout 900 python -m pytest tests/test_gpu_parity.py tests/test_gpu_multi.py -m gpu -x -q > $O/gputests.txt 2>&1; echo "gpu tests rc=$?"; tail -8 $O/gputests.txt | cut -c1-300
S="PMC_NUM_GROUPS=1,PMC_SERIAL_WALKS=1,PMC_TIMING_DUMP=1,PMC_WALK_BLOCKS_PER_CU=3"
python tools/sweep.py --packets 1e8 default,PMC_STAT_ATOMICS=1 default default,PMC_STAT_ATOMICS=1 default default,PMC_STAT_ATOMICS=1,$S default,$S > $O/sweep.txt 2>&1
grep -v "amdgpu.ids" $O/sweep.txt | grep "pkt/s\|PMC_TIMING" | awk '/pkt/ {print last} !/PMC_TIMING/ {print} {last=$0}' | cut -c1-200
(cd /tmp && PMC_NUM_GROUPS=1 PMC_SERIAL_WALKS=1 PMC_WALK_BLOCKS_PER_CU=3 timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d $O/kt -- python $R/bench.py --steps 1 --warmup 1 --packets 5e7 --no-cpu-baseline --no-secondary > $O/kt.log 2>&1)
find $O -name "*kernel_trace.csv" -delete
