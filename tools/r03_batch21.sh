#!/bin/bash
# round 3, batch 21 (GPU box): kernel trace of config 5 (Voronoi), one slot group
export TMPDIR=/tmp
R=$PWD
O=$R/gpurun_out/r03_batch21; mkdir -p $O
(cd /tmp && PMC_NUM_GROUPS=1 timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d $O/kt5 -- python $R/bench.py --config 5 --steps 1 --warmup 0 --packets 2e7 --no-cpu-baseline > $O/kt5.log 2>&1)
find $O -name "*kernel_trace.csv" -delete
python tools/sweep.py --packets 1e8 default default > $O/sweep.txt 2>&1; grep -v "amdgpu.ids" $O/sweep.txt | grep "pkt/s" | cut -c1-160
