#!/bin/bash
# round 3, batch 31 (GPU box): task records without RN(1/k) and the cross section (recomputed / read from the slot): parity, A/B, kernel trace
export TMPDIR=/tmp
R=$PWD
O=$R/gpurun_out/r03_batch31; mkdir -p $O
timeout 1200 python -m pytest tests -m gpu -x -q > $O/gputests.txt 2>&1; echo "gpu tests rc=$?"; grep -E "passed|failed|Error" $O/gputests.txt | tail -3
python tools/sweep.py --packets 1e8 libpmc_prev.so default libpmc_prev.so default libpmc_prev.so default > $O/sweep.txt 2>&1
grep -v "amdgpu.ids" $O/sweep.txt | grep "pkt/s" | cut -c1-180
(cd /tmp && PMC_NUM_GROUPS=1 PMC_SERIAL_WALKS=1 timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d $O/kt -- python $R/bench.py --steps 1 --warmup 1 --packets 5e7 --no-cpu-baseline --no-secondary > $O/kt.log 2>&1)
find $O -name "*kernel_trace.csv" -delete
