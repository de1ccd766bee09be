#!/bin/bash
# round 3, batch 38 (GPU box): statistics flush with the entries gathered in LDS, twelve per atomic instruction
export TMPDIR=/tmp
R=$PWD
O=$R/gpurun_out/r03_batch38; mkdir -p $O
timeout 900 python -m pytest tests/test_gpu_parity.py tests/test_gpu_multi.py -m gpu -x -q > $O/gputests.txt 2>&1; echo "gpu tests rc=$?"; grep -E "passed|failed" $O/gputests.txt | tail -1
python tools/sweep.py --packets 1e8 libpmc_prev.so default libpmc_prev.so default libpmc_prev.so default > $O/sweep.txt 2>&1
grep -v "amdgpu.ids" $O/sweep.txt | grep "pkt/s" | cut -c1-180
for lib in libpmc_prev.so libpmc.so; do
(cd /tmp && PMC_LIBRARY=$R/skirt9_amd/lib/$lib PMC_NUM_GROUPS=1 PMC_SERIAL_WALKS=1 timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d $O/kt_$lib -- python $R/bench.py --steps 1 --warmup 1 --packets 5e7 --no-cpu-baseline --no-secondary > $O/kt_$lib.log 2>&1)
find $O -name "*kernel_trace.csv" -delete
done
