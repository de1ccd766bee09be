#!/bin/bash
# round 3, batch 39 (GPU box): what the flux additions cost in the transition kernel (ablation builds)
export TMPDIR=/tmp
R=$PWD
O=$R/gpurun_out/r03_batch39; mkdir -p $O
for lib in libpmc.so libpmc_abl_fa.so libpmc_abl_hot.so; do
(cd /tmp && PMC_LIBRARY=$R/skirt9_amd/lib/$lib PMC_NUM_GROUPS=1 PMC_SERIAL_WALKS=1 timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d $O/kt_$lib -- python $R/bench.py --steps 1 --warmup 1 --packets 5e7 --no-cpu-baseline --no-secondary > $O/kt_$lib.log 2>&1)
find $O -name "*kernel_trace.csv" -delete
done
