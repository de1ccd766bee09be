#!/bin/bash
# round 3, batch 18 (GPU box): statistics log, entries per workgroup of the reduce kernel
export TMPDIR=/tmp
R=$PWD
O=$R/gpurun_out/r03_batch18; mkdir -p $O
for lib in libpmc.so libpmc_span12.so libpmc_span16.so; do
(cd /tmp && PMC_LIBRARY=$R/skirt9_amd/lib/$lib PMC_NUM_GROUPS=1 PMC_SERIAL_WALKS=1 PMC_WALK_BLOCKS_PER_CU=3 timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d $O/kt_$lib -- python $R/bench.py --steps 1 --warmup 1 --packets 5e7 --no-cpu-baseline --no-secondary > $O/kt_$lib.log 2>&1)
find $O -name "*kernel_trace.csv" -delete
done
python tools/sweep.py --packets 1e8 default,PMC_STAT_ATOMICS=1 default libpmc_span12.so default,PMC_STAT_ATOMICS=1 default libpmc_span12.so > $O/sweep.txt 2>&1
grep -v "amdgpu.ids" $O/sweep.txt | grep "pkt/s" | cut -c1-160
