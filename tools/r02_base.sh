#!/bin/bash
# round-2 baseline of HEAD on the GPU box: parity tests, headline bench, kernel trace, in-kernel phase stamps
export TMPDIR=/tmp
R=$PWD
O=$R/gpurun_out/${RUN_NAME:-r02_base}; rm -rf $O; mkdir -p $O
timeout 900 python -m pytest tests -m gpu -x -q > $O/pytest.log 2>&1; echo "pytest rc $?" >> $O/pytest.log; tail -3 $O/pytest.log
timeout 500 python bench.py --steps 3 --warmup 1 > $O/bench.json 2> $O/bench.err; cat $O/bench.json
(cd /tmp && timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d $O/kt -- python $R/bench.py --steps 2 --warmup 1 --packets 2e7 --no-cpu-baseline > $O/kt.log 2>&1)
find $O/kt -name "*kernel_stats.csv" -exec cat {} \; | head -12
find $O -name "*kernel_trace.csv" -size +20M -delete
timeout 600 python tools/sweep.py libpmc_prof.so,PMC_NUM_GROUPS=1,PMC_PROFILE_DUMP=1 default,PMC_NUM_GROUPS=1 default default,PMC_NUM_GROUPS=1,PMC_SERIAL_WALKS=1 > $O/sweep.txt 2> $O/sweep.err
cat $O/sweep.txt; grep PMC_PROFILE $O/sweep.err
