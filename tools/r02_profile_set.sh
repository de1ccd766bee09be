#!/bin/bash
# round-2 profile set of the current build (GPU box): parity tests, headline bench, kernel trace, HBM counter passes,
# SQ / TCP / TCC counters, the other workloads.  usage: RUN_NAME=r02_v1 bash tools/r02_profile_set.sh
export TMPDIR=/tmp
R=$PWD
O=$R/gpurun_out/${RUN_NAME:-r02_v1}; rm -rf $O; mkdir -p $O
timeout 900 python -m pytest tests -m gpu -x -q > $O/pytest.log 2>&1; echo "pytest rc $?" >> $O/pytest.log; tail -3 $O/pytest.log
timeout 600 python bench.py > $O/bench.json 2> $O/bench.err; cat $O/bench.json
(cd /tmp && timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d $O/kt -- python $R/bench.py --steps 2 --warmup 1 --packets 2e7 --no-cpu-baseline --no-secondary > $O/kt.log 2>&1)
(cd /tmp && timeout 300 rocprofv3 --pmc FETCH_SIZE --output-format csv -d $O/pmc_FETCH_SIZE -- python $R/bench.py --steps 1 --warmup 0 --packets 2e7 --no-cpu-baseline --no-secondary > $O/pmc_f.log 2>&1)
(cd /tmp && timeout 300 rocprofv3 --pmc WRITE_SIZE --output-format csv -d $O/pmc_WRITE_SIZE -- python $R/bench.py --steps 1 --warmup 0 --packets 2e7 --no-cpu-baseline --no-secondary > $O/pmc_w.log 2>&1)
python tools/pmc_hbm_summary.py $O > $O/pmc_hbm.csv; cat $O/pmc_hbm.csv
bash tools/pmc_passes.sh --steps 1 --warmup 0 --packets 2e7 --no-cpu-baseline --no-secondary > $O/pmc_passes.log 2>&1; cp gpurun_out/pmc/summary.txt $O/pmc_counters.txt; cat $O/pmc_counters.txt
find $O -name "*kernel_stats.csv" -exec cp {} $O/kernel_stats.csv \;
find $O -name "*kernel_trace.csv" -size +20M -delete
head -8 $O/kernel_stats.csv
