set -x
export TMPDIR=/tmp
R=$PWD
O=$R/gpurun_out/${RUN_NAME:-r01_v7}
rm -rf $O; mkdir -p $O
timeout 600 python -m pytest tests -m gpu -x -q > $O/pytest.log 2>&1; echo "pytest rc $?" >> $O/pytest.log
tail -3 $O/pytest.log
timeout 400 python bench.py > $O/bench.json 2> $O/bench.err
cat $O/bench.json
(cd /tmp && timeout 200 rocprofv3 --kernel-trace --stats --output-format csv -d $O/kt -- python $R/bench.py --steps 2 --warmup 1 --packets 2e7 --no-cpu-baseline > $O/kt.log 2>&1)
(cd /tmp && timeout 200 rocprofv3 --pmc FETCH_SIZE --output-format csv -d $O/pmc_FETCH_SIZE -- python $R/bench.py --steps 1 --warmup 0 --packets 2e7 --no-cpu-baseline > $O/pmc_f.log 2>&1)
(cd /tmp && timeout 200 rocprofv3 --pmc WRITE_SIZE --output-format csv -d $O/pmc_WRITE_SIZE -- python $R/bench.py --steps 1 --warmup 0 --packets 2e7 --no-cpu-baseline > $O/pmc_w.log 2>&1)
find $O -name "*kernel_trace.csv" -size +20M -delete
timeout 600 python bench.py --config 3 --steps 2 --warmup 1 --no-cpu-baseline > $O/bench_config3.json 2> $O/bench_config3.err; cat $O/bench_config3.json
timeout 600 python bench.py --store-radiation-field --steps 2 --warmup 1 --no-cpu-baseline > $O/bench_rf.json 2> $O/bench_rf.err; cat $O/bench_rf.json
mkdir -p $O/cli && timeout 300 ./skirt9_amd/lib/skirt_mi355x -o $O/cli tests/ski/cfg1rf.ski > $O/cli.log 2>&1; tail -4 $O/cli.log; ls $O/cli | head -20; head -5 $O/cli/cfg1rf_rf_J.dat
ls -R $O | head -40
