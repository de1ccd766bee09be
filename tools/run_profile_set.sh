set -x
export TMPDIR=/tmp
R=$PWD
O=$R/gpurun_out/r01_v6
rm -rf $O; mkdir -p $O
timeout 600 python -m pytest tests -m gpu -x -q > $O/pytest.log 2>&1; echo "pytest rc $?" >> $O/pytest.log
tail -3 $O/pytest.log
timeout 400 python bench.py > $O/bench.json 2> $O/bench.err
cat $O/bench.json
(cd /tmp && timeout 200 rocprofv3 --kernel-trace --stats --output-format csv -d $O/kt -- python $R/bench.py --steps 2 --warmup 1 --packets 2e7 --no-cpu-baseline > $O/kt.log 2>&1)
(cd /tmp && timeout 200 rocprofv3 --pmc FETCH_SIZE --output-format csv -d $O/pmc_FETCH_SIZE -- python $R/bench.py --steps 1 --warmup 0 --packets 2e7 --no-cpu-baseline > $O/pmc_f.log 2>&1)
(cd /tmp && timeout 200 rocprofv3 --pmc WRITE_SIZE --output-format csv -d $O/pmc_WRITE_SIZE -- python $R/bench.py --steps 1 --warmup 0 --packets 2e7 --no-cpu-baseline > $O/pmc_w.log 2>&1)
find $O -name "*kernel_trace.csv" -size +20M -delete
ls -R $O | head -40
