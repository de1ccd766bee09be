cd $GRAFT_REPO_ROOT
export TMPDIR=/tmp
mkdir -p gpurun_out/r01_v1
python bench.py > gpurun_out/r01_v1/bench_default.json 2> gpurun_out/r01_v1/bench_default.err
tail -c 1500 gpurun_out/r01_v1/bench_default.json
rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/r01_v1/trace -o r01 -- python bench.py --steps 3 --warmup 1 --packets 2e7 --no-cpu-baseline > gpurun_out/r01_v1/trace.log 2>&1
rocprofv3 --pmc FETCH_SIZE --output-format csv -d gpurun_out/r01_v1/pmc_fetch -o r01 -- python bench.py --steps 1 --warmup 0 --packets 2e7 --no-cpu-baseline > gpurun_out/r01_v1/pmc_fetch.log 2>&1
rocprofv3 --pmc WRITE_SIZE --output-format csv -d gpurun_out/r01_v1/pmc_write -o r01 -- python bench.py --steps 1 --warmup 0 --packets 2e7 --no-cpu-baseline > gpurun_out/r01_v1/pmc_write.log 2>&1
find gpurun_out/r01_v1 -type f | head -30
