timeout 900 python -m pytest tests -m gpu -x -q > gpurun_out/pytest26.txt 2>&1; grep -E "passed|failed|rror" gpurun_out/pytest26.txt | tail -3
timeout 900 python bench.py --store-radiation-field --steps 2 --warmup 1 --no-cpu-baseline > gpurun_out/bench_rf.json 2> gpurun_out/bench_rf.err; cut -c1-300 gpurun_out/bench_rf.json
