timeout 900 python -m pytest tests/test_gpu_parity.py tests/test_gpu_config5.py -m gpu -x -q -k "cfg5 or config5 or voronoi" > gpurun_out/pytest39.txt 2>&1; grep -E "passed|failed|rror|assert" gpurun_out/pytest39.txt | tail -6
mkdir -p /tmp/sites && python tools/make_sites.py --n 100000 --seed 1 /tmp/sites/cfg5_sites.txt && export SKH_INPUT_PATH=/tmp/sites
timeout 900 python tools/sweep.py --ski tests/ski/cfg5.ski --packets 2e7 default default libpmc_nohead.so libpmc_nohead.so default,PMC_NUM_GROUPS=1 libpmc_nohead.so,PMC_NUM_GROUPS=1 2>&1 | grep pkt | tee gpurun_out/sweep39.txt
