mkdir -p /tmp/sites && python tools/make_sites.py --n 100000 --seed 1 /tmp/sites/cfg5_sites.txt && export SKH_INPUT_PATH=/tmp/sites
timeout 900 python tools/sweep.py --ski tests/ski/cfg5.ski --packets 2e7 default default libpmc_w3.so libpmc_w3r4.so libpmc_w4r4.so libpmc_w3.so,PMC_NUM_GROUPS=1 default,PMC_NUM_GROUPS=1 2>&1 | grep pkt | tee gpurun_out/sweep41.txt
