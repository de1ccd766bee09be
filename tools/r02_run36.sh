mkdir -p /tmp/sites && python tools/make_sites.py --n 100000 --seed 1 /tmp/sites/cfg5_sites.txt && export SKH_INPUT_PATH=/tmp/sites
timeout 900 python tools/sweep.py --ski tests/ski/cfg5.ski --packets 2e7 default default libpmc_cr8.so libpmc_cr5.so libpmc_cr4.so default,PMC_VORO_NO_CULL=1 2>&1 | grep pkt | tee gpurun_out/sweep36.txt
