mkdir -p /tmp/sites && python tools/make_sites.py --n 100000 --seed 1 /tmp/sites/cfg5_sites.txt && export SKH_INPUT_PATH=/tmp/sites
V="default default"
for g in 1 2 3; do for w in 1 2 3; do V="$V default,PMC_NUM_GROUPS=$g,PMC_WALK_BLOCKS_PER_CU=$w"; done; done
timeout 1500 python tools/sweep.py --ski tests/ski/cfg5.ski --packets 2e7 $V 2>&1 | grep pkt | tee gpurun_out/sweep30.txt
