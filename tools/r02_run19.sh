export PMC_PROFILE_DUMP=1
mkdir -p /tmp/sph && python tools/make_sph.py --n 1000000 --seed 1 /tmp/sph/cfg4_sph.txt && export SKH_INPUT_PATH=/tmp/sph
timeout 1200 python tools/sweep.py --ski tests/ski/cfg4.ski --packets 5e7 default default default,PMC_WALK_BLOCKS_PER_CU=2 default,PMC_WALK_BLOCKS_PER_CU=3 default,PMC_NUM_GROUPS=2,PMC_WALK_BLOCKS_PER_CU=3 default,PMC_NUM_GROUPS=2,PMC_WALK_BLOCKS_PER_CU=2 2>&1 | grep -v "PMC_GEN\|amdgpu.ids\|census" | tee gpurun_out/sweep19.txt
