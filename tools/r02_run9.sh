export PMC_PROFILE_DUMP=1 PMC_TIMING_DUMP=1
S="PMC_NUM_GROUPS=1,PMC_WALK_BLOCKS_PER_CU=3,PMC_SERIAL_WALKS=1"
timeout 600 python tools/sweep.py --packets 1e8 default default,PMC_PEEL_V1=1 default,$S default,$S,PMC_PEEL_V1=1 default,$S,PMC_PEEL_BLOCKS_PER_CU=2 default 2>&1 | grep -v "PMC_GEN\|amdgpu.ids" | tee gpurun_out/sweep9.txt
timeout 900 python -m pytest tests -m gpu -x -q > gpurun_out/pytest9.txt 2>&1; grep -E "passed|failed|rror" gpurun_out/pytest9.txt | tail -5
