export PMC_PROFILE_DUMP=1 PMC_TIMING_DUMP=1
S="PMC_NUM_GROUPS=1,PMC_WALK_BLOCKS_PER_CU=3,PMC_SERIAL_WALKS=1"
timeout 600 python tools/sweep.py --ski tests/ski/cfg2small.ski --packets 1e8 default,$S default,$S,PMC_PEEL_BLOCKS_PER_CU=2 default,$S,PMC_PEEL_BLOCKS_PER_CU=3 default,$S,PMC_CELL_SHUFFLE=0 2>&1 | grep -v "PMC_GEN\|amdgpu.ids\|census" | tee gpurun_out/sweep13.txt
timeout 600 python tools/sweep.py --packets 1e8 default,$S,PMC_CELL_SHUFFLE=0 default,$S,PMC_CELL_SHUFFLE=3 2>&1 | grep -v "PMC_GEN\|amdgpu.ids\|census" | tee -a gpurun_out/sweep13.txt
