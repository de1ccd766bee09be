export PMC_PROFILE_DUMP=1 PMC_TIMING_DUMP=1
S="PMC_NUM_GROUPS=1,PMC_WALK_BLOCKS_PER_CU=3,PMC_SERIAL_WALKS=1"
timeout 600 python tools/sweep.py --packets 1e8 default default,$S libpmc_q4.so,$S libpmc_q16.so,$S libpmc_q24.so,$S default libpmc_q16.so 2>&1 | grep -v "PMC_GEN\|amdgpu.ids\|PROFILE prop\|census" | tee gpurun_out/sweep12.txt
PMC_PEEL_V1=1 timeout 600 python tools/sweep.py --packets 1e8 default default,$S default 2>&1 | grep -v "PMC_GEN\|amdgpu.ids\|PROFILE prop\|census" | tee gpurun_out/sweep12b.txt
