B="PMC_NUM_GROUPS=3,PMC_WALK_BLOCKS_PER_CU=1"
V="default,$B default,$B libpmc_pr128.so,$B libpmc_pr128.so,PMC_NUM_GROUPS=3,PMC_WALK_BLOCKS_PER_CU=2 libpmc_pr128.so,PMC_NUM_GROUPS=3,PMC_WALK_BLOCKS_PER_CU=3 libpmc_rf32.so,$B libpmc_rf24.so,$B libpmc_rf48.so,$B libpmc_ws8.so,$B libpmc_ws2.so,$B default,$B"
timeout 800 python tools/sweep.py --packets 1e8 $V 2>&1 | grep -v "PMC_GEN\|amdgpu.ids" | tee gpurun_out/sweep4.txt
