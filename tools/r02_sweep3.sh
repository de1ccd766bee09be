B="PMC_NUM_GROUPS=3,PMC_WALK_BLOCKS_PER_CU=1"
V="default,$B libpmc_pb256.so,$B libpmc_pb256.so,$B,PMC_PEEL_BLOCKS_PER_CU=2 libpmc_pb128.so,$B libpmc_pb128.so,$B,PMC_PEEL_BLOCKS_PER_CU=2 libpmc_pb128.so,$B,PMC_PEEL_BLOCKS_PER_CU=4 libpmc_pb256.so,PMC_NUM_GROUPS=2,PMC_WALK_BLOCKS_PER_CU=1 libpmc_pb256.so,PMC_NUM_GROUPS=4,PMC_WALK_BLOCKS_PER_CU=1 default,$B default"
timeout 800 python tools/sweep.py --packets 1e8 $V 2>&1 | grep -v "PMC_GEN\|amdgpu.ids" | tee gpurun_out/sweep3.txt
