V="default"
for w in 1 2 3; do for p in 1 2; do V="$V default,PMC_NUM_GROUPS=1,PMC_WALK_BLOCKS_PER_CU=$w,PMC_PEEL_BLOCKS_PER_CU=$p"; done; done
V="$V default,PMC_NUM_GROUPS=1,PMC_WALK_BLOCKS_PER_CU=3,PMC_SERIAL_WALKS=1 default,PMC_NUM_GROUPS=1,PMC_WALK_BLOCKS_PER_CU=2,PMC_SERIAL_WALKS=1 default,PMC_NUM_GROUPS=2,PMC_WALK_BLOCKS_PER_CU=2 default,PMC_NUM_GROUPS=2,PMC_WALK_BLOCKS_PER_CU=1 default,PMC_NUM_GROUPS=2,PMC_WALK_BLOCKS_PER_CU=3 default"
timeout 800 python tools/sweep.py --packets 1e8 $V 2>&1 | grep -v "PMC_GEN\|amdgpu.ids" | tee gpurun_out/sweep8.txt
