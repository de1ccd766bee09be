#!/bin/bash
# round 3, batch 1 (GPU box): perturbation sweep of the octree walk kernels + occupancy + a PC-sampling attempt
export TMPDIR=/tmp
R=$PWD
O=$R/gpurun_out/r03_batch1; mkdir -p $O
S="PMC_NUM_GROUPS=1,PMC_SERIAL_WALKS=1,PMC_TIMING_DUMP=1"
python tools/sweep.py --packets 5e7 \
  default,$S,PMC_WALK_BLOCKS_PER_CU=3 \
  libpmc_pert_valu_48.so,$S,PMC_WALK_BLOCKS_PER_CU=3 \
  libpmc_pert_gather_1.so,$S,PMC_WALK_BLOCKS_PER_CU=3 \
  libpmc_pert_gather_2.so,$S,PMC_WALK_BLOCKS_PER_CU=3 \
  libpmc_pert_lds_6.so,$S,PMC_WALK_BLOCKS_PER_CU=3 \
  default,$S,PMC_WALK_BLOCKS_PER_CU=1 \
  libpmc_pert_valu_48.so,$S,PMC_WALK_BLOCKS_PER_CU=1 \
  libpmc_pert_gather_2.so,$S,PMC_WALK_BLOCKS_PER_CU=1 \
  default,$S,PMC_WALK_BLOCKS_PER_CU=2 \
  default,$S,PMC_WALK_BLOCKS_PER_CU=3,PMC_PEEL_BLOCKS_PER_CU=2 \
  default \
  libpmc_pert_valu_48.so \
  libpmc_pert_gather_1.so \
  libpmc_pert_gather_2.so \
  libpmc_pert_lds_6.so \
  > $O/sweep.txt 2>&1
tail -60 $O/sweep.txt
# PC sampling (beta): stochastic first, host trap second; short run, own timeout
cd /tmp
for m in stochastic host_trap; do
  if [ $m = stochastic ]; then U="--pc-sampling-unit cycles --pc-sampling-interval 1048576"; else U="--pc-sampling-unit time --pc-sampling-interval 100"; fi
  ROCPROFILER_PC_SAMPLING_BETA_ENABLED=1 timeout 240 rocprofv3 --pc-sampling-beta-enabled --pc-sampling-method $m $U --kernel-trace --output-format csv -d $O/pcs_$m -- python $R/bench.py --steps 1 --warmup 0 --packets 1e7 --no-cpu-baseline --no-secondary > $O/pcs_$m.log 2>&1
  echo "pc sampling $m rc=$?"; tail -5 $O/pcs_$m.log
  find $O/pcs_$m -type f | head; 
done
find $O -name "*kernel_trace.csv" -size +20M -delete
find $O -name "*pc_sampling*.csv" -size +40M -exec sh -c 'head -c 40000000 "$1" > "$1.head"; rm "$1"' _ {} \;
du -sh $O
cd $R
timeout 600 python -m pytest tests -m gpu -x -q > $O/gputests.txt 2>&1; echo "gpu tests rc=$?"; tail -3 $O/gputests.txt
