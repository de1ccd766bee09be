#!/bin/bash
# GPU box: the profile set of a build -> gpurun_out/$RUN_NAME (summaries are copied to profiles/ by hand)
#   RUN_NAME=r06_v2 tools/run_profile_set.sh
export TMPDIR=/tmp
R=$PWD
O=$R/gpurun_out/${RUN_NAME:-r06_v2}
rm -rf $O; mkdir -p $O
timeout 900 python -m pytest tests -m gpu -x -q > $O/pytest.log 2>&1; echo "pytest rc $?" >> $O/pytest.log; tail -3 $O/pytest.log
# (the headline run: with the driver's step counts, and with its own counter passes behind the timed region)
timeout 900 python bench.py --steps 20 --warmup 5 > $O/bench.json 2> $O/bench.err; tail -c 600 $O/bench.json
# kernel trace of the bench command itself (three slot groups: launches overlap), and without overlap
(cd /tmp && timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d $O/kt -- python $R/bench.py --steps 2 --warmup 1 --packets 2e7 --no-cpu-baseline --no-secondary --no-breakdown --no-counters > $O/kt.log 2>&1)
(cd /tmp && PMC_NUM_GROUPS=1 PMC_SERIAL_WALKS=1 PMC_WALK_BLOCKS_PER_CU=3 timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d $O/kt_one_group -- python $R/bench.py --steps 1 --warmup 0 --packets 5e7 --no-cpu-baseline --no-secondary --no-breakdown --no-counters > $O/kt1.log 2>&1)
# HBM traffic: separate counter passes (MI355X_MICROARCH.md), one step of 2e7 packets each
(cd /tmp && timeout 300 rocprofv3 --pmc FETCH_SIZE --output-format csv -d $O/pmc_FETCH_SIZE -- python $R/bench.py --steps 1 --warmup 0 --packets 2e7 --no-cpu-baseline --no-secondary --no-breakdown --no-counters > $O/pmc_f.log 2>&1)
(cd /tmp && timeout 300 rocprofv3 --pmc WRITE_SIZE --output-format csv -d $O/pmc_WRITE_SIZE -- python $R/bench.py --steps 1 --warmup 0 --packets 2e7 --no-cpu-baseline --no-secondary --no-breakdown --no-counters > $O/pmc_w.log 2>&1)
python tools/pmc_hbm_summary.py $O > $O/pmc_hbm.csv; cat $O/pmc_hbm.csv
# L2 requests / hits / misses and VALU instructions per kernel (a pass of its own): the line-rate evidence of bench.py's roofline.dominant_kernel
(cd /tmp && timeout 300 rocprofv3 --pmc TCC_HIT_sum TCC_MISS_sum TCC_REQ_sum SQ_INSTS_VALU SQ_WAVES --output-format csv -d $O/pmc_l2 -- python $R/bench.py --steps 1 --warmup 0 --packets 2e7 --no-cpu-baseline --no-secondary --no-breakdown --no-counters > $O/pmc_l2.log 2>&1)
python tools/pmc_l2_summary.py $O/pmc_l2 > $O/pmc_l2.csv; cat $O/pmc_l2.csv
# configs[4] (Voronoi): kernel trace and the counters of the CU's vector memory unit (each in a run of its own)
(cd /tmp && timeout 400 rocprofv3 --kernel-trace --stats --output-format csv -d $O/kt_voronoi -- python $R/bench.py --config 5 --steps 1 --warmup 0 --packets 2e7 --no-cpu-baseline --no-secondary --no-counters > $O/ktv.log 2>&1)
PMC_PASS_ARGS="--config 5" tools/pmc_pass.sh ${RUN_NAME:-r06_v2}_voronoi_l2 "TCC_HIT_sum TCC_MISS_sum TCC_REQ_sum SQ_INSTS_VALU SQ_WAVES" > /dev/null
PMC_PASS_ARGS="--config 5" tools/pmc_pass.sh ${RUN_NAME:-r06_v2}_voronoi_ta "TA_TA_BUSY_sum TCP_TOTAL_CACHE_ACCESSES_sum GRBM_GUI_ACTIVE SQ_ACTIVE_INST_VALU SQ_INSTS_VALU" > /dev/null
cat $R/gpurun_out/${RUN_NAME:-r06_v2}_voronoi_l2.txt $R/gpurun_out/${RUN_NAME:-r06_v2}_voronoi_ta.txt > $O/pmc_voronoi.txt; cat $O/pmc_voronoi.txt
find $O -name "*kernel_trace.csv" -delete; find $O -name "*counter_collection.csv" -size +8M -delete
# the other workloads
timeout 600 python bench.py --config 3 --steps 2 --warmup 1 --no-cpu-baseline --no-counters > $O/bench_config3.json 2> $O/c3.err
timeout 900 python bench.py --config 4 --steps 2 --warmup 1 --packets 5e7 --no-counters > $O/bench_config4.json 2> $O/c4.err
timeout 900 python bench.py --config 5 --steps 2 --warmup 1 > $O/bench_config5.json 2> $O/c5.err
timeout 600 python bench.py --source uniform --steps 2 --warmup 1 --no-cpu-baseline --no-counters > $O/bench_uniform_source.json 2> $O/u.err
timeout 600 python bench.py --store-radiation-field --steps 2 --warmup 1 --no-cpu-baseline --no-counters > $O/bench_radiation_field.json 2> $O/rf.err
timeout 900 python bench.py --steps 1 --warmup 0 --packets 1e9 --no-cpu-baseline --no-secondary --no-breakdown --no-counters > $O/bench_one_segment_1e9.json 2> $O/b9.err
for f in $O/bench*.json; do python -c "
import json,sys
d=json.loads([l for l in open('$f') if l.startswith('{')][-1]); r=d['roofline']; print('$f'.split('/')[-1], '%.4g'%d['value'], 'ms %.1f'%d['ms_per_step'], 'frac %.4f'%r['frac'], (d.get('cpu_baseline') or {}).get('value'))"; done
du -sh $O
