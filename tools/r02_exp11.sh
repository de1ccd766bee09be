#!/bin/bash
# round 2, experiment 11: synthetic walk step, feature by feature (profiles/microbench/synth_walk.hip)
export TMPDIR=/tmp
O=gpurun_out/r02_exp11; rm -rf $O; mkdir -p $O
timeout 300 ./profiles/microbench/synth_walk > $O/synth_walk.txt 2>&1; cat $O/synth_walk.txt
