#!/bin/bash
# usage: scratch/ab_cfg23.sh "ENV1=.." "ENV2=.." : BASELINE configs[1] / [2] (expert training) img/s, 3 interleaved rounds
for r in 1 2 3; do
  for e in "$@"; do
    v=$(env $e python -c "
import bench, torch
from self_driving_model_amd import runtime
runtime.set_compute_dtype(torch.float16)
print('cfg2', bench.bench_drivable(16, 10, 4), 'cfg3', bench.bench_detection(8, 12, 4))" 2>/dev/null | tail -1)
    echo "$e -> $v"
  done
done
