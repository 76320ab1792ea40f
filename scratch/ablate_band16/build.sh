#!/bin/bash
# builds scratch/ablate_band16/band16_abl<mask>.so for every ablation mask (cross-compiles without a GPU)
cd "$(dirname "$0")"
for m in 0 1 2 3 4 5 6 7 8 9 10 12 14; do
  /opt/rocm/bin/hipcc -O3 -std=c++17 -fPIC --offload-arch=gfx950 -ffp-contract=off -DAMB_ABL=$m -shared -o band16_abl$m.so wrap.hip &
  if (( $(jobs -r | wc -l) >= 6 )); then wait -n; fi
done
wait
ls -la *.so | wc -l
