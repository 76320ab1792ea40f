#!/bin/bash
cd "$(dirname "$0")"
for m in 0 1 2 4 8 3 6 9 11 13 14 7; do
  /opt/rocm/bin/hipcc -O3 -std=c++17 -fPIC --offload-arch=gfx950 -ffp-contract=off -DAMP3_ABL=$m -shared -o duo_abl$m.so wrap.hip &
  if (( $(jobs -r | wc -l) >= 6 )); then wait -n; fi
done
wait
ls *.so | wc -l
