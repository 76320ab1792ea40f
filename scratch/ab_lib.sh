#!/bin/bash
# usage: scratch/ab_lib.sh <script.py> <libA.so> [libB.so ...]; runs the script once per library (AUTOMOE_HIP_LIB), twice over
for r in 1 2; do
  for l in "${@:2}"; do
    echo "== $l (round $r)"
    if [ "$l" = "default" ]; then python "$1" 2>&1 | grep -v amdgpu.ids; else AUTOMOE_HIP_LIB=$l python "$1" 2>&1 | grep -v amdgpu.ids; fi
  done
done
