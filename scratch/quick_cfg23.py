"""cfg2 (drivable expert B=16) and cfg3 (detection expert + Hungarian, B=8) throughput, A/B over one tuning key given as KEY=value,value."""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch, bench
from self_driving_model_amd import runtime
from self_driving_model_amd.hip import lib
runtime.set_compute_dtype(torch.float16)
key, vals = (sys.argv[1].split("=") + [""])[:2] if len(sys.argv) > 1 else ("", "")
vals = [int(v) for v in vals.split(",")] if vals else [None]
for rnd in range(2):
    for v in vals:
        if v is not None:
            lib.get().am_set_tuning(getattr(lib, "AM_TUNE_" + key), v)
        print(f"round {rnd} {key}={v}: cfg2 {bench.bench_drivable(16, 12, 4)} img/s, cfg3 {bench.bench_detection(8, 12, 4)} img/s", flush=True)
