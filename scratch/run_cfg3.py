"""BASELINE configs[2]: detection expert + Hungarian matcher train step, B=8 (for rocprofv3)."""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch, bench
from self_driving_model_amd import runtime
runtime.set_compute_dtype(torch.float16)
print("cfg3 img/s", bench.bench_detection(8, int(os.environ.get("STEPS", 12)), 4), flush=True)
