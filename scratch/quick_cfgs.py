"""Quick throughput of the trainable configurations: cfg2 (drivable expert B=16), cfg1-like (segmentation expert B=4), 4b (optional)."""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch, bench
from self_driving_model_amd import runtime
runtime.set_compute_dtype(torch.float16)
print("cfg2 drivable B16 img/s", bench.bench_drivable(16, 12, 4), flush=True)
if "4b" in sys.argv:
    print("4b", bench.bench_unfrozen(32, 8, 3) if hasattr(bench, "bench_unfrozen") else "n/a", flush=True)
