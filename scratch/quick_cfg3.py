"""Quick throughput of cfg3 (detection expert + Hungarian matcher, B = 8, 720p) with the split LSAP solver on and off, and of the
fp32 parity mode of the 4a step."""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch, bench
from self_driving_model_amd import runtime
from self_driving_model_amd.hip import matcher as hm
runtime.set_compute_dtype(torch.float16)
for split in (True, False, True, False):
    hm.USE_SPLIT_SOLVER = split
    print(f"cfg3 detection B8 img/s (split solver {split})", bench.bench_detection(8, 16, 4), flush=True)
hm.USE_SPLIT_SOLVER = True
if "fp32" in sys.argv:
    from self_driving_model_amd.training import synthetic
    batch = synthetic.carla_sequence_batch(32, 720, 1280, 10, torch.device("cuda:0"), seed=0)
    print("4a fp32 mode B32 img/s", bench.bench_fp32_mode(32, batch), flush=True)
