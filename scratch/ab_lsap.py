"""lsap_k timing on cost matrices of controlled structure: random (short augmenting paths) vs nearly identical columns (an
untrained detection head: every query wants the same targets, paths of length ~N)."""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from self_driving_model_amd.hip import matcher as hm
dev = torch.device("cuda:0")
def t(f, n=5):
    for _ in range(2): f()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(n): f()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / n * 1e3
g = torch.Generator(device="cpu").manual_seed(0)
for (B, Q, N) in [(8, 920, 18), (8, 920, 32), (64, 920, 64), (8, 920, 64)]:
    rnd = torch.rand(B, N, Q, generator=g)
    same = torch.rand(B, N, 1, generator=g).expand(B, N, Q) + 1e-4 * torch.rand(B, N, Q, generator=g)
    n = torch.full((B,), N, dtype=torch.int32)
    out = []
    rows_alike = torch.rand(B, 1, Q, generator=g).expand(B, N, Q) + 1e-4 * torch.rand(B, N, Q, generator=g)
    for name, c in (("random", rnd), ("near-identical columns", same), ("near-identical rows", rows_alike)):
        cd = c.contiguous().to(dev)
        nd = n.to(dev)
        ts = []
        for split in (True, False):  # round 3: the split solver (am_lsap_batched_ws) against the general kernels alone
            hm.USE_SPLIT_SOLVER = split
            ts.append(t(lambda: hm.lsap_batched(cd, nd, transposed_storage=True)))
        hm.USE_SPLIT_SOLVER = True
        out.append(f"{name} split {ts[0]:8.1f} us / general {ts[1]:8.1f} us")
    print(f"B={B} Q={Q} N={N}: " + "   ".join(out), flush=True)
