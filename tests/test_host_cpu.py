"""CPU tests of the host side: the C-ABI library loads and exports every symbol include/automoe_hip.h declares,
geometry / packing logic, drop-in API surface (class names, state_dict keys, error behaviour), the product path
refusing to run without a HIP device, and the data-parallel gradient reducer under gloo with world_size 2."""
import ctypes
import os
import re
import subprocess
import sys

import numpy as np
import pytest
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


@pytest.fixture(scope="module")
def lib():
    so = os.path.join(ROOT, "self-driving-model_amd", "csrc", "libautomoe_hip.so")
    if not os.path.exists(so):
        subprocess.check_call(["make", "-j8", "-C", os.path.dirname(so)])
    from self_driving_model_amd.hip import lib as L
    return L


def test_abi_exports_every_declared_symbol(lib):
    L = lib.get()
    header = open(lib.HEADER).read()
    declared = set(re.findall(r"\bint\s+(am_\w+)\s*\(", re.sub(r"/\*.*?\*/", "", header, flags=re.S)))
    assert len(declared) >= 36
    out = subprocess.check_output(["nm", "-D", "--defined-only", lib.LIB_PATH], text=True)
    exported = {l.split()[-1] for l in out.splitlines() if " T " in l}
    assert declared <= exported, declared - exported
    assert set(L.protos) == declared
    assert L.am_version() >= 1
    assert [L.am_conv_npad(n) for n in (3, 14, 32, 33, 64, 65, 128, 192, 512)] == [32, 32, 32, 64, 64, 128, 128, 256, 512]
    assert ctypes.sizeof(lib.ConvGeom) == 21 * 4 + 2 * 2 * lib.AM_MAX_TAPS + 2 * 4  # matches struct am_conv_geom (+ osplit, osplit_stride)


def test_abi_rejects_bad_arguments_without_a_gpu(lib):
    L = lib.get()
    g = lib.ConvGeom()
    raw = L._raw_am_conv_gemm
    assert raw(None, 1, None, None, None, 0, None, None, None) == -1          # null geometry
    g.ntaps, g.krun, g.N, g.ldi = 1, 24, 64, 24                                # 48-byte run: not a 64-byte multiple
    assert raw(ctypes.byref(g), 1, None, None, None, 0, None, None, None) == -1
    assert L._raw_am_linear_fwd(None, 0, None, None, None, 0, 1, 1, 1, 0, None) == -1
    assert L._raw_am_lsap_batched(None, 1, 4, None, 4, 16, 4, 1, None, None, 4, None, None, None) == -1
    with pytest.raises(RuntimeError):
        L.am_linear_fwd(None, 0, None, None, None, 0, 1, 1, 1, 0, None)        # checked wrapper raises


def test_geometry_and_packing(lib):
    from self_driving_model_amd.hip import conv as hc
    s = hc.ConvSpec(64, 128, 3, 2, 1)
    g = hc.fwd_geom(s, 2, 45, 80, 64, 128, 2)
    assert (g.MH, g.MW, g.ntaps, g.krun, g.N) == (23, 40, 9, 64, 128)
    assert list(g.dy)[:9] == [-1, -1, -1, 0, 0, 0, 1, 1, 1] and list(g.dx)[:3] == [-1, 0, 1]
    plans = hc.dgrad_plans(s, 2, 45, 80, 64, 128, 2)
    assert [len(t) for _, t in plans] == [1, 2, 2, 4] and sum(len(t) for _, t in plans) == 9
    assert sum(p.MH * p.MW for p, _ in plans) == 45 * 80
    # 1x1 stride-2 projection: only the even/even class has a tap, the other three write zeros
    plans = hc.dgrad_plans(hc.ConvSpec(64, 128, 1, 2, 0), 1, 46, 80, 64, 128, 2)
    assert [len(t) for _, t in plans] == [1, 0, 0, 0]
    # 3-channel stride-2 first layers run on the space-to-depth(2) image: 7x7/p3 -> 4x4 taps, 5x5/p2 -> 3x3 taps
    first = hc.ConvSpec(3, 64, 7, 2, 3, first=True)
    assert hc.first_layer_taps(first) == (-2, 4) and hc.first_layer_taps(hc.ConvSpec(3, 32, 5, 2, 2, first=True)) == (-1, 3)
    gf = hc.fwd_geom(first, 1, 360, 640, 16, 64, 2, orig_hw=(720, 1280))
    assert (gf.MH, gf.MW, gf.IH, gf.IW, gf.ntaps, gf.krun, gf.pix_shift) == (360, 640, 360, 640, 4, 64, 4)
    assert list(gf.dy)[:4] == [-2, -1, 0, 1] and list(gf.dx)[:4] == [-2] * 4
    w = torch.arange(64 * 3 * 7 * 7, dtype=torch.float32).reshape(64, 3, 7, 7) + 1
    wp = hc.pack_fwd(w, first, torch.float32)
    assert wp.shape == (64, 4 * 4 * 16) and int((wp != 0).sum()) == 64 * 147   # every weight lands exactly once
    assert torch.equal(hc.unpack_wgrad(wp, first, torch.float32), w)                      # pack/unpack are inverse
    p5 = hc.ConvSpec(3, 32, 5, 2, 2, first=True)
    w5 = torch.randn(32, 3, 5, 5)
    wp5 = hc.pack_fwd(w5, p5, torch.float32)
    assert wp5.shape == (32, 3 * 4 * 16) and torch.equal(hc.unpack_wgrad(wp5, p5, torch.float32), w5)
    # space-to-depth conv == the original conv (CPU check of the index algebra)
    img = torch.randn(2, 3, 20, 28)
    ref = torch.nn.functional.conv2d(img, w5, stride=2, padding=2)
    s2d = img.reshape(2, 3, 10, 2, 14, 2).permute(0, 2, 4, 3, 5, 1).reshape(2, 10, 14, 12)
    s2d = torch.nn.functional.pad(s2d, (0, 4))
    off0, taps = hc.first_layer_taps(p5)
    pad = torch.nn.functional.pad(s2d, (0, 0, -off0, 4, -off0, 4))
    wk = wp5.reshape(32, taps, 4, 16)
    out = torch.zeros(2, 10, 14, 32)
    for i in range(taps):
        for j in range(4):
            out += torch.einsum("byxc,nc->byxn", pad[:, i:i + 10, j:j + 14, :], wk[:, i, j, :])
    assert torch.allclose(out.permute(0, 3, 1, 2), ref, atol=1e-4)
    w3 = torch.randn(128, 64, 3, 3)
    assert torch.equal(hc.unpack_wgrad(hc.pack_fwd(w3, s, torch.float32), s, torch.float32), w3)
    assert hc.channel_ld(14, 2) == 32 and hc.channel_ld(19, 4) == 32 and hc.channel_ld(3, 4) == 16 and hc.channel_ld(256, 2) == 256


AUTOMOE_CFG = {"experts": [{"type": "detection", "num_classes": 10, "output_dim": 256, "pretrained_backbone": False},
                           {"type": "segmentation", "num_classes": 19, "output_dim": 256, "pretrained_backbone": False},
                           {"type": "drivable", "num_classes": 3, "output_dim": 256, "pretrained_backbone": False}],
               "gating": {"processed_dim": 256, "hidden_dim": 128, "temperature": 1.0, "use_softmax": True},
               "context": {"type": "simple", "context_dim": 64}, "policy": {"num_waypoints": 10}}


def test_dropin_surface_and_state_dict_keys(lib):
    from oracle import torch_ref as oref
    from self_driving_model_amd.models.automoe import AutoMoE, create_automoe_model
    from self_driving_model_amd.models.experts import BDDDetectionExpert, BDDDrivableExpert, BDDSegmentationExpert
    from self_driving_model_amd.training import HungarianMatcher
    m = create_automoe_model(AUTOMOE_CFG, "cpu")
    ref = oref.create_automoe_model(AUTOMOE_CFG, "cpu")
    assert isinstance(m, AutoMoE)
    assert list(m.state_dict().keys()) == list(ref.state_dict().keys())
    assert [tuple(v.shape) for v in m.state_dict().values()] == [tuple(v.shape) for v in ref.state_dict().values()]
    m.load_state_dict(ref.state_dict(), strict=True)
    assert sum(p.numel() for p in BDDDetectionExpert(10, False).parameters()) == 12_360_014
    assert sum(p.numel() for p in BDDSegmentationExpert(19, False).parameters()) == 12_361_299
    assert sum(p.numel() for p in BDDDrivableExpert(3, False).parameters()) == 12_357_187
    assert BDDDetectionExpert(7, False).num_classes == 7
    for k in ("backbone.0.weight", "backbone.1.num_batches_tracked", "backbone.5.0.downsample.1.running_var", "head.2.bias"):
        assert k in BDDDetectionExpert(10, False).state_dict()
    assert "decoder.0.weight" in BDDDrivableExpert(3, False).state_dict()
    # reference error behaviour
    with pytest.raises(ValueError):
        create_automoe_model(dict(AUTOMOE_CFG, experts=[{"type": "lidar"}]), "cpu")
    with pytest.raises(ValueError):
        create_automoe_model(dict(AUTOMOE_CFG, context={"type": "bogus"}), "cpu")
    with pytest.raises(ValueError):
        m.load_expert_checkpoints(["only-one"])
    with pytest.raises(RuntimeError):
        BDDDetectionExpert()  # pretrained_backbone=True needs a network fetch; no local weights configured
    with pytest.raises(AssertionError):
        HungarianMatcher(0, 0, 0)
    m.freeze_experts()
    assert not any(p.requires_grad for p in m.experts.parameters()) and all(p.requires_grad for p in m.gating_network.parameters())
    m.unfreeze_experts()
    assert all(p.requires_grad for p in m.experts.parameters())
    # wrapper keeps the reference's 'module.' checkpoint prefix
    from self_driving_model_amd.training.ddp import DataParallel
    assert all(k.startswith("module.") for k in DataParallel(m).state_dict())


def test_product_path_refuses_cpu_tensors(lib):
    """No CPU fallback: running the product modules on CPU tensors must fail loudly, never route elsewhere."""
    from self_driving_model_amd.models.automoe import create_automoe_model
    from self_driving_model_amd.training import HungarianMatcher
    m = create_automoe_model(AUTOMOE_CFG, "cpu")
    with pytest.raises(RuntimeError, match="no CPU fallback"):
        m({"image": torch.zeros(1, 3, 64, 64), "speed": torch.zeros(1, 1)})
    with pytest.raises(RuntimeError, match="no CPU fallback"):
        HungarianMatcher()({"pred_logits": torch.zeros(1, 5, 3), "pred_boxes": torch.zeros(1, 5, 4)},
                           [{"boxes": torch.zeros(1, 4), "labels": torch.zeros(1, dtype=torch.int64)}])
    src = "".join(open(os.path.join(dp, f)).read() for dp, _, fs in os.walk(os.path.join(ROOT, "self-driving-model_amd"))
                  for f in fs if f.endswith(".py"))
    assert "import oracle" not in src and "from oracle" not in src


def test_gating_losses_host_logic_matches_golden(golden_dir):
    """compute_gating_losses is device-agnostic glue: check it on CPU against the reference's own values."""
    from _seeded import seeded_tensor
    from self_driving_model_amd.training.train_gating_network import compute_gating_losses
    g = np.load(os.path.join(golden_dir, "gating_losses.npz"))
    B, H, E = 6, 10, 3
    w = torch.softmax(seeded_tensor((B, E), 600), dim=1)
    pred = {"waypoints": seeded_tensor((B, H, 2), 601), "speed_seq": seeded_tensor((B, H), 602), "expert_weights": w}
    pred["speed"] = pred["speed_seq"][:, -1:].contiguous()
    twp, tspd = seeded_tensor((B, H, 2), 603), seeded_tensor((B, H), 604)
    cfg = {"ade_weight": 1.0, "fde_weight": 2.0, "speed_weight": 0.2, "smoothness_weight": 0.1,
           "load_balancing_weight": 0.01, "entropy_weight": 0.001}
    cases = {"seq": (pred, cfg), "last": ({k: v for k, v in pred.items() if k != "speed_seq"}, cfg),
             "noaux": (pred, dict(cfg, use_load_balancing=False, use_entropy_loss=False))}
    for tag, (p, c) in cases.items():
        for k, v in compute_gating_losses(p, twp, tspd, c).items():
            np.testing.assert_allclose(float(v), float(g[f"{tag}/{k}"]), rtol=1e-5, atol=1e-7, err_msg=f"{tag}/{k}")


_DDP_WORKER = r'''
import os, sys, torch, torch.distributed as dist
sys.path.insert(0, sys.argv[1])
from self_driving_model_amd.training.ddp import GradBucketReducer
rank, world = int(os.environ["RANK"]), int(os.environ["WORLD_SIZE"])
dist.init_process_group("gloo", init_method="env://")
torch.manual_seed(0)
model = torch.nn.Sequential(torch.nn.Linear(16, 64), torch.nn.ReLU(), torch.nn.Linear(64, 64), torch.nn.ReLU(), torch.nn.Linear(64, 4))
unused = torch.nn.Parameter(torch.zeros(7))            # never touched by forward: its bucket is reduced in finish()
params = list(model.parameters()) + [unused]
offsets, off = [], 0
for p in params:
    offsets.append(off); off += (p.numel() + 3) // 4 * 4
flat_p, flat_g = torch.zeros(off), torch.zeros(off)
for p, o in zip(params, offsets):
    flat_p[o:o + p.numel()].copy_(p.data.reshape(-1) + rank)   # ranks start different; broadcast must fix it
    p.data = flat_p[o:o + p.numel()].view(p.shape)
    p.grad = flat_g[o:o + p.numel()].view(p.shape)
red = GradBucketReducer(params, offsets, flat_g, bucket_bytes=8 * 1024, broadcast_from=flat_p)
assert red.enabled and red.world == world and len(red.buckets) >= 2
ref = [p.detach().clone() for p in params]
gathered = [torch.zeros_like(flat_p) for _ in range(world)]
dist.all_gather(gathered, flat_p)
assert all(torch.equal(g, gathered[0]) for g in gathered), "constructor broadcast failed"
for it in range(2):                                     # two steps: reducer state must reset
    flat_g.zero_()
    g = torch.Generator().manual_seed(100 * it + rank)  # per-rank shard of the global batch
    x, y = torch.randn(8, 16, generator=g), torch.randn(8, 4, generator=g)
    ((model(x) - y) ** 2).mean().backward()
    red.finish()
    mean_grad = flat_g / world
    # reference: the same model on the concatenated global batch
    tot = torch.zeros(off)
    for r in range(world):
        g2 = torch.Generator().manual_seed(100 * it + r)
        x2, y2 = torch.randn(8, 16, generator=g2), torch.randn(8, 4, generator=g2)
        m2 = torch.nn.Sequential(torch.nn.Linear(16, 64), torch.nn.ReLU(), torch.nn.Linear(64, 64), torch.nn.ReLU(), torch.nn.Linear(64, 4))
        m2.load_state_dict(model.state_dict())
        ((m2(x2) - y2) ** 2).mean().backward()
        for p2, o in zip(m2.parameters(), offsets):
            tot[o:o + p2.numel()] += p2.grad.reshape(-1)
    assert torch.allclose(mean_grad, tot / world, rtol=1e-5, atol=1e-6), float((mean_grad - tot / world).abs().max())
dist.barrier()
dist.destroy_process_group()
print("ddp-ok", rank)
'''


def test_grad_bucket_reducer_gloo_world2(tmp_path):
    script = tmp_path / "ddp_worker.py"
    script.write_text(_DDP_WORKER)
    env = dict(os.environ, MASTER_ADDR="127.0.0.1", OMP_NUM_THREADS="1")
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node=2", "--master-addr", "127.0.0.1",
           "--master-port", "29641", str(script), ROOT]
    r = subprocess.run(cmd, env=env, capture_output=True, text=True, timeout=300)
    assert r.returncode == 0, r.stdout[-2000:] + r.stderr[-3000:]
    assert r.stdout.count("ddp-ok") == 2


def test_validation_metrics_match_the_reference_loops():
    """training/metrics.py (batched device arithmetic, no host sync) against the oracle's restatement of the reference's per-image /
    per-class loops (training/train_bdd100k_ddp.py:267-291, 299-325): detection avg IoU + recall@0.5 with ragged target counts,
    an image without targets and degenerate boxes; segmentation pixel accuracy + mean IoU with ignored pixels and an absent class."""
    import torch
    from oracle import losses as ol
    from oracle.matcher import HungarianMatcher as OM, box_cxcywh_to_xyxy
    from self_driving_model_amd.training import metrics as M
    g = torch.Generator().manual_seed(5)
    B, Q, Nmax = 4, 40, 6
    pred = torch.rand(B, Q, 4, generator=g) * torch.tensor([100.0, 60.0, 40.0, 30.0]) + torch.tensor([0.0, 0.0, 2.0, 2.0])
    n_tgt = torch.tensor([6, 3, 0, 1])
    tgt = torch.rand(B, Nmax, 4, generator=g) * torch.tensor([100.0, 60.0, 40.0, 30.0]) + torch.tensor([0.0, 0.0, 2.0, 2.0])
    tgt[1, 0] = pred[1, 7]       # an exact hit: IoU 1
    tgt[0, 1] = pred[0, 3] + torch.tensor([1.0, -1.0, 0.5, 0.0])
    logits = torch.randn(B, Q, 5, generator=g)
    labels = torch.randint(0, 5, (B, Nmax), generator=g)
    targets = [{"boxes": tgt[b, : int(n_tgt[b])], "labels": labels[b, : int(n_tgt[b])]} for b in range(B)]
    indices = OM(1.0, 5.0, 2.0)({"pred_logits": logits, "pred_boxes": pred}, targets)
    k = max(1, int(n_tgt.max()))
    rows = torch.full((B, k), -1, dtype=torch.int64); cols = torch.full((B, k), -1, dtype=torch.int64); count = torch.zeros(B, dtype=torch.int64)
    for b, (pi, ti) in enumerate(indices):
        rows[b, : pi.numel()], cols[b, : ti.numel()], count[b] = pi, ti, pi.numel()
    got = M.detection_metrics(pred, tgt, n_tgt, rows, cols, count)
    want = ol.detection_val_metrics(pred, targets, indices)
    for key in want:
        assert abs(float(got[key]) - want[key]) < 1e-6, (key, float(got[key]), want[key])
    assert want["avg_iou"] > 0 and want["recall_0.5"] > 0
    none = M.detection_metrics(pred, tgt, torch.zeros(B, dtype=torch.int64), rows, cols, torch.zeros(B, dtype=torch.int64))
    assert float(none["avg_iou"]) == 0.0 and float(none["recall_0.5"]) == 0.0
    # segmentation
    out = torch.randn(2, 6, 17, 23, generator=g)
    masks = torch.randint(0, 5, (2, 17, 23), generator=g)  # class 5 never present
    masks[torch.rand(2, 17, 23, generator=g) < 0.1] = 255
    got = M.segmentation_metrics(out, masks)
    want = ol.segmentation_val_metrics(out, masks)
    for key in want:
        assert abs(float(got[key]) - want[key]) < 1e-6, (key, float(got[key]), want[key])
