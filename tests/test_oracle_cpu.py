"""CPU tests: the oracle against the golden vectors generated from the reference
(tests/golden/make_golden.py) and against scipy.  These pin the oracle; the HIP parity tests
(-m gpu) then compare the product against the oracle."""
import os

import numpy as np
import pytest
import torch

from _seeded import seed_module_, seeded_tensor
from oracle import losses as olosses
from oracle import matcher as omatcher
from oracle import torch_ref as oref

RTOL, ATOL = 1e-5, 1e-6  # same library, same arithmetic: only thread-count reassociation may differ


def _load(golden_dir, name):
    return np.load(os.path.join(golden_dir, f"{name}.npz"))


def _close(a, b, rtol=RTOL, atol=ATOL):
    np.testing.assert_allclose(np.asarray(a, dtype=np.float64), np.asarray(b, dtype=np.float64), rtol=rtol, atol=atol)


def _check_grad_summary(module, g, tag):
    for n, p in module.named_parameters():
        grad = p.grad if p.grad is not None else torch.zeros_like(p)
        _close(grad.double().sum().numpy(), g[f"{tag}/gsum/{n}"], rtol=1e-4, atol=1e-5)
        _close(grad.double().pow(2).sum().sqrt().numpy(), g[f"{tag}/gl2/{n}"], rtol=1e-4, atol=1e-6)


GATING_VARIANTS = {
    "e3": dict(num_experts=3), "e4": dict(num_experts=4),
    "e3_sigmoid": dict(num_experts=3, use_softmax=False, temperature=1.0),
    "e3_temp": dict(num_experts=3, temperature=0.5),
    "e4_topk2": dict(num_experts=4, top_k=2, noise_scale=0.0, apply_topk_at_eval=True),
}


@pytest.mark.parametrize("tag", list(GATING_VARIANTS))
def test_gating_matches_reference(golden_dir, tag):
    g = _load(golden_dir, "gating")
    kw = GATING_VARIANTS[tag]
    E = kw["num_experts"]
    m = oref.GatingNetwork(context_dim=64, expert_output_dims=[256] * E, processed_dim=256, hidden_dim=128, **kw)
    seed_module_(m, 100 + E).eval()
    xs = [seeded_tensor((4, 256), 200 + i).requires_grad_() for i in range(E)]
    ctx = seeded_tensor((4, 64), 300).requires_grad_()
    o = m(xs, ctx)
    loss = (o["combined_output"] * seeded_tensor((4, 256), 301)).sum() + (o["expert_weights"] * seeded_tensor((4, E), 302)).sum()
    loss.backward()
    _close(o["combined_output"].detach(), g[f"{tag}/combined_output"])
    _close(o["expert_weights"].detach(), g[f"{tag}/expert_weights"])
    _close(o["gate_logits"].detach(), g[f"{tag}/gate_logits"])
    _close(torch.stack(o["processed_expert_outputs"]).detach(), g[f"{tag}/processed"])
    _close(ctx.grad, g[f"{tag}/d_ctx"], rtol=1e-4)
    _close(torch.stack([x.grad for x in xs]), g[f"{tag}/d_x"], rtol=1e-4)
    _check_grad_summary(m, g, tag)
    _close(m.get_expert_weights(ctx.detach()).detach(), g[f"{tag}/ctx_only_weights"])
    _close(m.get_gating_logits(ctx.detach()).detach(), g[f"{tag}/ctx_only_logits"])
    # the reference's own invariants (tests/test_gating_network.py:76-80)
    w = o["expert_weights"].detach()
    assert torch.allclose(w.sum(dim=1), torch.ones(4), atol=1e-6) and (w >= 0).all()


@pytest.mark.parametrize("tag,shape,train", [("small_train", (2, 64, 96), True), ("small_eval", (2, 64, 96), False)])
def test_policy_matches_reference(golden_dir, tag, shape, train):
    g = _load(golden_dir, "policy")
    B, H, W = shape
    m = seed_module_(oref.TrajectoryPolicy(horizon=10, context_dim=256, backbone_dim=512), 400)
    m.train(train)
    ctx = seeded_tensor((B, 256), 402).requires_grad_()
    o = m(seeded_tensor((B, 3, H, W), 401), context=ctx)
    _close(o["waypoints"].detach(), g[f"{tag}/waypoints"], rtol=1e-4, atol=1e-5)
    _close(o["speed"].detach(), g[f"{tag}/speed"], rtol=1e-4, atol=1e-5)
    ((o["waypoints"] * seeded_tensor((B, 10, 2), 403)).sum() + (o["speed"] * seeded_tensor((B, 10), 404)).sum()).backward()
    _close(ctx.grad, g[f"{tag}/d_ctx"], rtol=1e-4, atol=1e-5)
    _close(m.backbone.net[0].weight.grad, g[f"{tag}/d_conv0_w"], rtol=1e-3, atol=1e-4)
    _close(m.backbone.net[1].weight.grad, g[f"{tag}/d_bn0_w"], rtol=1e-3, atol=1e-4)
    _close(m.backbone.net[10].running_mean, g[f"{tag}/bn3_running_mean"], rtol=1e-4, atol=1e-5)
    _close(m.backbone.net[10].running_var, g[f"{tag}/bn3_running_var"], rtol=1e-4, atol=1e-5)


def test_policy_hd_eval_matches_reference(golden_dir):
    g = _load(golden_dir, "policy")
    m = seed_module_(oref.TrajectoryPolicy(horizon=10, context_dim=256, backbone_dim=512), 400).eval()
    with torch.no_grad():
        o = m(seeded_tensor((1, 3, 720, 1280), 401), context=seeded_tensor((1, 256), 402))
    _close(o["waypoints"], g["hd_eval/waypoints"], rtol=1e-4, atol=1e-5)
    _close(o["speed"], g["hd_eval/speed"], rtol=1e-4, atol=1e-5)


def test_extractors_and_context_match_reference(golden_dir):
    g = _load(golden_dir, "extractors")
    det = seed_module_(oref.DetectionExpertExtractor(256, 10), 500).eval()
    seg = seed_module_(oref.SegmentationExpertExtractor(256, 19), 501).eval()
    drv = seed_module_(oref.DrivableExpertExtractor(256, 3), 502).eval()
    cl, bd = seeded_tensor((3, 10, 6, 10), 510).requires_grad_(), seeded_tensor((3, 4, 6, 10), 511).requires_grad_()
    sx, dx = seeded_tensor((3, 19, 24, 40), 512).requires_grad_(), seeded_tensor((3, 3, 24, 40), 513).requires_grad_()
    probe = seeded_tensor((3, 256), 514)
    for tag, m, y, ins in (("det", det, det({"class_logits": cl, "bbox_deltas": bd}), (cl, bd)),
                           ("seg", seg, seg(sx), (sx,)), ("drv", drv, drv(dx), (dx,))):
        (y * probe).sum().backward()
        _close(y.detach(), g[f"{tag}/features"])
        for i, t in enumerate(ins):
            _close(t.grad, g[f"{tag}/d_in{i}"], rtol=1e-4)
        _check_grad_summary(m, g, tag)
    c = seed_module_(oref.SimpleContextExtractor(64), 520).eval()
    ins = [seeded_tensor((5, 1), 521 + i).requires_grad_() for i in range(4)]
    y = c(*ins)
    (y * seeded_tensor((5, 64), 530)).sum().backward()
    _close(y.detach(), g["ctx/features"])
    _close(torch.cat([t.grad for t in ins], dim=1), g["ctx/d_in"], rtol=1e-4)


def test_gating_losses_match_reference(golden_dir):
    g = _load(golden_dir, "gating_losses")
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
        r = olosses.gating_losses(p, twp, tspd, c)
        for k, v in r.items():
            _close(v.detach(), g[f"{tag}/{k}"], rtol=1e-6, atol=1e-7)


def _lsap_case_names(g):
    return sorted({k.split("/")[0] for k in g.files})


def test_lsap_c_matches_scipy_golden(golden_dir):
    g = _load(golden_dir, "lsap_cases")
    for name in _lsap_case_names(g):
        r, c = omatcher.lsap_c(g[f"{name}/cost"])
        assert np.array_equal(r, g[f"{name}/rows"]) and np.array_equal(c, g[f"{name}/cols"]), name
        assert r.dtype == np.int64 and c.dtype == np.int64


def test_lsap_c_matches_scipy_live():
    from scipy.optimize import linear_sum_assignment
    rng = np.random.default_rng(123)
    for trial in range(200):
        nr, nc = int(rng.integers(1, 80)), int(rng.integers(1, 80))
        kind = trial % 4
        if kind == 0:
            m = rng.standard_normal((nr, nc))
        elif kind == 1:
            m = rng.integers(0, 3, size=(nr, nc)).astype(np.float64)  # heavy ties
        elif kind == 2:
            m = rng.standard_normal((nr, nc)).astype(np.float32)
        else:
            m = np.round(rng.standard_normal((nr, nc)), 1)
        r0, c0 = linear_sum_assignment(m)
        r1, c1 = omatcher.lsap_c(m)
        assert np.array_equal(r0, r1) and np.array_equal(c0, c1), (trial, nr, nc)


def test_lsap_c_invalid_entries():
    m = np.zeros((3, 3)); m[1, 1] = np.nan
    with pytest.raises(ValueError):
        omatcher.lsap_c(m)
    m[1, 1] = -np.inf
    with pytest.raises(ValueError):
        omatcher.lsap_c(m)
    m = np.full((2, 2), np.inf)
    with pytest.raises(ValueError):
        omatcher.lsap_c(m)  # infeasible, as scipy
    r, c = omatcher.lsap_c(np.zeros((5, 0)))
    assert r.size == 0 and c.size == 0 and r.dtype == np.int64


def test_box_ops_hand_cases():
    b = torch.tensor([[10.0, 20.0, 4.0, 6.0]])
    assert torch.equal(omatcher.box_cxcywh_to_xyxy(b), torch.tensor([[8.0, 17.0, 12.0, 23.0]]))
    assert torch.equal(omatcher.box_xyxy_to_cxcywh(omatcher.box_cxcywh_to_xyxy(b)), b)
    a = torch.tensor([[0.0, 0.0, 2.0, 2.0]])
    c = torch.tensor([[1.0, 1.0, 3.0, 3.0], [4.0, 4.0, 5.0, 5.0], [0.0, 0.0, 2.0, 2.0]])
    giou = omatcher.generalized_box_iou(a, c)
    # overlap 1, union 7, hull 9 -> 1/7 - 2/9 ; disjoint: 0 - (25-5)/25 ; identical: 1
    _close(giou, [[1 / 7 - 2 / 9, -20 / 25, 1.0]], rtol=1e-6)


def test_resnet_trunk_structure():
    """torchvision is absent: structural pins only: param count and key names."""
    e = oref.BDDDetectionExpert(10, pretrained_backbone=False)
    n_trunk = sum(p.numel() for p in e.backbone.parameters())
    assert n_trunk == 11_176_512
    assert sum(p.numel() for p in e.parameters()) == 12_360_014
    assert sum(p.numel() for p in oref.BDDSegmentationExpert(19, False).parameters()) == 12_361_299
    assert sum(p.numel() for p in oref.BDDDrivableExpert(3, False).parameters()) == 12_357_187
    keys = set(e.state_dict().keys())
    for k in ("backbone.0.weight", "backbone.1.running_mean", "backbone.1.num_batches_tracked",
              "backbone.4.0.conv1.weight", "backbone.4.1.bn2.bias", "backbone.5.0.downsample.0.weight",
              "backbone.5.0.downsample.1.running_var", "backbone.7.1.conv2.weight", "head.0.weight", "head.2.bias"):
        assert k in keys, k
    assert "backbone.4.0.downsample.0.weight" not in keys
    with torch.no_grad():
        y = e.eval()(torch.zeros(1, 3, 64, 96))
    assert y["class_logits"].shape == (1, 10, 2, 3) and y["bbox_deltas"].shape == (1, 4, 2, 3)
    with pytest.raises(RuntimeError):
        oref.BDDDetectionExpert()  # pretrained default needs a fetch


def test_config1_seg_expert_forward_cpu():
    """BASELINE config 1: segmentation expert forward on one 3x256x256 tensor, CPU."""
    torch.manual_seed(0)
    m = oref.BDDSegmentationExpert(19, pretrained_backbone=False).eval()
    with torch.no_grad():
        y = m(torch.randn(1, 3, 256, 256))
    assert y.shape == (1, 19, 256, 256) and torch.isfinite(y).all()


def test_automoe_oracle_shapes_and_invariants():
    cfg = {"experts": [{"type": "detection", "num_classes": 10, "output_dim": 256, "pretrained_backbone": False},
                       {"type": "segmentation", "num_classes": 19, "output_dim": 256, "pretrained_backbone": False},
                       {"type": "drivable", "num_classes": 3, "output_dim": 256, "pretrained_backbone": False}],
           "gating": {"processed_dim": 256, "hidden_dim": 128, "temperature": 1.0, "use_softmax": True},
           "context": {"type": "simple", "context_dim": 64}, "policy": {"num_waypoints": 10}}
    m = oref.create_automoe_model(cfg, "cpu").eval()
    B = 2
    batch = {"image": torch.randn(B, 3, 64, 96), "speed": torch.randn(B, 10), "steering": torch.randn(B, 10),
             "throttle": torch.randn(B, 10), "brake": torch.randn(B, 10)}
    with torch.no_grad():
        o = m(batch)
    assert o["waypoints"].shape == (B, 10, 2) and o["speed"].shape == (B, 1) and o["speed_seq"].shape == (B, 10)
    assert o["expert_weights"].shape == (B, 3) and o["context_features"].shape == (B, 64)
    assert o["combined_features"].shape == (B, 256)
    assert torch.allclose(o["expert_weights"].sum(dim=1), torch.ones(B), atol=1e-6)
    n_train = sum(p.numel() for n, p in m.named_parameters() if not n.startswith("experts."))
    assert n_train == 415_488 + 2_400 + 602_115 + 1_850_654
    with pytest.raises(ValueError):
        oref.create_automoe_model(dict(cfg, experts=[{"type": "lidar"}]), "cpu")


# ---- NuScenes expert head / extractor, matcher box dimensions ----
def _check_param_grads(m, g, tag):
    for n, p in m.named_parameters():
        l2 = float(g[f"{tag}/gl2/{n}"])
        got = p.grad if p.grad is not None else torch.zeros_like(p)
        np.testing.assert_allclose(got.double().sum().numpy(), g[f"{tag}/gsum/{n}"], rtol=1e-4, atol=1e-5 * (1 + l2 * p.numel() ** 0.5), err_msg=n)
        np.testing.assert_allclose(got.double().pow(2).sum().sqrt().numpy(), g[f"{tag}/gl2/{n}"], rtol=1e-4, atol=1e-6, err_msg=n)


def test_nuscenes_extractor_and_head_match_reference(golden_dir):
    """oracle NuScenesExpertExtractor vs the reference module; oracle NuScenesExpert decoder + heads vs the reference class
    compiled from source with a caller-supplied image backbone (tests/golden/make_golden_nuscenes.py)."""
    g = _load(golden_dir, "nuscenes")
    for D in (4, 7):
        m = seed_module_(oref.NuScenesExpertExtractor(256, num_queries=12, num_classes=10, bbox_dim=D), 700 + D).eval()
        cl, bb = seeded_tensor((3, 12, 10), 710 + D).requires_grad_(), seeded_tensor((3, 12, D), 720 + D).requires_grad_()
        y = m({"class_logits": cl, "bbox_preds": bb})
        (y * seeded_tensor((3, 256), 730)).sum().backward()
        np.testing.assert_allclose(y.detach().numpy(), g[f"ext{D}/features"], rtol=1e-5, atol=1e-6)
        np.testing.assert_allclose(cl.grad.numpy(), g[f"ext{D}/d_cls"], rtol=1e-4, atol=1e-7)
        np.testing.assert_allclose(bb.grad.numpy(), g[f"ext{D}/d_box"], rtol=1e-4, atol=1e-7)
        _check_param_grads(m, g, f"ext{D}")
    for D, Q in ((7, 12), (4, 196)):
        m = seed_module_(oref.NuScenesExpert(image_backbone=torch.nn.Identity(), num_queries=Q, bbox_dim=D), 740 + D).eval()
        feat = seeded_tensor((3, 256), 750 + D).requires_grad_()
        o = m({"image": feat})
        (o["class_logits"] * seeded_tensor((3, Q, 10), 760)).sum().add((o["bbox_preds"] * seeded_tensor((3, Q, D), 761)).sum()).backward()
        np.testing.assert_allclose(o["class_logits"].detach().numpy(), g[f"head{D}/class_logits"], rtol=1e-5, atol=1e-6)
        np.testing.assert_allclose(o["bbox_preds"].detach().numpy(), g[f"head{D}/bbox_preds"], rtol=1e-5, atol=1e-6)
        np.testing.assert_allclose(feat.grad.numpy(), g[f"head{D}/d_feat"], rtol=1e-4, atol=1e-6)
        _check_param_grads(m, g, f"head{D}")


@pytest.mark.parametrize("D", [7, 5, 4])
def test_matcher_box_dimensions_match_reference(golden_dir, D):
    """oracle cost matrix + C LSAP vs the reference HungarianMatcher.forward compiled from source (D = 7 BEV branch,
    D = 5 no-GIoU branch, D = 4): cost matrices as handed to scipy, and the indices it returned."""
    g = _load(golden_dir, "matcher_dims")
    logits, boxes = torch.from_numpy(g[f"d{D}/logits"]), torch.from_numpy(g[f"d{D}/boxes"])
    targets = [{"boxes": torch.from_numpy(g[f"d{D}/tgt_boxes{b}"]), "labels": torch.from_numpy(g[f"d{D}/tgt_labels{b}"])} for b in range(3)]
    for b in range(3):
        C = omatcher.cost_matrix(logits[b], boxes[b], targets[b]["labels"], targets[b]["boxes"], 1.0, 5.0, 2.0).numpy()
        np.testing.assert_allclose(C, g[f"d{D}/cost{b}"], rtol=1e-5, atol=1e-5)
    idx = omatcher.HungarianMatcher(1.0, 5.0, 2.0)({"pred_logits": logits, "pred_boxes": boxes}, targets)
    for b in range(3):
        np.testing.assert_array_equal(idx[b][0].numpy(), g[f"d{D}/rows{b}"])
        np.testing.assert_array_equal(idx[b][1].numpy(), g[f"d{D}/cols{b}"])


def test_four_expert_reference_config_oracle():
    """The reference's own 4-expert model_config.json shape (with the pretrained fetch turned off): construction, keys,
    one forward at a small size."""
    cfg = {"experts": [{"type": "detection", "num_classes": 10, "output_dim": 256, "pretrained_backbone": False},
                       {"type": "segmentation", "num_classes": 19, "output_dim": 256, "pretrained_backbone": False},
                       {"type": "drivable", "num_classes": 3, "output_dim": 256, "pretrained_backbone": False},
                       {"type": "nuscenes", "num_queries": 196, "num_classes": 10, "output_dim": 256, "fusion": "sum", "use_lidar": False,
                        "use_tnet": False, "bbox_dim": 4, "pretrained_backbone": False}],
           "gating": {"processed_dim": 256, "hidden_dim": 128, "temperature": 1.0, "use_softmax": True, "top_k": 2},
           "context": {"type": "simple", "context_dim": 64}, "policy": {"hidden_dim": 256, "num_waypoints": 10, "waypoint_dim": 2}}
    m = oref.create_automoe_model(cfg).eval()
    assert sum(p.numel() for p in m.parameters()) == 53_109_940
    sd = m.state_dict()
    for k in ("experts.3.image_backbone.0.weight", "experts.3.image_projection.weight", "experts.3.query_embed.weight",
              "experts.3.decoder.3.bias", "experts.3.class_head.weight", "experts.3.bbox_head.bias",
              "expert_extractors.extractors.3.feature_extractor.0.weight", "expert_extractors.extractors.3.feature_extractor.4.bias"):
        assert k in sd, k
    with torch.no_grad():
        o = m({"image": seeded_tensor((1, 3, 64, 96), 5), "speed": seeded_tensor((1, 10), 6)})
    assert o["expert_weights"].shape == (1, 4) and abs(float(o["expert_weights"].sum()) - 1) < 1e-6
    assert o["expert_outputs"][3]["class_logits"].shape == (1, 196, 10) and o["expert_outputs"][3]["bbox_preds"].shape == (1, 196, 4)


# ---- CARLA trainers' loss glue ----
def test_policy_losses_match_reference(golden_dir):
    """oracle.policy_losses and the product's torch-op `compute_losses` vs the reference function compiled from source."""
    from self_driving_model_amd.training.train_carla_policy import compute_losses
    g = _load(golden_dir, "policy_losses")
    for tag, (B, T) in {"b6t8": (6, 8), "b32t10": (32, 10), "b3t3": (3, 3)}.items():
        for fn in (olosses.policy_losses, compute_losses):
            wp = seeded_tensor((B, T, 2), 900 + B).requires_grad_()
            spd = seeded_tensor((B, T), 901 + B).requires_grad_()
            r = fn({"waypoints": wp, "speed": spd}, seeded_tensor((B, T, 2), 902 + B), seeded_tensor((B, T), 903 + B))
            r["loss"].backward()
            for k in ("loss", "ade", "fde", "speed", "smooth"):
                np.testing.assert_allclose(r[k].detach().double().numpy(), g[f"{tag}/{k}"], rtol=1e-6, atol=1e-7, err_msg=f"{tag}/{k}")
            np.testing.assert_allclose(wp.grad.numpy(), g[f"{tag}/d_wp"], rtol=1e-5, atol=1e-8)
            np.testing.assert_allclose(spd.grad.numpy(), g[f"{tag}/d_spd"], rtol=1e-5, atol=1e-8)


def test_carla_mask_sanitising_and_empty_detection_batch():
    from self_driving_model_amd.training.train_carla_bdd_experts_ddp import sanitize_mask
    m = torch.tensor([[[0, 2, 3, -1], [255, 1, 7, 2]]])
    exp = olosses.carla_sanitize_mask(m, 3)
    assert torch.equal(sanitize_mask(m, 3), exp) and exp.tolist() == [[[0, 2, 255, 255], [255, 1, 255, 2]]]
    assert torch.equal(sanitize_mask(m[..., None].expand(-1, -1, -1, 3), 3), exp)  # trailing channel axis dropped
    # a batch without any ground-truth box: both loss terms are exactly 0.0 (the BDD100K trainer's CE would be NaN)
    out = {"class_logits": seeded_tensor((2, 10, 3, 4), 1), "bbox_deltas": seeded_tensor((2, 4, 3, 4), 2)}
    total, cls, box, _ = olosses.carla_detection_loss(out, -torch.ones(2, 5, 4), -torch.ones(2, 5, dtype=torch.int64), 10,
                                                      omatcher.HungarianMatcher())
    assert float(total) == 0.0 and float(cls) == 0.0 and float(box) == 0.0
