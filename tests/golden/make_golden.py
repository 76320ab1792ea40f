#!/usr/bin/env python3
"""Generate the committed golden fixtures (run in the build container, where /root/reference is
mounted; the GPU box never has it).

    python tests/golden/make_golden.py            # writes tests/golden/*.npz

What it pins:
  * reference modules that import here with plain torch -- models/gating/gating_network.py,
    models/policy/trajectory_head.py, models/context/context_features.py,
    models/experts/expert_extractors.py (by file path: its package __init__ pulls torchvision) --
    are run on seeded weights/inputs; outputs and gradients are stored.
  * `compute_gating_losses` (training/train_gating_network.py:21-74): the module itself cannot be
    imported (tensorboard / torchvision missing -- an ordinary ModuleNotFoundError), so that one
    pure-torch function is compiled from the source text and run.
  * scipy.optimize.linear_sum_assignment 1.15.3 (the reference's third-party solver,
    training/hungarian_matcher.py:79) on the edge cases of SURVEY.md section 8(c)(vi).
Only data is written: inputs (or the seed that regenerates them) and expected outputs.
"""
from __future__ import annotations

import ast
import importlib.util
import os
import sys

import numpy as np
import torch
import torch.nn.functional as F

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, HERE)
from _seeded import seed_module_, seeded_tensor  # noqa: E402

REF = os.environ.get("AUTOMOE_REFERENCE", "/root/reference")


def _load(path, name):
    spec = importlib.util.spec_from_file_location(name, os.path.join(REF, path))
    mod = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(mod)
    return mod


def _grad_summary(module):
    out = {}
    for n, p in module.named_parameters():
        g = p.grad if p.grad is not None else torch.zeros_like(p)
        out[f"gsum/{n}"] = g.double().sum().numpy()
        out[f"gl2/{n}"] = g.double().pow(2).sum().sqrt().numpy()
    return out


def gating_cases(gn):
    out = {}
    variants = {
        "e3": dict(num_experts=3), "e4": dict(num_experts=4),
        "e3_sigmoid": dict(num_experts=3, use_softmax=False, temperature=1.0),
        "e3_temp": dict(num_experts=3, temperature=0.5),
        "e4_topk2": dict(num_experts=4, top_k=2, noise_scale=0.0, apply_topk_at_eval=True),
    }
    for tag, kw in variants.items():
        E = kw["num_experts"]
        m = gn.GatingNetwork(context_dim=64, expert_output_dims=[256] * E, processed_dim=256, hidden_dim=128, **kw)
        seed_module_(m, 100 + E)
        m.eval()
        xs = [seeded_tensor((4, 256), 200 + i).requires_grad_() for i in range(E)]
        ctx = seeded_tensor((4, 64), 300).requires_grad_()
        o = m(xs, ctx)
        probe = seeded_tensor((4, 256), 301)
        wprobe = seeded_tensor((4, E), 302)
        loss = (o["combined_output"] * probe).sum() + (o["expert_weights"] * wprobe).sum()
        loss.backward()
        out[f"{tag}/combined_output"] = o["combined_output"].detach().numpy()
        out[f"{tag}/expert_weights"] = o["expert_weights"].detach().numpy()
        out[f"{tag}/gate_logits"] = o["gate_logits"].detach().numpy()
        out[f"{tag}/processed"] = torch.stack(o["processed_expert_outputs"]).detach().numpy()
        out[f"{tag}/d_ctx"] = ctx.grad.numpy()
        out[f"{tag}/d_x"] = torch.stack([x.grad for x in xs]).numpy()
        for k, v in _grad_summary(m).items():
            out[f"{tag}/{k}"] = v
        out[f"{tag}/ctx_only_weights"] = m.get_expert_weights(ctx.detach()).detach().numpy()
        out[f"{tag}/ctx_only_logits"] = m.get_gating_logits(ctx.detach()).detach().numpy()
    return out


def policy_cases(th):
    out = {}
    for tag, (B, H, W, train) in {"small_train": (2, 64, 96, True), "small_eval": (2, 64, 96, False),
                                  "hd_eval": (1, 720, 1280, False)}.items():
        m = th.TrajectoryPolicy(horizon=10, context_dim=256, backbone_dim=512)
        seed_module_(m, 400)
        m.train(train)
        img = seeded_tensor((B, 3, H, W), 401)
        ctx = seeded_tensor((B, 256), 402).requires_grad_()
        o = m(img, context=ctx)
        out[f"{tag}/waypoints"] = o["waypoints"].detach().numpy()
        out[f"{tag}/speed"] = o["speed"].detach().numpy()
        if tag != "hd_eval":
            loss = (o["waypoints"] * seeded_tensor((B, 10, 2), 403)).sum() + (o["speed"] * seeded_tensor((B, 10), 404)).sum()
            loss.backward()
            out[f"{tag}/d_ctx"] = ctx.grad.numpy()
            for k, v in _grad_summary(m).items():
                out[f"{tag}/{k}"] = v
            out[f"{tag}/d_conv0_w"] = m.backbone.net[0].weight.grad.numpy()
            out[f"{tag}/d_bn0_w"] = m.backbone.net[1].weight.grad.numpy()
            out[f"{tag}/d_bn0_b"] = m.backbone.net[1].bias.grad.numpy()
            out[f"{tag}/bn3_running_mean"] = m.backbone.net[10].running_mean.numpy().copy()
            out[f"{tag}/bn3_running_var"] = m.backbone.net[10].running_var.numpy().copy()
    return out


def extractor_cases(ex, cf):
    out = {}
    det = seed_module_(ex.DetectionExpertExtractor(256, 10), 500).eval()
    seg = seed_module_(ex.SegmentationExpertExtractor(256, 19), 501).eval()
    drv = seed_module_(ex.DrivableExpertExtractor(256, 3), 502).eval()
    cl, bd = seeded_tensor((3, 10, 6, 10), 510).requires_grad_(), seeded_tensor((3, 4, 6, 10), 511).requires_grad_()
    sx, dx = seeded_tensor((3, 19, 24, 40), 512).requires_grad_(), seeded_tensor((3, 3, 24, 40), 513).requires_grad_()
    probe = seeded_tensor((3, 256), 514)
    for tag, m, y, ins in (("det", det, det({"class_logits": cl, "bbox_deltas": bd}), (cl, bd)),
                           ("seg", seg, seg(sx), (sx,)), ("drv", drv, drv(dx), (dx,))):
        (y * probe).sum().backward()
        out[f"{tag}/features"] = y.detach().numpy()
        for i, t in enumerate(ins):
            out[f"{tag}/d_in{i}"] = t.grad.numpy()
        for k, v in _grad_summary(m).items():
            out[f"{tag}/{k}"] = v
    c = seed_module_(cf.SimpleContextExtractor(64), 520).eval()
    ins = [seeded_tensor((5, 1), 521 + i).requires_grad_() for i in range(4)]
    y = c(*ins)
    (y * seeded_tensor((5, 64), 530)).sum().backward()
    out["ctx/features"] = y.detach().numpy()
    out["ctx/d_in"] = torch.cat([t.grad for t in ins], dim=1).numpy()
    for k, v in _grad_summary(c).items():
        out[f"ctx/{k}"] = v
    return out


def gating_loss_cases():
    src = open(os.path.join(REF, "training/train_gating_network.py")).read()
    tree = ast.parse(src)
    fn = next(n for n in tree.body if isinstance(n, ast.FunctionDef) and n.name == "compute_gating_losses")
    ns = {"torch": torch, "F": F, "Dict": dict}
    exec(compile(ast.Module(body=[fn], type_ignores=[]), "train_gating_network.py", "exec"), ns)
    f = ns["compute_gating_losses"]
    out = {}
    B, H, E = 6, 10, 3
    w = torch.softmax(seeded_tensor((B, E), 600), dim=1)
    pred = {"waypoints": seeded_tensor((B, H, 2), 601), "speed_seq": seeded_tensor((B, H), 602), "expert_weights": w}
    pred["speed"] = pred["speed_seq"][:, -1:].contiguous()
    twp, tspd = seeded_tensor((B, H, 2), 603), seeded_tensor((B, H), 604)
    cfg = {"ade_weight": 1.0, "fde_weight": 2.0, "speed_weight": 0.2, "smoothness_weight": 0.1,
           "load_balancing_weight": 0.01, "entropy_weight": 0.001}
    for tag, (p, ts, c) in {
        "seq": (pred, tspd, cfg),
        "last": ({k: v for k, v in pred.items() if k != "speed_seq"}, tspd, cfg),
        "noaux": (pred, tspd, dict(cfg, use_load_balancing=False, use_entropy_loss=False)),
    }.items():
        r = f(p, twp, ts, c)
        for k, v in r.items():
            out[f"{tag}/{k}"] = v.detach().double().numpy()
    return out


def lsap_cases():
    from scipy.optimize import linear_sum_assignment
    rng = np.random.default_rng(7)
    out = {}
    mats = {}
    for n in (0, 1, 18, 64, 100):
        mats[f"rand920x{n}"] = rng.standard_normal((920, n)).astype(np.float32)
    mats["rand196x20"] = rng.standard_normal((196, 20)).astype(np.float32)
    mats["square64"] = rng.standard_normal((64, 64)).astype(np.float32)
    mats["wide20x196"] = rng.standard_normal((20, 196)).astype(np.float32)
    mats["ties4x3"] = np.ones((4, 3), dtype=np.float32)
    mats["ties920x7"] = np.zeros((920, 7), dtype=np.float32)
    mats["intcost50x9"] = rng.integers(0, 4, size=(50, 9)).astype(np.float32)
    dup = rng.standard_normal((920, 12)).astype(np.float32)
    dup[:, 5] = dup[:, 2]; dup[:, 9] = dup[:, 2]  # duplicate GT boxes -> identical columns
    mats["dupcols920x12"] = dup
    inf = rng.standard_normal((40, 6)).astype(np.float32)
    inf[::3, 1] = np.inf; inf[5, :] = np.inf
    mats["inf40x6"] = inf
    for k, m in mats.items():
        r, c = linear_sum_assignment(m)
        out[f"{k}/cost"], out[f"{k}/rows"], out[f"{k}/cols"] = m, r.astype(np.int64), c.astype(np.int64)
    return out


def main():
    torch.manual_seed(0)
    torch.set_num_threads(8)
    sys.path.insert(0, REF)
    gn = _load("models/gating/gating_network.py", "ref_gating_network")
    th = _load("models/policy/trajectory_head.py", "ref_trajectory_head")
    cf = _load("models/context/context_features.py", "ref_context_features")
    ex = _load("models/experts/expert_extractors.py", "ref_expert_extractors")
    for name, data in (("gating", gating_cases(gn)), ("policy", policy_cases(th)), ("extractors", extractor_cases(ex, cf)),
                       ("gating_losses", gating_loss_cases()), ("lsap_cases", lsap_cases())):
        path = os.path.join(HERE, f"{name}.npz")
        np.savez_compressed(path, **data)
        print(f"wrote {path}: {len(data)} arrays, {os.path.getsize(path) / 1024:.1f} KiB")


if __name__ == "__main__":
    main()
