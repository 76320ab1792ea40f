#!/usr/bin/env python3
"""Golden fixtures for the GLUE rows of SURVEY.md section 8 that the oracle so far only restated (missing #1):
the `AutoMoE` composition (A8), the three BDD expert wrappers (A2 / A3) and the trainer's set-loss / segmentation-loss
assembly (A10 / A11).  Run in the build container (where /root/reference is mounted); writes
tests/golden/automoe.npz, experts.npz and set_loss.npz -- data only (seeds regenerate the inputs and weights).

None of the modules involved imports here (torchvision / tensorboard are absent: an ordinary ModuleNotFoundError, not a denial),
so -- the technique make_golden_nuscenes.py already uses for `HungarianMatcher` -- each CLASS is compiled from the reference's
source text at generation time and run with the names it needs bound as follows:

  * `BDDDetectionExpert`, `BDDSegmentationExpert`, `BDDDrivableExpert` (models/experts/bdd_*_expert.py): `models.resnet18` is
    bound to a factory returning a module with torchvision's child order (conv1, bn1, relu, maxpool, layer1-4, avgpool, fc)
    built from the oracle's restated trunk, so `children()[:-2]`, the head / decoder, the channel slicing and
    `F.interpolate(..., align_corners=False)` are the reference's own lines.  The trunk itself stays "parity unpinned".
  * `AutoMoE` (models/automoe.py:13-279): the reference's importable `GatingNetwork`, `TrajectoryPolicy`,
    `create_expert_extractors`, `create_context_extractor` modules and the three compiled expert classes above.  Context
    slicing (:101-135), the expert loop (:156-187) and the output dict (`speed = speed_seq[:, -1:]`, :216-233) are the reference's.
  * `BDDTrainer._train_detection_batch` / `_train_segmentation_batch` (training/train_bdd100k_ddp.py:117-194): the class is
    compiled from source and the two methods run on a minimal `self` (model, device, config, the loss modules and the compiled
    `HungarianMatcher` with scipy's real solver); `box_convert` / `generalized_box_iou` are the oracle's restatements
    (torchvision.ops is absent: those two formulas stay unpinned).
"""
from __future__ import annotations

import ast
import importlib.util
import os
import sys
import types
import typing

import numpy as np
import torch
import torch.nn as nn
import torch.nn.functional as F

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, HERE)
sys.path.insert(0, os.path.dirname(os.path.dirname(HERE)))
from _seeded import seed_module_, seeded_tensor  # noqa: E402

REF = os.environ.get("AUTOMOE_REFERENCE", "/root/reference")


def _class_from_source(path, name, ns):
    tree = ast.parse(open(os.path.join(REF, path)).read())
    node = next(n for n in tree.body if isinstance(n, ast.ClassDef) and n.name == name)
    exec(compile(ast.Module(body=[node], type_ignores=[]), os.path.basename(path), "exec"), ns)
    return ns[name]


def _module_by_path(path, name):
    spec = importlib.util.spec_from_file_location(name, os.path.join(REF, path))
    m = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(m)
    return m


class _TorchvisionShapedResNet18(nn.Module):
    """torchvision's child ORDER around the oracle's trunk: the reference slices `list(resnet.children())[:-2]`."""

    def __init__(self):
        super().__init__()
        from oracle.torch_ref import resnet18_trunk
        t = resnet18_trunk()
        self.conv1, self.bn1, self.relu, self.maxpool = t[0], t[1], t[2], t[3]
        self.layer1, self.layer2, self.layer3, self.layer4 = t[4], t[5], t[6], t[7]
        self.avgpool = nn.AdaptiveAvgPool2d((1, 1))
        self.fc = nn.Linear(512, 1000)


def reference_expert_classes():
    models_ns = types.SimpleNamespace(resnet18=lambda pretrained=False: _TorchvisionShapedResNet18())
    ns = {"nn": nn, "F": F, "models": models_ns}
    return {n: _class_from_source(f"models/experts/{f}.py", n, dict(ns)) for n, f in (
        ("BDDDetectionExpert", "bdd_detection_expert"), ("BDDSegmentationExpert", "bdd_segmentation_expert"),
        ("BDDDrivableExpert", "bdd_drivable_expert"))}


def reference_automoe_class(experts):
    sys.path.insert(0, REF)
    from models.context.context_features import create_context_extractor  # importable with plain torch
    from models.gating.gating_network import GatingNetwork
    from models.policy.trajectory_head import TrajectoryPolicy
    ex = _module_by_path("models/experts/expert_extractors.py", "ref_expert_extractors")

    class NuScenesExpert(nn.Module):  # only named in an isinstance() test on the checkpoint path
        pass

    import warnings
    ns = {"nn": nn, "torch": torch, "F": F, "warnings": warnings, "Dict": typing.Dict, "List": typing.List,
          "Optional": typing.Optional, "Tuple": typing.Tuple, "NuScenesExpert": NuScenesExpert,
          "TrajectoryPolicy": TrajectoryPolicy, "GatingNetwork": GatingNetwork,
          "create_expert_extractors": ex.create_expert_extractors, "create_context_extractor": create_context_extractor}
    ns.update(experts)
    return _class_from_source("models/automoe.py", "AutoMoE", ns)


def reference_matcher_class(captured):
    from scipy.optimize import linear_sum_assignment as scipy_lsa
    from oracle import matcher as om

    def lsa(C):
        c = C.detach().cpu().numpy() if isinstance(C, torch.Tensor) else np.asarray(C)
        captured.append(c.copy())
        return scipy_lsa(c)

    def box_convert(b, i, o):
        assert (i, o) == ("cxcywh", "xyxy")
        return om.box_cxcywh_to_xyxy(b)

    ns = {"nn": nn, "torch": torch, "linear_sum_assignment": lsa, "box_convert": box_convert,
          "generalized_box_iou": om.generalized_box_iou}
    return _class_from_source("training/hungarian_matcher.py", "HungarianMatcher", ns)


def reference_trainer_class(matcher_cls):
    from oracle import matcher as om
    import torch.distributed as dist
    import torch.optim as optim

    def box_convert(b, i, o):
        assert (i, o) == ("xyxy", "cxcywh")
        return om.box_xyxy_to_cxcywh(b)

    ns = {"nn": nn, "torch": torch, "optim": optim, "dist": dist, "DDP": nn.parallel.DistributedDataParallel,
          "HungarianMatcher": matcher_cls, "box_convert": box_convert, "tqdm": lambda x, **k: x, "SummaryWriter": None}
    return _class_from_source("training/train_bdd100k_ddp.py", "BDDTrainer", ns)


def _grad_summary(module, tag, out):
    for n, p in module.named_parameters():
        g = p.grad if p.grad is not None else torch.zeros_like(p)
        out[f"{tag}/gsum/{n}"] = g.double().sum().numpy()
        out[f"{tag}/gl2/{n}"] = g.double().pow(2).sum().sqrt().numpy()


def _f64_twin(m):
    """The same module in double: the arbiter for stored gradients (a ReLU whose pre-activation is within fp32 rounding of zero
    takes either branch in fp32 -- torch's own included; the tests accept a gradient that is closer to this run than the fp32
    reference run is)."""
    import copy
    return copy.deepcopy(m).double()


class _default_f64:
    """torch.zeros(...) & co. inside the reference's forward (gating_network.py:163 builds its accumulator that way) follow
    the default dtype: double for the arbiter run."""

    def __enter__(self):
        self.prev = torch.get_default_dtype()
        torch.set_default_dtype(torch.float64)

    def __exit__(self, *exc):
        torch.set_default_dtype(self.prev)
        return False


def _dbl(batch):
    return {k: (v.double() if isinstance(v, torch.Tensor) and v.dtype.is_floating_point else v) for k, v in batch.items()}


def _no_dropout(m):
    for d in m.modules():
        if isinstance(d, nn.Dropout):
            d.p = 0.0
    return m


MODEL_CFG = {
    "experts": [{"type": "detection", "num_classes": 10, "output_dim": 256, "pretrained_backbone": False},
                {"type": "segmentation", "num_classes": 19, "output_dim": 256, "pretrained_backbone": False},
                {"type": "drivable", "num_classes": 3, "output_dim": 256, "pretrained_backbone": False}],
    "gating": {"processed_dim": 256, "hidden_dim": 128, "temperature": 1.0, "use_softmax": True},
    "context": {"type": "simple", "context_dim": 64},
    "policy": {"num_waypoints": 10},
}


def automoe_batches(B=3, H=64, W=96):
    """Three batches exercising every branch of automoe.py:101-135: [B,T] sequences, a 3-D control tensor (view(B,-1)[:, -1:]),
    [B,1] inputs, and a batch without steering / throttle / brake (zeros)."""
    img = seeded_tensor((B, 3, H, W), 1100)
    return {
        "seq": {"image": img, "speed": seeded_tensor((B, 10), 1101), "steering": seeded_tensor((B, 10), 1102),
                "throttle": seeded_tensor((B, 1), 1103), "brake": seeded_tensor((B, 2, 5), 1104)},
        "last": {"image": img, "speed": seeded_tensor((B, 1), 1105), "steering": seeded_tensor((B, 1), 1106),
                 "throttle": seeded_tensor((B, 1), 1107), "brake": seeded_tensor((B, 1), 1108)},
        "speed_only": {"image": img, "speed": seeded_tensor((B, 10), 1109)},
    }


def automoe_cases(experts):
    cls = reference_automoe_class(experts)
    out = {}
    for mode in ("eval", "train"):
        m = seed_module_(cls(MODEL_CFG["experts"], MODEL_CFG["gating"], MODEL_CFG["context"], MODEL_CFG["policy"], device="cpu"), 1)
        _no_dropout(m)
        m.train(mode == "train")
        for frozen in ((True, False) if mode == "eval" else (True,)):
            (m.freeze_experts if frozen else m.unfreeze_experts)()
            for bname, batch in automoe_batches().items():
                if not (bname == "seq" or (mode == "eval" and frozen)):
                    continue
                tag = f"{mode}/{'frozen' if frozen else 'unfrozen'}/{bname}"
                m.zero_grad(set_to_none=True)
                o = m(batch)
                B = batch["image"].size(0)
                loss = ((o["waypoints"] * seeded_tensor((B, 10, 2), 1110)).sum() + (o["speed_seq"] * seeded_tensor((B, 10), 1111)).sum()
                        + (o["speed"] * seeded_tensor((B, 1), 1112)).sum() + (o["expert_weights"] * seeded_tensor((B, 3), 1113)).sum()
                        + (o["gate_logits"] * seeded_tensor((B, 3), 1114)).sum() + (o["combined_features"] * seeded_tensor((B, 256), 1115)).sum())
                loss.backward()
                for k in ("waypoints", "speed", "speed_seq", "expert_weights", "context_features", "combined_features", "gate_logits"):
                    out[f"{tag}/{k}"] = o[k].detach().numpy()
                eo = o["expert_outputs"]
                out[f"{tag}/expert0_class_logits"] = eo[0]["class_logits"].detach().numpy()
                out[f"{tag}/expert0_bbox_deltas"] = eo[0]["bbox_deltas"].detach().numpy()
                out[f"{tag}/expert1_mean"] = eo[1].detach().double().mean(dim=(2, 3)).numpy()  # [B,19] (the full map is 19x64x96 per image)
                out[f"{tag}/expert1_corner"] = eo[1].detach()[:, :, :4, :4].numpy()
                out[f"{tag}/expert2"] = eo[2].detach().numpy()
                out[f"{tag}/loss"] = loss.detach().double().numpy()
                _grad_summary(m, tag, out)
                out[f"{tag}/d_policy_conv0"] = m.policy_head.backbone.net[0].weight.grad.numpy()
                out[f"{tag}/d_gate_out"] = m.gating_network.gate_network[3].weight.grad.numpy()
                if not frozen:
                    out[f"{tag}/d_expert0_head2"] = m.experts[0].head[2].weight.grad.numpy()
                    out[f"{tag}/d_expert2_conv1"] = m.experts[2].backbone[0].weight.grad.numpy()
                m64 = _f64_twin(m)
                m64.zero_grad(set_to_none=True)
                with _default_f64():
                    o64 = m64(_dbl(batch))
                l64 = ((o64["waypoints"] * seeded_tensor((B, 10, 2), 1110).double()).sum() + (o64["speed_seq"] * seeded_tensor((B, 10), 1111).double()).sum()
                       + (o64["speed"] * seeded_tensor((B, 1), 1112).double()).sum() + (o64["expert_weights"] * seeded_tensor((B, 3), 1113).double()).sum()
                       + (o64["gate_logits"] * seeded_tensor((B, 3), 1114).double()).sum() + (o64["combined_features"] * seeded_tensor((B, 256), 1115).double()).sum())
                l64.backward()
                out[f"{tag}/d_policy_conv0_f64"] = m64.policy_head.backbone.net[0].weight.grad.numpy()
                out[f"{tag}/d_gate_out_f64"] = m64.gating_network.gate_network[3].weight.grad.numpy()
                if not frozen:
                    out[f"{tag}/d_expert0_head2_f64"] = m64.experts[0].head[2].weight.grad.numpy()
                    out[f"{tag}/d_expert2_conv1_f64"] = m64.experts[2].backbone[0].weight.grad.numpy()
        if mode == "eval":
            out["eval/ctx_only_weights"] = m.get_expert_weights(automoe_batches()["seq"]).detach().numpy()
    return out


def expert_cases(experts):
    out = {}
    for name, ncls in (("BDDDetectionExpert", 10), ("BDDSegmentationExpert", 19), ("BDDDrivableExpert", 3)):
        for mode in ("eval", "train"):
            m = seed_module_(experts[name](num_classes=ncls, pretrained_backbone=False), 1200 + ncls)
            m.train(mode == "train")
            B, H, W = (4, 64, 96)
            x = seeded_tensor((B, 3, H, W), 1210 + ncls).requires_grad_()
            tag = f"{name}/{mode}"
            o = m(x)
            if isinstance(o, dict):
                loss = (o["class_logits"] * seeded_tensor(tuple(o["class_logits"].shape), 1220)).sum() + \
                       (o["bbox_deltas"] * seeded_tensor(tuple(o["bbox_deltas"].shape), 1221)).sum()
                out[f"{tag}/class_logits"], out[f"{tag}/bbox_deltas"] = o["class_logits"].detach().numpy(), o["bbox_deltas"].detach().numpy()
                if mode == "eval":
                    p = m.predict(x.detach())
                    out[f"{tag}/class_probs"], out[f"{tag}/bbox_sigmoid"] = p["class_probs"].detach().numpy(), p["bbox_deltas"].detach().numpy()
            else:
                loss = (o * seeded_tensor(tuple(o.shape), 1222)).sum() / (H * W)
                out[f"{tag}/logits_mean"] = o.detach().double().mean(dim=(2, 3)).numpy()
                out[f"{tag}/logits_rows"] = o.detach()[:, :, ::16, :].numpy()  # every 16th row of the full-resolution map
            loss.backward()
            out[f"{tag}/loss"] = loss.detach().double().numpy()
            out[f"{tag}/d_x_mean"] = x.grad.double().mean(dim=(2, 3)).numpy()
            _grad_summary(m, tag, out)
            last = m.head[2] if hasattr(m, "head") else m.decoder[2]
            out[f"{tag}/d_last_w"] = last.weight.grad.numpy()
            out[f"{tag}/d_conv1_w"] = m.backbone[0].weight.grad.numpy()
            m64 = _f64_twin(m)
            m64.zero_grad(set_to_none=True)
            x64 = x.detach().double().requires_grad_()
            o64 = m64(x64)
            if isinstance(o64, dict):
                l64 = (o64["class_logits"] * seeded_tensor(tuple(o64["class_logits"].shape), 1220).double()).sum() + \
                      (o64["bbox_deltas"] * seeded_tensor(tuple(o64["bbox_deltas"].shape), 1221).double()).sum()
            else:
                l64 = (o64 * seeded_tensor(tuple(o64.shape), 1222).double()).sum() / (H * W)
            l64.backward()
            last64 = m64.head[2] if hasattr(m64, "head") else m64.decoder[2]
            out[f"{tag}/d_last_w_f64"] = last64.weight.grad.numpy()
            out[f"{tag}/d_conv1_w_f64"] = m64.backbone[0].weight.grad.numpy()
            out[f"{tag}/d_x_mean_f64"] = x64.grad.mean(dim=(2, 3)).numpy()
    return out


def detection_batch(B=3, H=128, W=160, counts=(5, 0, 9), nmax=12, seed=1300):
    rng = np.random.default_rng(seed)
    boxes = -np.ones((B, nmax, 4), np.float32)
    labels = -np.ones((B, nmax), np.int64)
    for b, n in enumerate(counts):
        x1, y1 = rng.random(n) * W * 0.8, rng.random(n) * H * 0.8
        w, h = (0.02 + 0.18 * rng.random(n)) * W, (0.02 + 0.18 * rng.random(n)) * H
        boxes[b, :n] = np.stack([x1, y1, x1 + w, y1 + h], 1).astype(np.float32)
        labels[b, :n] = rng.integers(0, 10, n)
    if counts[0] >= 2:
        boxes[0, 1] = boxes[0, 0]  # duplicate ground-truth box: an exact tie for the solver
    return {"image": seeded_tensor((B, 3, H, W), seed + 1), "bboxes": torch.from_numpy(boxes), "labels": torch.from_numpy(labels)}


def set_loss_cases(experts):
    captured = []
    matcher_cls = reference_matcher_class(captured)
    trainer_cls = reference_trainer_class(matcher_cls)
    out = {}
    for mode in ("eval", "train"):
        model = seed_module_(experts["BDDDetectionExpert"](num_classes=10, pretrained_backbone=False), 1310)
        model.train(mode == "train")
        t = object.__new__(trainer_cls)  # a minimal `self`: what the two methods read
        t.model, t.device, t.config, t.task = model, "cpu", {"bbox_loss_weight": 2.0}, "detection"
        t.class_loss_fn, t.bbox_loss_fn = nn.CrossEntropyLoss(ignore_index=10), nn.SmoothL1Loss(reduction="mean")
        t.matcher = matcher_cls(cost_class=1.0, cost_bbox=5.0, cost_giou=2.0)
        for bname, kw in (("mixed", {}), ("empty", {"counts": (0, 0, 0)})):
            tag = f"det/{mode}/{bname}"
            batch = detection_batch(**kw)
            captured.clear()
            model.zero_grad(set_to_none=True)
            loss = t._train_detection_batch(batch)
            loss.backward()
            out[f"{tag}/loss"] = loss.detach().double().numpy()
            for b, c in enumerate(captured):
                out[f"{tag}/cost{b}"] = c
            _grad_summary(model, tag, out)
            out[f"{tag}/d_head2_w"] = model.head[2].weight.grad.numpy()
            out[f"{tag}/d_head2_b"] = model.head[2].bias.grad.numpy()
            if bname == "mixed":
                # arbiter run in double on the SAME assignment (the matcher's decision is part of the fp32 step being pinned)
                with torch.no_grad():
                    o32 = model(batch["image"])
                    B_, C_, H_, W_ = o32["class_logits"].shape
                    tg = []
                    for b in range(B_):
                        keep = batch["labels"][b] != -1
                        bx = batch["bboxes"][b][keep]
                        from oracle import matcher as om
                        tg.append({"boxes": om.box_xyxy_to_cxcywh(bx) if bx.numel() else bx, "labels": batch["labels"][b][keep]})
                    idx32 = t.matcher({"pred_logits": o32["class_logits"].permute(0, 2, 3, 1).reshape(B_, H_ * W_, C_),
                                       "pred_boxes": o32["bbox_deltas"].permute(0, 2, 3, 1).reshape(B_, H_ * W_, 4)}, tg)
                t64 = object.__new__(trainer_cls)
                t64.model, t64.device, t64.config, t64.task = _f64_twin(model), "cpu", {"bbox_loss_weight": 2.0}, "detection"
                t64.model.zero_grad(set_to_none=True)
                t64.class_loss_fn, t64.bbox_loss_fn = t.class_loss_fn, t.bbox_loss_fn
                t64.matcher = lambda outputs, targets: idx32
                with _default_f64():
                    l64 = t64._train_detection_batch(dict(batch, image=batch["image"].double()))  # (targets stay fp32: the reference builds its target buffers in fp32)
                l64.backward()
                out[f"{tag}/loss_f64"] = l64.detach().double().numpy()
                out[f"{tag}/d_head2_w_f64"] = t64.model.head[2].weight.grad.numpy()
                out[f"{tag}/d_head2_b_f64"] = t64.model.head[2].bias.grad.numpy()
    for name, ncls in (("BDDSegmentationExpert", 19), ("BDDDrivableExpert", 3)):
        for mode in ("eval", "train"):
            model = seed_module_(experts[name](num_classes=ncls, pretrained_backbone=False), 1320 + ncls)
            model.train(mode == "train")
            t = object.__new__(trainer_cls)
            t.model, t.device, t.config, t.task = model, "cpu", {}, "segmentation"
            t.loss_fn = nn.CrossEntropyLoss(ignore_index=255)
            B, H, W = 3, 64, 96
            rng = np.random.default_rng(1330 + ncls)
            mask = rng.integers(0, ncls, (B, H, W)).astype(np.int64)
            mask[rng.random((B, H, W)) < 0.05] = 255
            batch = {"image": seeded_tensor((B, 3, H, W), 1331 + ncls), "mask": torch.from_numpy(mask)}
            tag = f"seg{ncls}/{mode}"
            loss = t._train_segmentation_batch(batch)
            loss.backward()
            out[f"{tag}/loss"] = loss.detach().double().numpy()
            _grad_summary(model, tag, out)
            out[f"{tag}/d_dec2_w"] = model.decoder[2].weight.grad.numpy()
            t64 = object.__new__(trainer_cls)
            t64.model, t64.device, t64.config, t64.task, t64.loss_fn = _f64_twin(model), "cpu", {}, "segmentation", t.loss_fn
            t64.model.zero_grad(set_to_none=True)
            l64 = t64._train_segmentation_batch(dict(batch, image=batch["image"].double()))
            l64.backward()
            out[f"{tag}/loss_f64"] = l64.detach().double().numpy()
            out[f"{tag}/d_dec2_w_f64"] = t64.model.decoder[2].weight.grad.numpy()
    return out


def main():
    torch.manual_seed(0)
    torch.set_num_threads(8)
    experts = reference_expert_classes()
    for name, fn in (("experts", expert_cases), ("automoe", automoe_cases), ("set_loss", set_loss_cases)):
        data = fn(experts)
        path = os.path.join(HERE, f"{name}.npz")
        np.savez_compressed(path, **data)
        print(f"wrote {path}: {len(data)} arrays, {os.path.getsize(path) / 1024:.1f} KiB")


if __name__ == "__main__":
    main()
