#!/usr/bin/env python3
"""Golden fixtures for the SURVEY.md section 8(f) row 3 widening (NuScenes expert, its extractor, the D = 7 matcher branch).
Run in the build container (where /root/reference is mounted); writes tests/golden/nuscenes.npz and matcher_dims.npz.

What it pins:
  * `NuScenesExpertExtractor` -- models/experts/expert_extractors.py imports by file path with plain torch: run on seeded
    weights / inputs, outputs + gradients stored.
  * `NuScenesExpert`'s query decoder and heads (nuscenes_expert.py:96-190) -- the module imports torchvision at the top (an
    ordinary ModuleNotFoundError here), so that one class is compiled from the source text and built with a caller-supplied
    `image_backbone` (the constructor then never touches torchvision): the broadcast over queries, the decoder and both
    heads are the reference's own arithmetic.  The ResNet-18 trunk in front of it stays "parity unpinned" (DESIGN.md).
  * `HungarianMatcher.forward` (training/hungarian_matcher.py:13-85) for D = 7 (BEV branch), D = 5 (no-GIoU branch) and
    D = 4 -- same import problem (torchvision.ops), so the class is compiled from the source text with `box_convert` /
    `generalized_box_iou` bound to the oracle's restatements (still unpinned themselves) and scipy's real
    linear_sum_assignment; the cost matrices it hands to scipy are captured together with the returned indices.
Only data is written: inputs (or the seed that regenerates them) and expected outputs.
"""
from __future__ import annotations

import ast
import importlib.util
import os
import sys

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


def _grad_summary(module):
    out = {}
    for n, p in module.named_parameters():
        g = p.grad if p.grad is not None else torch.zeros_like(p)
        out[f"gsum/{n}"] = g.double().sum().numpy()
        out[f"gl2/{n}"] = g.double().pow(2).sum().sqrt().numpy()
    return out


def nuscenes_cases():
    spec = importlib.util.spec_from_file_location("ref_expert_extractors", os.path.join(REF, "models/experts/expert_extractors.py"))
    ex = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(ex)
    out = {}
    for D in (4, 7):
        m = seed_module_(ex.NuScenesExpertExtractor(256, num_queries=12, num_classes=10, bbox_dim=D), 700 + D).eval()
        cl, bb = seeded_tensor((3, 12, 10), 710 + D).requires_grad_(), seeded_tensor((3, 12, D), 720 + D).requires_grad_()
        y = m({"class_logits": cl, "bbox_preds": bb})
        (y * seeded_tensor((3, 256), 730)).sum().backward()
        out[f"ext{D}/features"] = y.detach().numpy()
        out[f"ext{D}/d_cls"], out[f"ext{D}/d_box"] = cl.grad.numpy(), bb.grad.numpy()
        for k, v in _grad_summary(m).items():
            out[f"ext{D}/{k}"] = v
    cls = _class_from_source("models/experts/nuscenes_expert.py", "NuScenesExpert", {"nn": nn, "torch": torch, "F": F})
    for D, Q in ((7, 12), (4, 196)):
        m = seed_module_(cls(image_backbone=nn.Identity(), num_queries=Q, bbox_dim=D), 740 + D).eval()
        feat = seeded_tensor((3, 256), 750 + D).requires_grad_()  # what trunk + pool + projection hand to the decoder
        o = m({"image": feat})
        (o["class_logits"] * seeded_tensor((3, Q, 10), 760)).sum().add((o["bbox_preds"] * seeded_tensor((3, Q, D), 761)).sum()).backward()
        out[f"head{D}/class_logits"], out[f"head{D}/bbox_preds"] = o["class_logits"].detach().numpy(), o["bbox_preds"].detach().numpy()
        out[f"head{D}/d_feat"] = feat.grad.numpy()
        for k, v in _grad_summary(m).items():
            out[f"head{D}/{k}"] = v
    return out


def matcher_cases():
    from scipy.optimize import linear_sum_assignment as scipy_lsa
    from oracle import matcher as om
    captured = []

    def lsa(C):
        c = C.detach().cpu().numpy() if isinstance(C, torch.Tensor) else np.asarray(C)
        captured.append(c.copy())
        return scipy_lsa(c)

    def box_convert(b, i, o):
        assert (i, o) == ("cxcywh", "xyxy")
        return om.box_cxcywh_to_xyxy(b)

    ns = {"nn": nn, "torch": torch, "linear_sum_assignment": lsa, "box_convert": box_convert,
          "generalized_box_iou": om.generalized_box_iou}
    cls = _class_from_source("training/hungarian_matcher.py", "HungarianMatcher", ns)
    out = {}
    for D in (7, 5, 4):
        B, Q, C = 3, 40, 10
        logits = seeded_tensor((B, Q, C), 800 + D)
        boxes = seeded_tensor((B, Q, D), 810 + D)
        boxes[..., 3 if D == 7 else 2:5 if D == 7 else 4] = boxes[..., 3 if D == 7 else 2:5 if D == 7 else 4].abs() + 0.1  # positive extents
        counts = [7, 0, 13]
        targets = []
        for b, n in enumerate(counts):
            tb = seeded_tensor((n, D), 820 + D + b) if n else torch.zeros(0, D)
            if n:
                tb[:, 3 if D == 7 else 2:5 if D == 7 else 4] = tb[:, 3 if D == 7 else 2:5 if D == 7 else 4].abs() + 0.1
            lab = (seeded_tensor((n,), 830 + D + b).abs() * 3).long().clamp(0, C - 1) if n else torch.zeros(0, dtype=torch.int64)
            targets.append({"boxes": tb, "labels": lab})
        captured.clear()
        idx = cls(1.0, 5.0, 2.0)({"pred_logits": logits, "pred_boxes": boxes}, targets)
        out[f"d{D}/logits"], out[f"d{D}/boxes"] = logits.numpy(), boxes.numpy()
        for b in range(B):
            out[f"d{D}/tgt_boxes{b}"], out[f"d{D}/tgt_labels{b}"] = targets[b]["boxes"].numpy(), targets[b]["labels"].numpy()
            out[f"d{D}/cost{b}"] = captured[b]
            out[f"d{D}/rows{b}"], out[f"d{D}/cols{b}"] = idx[b][0].numpy(), idx[b][1].numpy()
    return out


def main():
    torch.manual_seed(0)
    torch.set_num_threads(8)
    for name, data in (("nuscenes", nuscenes_cases()), ("matcher_dims", matcher_cases())):
        path = os.path.join(HERE, f"{name}.npz")
        np.savez_compressed(path, **data)
        print(f"wrote {path}: {len(data)} arrays, {os.path.getsize(path) / 1024:.1f} KiB")


if __name__ == "__main__":
    main()
