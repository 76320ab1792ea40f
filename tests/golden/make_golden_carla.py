#!/usr/bin/env python3
"""Golden fixture for the CARLA policy objective.  Run in the build container (where
/root/reference is mounted); writes tests/golden/policy_losses.npz.

`compute_losses` (training/train_carla_policy.py:22-30) is pure torch, but its module imports the dataloaders package
(torchvision: an ordinary ModuleNotFoundError here), so that one function is compiled from the source text and run on seeded
inputs.  Only data is written: the five loss values and d loss / d prediction for each case.
"""
import ast
import os
import sys

import numpy as np
import torch
import torch.nn.functional as F

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, HERE)
from _seeded import seeded_tensor  # noqa: E402

REF = os.environ.get("AUTOMOE_REFERENCE", "/root/reference")


def main():
    tree = ast.parse(open(os.path.join(REF, "training/train_carla_policy.py")).read())
    fn = next(n for n in tree.body if isinstance(n, ast.FunctionDef) and n.name == "compute_losses")
    ns = {"torch": torch, "F": F, "Dict": dict}
    exec(compile(ast.Module(body=[fn], type_ignores=[]), "train_carla_policy.py", "exec"), ns)
    f = ns["compute_losses"]
    out = {}
    for tag, (B, T) in {"b6t8": (6, 8), "b32t10": (32, 10), "b3t3": (3, 3)}.items():
        wp = seeded_tensor((B, T, 2), 900 + B).requires_grad_()
        spd = seeded_tensor((B, T), 901 + B).requires_grad_()
        twp, tspd = seeded_tensor((B, T, 2), 902 + B), seeded_tensor((B, T), 903 + B)
        r = f({"waypoints": wp, "speed": spd}, twp, tspd)
        r["loss"].backward()
        for k, v in r.items():
            out[f"{tag}/{k}"] = v.detach().double().numpy()
        out[f"{tag}/d_wp"], out[f"{tag}/d_spd"] = wp.grad.numpy(), spd.grad.numpy()
    path = os.path.join(HERE, "policy_losses.npz")
    np.savez_compressed(path, **out)
    print(f"wrote {path}: {len(out)} arrays, {os.path.getsize(path) / 1024:.1f} KiB")


if __name__ == "__main__":
    main()
