"""load_model / model_infer -- drop-in for inference/run_automoe.py:34-53,144-156 (BASELINE config 5).
The CARLA simulator loop, PID / pure-pursuit control and GIF export of the reference script need a simulator and
are outside the accelerated hot path."""
import json
from pathlib import Path
from typing import Any, Dict

import numpy as np
import torch
import torch.nn as nn

from .. import runtime
from ..models.automoe import create_automoe_model



def load_model(model_config_path: str, checkpoint_path: str, device: torch.device) -> nn.Module:
    cfg = json.loads(Path(model_config_path).read_text())
    model = create_automoe_model(cfg, device)
    state = torch.load(checkpoint_path, map_location=device, weights_only=True)
    state_dict = state.get("model_state_dict", state)
    if any(k.startswith("module.") for k in state_dict.keys()):  # strip DDP prefixes
        state_dict = {k[len("module."):]: v for k, v in state_dict.items()}
    missing, unexpected = model.load_state_dict(state_dict, strict=False)
    if missing or unexpected:
        print(f"Loaded with relaxed matching. Missing={len(missing)} Unexpected={len(unexpected)}")
    model.eval()
    return model


@torch.no_grad()
def model_infer(model: nn.Module, image_rgb: np.ndarray, last_speed_kmh: float, device: torch.device, img_tf=None) -> Dict[str, torch.Tensor]:
    """image_rgb: [H,W,3] uint8.  The reference runs under torch.autocast (fp16 on GPU): here the fp16 MFMA mode."""
    if img_tf is not None:
        tensor = img_tf(image_rgb).unsqueeze(0).to(device)
    else:
        # raw frame: /255 and the ImageNet normalisation run inside the boundary layout kernel (hip.ops.image_to_s2d)
        tensor = torch.from_numpy(np.ascontiguousarray(image_rgb)).to(device).permute(2, 0, 1).unsqueeze(0).contiguous()
    batch: Dict[str, Any] = {"image": tensor, "speed": torch.tensor([[last_speed_kmh]], dtype=torch.float32, device=device),
                             "steering": torch.zeros(1, 1, device=device), "throttle": torch.zeros(1, 1, device=device),
                             "brake": torch.zeros(1, 1, device=device)}
    prev = runtime.input_normalization()
    try:
        if tensor.dtype == torch.uint8:
            runtime.set_input_normalization(runtime.IMAGENET_MEAN, runtime.IMAGENET_STD)
        with runtime.precision(torch.float16):
            return model(batch)
    finally:
        runtime.set_input_normalization(*(prev if prev is not None else (None, None)))
