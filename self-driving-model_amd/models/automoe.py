"""AutoMoE composition -- drop-in for models/automoe.py:13-298 (API surface: create_automoe_model,
AutoMoE.forward(batch) -> 8-key dict, load_expert_checkpoints, freeze_experts / unfreeze_experts,
get_expert_weights).

MI355X-first differences in HOW (results are the reference's):
  * the image is converted once to NHWC (16 B per pixel) and shared by the three expert trunks and
    the policy backbone, instead of four independent NCHW reads;
  * an expert failure is not silently replaced by zeros unless AUTOMOE_SWALLOW_EXPERT_ERRORS=1
    (the reference's try/except at automoe.py:181-185 hides shape bugs): it warns and re-raises.
"""
import os
import warnings
from typing import Dict, List, Optional

import torch
import torch.nn as nn

from .. import runtime
from ..hip import conv as hconv
from ..hip import ops as hops
from .context.context_features import create_context_extractor
from .experts import BDDDetectionExpert, BDDDrivableExpert, BDDSegmentationExpert, NuScenesExpert
from .experts.expert_extractors import create_expert_extractors
from .gating.gating_network import GatingNetwork
from .policy.trajectory_head import TrajectoryPolicy
from ._nn import grouped_layernorm, grouped_linear, mlp5


def _last_step(t: torch.Tensor) -> torch.Tensor:
    if t.dim() == 2 and t.size(1) > 1:
        return t[:, -1:].contiguous()
    if t.dim() > 2:
        return t.view(t.size(0), -1)[:, -1:].contiguous()
    return t


class AutoMoE(nn.Module):
    """Complete AutoMoE: Mixture of Experts Self-Driving Model"""

    def __init__(self, expert_configs: List[Dict], gating_config: Dict, context_config: Dict, policy_config: Dict,
                 device: str = "cuda"):
        super().__init__()
        self.device = device
        self.expert_configs = expert_configs
        self.gating_config = gating_config
        self.context_config = context_config
        self.policy_config = policy_config
        self.experts = self._create_experts()
        self.expert_extractors = create_expert_extractors(expert_configs)
        self.context_extractor = create_context_extractor(context_config)
        self.gating_network = self._create_gating_network()
        self.policy_head = self._create_policy_head()
        # When True, segmentation / drivable experts feed their extractor through the fused upsample+average-pool
        # kernel (identical features, no 70 MB/img logits round trip); `expert_outputs` then holds their LOW-RES
        # [B,C,h,w] logits instead of the upsampled ones.  Default False = the reference's dict contents exactly.
        self.fuse_expert_pooling = False
        self.overlap_policy_backbone = os.environ.get("AUTOMOE_OVERLAP_BACKBONE", "1") != "0"
        self._side_stream = None
        self._expert_streams = []
        self.parallel_experts = os.environ.get("AUTOMOE_PARALLEL_EXPERTS", "1") != "0"
        # the parallel branches of the MoE tail (per-expert extractor / processor MLPs, context encoders, policy heads) as grouped
        # launches: ~15 launches forward instead of ~46 (hip/ops.py GroupedLinear / GroupedLayerNorm); off = one launch per layer
        self.group_tail = os.environ.get("AUTOMOE_GROUP_TAIL", "1") != "0"
        self.to(device)

    def _create_experts(self) -> nn.ModuleList:
        experts = nn.ModuleList()
        for config in self.expert_configs:
            t = config["type"]
            if t == "detection":
                e = BDDDetectionExpert(num_classes=config.get("num_classes", 10),
                                       pretrained_backbone=config.get("pretrained_backbone", True))
            elif t == "segmentation":
                e = BDDSegmentationExpert(num_classes=config.get("num_classes", 19),
                                          pretrained_backbone=config.get("pretrained_backbone", True))
            elif t == "drivable":
                e = BDDDrivableExpert(num_classes=config.get("num_classes", 3),
                                      pretrained_backbone=config.get("pretrained_backbone", True))
            elif t == "nuscenes":
                e = NuScenesExpert(num_queries=config.get("num_queries", 100), fusion=config.get("fusion", "concat"),
                                   use_lidar=config.get("use_lidar", False), use_tnet=config.get("use_tnet", False),
                                   bbox_dim=config.get("bbox_dim", 7),
                                   pretrained_backbone=config.get("pretrained_backbone", True))  # reference: always pretrained
            else:
                raise ValueError(f"Unknown expert type: {t}")
            experts.append(e)
        return experts

    def _create_gating_network(self) -> GatingNetwork:
        return GatingNetwork(
            num_experts=len(self.expert_configs),
            context_dim=self.context_config.get("context_dim", 64),
            expert_output_dims=[c.get("output_dim", 256) for c in self.expert_configs],
            processed_dim=self.gating_config.get("processed_dim", 256),
            hidden_dim=self.gating_config.get("hidden_dim", 128),
            temperature=self.gating_config.get("temperature", 1.0),
            use_softmax=self.gating_config.get("use_softmax", True))  # top_k / noise keys are ignored, as in the reference

    def _create_policy_head(self) -> TrajectoryPolicy:
        return TrajectoryPolicy(horizon=self.policy_config.get("num_waypoints", 10),
                                context_dim=self.gating_config.get("processed_dim", 256),
                                backbone_dim=self.policy_config.get("backbone_dim", 512))

    def _vehicle_state(self, batch: Dict[str, torch.Tensor]):
        """(speed, steering, throttle, brake) at the last step, [B,1] each (reference automoe.py:101-135)."""
        if self.context_config.get("type", "simple") != "simple":
            raise ValueError("only the 'simple' context extractor is on the accelerated path")
        speed = batch["speed"]
        speed_in = speed[:, -1:].contiguous() if speed.dim() == 2 and speed.size(1) > 1 else speed
        if all(k in batch for k in ("speed", "steering", "throttle", "brake")):
            steering, throttle, brake = (_last_step(batch[k]) for k in ("steering", "throttle", "brake"))
        else:
            bsz, device = speed_in.size(0), speed_in.device
            steering = torch.zeros(bsz, 1, device=device)
            throttle = torch.zeros(bsz, 1, device=device)
            brake = torch.zeros(bsz, 1, device=device)
        return speed_in, steering, throttle, brake

    def _extract_context_features(self, batch: Dict[str, torch.Tensor]) -> torch.Tensor:
        return self.context_extractor(*self._vehicle_state(batch))

    def _tail_grouped(self, mlp_inputs: List[torch.Tensor], batch: Dict[str, torch.Tensor]):
        """Extractor MLPs + context extractor + GatingNetwork.forward (gating_network.py:122-175) with every stage's independent
        branches in one launch.  Same modules, same arithmetic per layer as the ungrouped path (a Linear -> ReLU -> Dropout triple
        is one epilogue; its mask comes from the same counter-based generator).  Returns (context_features, gating_output)."""
        gn = self.gating_network
        E = len(mlp_inputs)
        ext = [mlp5(ex.feature_extractor) for ex in self.expert_extractors.extractors]
        cl1, cdr, cl2, cln = mlp5(self.context_extractor.encoder)
        vehicle_state = torch.cat(self._vehicle_state(batch), dim=-1).float()
        h = grouped_linear([e[0] for e in ext] + [cl1], mlp_inputs + [vehicle_state], True, [e[1] for e in ext] + [cdr])
        h = grouped_linear([e[2] for e in ext] + [cl2], h, False)
        h = grouped_layernorm([e[3] for e in ext] + [cln], h)
        feats, context_features = h[:E], h[E]
        proc = [mlp5(p.processor) for p in gn.expert_processors]
        ce = list(gn.context_encoder.context_encoder)  # Linear, ReLU, Dropout, Linear, ReLU, Dropout
        h = grouped_linear([p[0] for p in proc] + [ce[0]], feats + [context_features], True, [p[1] for p in proc] + [ce[2]])
        # second layers: the processors' have no activation, the context encoder's has ReLU + Dropout -> two groups by epilogue
        hp = grouped_linear([p[2] for p in proc], h[:E], False)
        (ctx_enc,) = grouped_linear([ce[3]], [h[E]], True, [ce[5]])
        processed = grouped_layernorm([p[3] for p in proc], hp)
        gate_input = torch.cat([ctx_enc] + processed, dim=1)
        g1, gdr, g2 = gn.gate_network[0], gn.gate_network[2], gn.gate_network[3]
        (hid,) = grouped_linear([g1], [gate_input], True, [gdr])
        (gate_logits,) = grouped_linear([g2], [hid], False)
        apply_topk = (gn.top_k > 0) and (gn.training or gn.apply_topk_at_eval)
        gate_weights, combined = gn._gate(gate_logits, processed, apply_topk)
        (out,) = grouped_linear([gn.output_projection], [combined], False)
        return context_features, {"combined_output": out, "expert_weights": gate_weights, "processed_expert_outputs": processed,
                                  "gate_logits": gate_logits}

    def _run_experts(self, batch: Dict[str, torch.Tensor], nhwc: torch.Tensor) -> List:
        outs = []
        for i, expert in enumerate(self.experts):
            try:
                if self.expert_configs[i]["type"] == "nuscenes":
                    outs.append(expert({"image": batch["image"], "lidar": batch.get("lidar")}, nhwc_input=nhwc))
                else:
                    outs.append(expert(batch["image"], nhwc_input=nhwc))
            except Exception as e:  # noqa: BLE001
                warnings.warn(f"Error running expert {i} ({self.expert_configs[i]['type']}): {e}")
                if os.environ.get("AUTOMOE_SWALLOW_EXPERT_ERRORS", "0") != "1":
                    raise
                outs.append(torch.zeros(batch["image"].size(0), self.expert_configs[i].get("output_dim", 256),
                                        device=batch["image"].device))
        return outs

    def _run_expert_trunks(self, batch, nhwc):
        """The experts themselves (trunk + head + fused upsample/pool): [(tensor for the extractor, pooled?)], outputs."""
        # Frozen experts in train-mode BatchNorm are chains of conv -> 5-us statistics finalize -> normalise passes: every
        # finalize drains the chip.  The experts are independent, so each runs on its own stream and the bubbles of one
        # are filled by the others' kernels.
        outs, pend = [], []
        par = self.parallel_experts and batch["image"].is_cuda
        main = torch.cuda.current_stream() if par else None
        forked = main.record_event() if par else None  # experts 1.. start here, beside expert 0 (not behind it)
        for i, expert in enumerate(self.experts):
            if hasattr(expert, "pooled_logits"):
                if par and i > 0:
                    while len(self._expert_streams) < i:
                        self._expert_streams.append(torch.cuda.Stream(device=batch["image"].device))
                    st = self._expert_streams[i - 1]
                    st.wait_event(forked)
                    with torch.cuda.stream(st):
                        pooled, low = expert.pooled_logits(batch["image"], nhwc_input=nhwc)
                    pooled.record_stream(main); low.record_stream(main)
                else:
                    pooled, low = expert.pooled_logits(batch["image"], nhwc_input=nhwc)
                pend.append((pooled, True))
                outs.append(low.detach()[..., : expert.num_classes].permute(0, 3, 1, 2))
            else:
                if self.expert_configs[i]["type"] == "nuscenes":
                    out = expert({"image": batch["image"], "lidar": batch.get("lidar")}, nhwc_input=nhwc)
                else:
                    out = expert(batch["image"], nhwc_input=nhwc)
                outs.append(out)
                pend.append((out, False))
        if par:
            for st in self._expert_streams:
                main.wait_stream(st)
        return pend, outs

    def _run_experts_fused(self, batch, nhwc, fork=None, expert_cache=None):
        """Experts, then their extractor MLPs.  `fork` (a callable) runs between the two: the launch-latency-bound MLP tail
        that starts here can then overlap whatever `fork` put on another stream.  `expert_cache`: the experts' results when
        they were computed ahead of this step (forward_experts)."""
        if expert_cache is not None:
            pend, outs = expert_cache["pend"], expert_cache["outs"]
        else:
            pend, outs = self._run_expert_trunks(batch, nhwc)
        if fork is not None:
            fork()
        if getattr(self, "_grouped_now", self.group_tail):  # the extractor MLPs run grouped with the rest of the tail: hand back their [B, C] inputs
            return outs, [t if pooled else extractor.pre_mlp(t) for extractor, (t, pooled) in zip(self.expert_extractors.extractors, pend)]
        feats = []
        for extractor, (t, pooled) in zip(self.expert_extractors.extractors, pend):
            feats.append(extractor.feature_extractor(t, start=2) if pooled else extractor(t))  # pooled: skip pool + flatten
        return outs, feats

    def experts_frozen(self) -> bool:
        return not any(p.requires_grad for e in self.experts for p in e.parameters())

    @torch.no_grad()
    def forward_experts(self, batch: Dict[str, torch.Tensor]) -> Dict:
        """The frozen-expert phase of forward() on its own (own statistics arena, nothing of the trainable part): a trainer
        may run it for the NEXT batch while the rest of the current step (policy, gating, backward, optimizer) is still on
        the GPU -- frozen experts do not depend on the update.  Pass the result to forward(batch, expert_cache=...)."""
        if not (self.fuse_expert_pooling and self.experts_frozen()):
            raise RuntimeError("forward_experts: needs frozen experts and fuse_expert_pooling")
        runtime.begin_step(batch["image"].device, phase="experts")
        nhwc = hops.image_to_nhwc(batch["image"], runtime.compute_dtype())
        pend, outs = self._run_expert_trunks(batch, nhwc)
        hconv.flush_bn_counters()
        return {"pend": pend, "outs": outs}

    def forward(self, batch: Dict[str, torch.Tensor], expert_cache: Optional[Dict] = None) -> Dict[str, torch.Tensor]:
        runtime.begin_step(batch["image"].device)
        # (swallowed expert failures put ready-made zero features in place of expert outputs: that path keeps one launch per layer)
        grouped = self._grouped_now = self.group_tail and os.environ.get("AUTOMOE_SWALLOW_EXPERT_ERRORS", "0") != "1"
        context_features = None if grouped else self._extract_context_features(batch)
        nhwc = hops.image_to_nhwc(batch["image"], runtime.compute_dtype())  # one read of the image for 4 backbones
        # The policy backbone (a conv stack on the image) does not depend on the experts; the extractor / gating MLPs
        # that follow them are dozens of launch-latency-bound kernels on [B, <=512] tensors.  With overlap_policy_backbone
        # the backbone runs on a side stream forked AFTER the experts (so the big kernels do not fight each other) and
        # joined before the policy heads: the MLP tail hides under its convolutions, in forward and -- autograd replays
        # each node on its forward stream -- in backward.
        side = {}

        def fork_backbone():
            main = torch.cuda.current_stream()
            if self._side_stream is None:
                self._side_stream = torch.cuda.Stream(device=batch["image"].device)
            self._side_stream.wait_stream(main)
            with torch.cuda.stream(self._side_stream):
                side["feat"] = self.policy_head.backbone(batch["image"], nhwc_input=nhwc)

        overlap = self.overlap_policy_backbone and self.fuse_expert_pooling and batch["image"].is_cuda
        if self.fuse_expert_pooling:
            expert_outputs, expert_features = self._run_experts_fused(batch, nhwc, fork_backbone if overlap else None, expert_cache)
        else:
            expert_outputs = self._run_experts(batch, nhwc)
            if grouped:
                expert_features = [ex.pre_mlp(o) for ex, o in zip(self.expert_extractors.extractors, expert_outputs)]
            else:
                expert_features = self.expert_extractors.extract_features(expert_outputs)
        if grouped:
            context_features, gating_output = self._tail_grouped(expert_features, batch)
        else:
            gating_output = self.gating_network(expert_features, context_features)
        if "feat" in side:
            torch.cuda.current_stream().wait_stream(self._side_stream)
            side["feat"].record_stream(torch.cuda.current_stream())
            policy_output = self.policy_head(batch["image"], context=gating_output["combined_output"], nhwc_input=nhwc,
                                             backbone_feat=side["feat"])
        else:
            policy_output = self.policy_head(batch["image"], context=gating_output["combined_output"], nhwc_input=nhwc)
        hconv.flush_bn_counters()
        speed_seq = policy_output.get("speed")
        speed_out = speed_seq[:, -1:].contiguous() if speed_seq is not None and speed_seq.dim() == 2 else None
        return {
            "waypoints": policy_output["waypoints"],
            "speed": speed_out if speed_out is not None else speed_seq,
            "speed_seq": speed_seq,
            "expert_weights": gating_output["expert_weights"],
            "expert_outputs": expert_outputs,
            "context_features": context_features,
            "combined_features": gating_output["combined_output"],
            "gate_logits": gating_output["gate_logits"],
        }

    def get_expert_weights(self, batch: Dict[str, torch.Tensor]) -> torch.Tensor:
        return self.gating_network.get_expert_weights(self._extract_context_features(batch))

    def load_expert_checkpoints(self, checkpoint_paths: List[str]):
        if len(checkpoint_paths) != len(self.experts):
            raise ValueError(f"Expected {len(self.experts)} checkpoint paths, got {len(checkpoint_paths)}")
        for i, (expert, path) in enumerate(zip(self.experts, checkpoint_paths)):
            if path and path != "":
                try:
                    checkpoint = torch.load(path, map_location=self.device, weights_only=True)
                    state_dict = checkpoint.get("model_state_dict", checkpoint)
                    if isinstance(expert, NuScenesExpert):
                        # checkpoints of the older NuScenes expert name its query MLP `mlp.` and its box head `box_head.`
                        # (reference automoe.py:251-262): remapped, and loaded non-strictly as the reference does
                        state_dict = {("decoder." + k[len("mlp."):] if k.startswith("mlp.") else
                                       "bbox_head." + k[len("box_head."):] if k.startswith("box_head.") else k): v
                                      for k, v in state_dict.items()}
                        expert.load_state_dict(state_dict, strict=False)
                    else:
                        expert.load_state_dict(state_dict)
                    print(f"Loaded checkpoint for expert {i}: {path}")
                except Exception as e:  # noqa: BLE001
                    warnings.warn(f"Failed to load checkpoint for expert {i}: {e}")

    def freeze_experts(self):
        for expert in self.experts:
            for p in expert.parameters():
                p.requires_grad = False

    def unfreeze_experts(self):
        for expert in self.experts:
            for p in expert.parameters():
                p.requires_grad = True


def create_automoe_model(config: Dict, device: str = "cuda") -> AutoMoE:
    return AutoMoE(expert_configs=config["experts"], gating_config=config["gating"], context_config=config["context"],
                   policy_config=config["policy"], device=device)
