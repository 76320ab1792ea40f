#!/usr/bin/env python3
"""bench.py -- images/sec of the AutoMoE train step on synthetic 3x720x1280 batches (BASELINE.json metric).

    python bench.py [--gpus N] [--steps K] [--warmup W]
    python -m torch.distributed.run --nnodes=1 --nproc-per-node N --master-addr 127.0.0.1 --master-port P \
        bench.py --gpus N --steps K --warmup W

Workload (config.workload): BASELINE.json configs[3], "Full AutoMoE (3 experts + gating + policy) train step",
per-GPU batch 32 (weak scaling), variant 4a = experts frozen (the reference default,
training/train_gating_network.py:232; their BatchNorm still runs on batch statistics), fp16 MFMA compute with
fp32 accumulation and fp32 master weights.  A step = zero_grad, forward of the four backbones + MoE tail,
gating losses, backward (+ RCCL all-reduce overlapped on a side stream), clip 1.0 + AdamW.  Inputs are resident in
HBM before the timed region.  One JSON line is printed by rank 0; `value` is whole-job images/sec.

Extra legs (rank 0, N=1 only, after the timed region): `roofline` for the dominant kernel family (conv gather-GEMM)
from per-launch HIP events, `cpu_baseline` = the oracle (torch-CPU fp32 restatement of the reference) on a bounded
sample of the same workload, and `other_configs` (configs[1] drivable-expert train step, 4b unfrozen) for context.
"""
import argparse
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

import torch  # noqa: E402
import torch.distributed as dist  # noqa: E402

MODEL_CFG = {
    "experts": [{"type": "detection", "num_classes": 10, "output_dim": 256, "pretrained_backbone": False},
                {"type": "segmentation", "num_classes": 19, "output_dim": 256, "pretrained_backbone": False},
                {"type": "drivable", "num_classes": 3, "output_dim": 256, "pretrained_backbone": False}],
    "gating": {"processed_dim": 256, "hidden_dim": 128, "temperature": 1.0, "use_softmax": True},
    "context": {"type": "simple", "context_dim": 64},
    "policy": {"num_waypoints": 10},
}
TRAIN_CFG = {"learning_rate": 4e-4, "weight_decay": 1e-4}
PEAK_F16_TFLOPS = 2500.0  # dense MFMA f16, MI355X_MICROARCH.md chip table (spec; 2:1 sparsity figure NOT used)
H, W = 720, 1280


def timed_steps(step_fn, steps, warmup, distributed):
    for _ in range(warmup):
        step_fn()
    torch.cuda.synchronize()
    if distributed:
        dist.barrier()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(steps):
        step_fn()
    torch.cuda.synchronize()
    if distributed:
        dist.barrier()
    torch.cuda.synchronize()
    dt = time.perf_counter() - t0
    if distributed:
        t = torch.tensor([dt], dtype=torch.float64, device="cuda")
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        dt = float(t.item())
    return dt


def cpu_baseline(seconds_budget=25.0):
    """The oracle (oracle/torch_ref.py: torch-CPU fp32, the reference's arithmetic) on the same step at B=2."""
    from oracle import torch_ref as oref
    from oracle.losses import gating_losses
    cores = min(16, os.cpu_count() or 1)
    torch.set_num_threads(cores)
    torch.manual_seed(0)
    m = oref.create_automoe_model(MODEL_CFG, "cpu")
    m.freeze_experts()
    m.train()
    params = [p for p in m.parameters() if p.requires_grad]
    opt = torch.optim.AdamW(params, lr=4e-4, weight_decay=1e-4)
    B = 2
    batch = {"image": torch.randn(B, 3, H, W), "speed": torch.randn(B, 10), "steering": torch.randn(B, 10),
             "throttle": torch.randn(B, 10), "brake": torch.randn(B, 10), "waypoints": torch.randn(B, 10, 2)}

    def step():
        opt.zero_grad()
        out = m(batch)
        gating_losses(out, batch["waypoints"], batch["speed"], {})["total_loss"].backward()
        torch.nn.utils.clip_grad_norm_(params, 1.0)
        opt.step()

    step()  # warm-up
    times = []
    t_all = time.perf_counter()
    while len(times) < 3 and (time.perf_counter() - t_all) < seconds_budget:
        t0 = time.perf_counter()
        step()
        times.append(time.perf_counter() - t0)
    med = sorted(times)[len(times) // 2]
    return {"value": round(B / med, 3), "unit": "images/sec", "cores": cores, "kind": "port",
            "sample": f"oracle AutoMoE 4a train step (torch-CPU fp32), B={B} 3x720x1280, median of {len(times)} steps after 1 warm-up"}


def spawn_ranks(n: int) -> int:
    """`python -m torch.distributed.run --nnodes=1 --nproc-per-node n ... bench.py <same flags>` as a child; returns its exit code."""
    import socket
    import subprocess
    with socket.socket() as sk:
        sk.bind(("127.0.0.1", 0))
        port = sk.getsockname()[1]
    env = dict(os.environ)
    env.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")  # dmabuf IPC: what RCCL needs on this driver
    env.setdefault("OMP_NUM_THREADS", str(max(1, (os.cpu_count() or 8) // n)))
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", f"--nproc-per-node={n}", "--master-addr", "127.0.0.1",
           "--master-port", str(port), os.path.abspath(__file__)] + sys.argv[1:]
    return subprocess.call(cmd, env=env)


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=50)
    ap.add_argument("--warmup", type=int, default=10)
    ap.add_argument("--batch", type=int, default=32, help="per-GPU batch (BASELINE config 4: 32)")
    ap.add_argument("--precision", choices=["fp16", "fp32"], default="fp16")
    ap.add_argument("--no-extras", action="store_true", help="skip roofline / cpu_baseline / other_configs legs")
    args = ap.parse_args()

    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local = int(os.environ.get("LOCAL_RANK", "0"))
    distributed = world > 1
    if args.gpus > 1 and "WORLD_SIZE" not in os.environ:
        # plain `python bench.py --gpus N`: start the N ranks ourselves (one process per GPU, the reference's launcher line,
        # training/train_gating_network.sh:111) as CHILD processes, before this process has made any GPU call, and relay
        # rank 0's JSON line.  Never an exec: a process that touched the GPU must not be replaced.
        raise SystemExit(spawn_ranks(args.gpus))
    torch.cuda.set_device(local % max(torch.cuda.device_count(), 1))
    local = local % max(torch.cuda.device_count(), 1)
    if distributed:
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        dist.init_process_group(backend=os.environ.get("AUTOMOE_DIST_BACKEND", "nccl"), init_method="env://")  # "nccl" = RCCL over xGMI

    from self_driving_model_amd import runtime
    from self_driving_model_amd.hip import conv as hconv
    from self_driving_model_amd.models.automoe import create_automoe_model
    from self_driving_model_amd.training import synthetic
    from self_driving_model_amd.training.train_gating_network import GatingTrainStep

    runtime.set_compute_dtype(torch.float16 if args.precision == "fp16" else torch.float32)
    dev = torch.device("cuda", local)
    torch.manual_seed(0)
    model = create_automoe_model(MODEL_CFG, dev)
    model.freeze_experts()  # variant 4a: reference default
    model.train()
    step = GatingTrainStep(model, TRAIN_CFG)
    batch = synthetic.carla_sequence_batch(args.batch, H, W, 10, dev, seed=rank)

    def run():
        # once the step is captured the synthetic batch lives in the graph's input buffers (what a loader's H2D copy would
        # target): no per-step device-to-device copy of the 354 MB image batch
        # ... and with frozen experts the NEXT step's expert forward (its own graph on its own stream, same synthetic data in its
        # own input buffer) is launched behind this step's forward, so it shares the chip with this step's backward / optimizer
        step(step.input_buffers or batch, next_batch=True)

    dt = timed_steps(run, args.steps, args.warmup, distributed)
    value = world * args.batch * args.steps / dt
    out = {
        "metric": "images/sec AutoMoE train step, 3x720x1280 synthetic",
        "value": round(value, 2), "unit": "images/sec", "n_gpus": world, "steps": args.steps, "warmup": args.warmup,
        "ms_per_step": round(1e3 * dt / args.steps, 3), "higher_is_better": True, "scaling": "weak", "vs_baseline": None,
        "dtype": "f16" if args.precision == "fp16" else "f32", "data": "synthetic",
        "config": {"workload": "BASELINE configs[3] variant 4a: full AutoMoE (det+seg+drv ResNet-18 experts frozen, gating, "
                               "policy) train step, 3x720x1280", "per_gpu_batch": args.batch, "global_batch": args.batch * world,
                   "parallelism": f"dp{world}", "loss_scale": runtime.loss_scale(), "optimizer": "AdamW(4e-4,1e-4)+clip1.0",
                   "hipgraph": step._graph is not None, "fused_expert_pooling": bool(model.fuse_expert_pooling),
                   "expert_prefetch": step._graph_experts is not None, "expert_streams": bool(model.parallel_experts), "policy_backbone_stream": bool(model.overlap_policy_backbone)},
    }

    if rank == 0 and world == 1 and not args.no_extras:
        # ---- roofline of the dominant kernel (and of the whole conv forward family): per-launch HIP events on the launch stream ----
        saved = (step._graph, step.use_graph)  # per-launch events need eager launches, not a graph replay
        step._graph, step.use_graph = None, False
        # ... and one kernel on the chip at a time: in the timed step the experts and the policy backbone run on their own
        # streams, where an event pair around a launch would also count the kernels it shares the chip with
        saved_par = (model.parallel_experts, model.overlap_policy_backbone)
        model.parallel_experts = model.overlap_policy_backbone = False
        hconv.TIMER = hconv.KernelTimer()
        from self_driving_model_amd.hip import lib as hlib
        hlib.CALL_COUNTS = {}
        # the in-kernel cycle stamps live in a diagnostic instantiation of conv_ring16_k, selected for these two eager steps only
        old_diag = hconv._L().am_set_tuning(hlib.AM_TUNE_RING_DIAG, 1)
        for _ in range(2):
            run()
        torch.cuda.synchronize()
        hconv._L().am_set_tuning(hlib.AM_TUNE_RING_DIAG, old_diag)
        calls, hlib.CALL_COUNTS = hlib.CALL_COUNTS, None
        tail_prefixes = ("am_linear", "am_layernorm", "am_moe_tail", "am_gate", "am_dropout")
        out["moe_tail"] = {"grouped": bool(model.group_tail),
                           "abi_calls_per_step_fwd_bwd": sum(v for k, v in calls.items() if k.startswith(tail_prefixes)) // 2,
                           "all_abi_calls_per_step": sum(calls.values()) // 2}
        summ = hconv.TIMER.summary()
        per_kernel = hconv.TIMER.summary(by="kernel")
        hconv.TIMER = None
        model.parallel_experts, model.overlap_policy_backbone = saved_par
        # the clock the chip held inside the dominant kernel (s_memtime vs the 100 MHz s_memrealtime, workgroup 0 of the last
        # 256x256 ring launch): the 2.5 PFLOP/s spec peak is quoted at 2.4 GHz
        import ctypes
        clk = (ctypes.c_longlong * 8)()
        in_kernel = None
        try:
            hconv._L().am_diag_ring_clock(clk, hconv.stream())
            if clk[1] > 0 and clk[2] > 0:
                ghz = clk[0] / (clk[1] * 10.0)  # cycles per ns
                in_kernel = {"kernel": "conv_ring16_k<256,256,2,4> (diagnostic instantiation: the stride-2 entries / policy conv4 launches it still gets)",
                             "clock_ghz": round(ghz, 3), "cycles_per_kstep": round(clk[0] / clk[2], 1), "mfma_floor_cycles_per_kstep": 1024,
                             "peak_at_clock_tflops": round(PEAK_F16_TFLOPS * ghz / 2.4, 1), "ksteps": int(clk[2]),
                             "prologue_cycles": int(clk[3]), "epilogue_cycles": int(clk[4])}
        except Exception as e:  # noqa: BLE001
            in_kernel = {"error": repr(e)[:120]}
        step._graph, step.use_graph = saved
        # dominant kernel = the one with the largest share of the step (what the rocprof summary under profiles/ ranks first)
        dom_name, dom = max(per_kernel.items(), key=lambda kv: kv[1]["ms"])
        ach = dom["flops"] / (dom["ms"] * 1e-3) / 1e12
        fam = summ.get("conv_gemm", {"flops": 0.0, "ms": 1.0, "launches": 0})
        fam_ach = fam["flops"] / (fam["ms"] * 1e-3) / 1e12
        # HBM-side bytes per launch and the MFMA-busy share of the dominant kernel: PMC counters cannot be read from inside this
        # process; the figures come from separate `rocprofv3 --pmc` passes of this command (scratch/prof_r03.sh -> profiles/r03/
        # pmc_conv_kernels.json: FETCH_SIZE doubled per the gfx950 correction, raw per-dispatch rows inside).  They are reported
        # under `static` with the identity of what they were measured on, and `roofline.traffic` carries the number only while
        # the kernel's source file in this tree is byte-identical to the one profiled (sha256) -- otherwise null.
        traffic, static = None, {"measured_in_run": False, "source": "profiles/r03/pmc_conv_kernels.json", "stale": True}
        try:
            import hashlib
            pmc = json.load(open(os.path.join(ROOT, "profiles", "r03", "pmc_conv_kernels.json")))
            ent = next((v for k, v in pmc["kernels"].items() if dom_name.split("<")[0].split(" ")[0] == k.split("<")[0]), None)
            if ent is not None:
                cur = hashlib.sha256(open(os.path.join(ROOT, ent["source"]), "rb").read()).hexdigest()
                static.update({"kernel": dom_name, "kernel_source": ent["source"], "kernel_source_sha256_profiled": ent["source_sha256"],
                               "stale": cur != ent["source_sha256"] or args.batch != pmc.get("batch"),
                               "hbm_bytes_per_launch": ent.get("hbm_bytes_per_launch"), "mfma_busy_frac": ent.get("mfma_busy_frac"),
                               "share_of_wave_cycles": ent.get("share_of_wave_cycles")})
                if not static["stale"] and ent.get("hbm_bytes_per_launch"):
                    traffic = int(ent["hbm_bytes_per_launch"])
        except Exception as e:  # noqa: BLE001
            static["error"] = repr(e)[:120]
        # algorithmic bytes of the dominant kernel's launches (input + weights + output, each once, f16), from this run's launch list
        alg_bytes = dom.get("bytes", 0.0) / max(dom["launches"], 1) if dom.get("bytes") else None
        out["roofline"] = {"bound": "mfma", "achieved": round(ach, 2), "peak": PEAK_F16_TFLOPS, "unit": "TFLOP/s",
                           "frac": round(ach / PEAK_F16_TFLOPS, 4), "traffic": traffic,
                           "algorithmic_bytes_per_launch": alg_bytes, "static": static,
                           "kernel": dom_name, "in_kernel": in_kernel, "launches_per_step": dom["launches"] // 2,
                           "algorithmic_gflop_per_launch": round(dom["flops"] / max(dom["launches"], 1) / 1e9, 2),
                           "avg_launch_ms": round(dom["ms"] / max(dom["launches"], 1), 4),
                           "ms_per_step": round(dom["ms"] / 2, 3),
                           "conv_forward_family": {"tflops": round(fam_ach, 2), "frac": round(fam_ach / PEAK_F16_TFLOPS, 4),
                                                   "launches_per_step": fam["launches"] // 2,
                                                   "algorithmic_gflop_per_step": round(fam["flops"] / 2 / 1e9, 1),
                                                   "ms_per_step": round(fam["ms"] / 2, 3)},
                           "by_kernel": {k: {"tflops": round(v["flops"] / (v["ms"] * 1e-3) / 1e12, 2), "ms_per_step": round(v["ms"] / 2, 3),
                                             "launches_per_step": v["launches"] // 2} for k, v in sorted(per_kernel.items(), key=lambda kv: -kv[1]["ms"])},
                           "by_kind": {k: {"tflops": round(v["flops"] / (v["ms"] * 1e-3) / 1e12, 2), "ms_per_step": round(v["ms"] / 2, 3),
                                           "launches_per_step": v["launches"] // 2} for k, v in summ.items()}}
        out["step_flops_frac_of_peak"] = round(228.7e9 * args.batch * args.steps / dt / 1e12 / PEAK_F16_TFLOPS, 4)
        # ---- other configs for context (short) ----
        others = {}
        try:
            others["cfg2_drivable_expert_train_B16_img_s"] = bench_drivable(16, 6, 4)
            others["cfg3_detection_expert_hungarian_train_B8_img_s"] = bench_detection(8, 12, 4)
            others.update(bench_matcher())
            others.update(bench_inference(model, 64, 20, 3))
            model.unfreeze_experts()
            step_b = GatingTrainStep(model, TRAIN_CFG)
            dt_b = timed_steps(lambda: step_b(batch), 3, 4, False)  # warm-up covers the hipGraph capture
            others["cfg4b_unfrozen_B32_img_s"] = round(args.batch * 3 / dt_b, 2)
            del step_b
            # the parity-exact mode (exact-fp32 MFMA kernels, the ones held to rtol 1e-3 / atol 1e-5 against the oracle): what the
            # north_star tolerance costs in throughput on the same 4a step
            if args.precision == "fp16":
                others["cfg4a_fp32_mode_B32_img_s"] = bench_fp32_mode(args.batch, batch)
        except Exception as e:  # noqa: BLE001
            others["error"] = repr(e)[:200]
        out["other_configs"] = others
        del step, model
        torch.cuda.empty_cache()
        out["cpu_baseline"] = cpu_baseline()
    if rank == 0:
        try:  # the GPU test session's parity bookkeeping (tests/conftest.py), as last committed under profiles/: NOT measured in this run
            pa = json.load(open(os.path.join(ROOT, "profiles", "r03", "parity_arbitrations.json")))
            arb = pa["arbitrations"]
            out["static_parity"] = {"measured_in_run": False, "source": "profiles/r03/parity_arbitrations.json",
                                    "gradients_compared_elementwise": sum(c["parameters"] for c in pa["checks"]),
                                    "arbitrations": len(arb),
                                    "arbitrations_where_hip_is_at_least_as_close_to_fp64_as_torch_cpu_fp32":
                                        sum(1 for a in arb if a["hip_vs_fp64"] <= a["torch_fp32_vs_fp64"]),
                                    "strict_checks_without_arbitration": sum(1 for c in pa["checks"] if c["test"].startswith("strict/"))}
        except Exception:  # noqa: BLE001
            pass
        print(json.dumps(out), flush=True)
    if distributed:
        dist.destroy_process_group()


def bench_fp32_mode(B, batch):
    """The 4a train step of the timed line in fp32 compute mode (runtime.precision(float32): conv_gemm_k<float> on
    v_mfma_f32_32x32x2_f32, fp32 activations) -- the kernels the parity tests hold to the north_star tolerance."""
    from self_driving_model_amd import runtime
    from self_driving_model_amd.models.automoe import create_automoe_model
    from self_driving_model_amd.training.train_gating_network import GatingTrainStep
    dev = batch["image"].device
    with runtime.precision(torch.float32):
        torch.manual_seed(0)
        m = create_automoe_model(MODEL_CFG, dev)
        m.freeze_experts()
        m.train()
        st = GatingTrainStep(m, TRAIN_CFG)
        dt = timed_steps(lambda: st(st.input_buffers or batch, next_batch=True), 3, 4, False)
    del st, m
    torch.cuda.empty_cache()
    return round(B * 3 / dt, 2)


def bench_drivable(B, steps, warmup):
    """BASELINE configs[1]: drivable-area expert train step (training/train_bdd100k_ddp.py trainer), batch 16, 3x720x1280, fp16."""
    from self_driving_model_amd.models.experts import BDDDrivableExpert
    from self_driving_model_amd.training import synthetic
    from self_driving_model_amd.training.train_bdd100k_ddp import BDDTrainer
    dev = torch.device("cuda", torch.cuda.current_device())
    m = BDDDrivableExpert(3, pretrained_backbone=False).to(dev).train()
    b = synthetic.bdd_drivable_batch(B, H, W, 3, dev, seed=0)
    loader = synthetic.SyntheticLoader(b, steps)
    tr = BDDTrainer("drivable", m, loader, loader, dev, {"learning_rate": 2e-4, "weight_decay": 1e-5, "epochs": 1, "run_name": "bench"})
    dt = timed_steps(lambda: tr.train_step(tr.input_buffers or b), steps, warmup, False)  # no per-step device-to-device copy of the batch
    return round(B * steps / dt, 2)


def bench_detection(B, steps, warmup):
    """BASELINE configs[2]: detection expert + Hungarian matcher train step, batch 8 synthetic boxes, fp16."""
    from self_driving_model_amd.models.experts import BDDDetectionExpert
    from self_driving_model_amd.training import synthetic
    from self_driving_model_amd.training.train_bdd100k_ddp import BDDTrainer
    dev = torch.device("cuda", torch.cuda.current_device())
    m = BDDDetectionExpert(10, pretrained_backbone=False).to(dev).train()
    b = synthetic.bdd_detection_batch(B, H, W, 10, 32, dev, seed=0)
    loader = synthetic.SyntheticLoader(b, steps)
    tr = BDDTrainer("detection", m, loader, loader, dev, {"learning_rate": 2e-4, "weight_decay": 1e-5, "epochs": 1, "run_name": "bench"})
    dt = timed_steps(lambda: tr.train_step(tr.input_buffers or b), steps, warmup, False)  # no per-step device-to-device copy of the batch
    return round(B * steps / dt, 2)


def bench_matcher():
    """Matcher-only micro-benchmark: device cost + batched LSAP vs scipy on the host, Q = 920."""
    import numpy as np
    from scipy.optimize import linear_sum_assignment
    from self_driving_model_amd.hip import matcher as hm
    dev = torch.device("cuda", torch.cuda.current_device())
    out = {}
    for B, ni in ((8, 18), (64, 64)):
        cost = torch.randn(B, ni, 920, device=dev)
        n = torch.full((B,), ni, dtype=torch.int32, device=dev)
        for _ in range(2):
            hm.lsap_batched(cost, n, transposed_storage=True)
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        for _ in range(5):
            hm.lsap_batched(cost, n, transposed_storage=True)
        torch.cuda.synchronize()
        gpu_ms = (time.perf_counter() - t0) / 5 * 1e3
        c = cost.cpu().numpy().transpose(0, 2, 1)
        t0 = time.perf_counter()
        for b in range(B):
            linear_sum_assignment(c[b])
        cpu_ms = (time.perf_counter() - t0) * 1e3
        out[f"lsap_B{B}_920x{ni}_ms_gpu_vs_scipy"] = [round(gpu_ms, 3), round(cpu_ms, 3)]
    return out


@torch.no_grad()
def bench_inference(model, B, runs, warmup):
    """BASELINE configs[4]: AutoMoE inference (inference/run_automoe.py path), batch 64, fp16: p50 / p95 latency."""
    from self_driving_model_amd.training import synthetic
    dev = torch.device("cuda", torch.cuda.current_device())
    model.eval()
    batch = synthetic.carla_sequence_batch(B, H, W, 10, dev, seed=1)
    batch = {k: v for k, v in batch.items() if k != "waypoints"}
    lat = []
    for i in range(warmup + runs):
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        with torch.no_grad():  # SURVEY 8(d) config 5 / run_automoe.py:34-53 model_infer
            model(batch)
        torch.cuda.synchronize()
        if i >= warmup:
            lat.append((time.perf_counter() - t0) * 1e3)
    lat.sort()
    model.train()
    return {"cfg5_inference_B64_p50_ms": round(lat[len(lat) // 2], 2), "cfg5_inference_B64_p95_ms": round(lat[int(len(lat) * 0.95) - 1], 2)}


if __name__ == "__main__":
    main()
