import os
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
for p in (ROOT, os.path.join(ROOT, "tests", "golden")):
    if p not in sys.path:
        sys.path.insert(0, p)


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")


def _has_gpu():
    try:
        import torch
        return torch.cuda.is_available()
    except Exception:
        return False


def pytest_collection_modifyitems(config, items):
    if _has_gpu():
        return
    skip = pytest.mark.skip(reason="no GPU in this container")
    for item in items:
        if "gpu" in item.keywords:
            item.add_marker(skip)


@pytest.fixture(scope="session")
def golden_dir():
    return os.path.join(ROOT, "tests", "golden")


@pytest.fixture(scope="session", autouse=True)
def _oracle_built():
    """The C part of the oracle is a build product (oracle/_ref/, git-ignored): build it on demand."""
    so = os.path.join(ROOT, "oracle", "_ref", "liblsap_oracle.so")
    if not os.path.exists(so):
        import subprocess
        subprocess.check_call(["make", "-C", os.path.join(ROOT, "oracle")])


# ---- which kernel ran (include/automoe_hip.h am_conv_last_variant): kernel tests pin the kernel their docstring names ----
KERNELS = {0: "none", 1: "conv_ring_k<256,256>", 2: "conv_ring_k<256,128>", 3: "conv3x3_c64n64_duo_k", 4: "(retired)",
           5: "(retired)", 6: "conv_gemm2_k", 7: "(retired)", 8: "conv_gemm_k", 9: "conv_s2d_k", 10: "conv_s2d_pool_k",
           11: "conv_ring16_k<256,256>", 12: "conv_ring16_k<256,128>", 13: "wgrad_ring_k", 14: "conv_wgrad_k", 15: "conv_s2d_wgrad_k", 16: "conv_halo_k",
           17: "conv_patch_wgrad_k", 18: "conv_band16_k", 19: "conv_ring16_k<128,256>"}


def launched_kernel(expect=None, what=""):
    """Name of the conv kernel the last am_conv_* call of this thread launched; asserts it is (one of) `expect`.
    AUTOMOE_TEST_RECORD_KERNELS=<file> appends (what, name) lines: how the expectations in the tests were first filled in."""
    from self_driving_model_amd.hip import lib
    name = KERNELS.get(lib.get().am_conv_last_variant(), "?")
    rec = os.environ.get("AUTOMOE_TEST_RECORD_KERNELS")
    if rec:
        with open(rec, "a") as f:
            f.write(f"{what}\t{name}\n")
    if expect is not None:
        expect = (expect,) if isinstance(expect, str) else tuple(expect)
        assert name in expect, f"{what}: launched {name}, the test is written for {expect}"
    return name


def pytest_terminal_summary(terminalreporter, exitstatus, config):
    """Parity bookkeeping of tests/test_hip_models.py::_grad_check: how many parameter gradients were compared element-wise
    with the fp32 oracle, and every one that needed the fp64 arbitration (ill-conditioned train-mode BatchNorm cases)."""
    mod = sys.modules.get("test_hip_models")
    if mod is not None and getattr(mod, "F16_DISTANCE", None):
        import json
        try:
            os.makedirs(os.path.join(ROOT, "gpurun_out"), exist_ok=True)
            with open(os.path.join(ROOT, "gpurun_out", "f16_distance.json"), "w") as f:
                json.dump({"bounds": mod.F16_BOUNDS, "measured": mod.F16_DISTANCE}, f, indent=1)
        except OSError:
            pass
        terminalreporter.write_sep("=", "f16 vs fp32-mode distance at full size")
        for tag, rec in mod.F16_DISTANCE.items():
            terminalreporter.write_line(f"{tag}: loss delta {rec['loss_rel_delta']:.2e}  cos {rec['all']['cos']:.4f}  rel-L2 {rec['all']['rel_l2']:.4f}")
            for k, v in rec["stages"].items():
                terminalreporter.write_line(f"    {k:24s} cos {v['cos']:.4f}  rel-L2 {v['rel_l2']:.4f}  |g| {v['norm_fp32']:.3e}  ({v['params']} tensors)")
    if mod is None or not getattr(mod, "COMPARED", None):
        return
    import json
    arb, cmp_ = mod.ARBITRATIONS, mod.COMPARED
    tr = terminalreporter
    tr.write_sep("=", "gradient parity: fp64 arbitrations")
    tr.write_line(f"parameters compared element-wise with the fp32 oracle: {sum(c[1] for c in cmp_)} in {len(cmp_)} checks; "
                  f"arbitrated against fp64: {len(arb)}, of which hip is at least as close to fp64 as torch-cpu-fp32: "
                  f"{sum(1 for a in arb if a['hip_vs_fp64'] <= a['torch_fp32_vs_fp64'])}")
    for a in arb:
        tr.write_line(f"  {a['test']}  {a['param']}: hip vs fp64 {a['hip_vs_fp64']:.2e}, torch-cpu-fp32 vs fp64 {a['torch_fp32_vs_fp64']:.2e} "
                      f"(missed rtol {a['rtol']:g} / atol {a['atol']:g} against torch-cpu-fp32)")
    out = os.path.join(ROOT, "gpurun_out")
    try:
        os.makedirs(out, exist_ok=True)
        with open(os.path.join(out, "parity_arbitrations.json"), "w") as f:
            json.dump({"checks": [{"test": c[0], "parameters": c[1], "arbitrated": c[2]} for c in cmp_], "arbitrations": arb}, f, indent=1)
    except OSError:
        pass
