"""ctypes binding of libautomoe_hip.so (include/automoe_hip.h).

The prototypes are read from the header itself, so the binding cannot drift from the C ABI: every
`int am_*(...)` declaration becomes a checked Python callable `lib.am_*`.  There is NO fallback:
if the shared library is missing or a symbol is absent, import of the product path fails loudly.
"""
from __future__ import annotations

import ctypes
import os
import re
from typing import Dict, List, Tuple

_PKG = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
_ROOT = os.path.dirname(_PKG)
HEADER = os.path.join(_ROOT, "include", "automoe_hip.h")
LIB_PATH = os.environ.get("AUTOMOE_HIP_LIB") or os.path.join(_PKG, "csrc", "libautomoe_hip.so")  # override: A/B builds

AM_F32, AM_F16 = 0, 1
AM_MAX_TAPS = 16
AM_STATS_REPLICAS = 16
AM_MAX_EXPERTS = 8
AM_TUNE_RING = 0
AM_TUNE_RING128_MIN_TILES = 1
AM_TUNE_WGRAD_RING = 2
AM_TUNE_WGRAD_MAX_SLABS = 3
AM_TUNE_RING_SHORT_K = 4
AM_TUNE_HALO_MIN_TILES = 5
AM_TUNE_PATCH_WGRAD_MIN_TILES = 6
AM_TUNE_PATCH_WGRAD_C128 = 7
AM_TUNE_DUO_MFMA16 = 8
AM_TUNE_BAND_MIN_TILES = 9
AM_TUNE_RING_DIAG = 10
AM_TUNE_RING16_M128_MIN_TILES = 11

_ERR = {-1: "AM_ERR_ARG (bad argument)", -2: "AM_ERR_LAUNCH (HIP runtime refused the launch)",
        -3: "AM_ERR_UNSUPPORTED (shape/dtype not built)"}


class ConvGeom(ctypes.Structure):
    """am_conv_geom of include/automoe_hip.h."""
    _fields_ = [(n, ctypes.c_int32) for n in (
        "B", "MH", "MW", "IH", "IW", "ldi", "x_coff", "OH", "OW", "ldo", "y_coff", "oys", "oy0", "oxs", "ox0",
        "iys", "ixs", "ntaps", "krun", "pix_shift", "N")] + [("dy", ctypes.c_int16 * AM_MAX_TAPS),
                                                               ("dx", ctypes.c_int16 * AM_MAX_TAPS),
                                                               ("osplit", ctypes.c_int32), ("osplit_stride", ctypes.c_int32)]


AM_TAIL_MAX_GROUP = 8


class TailLinear(ctypes.Structure):
    """am_tail_linear of include/automoe_hip.h."""
    _fields_ = ([(n, ctypes.c_void_p) for n in ("x", "W", "bias", "y", "dy", "yact", "dx", "dW", "dbias")]
                + [(n, ctypes.c_int32) for n in ("ldx", "ldy", "lddy", "ldya", "lddx", "N", "K", "relu", "dx_accumulate")]
                + [("drop_p", ctypes.c_float), ("gscale", ctypes.c_float), ("seed", ctypes.c_uint64)])


class TailLayerNorm(ctypes.Structure):
    """am_tail_layernorm of include/automoe_hip.h."""
    _fields_ = ([(n, ctypes.c_void_p) for n in ("x", "gamma", "beta", "y", "mean", "rstd", "dy", "dx", "dgamma", "dbeta")]
                + [(n, ctypes.c_int32) for n in ("ldx", "ldy", "lddy", "lddx", "D")] + [("eps", ctypes.c_float)])


_SCALARS = {"int": ctypes.c_int, "float": ctypes.c_float, "double": ctypes.c_double, "long long": ctypes.c_longlong,
            "unsigned long long": ctypes.c_ulonglong, "am_stream_t": ctypes.c_void_p}


def parse_header(path: str = HEADER) -> Dict[str, List[Tuple[str, object]]]:
    """{symbol: [(arg name, ctypes type), ...]} for every `int am_*(...)` prototype in the header."""
    text = open(path).read()
    text = re.sub(r"/\*.*?\*/", "", text, flags=re.S)
    protos = {}
    for m in re.finditer(r"\bint\s+(am_\w+)\s*\(([^;{]*?)\)\s*;", text, flags=re.S):
        name, args = m.group(1), " ".join(m.group(2).split())
        out = []
        if args and args != "void":
            for a in args.split(","):
                a = a.strip()
                if "*" in a:
                    out.append((a.split("*")[-1].strip() or "p", ctypes.c_void_p))
                    continue
                toks = a.replace("const ", "").split()
                ctype = " ".join(toks[:-1])
                if ctype not in _SCALARS:
                    raise RuntimeError(f"{path}: unknown C type '{ctype}' in {name}")
                out.append((toks[-1], _SCALARS[ctype]))
        protos[name] = out
    return protos


def _load_torch_hip_runtime():
    """libautomoe_hip.so needs libamdhip64.so.7.  PyTorch-ROCm ships its own copy under torch/lib with the same
    SONAME; the process must use exactly one HIP runtime (streams and device pointers are shared with torch), so
    torch's copy is loaded first and the dynamic linker then resolves our dependency to it."""
    import torch  # noqa: F401  (imports torch/lib/libamdhip64.so as a side effect)
    cand = os.path.join(os.path.dirname(torch.__file__), "lib", "libamdhip64.so")
    if os.path.exists(cand):
        ctypes.CDLL(cand, mode=ctypes.RTLD_GLOBAL)


class _Lib:
    def __init__(self):
        _load_torch_hip_runtime()
        if not os.path.exists(LIB_PATH):
            raise RuntimeError(
                f"{LIB_PATH} is missing: the HIP extension is not built. Run `python -c 'import __graft_entry__ as g; "
                f"g.build()'` (or `make -C self-driving-model_amd/csrc`). There is no CPU fallback for the product path.")
        self._dll = ctypes.CDLL(LIB_PATH)
        self.protos = parse_header()
        for name, args in self.protos.items():
            try:
                fn = getattr(self._dll, name)
            except AttributeError as e:
                raise RuntimeError(f"{LIB_PATH} does not export {name} declared in {HEADER}") from e
            fn.restype = ctypes.c_int
            fn.argtypes = [t for _, t in args]
            setattr(self, "_raw_" + name, fn)
            if name in ("am_version", "am_conv_npad", "am_conv_last_variant", "am_set_tuning", "am_get_tuning"):  # return values, not status codes
                setattr(self, name, fn)
            else:
                setattr(self, name, self._checked(name, fn))
        # A/B builds and CI legs pin kernels without code changes: AUTOMOE_TUNE_<KEY>=<int> (keys of am_set_tuning)
        for key, idx in (("RING", AM_TUNE_RING), ("RING128_MIN_TILES", AM_TUNE_RING128_MIN_TILES), ("WGRAD_RING", AM_TUNE_WGRAD_RING), ("WGRAD_MAX_SLABS", AM_TUNE_WGRAD_MAX_SLABS),
                         ("RING_SHORT_K", AM_TUNE_RING_SHORT_K), ("HALO_MIN_TILES", AM_TUNE_HALO_MIN_TILES),
                         ("PATCH_WGRAD_MIN_TILES", AM_TUNE_PATCH_WGRAD_MIN_TILES), ("PATCH_WGRAD_C128", AM_TUNE_PATCH_WGRAD_C128),
                         ("DUO_MFMA16", AM_TUNE_DUO_MFMA16), ("BAND_MIN_TILES", AM_TUNE_BAND_MIN_TILES), ("RING_DIAG", AM_TUNE_RING_DIAG),
                         ("RING16_M128_MIN_TILES", AM_TUNE_RING16_M128_MIN_TILES)):
            v = os.environ.get("AUTOMOE_TUNE_" + key)
            if v is not None:
                self.am_set_tuning(idx, int(v))

    @staticmethod
    def _checked(name, fn):
        def call(*args):
            if CALL_COUNTS is not None:
                CALL_COUNTS[name] = CALL_COUNTS.get(name, 0) + 1
            rc = fn(*args)
            if rc != 0:
                raise RuntimeError(f"{name} failed: {_ERR.get(rc, rc)}")
        call.__name__ = name
        return call


_LIB = None
CALL_COUNTS = None  # set to a dict to count ABI calls per entry point (bench.py: launches of the MoE tail per step)


def get() -> _Lib:
    global _LIB
    if _LIB is None:
        _LIB = _Lib()
    return _LIB
