"""ctypes binding of libxrface.so (the C ABI declared in include/xrface.h).

The product path has NO CPU fallback: if the shared library is missing, or an entry point
returns an error, a RuntimeError is raised.  Signatures are parsed from include/xrface.h so the
header stays the single source of truth (and the CPU test-suite can check every declared symbol
is exported).
"""
from __future__ import annotations

import ctypes
import os
import re

import torch

_HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.environ.get("XR_LIB") or os.path.join(_HERE, "libxrface.so")   # XR_LIB: A/B a differently built library
HEADER_PATH = os.path.normpath(os.path.join(_HERE, "..", "..", "include", "xrface.h"))

XR_BF16, XR_F32, XR_F32X2 = 0, 1, 2
ACT_NONE, ACT_PRELU, ACT_RELU, ACT_TANH = 0, 1, 2, 3

_CT = {
    "int": ctypes.c_int, "float": ctypes.c_float, "int64_t": ctypes.c_int64, "uint64_t": ctypes.c_uint64,
    "size_t": ctypes.c_size_t, "double": ctypes.c_double,
}


def parse_header(path: str = HEADER_PATH):
    """Return {name: (restype, [argtypes])} for every ``int xr_*(...)`` / ``const char* xr_*`` prototype."""
    src = open(path).read()
    src = re.sub(r"/\*.*?\*/", "", src, flags=re.S)
    protos = {}
    for m in re.finditer(r"(const char\*|int)\s+(xr_\w+)\s*\(([^)]*)\)\s*;", src):
        ret, name, args = m.group(1), m.group(2), m.group(3).strip()
        argtypes = []
        if args and args != "void":
            for a in args.split(","):
                a = a.strip()
                if "*" in a:
                    argtypes.append(ctypes.c_void_p)
                else:
                    ty = a.replace("const ", "").split()[0]
                    if ty == "unsigned":
                        ty = "uint64_t"
                    argtypes.append(_CT[ty])
        protos[name] = (ctypes.c_char_p if ret != "int" else ctypes.c_int, argtypes)
    return protos


class _Lib:
    def __init__(self):
        self._dll = None
        self._protos = None

    def load(self):
        if self._dll is not None:
            return self._dll
        if not os.path.exists(LIB_PATH):
            raise RuntimeError(
                f"xrface: HIP extension {LIB_PATH} is missing -- build it with `python __graft_entry__.py` "
                f"(or `make -C cross-resolution-face-recognition_amd/csrc`).  There is no CPU fallback.")
        dll = ctypes.CDLL(LIB_PATH)
        self._protos = parse_header()
        for name, (res, args) in self._protos.items():
            fn = getattr(dll, name)  # AttributeError if the library does not export a declared symbol
            fn.restype = res
            fn.argtypes = args
        self._dll = dll
        if os.environ.get("XR_DETERMINISTIC", "0") == "1":   # device half of ops.set_deterministic (the host half reads the same variable)
            dll.xr_set_deterministic(1)
        for kv in filter(None, os.environ.get("XR_TUNE", "").split(",")):  # e.g. XR_TUNE="7=0,6=18": kernel tuning knobs
            k, v = kv.split("=")
            dll.xr_tune(int(k), int(v))
        return dll

    def __getattr__(self, name):
        dll = self.load()
        fn = getattr(dll, name)
        if self._protos[name][0] is not ctypes.c_int:
            return fn

        def call(*args):
            rc = fn(*args)
            if rc < 0:
                raise RuntimeError(f"{name} failed ({rc}): {dll.xr_last_error().decode()}")
            return rc

        call.__name__ = name
        setattr(self, name, call)
        return call


lib = _Lib()


def ptr(t):
    """Device pointer of a tensor (None -> NULL)."""
    return None if t is None else t.data_ptr()


_raw_stream = getattr(torch._C, "_cuda_getCurrentRawStream", None)
_cur_device = getattr(torch._C, "_cuda_getDevice", None)


def stream():
    """Raw handle of the current stream of the current device.  (torch.cuda.current_stream() builds a Stream object through
    several Python layers: ~4 us, once per kernel launch -- 10 % of the host's enqueue time of a training step.)"""
    if _raw_stream is not None and _cur_device is not None:
        return _raw_stream(_cur_device())
    return torch.cuda.current_stream().cuda_stream


def stream_of(device_index: int):
    """Raw handle of the current stream of device ``device_index``."""
    if _raw_stream is not None:
        return _raw_stream(device_index)
    return torch.cuda.current_stream(device_index).cuda_stream


def dt(t) -> int:
    if t.dtype == torch.bfloat16:
        return XR_BF16
    if t.dtype == torch.float32:
        return XR_F32
    raise RuntimeError(f"xrface: unsupported activation dtype {t.dtype} (bf16 or fp32 only)")
