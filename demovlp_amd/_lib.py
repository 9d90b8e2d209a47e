"""ctypes binding of libdemovlp_hip.so (C ABI declared in include/demovlp_hip.h).

There is NO fallback: if the library is missing or a kernel reports an error, this raises.  The CPU oracle under
``oracle/`` is test infrastructure and is never imported from here.
"""
from __future__ import annotations

import ctypes
import os

_HERE = os.path.dirname(os.path.abspath(__file__))
# DEMOVLP_HIP_LIB: developer override for A/B timing of two builds of the library in one GPU session
LIB_PATH = os.environ.get("DEMOVLP_HIP_LIB") or os.path.join(_HERE, "lib", "libdemovlp_hip.so")

F32, BF16 = 0, 1
EPI_GELU, EPI_GELU_BWD, EPI_RELU_BWD, EPI_ACCUM, EPI_LEAKY = 1, 2, 4, 8, 16
ERRORS = {-1: "unsupported dtype", -2: "bad shape / alignment / missing workspace", -3: "HIP launch failed",
          -4: "configuration not supported by this kernel"}

HEADER_PATH = os.path.join(os.path.dirname(_HERE), "include", "demovlp_hip.h")
_CT = {"int64_t": ctypes.c_int64, "int": ctypes.c_int, "float": ctypes.c_float}


def _parse_header(path=HEADER_PATH):
    """{name: (restype, [argtypes])} for every function declared in include/demovlp_hip.h."""
    import re
    text = open(path).read()
    text = re.sub(r"/\*.*?\*/", " ", text, flags=re.S)
    sigs = {}
    for m in re.finditer(r"\b(int64_t|int)\s+(dvlp_\w+)\s*\(([^)]*)\)\s*;", text):
        ret, name, args = m.group(1), m.group(2), m.group(3)
        argtypes = []
        for a in [x.strip() for x in args.split(",") if x.strip() and x.strip() != "void"]:
            if "*" in a:
                argtypes.append(ctypes.c_void_p)
            else:
                base = a.replace("const", "").split()[0]
                argtypes.append(_CT[base])
        sigs[name] = (_CT[ret], argtypes)
    return sigs


_SIGS = _parse_header()
_RET64 = {n for n, (r, _) in _SIGS.items() if r is ctypes.c_int64}

_lib = None


class DemoVLPHipError(RuntimeError):
    pass


def load():
    """Load the shared library (once).  Raises if it has not been built (``python -m demovlp_amd.build``)."""
    global _lib
    if _lib is not None:
        return _lib
    # PyTorch-ROCm bundles its own libamdhip64; import it FIRST so this library binds to the same HIP runtime
    # (loading /opt/rocm's copy first gives the process two runtimes and ours then sees "no ROCm-capable device").
    import torch  # noqa: F401
    if not os.path.exists(LIB_PATH):
        raise DemoVLPHipError(f"{LIB_PATH} is missing: build it with `python -m demovlp_amd.build` "
                              "(hipcc --offload-arch=gfx950); there is no CPU fallback.")
    lib = ctypes.CDLL(LIB_PATH)
    for name, (ret, argtypes) in _SIGS.items():
        fn = getattr(lib, name)          # AttributeError if a declared symbol is not exported
        fn.argtypes = argtypes
        fn.restype = ret
    _lib = lib
    return lib


def exported_symbols():
    return list(_SIGS)


def call(name, *args):
    """Invoke an entry point; raise on a non-zero status."""
    fn = getattr(load(), name)
    rc = fn(*args)
    if name in _RET64:
        return rc
    if rc != 0:
        detail = ""
        if rc == -3:
            f = load().dvlp_last_error_string
            f.restype = ctypes.c_char_p
            detail = f" ({f().decode()})"
        raise DemoVLPHipError(f"{name} failed: {ERRORS.get(rc, rc)}{detail}")
    return rc
