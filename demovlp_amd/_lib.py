"""ctypes binding of libdemovlp_hip.so (C ABI declared in include/demovlp_hip.h).

There is NO fallback: if the library is missing or a kernel reports an error, this raises.  The CPU oracle under
``oracle/`` is test infrastructure and is never imported from here.
"""
from __future__ import annotations

import ctypes
import os

_HERE = os.path.dirname(os.path.abspath(__file__))
# The product library exports no developer switch (`dvlp_dev_*`); the same sources built with -DDVLP_DEV are libdemovlp_hip_dev.so
# (include/demovlp_hip_dev.h), which tests and tools switch to with `use_dev_library()` when they need to force a code path.
# DEMOVLP_HIP_LIB: developer override for A/B timing of two builds of the library in one GPU session
LIB_PATH = os.environ.get("DEMOVLP_HIP_LIB") or os.path.join(_HERE, "lib", "libdemovlp_hip.so")
DEV_LIB_PATH = os.path.join(_HERE, "lib", "libdemovlp_hip_dev.so")

F32, BF16 = 0, 1
EPI_GELU, EPI_GELU_BWD, EPI_RELU_BWD, EPI_ACCUM, EPI_LEAKY = 1, 2, 4, 8, 16
ERRORS = {-1: "unsupported dtype", -2: "bad shape / alignment / missing workspace", -3: "HIP launch failed",
          -4: "configuration not supported by this kernel"}

HEADER_PATH = os.path.join(os.path.dirname(_HERE), "include", "demovlp_hip.h")
DEV_HEADER_PATH = os.path.join(os.path.dirname(_HERE), "include", "demovlp_hip_dev.h")
_CT = {"int64_t": ctypes.c_int64, "int": ctypes.c_int, "float": ctypes.c_float}


def _parse_header(path=HEADER_PATH):
    """{name: (restype, [argtypes])} for every function declared in a header of include/."""
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
_DEV_SIGS = _parse_header(DEV_HEADER_PATH)
_RET64 = {n for n, (r, _) in {**_SIGS, **_DEV_SIGS}.items() if r is ctypes.c_int64}

_libs = {}             # path -> (CDLL, has developer switches)
_active = LIB_PATH
_switch_hooks = []     # called with the OLD library still active, before another one takes over (ops drops what it registered there)


class DemoVLPHipError(RuntimeError):
    pass


def load():
    """Load the active shared library (once per path).  Raises if it has not been built (``python -m demovlp_amd.build``)."""
    hit = _libs.get(_active)
    if hit is not None:
        return hit[0]
    # PyTorch-ROCm bundles its own libamdhip64; import it FIRST so this library binds to the same HIP runtime
    # (loading /opt/rocm's copy first gives the process two runtimes and ours then sees "no ROCm-capable device").
    import torch  # noqa: F401
    if not os.path.exists(_active):
        raise DemoVLPHipError(f"{_active} is missing: build it with `python -m demovlp_amd.build` "
                              "(hipcc --offload-arch=gfx950); there is no CPU fallback.")
    lib = ctypes.CDLL(_active)
    dev = hasattr(lib, next(iter(_DEV_SIGS)))
    for name, (ret, argtypes) in {**_SIGS, **(_DEV_SIGS if dev else {})}.items():
        fn = getattr(lib, name)          # AttributeError if a declared symbol is not exported
        fn.argtypes = argtypes
        fn.restype = ret
    _libs[_active] = (lib, dev)
    return lib


def is_dev_library():
    load()
    return _libs[_active][1]


def active_library():
    return _active


def on_library_switch(fn):
    _switch_hooks.append(fn)
    return fn


def use_library(path):
    """Make ``path`` the library every later call goes to (both stay loaded).  State the package registered with the old one -- split-K
    scratch, the deferred-reduction queue -- is dropped first and registered anew, lazily, with the new one."""
    global _active
    if path == _active:
        return
    if _active in _libs:
        for fn in _switch_hooks:
            fn()
    _active = path


def use_dev_library(on=True):
    """Tests and tools that force a code path: switch to (or back from) the -DDVLP_DEV build, the only one that exports `dvlp_dev_*`."""
    use_library(DEV_LIB_PATH if on else LIB_PATH)


def exported_symbols(dev=False):
    return list(_DEV_SIGS if dev else _SIGS)


def call(name, *args):
    """Invoke an entry point; raise on a non-zero status."""
    lib = load()
    if name.startswith("dvlp_dev_") and not _libs[_active][1]:
        raise DemoVLPHipError(f"{name} is a developer switch: the product library does not export it -- "
                              f"demovlp_amd._lib.use_dev_library() switches to {os.path.basename(DEV_LIB_PATH)}")
    fn = getattr(lib, name)
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
