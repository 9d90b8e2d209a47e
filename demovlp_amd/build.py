"""Build libdemovlp_hip.so in-tree with hipcc for gfx950 (cross-compiles without a GPU)."""
from __future__ import annotations

import os
import subprocess
import sys
from concurrent.futures import ThreadPoolExecutor

HERE = os.path.dirname(os.path.abspath(__file__))
CSRC = os.path.join(HERE, "csrc")
LIBDIR = os.path.join(HERE, "lib")
LIB = os.path.join(LIBDIR, "libdemovlp_hip.so")
DEV_LIB = os.path.join(LIBDIR, "libdemovlp_hip_dev.so")      # the same sources under -DDVLP_DEV: exports the dvlp_dev_* switches (tests, tools)
SOURCES = ["gemm.hip", "gemm_rb.hip", "norm.hip", "attention.hip", "embed.hip", "xattn.hip", "xfused.hip", "losses.hip", "select.hip", "optim.hip", "dropout.hip"]
FLAGS = ["--offload-arch=gfx950", "-O3", "-std=c++17", "-fPIC", "-Wno-unused-value"]


def _hipcc() -> str:
    for c in (os.environ.get("HIPCC"), "/opt/rocm/bin/hipcc", "hipcc"):
        if c and (os.path.isabs(c) and os.path.exists(c) or not os.path.isabs(c)):
            return c
    raise RuntimeError("hipcc not found")


def _stale(out: str, deps) -> bool:
    if not os.path.exists(out):
        return True
    t = os.path.getmtime(out)
    return any(os.path.getmtime(d) > t for d in deps)


def build(force: bool = False, verbose: bool = True, dev: bool = True) -> str:
    """Product library (no developer switch exported) and, with ``dev``, libdemovlp_hip_dev.so beside it."""
    os.makedirs(LIBDIR, exist_ok=True)
    hipcc = _hipcc()
    hdr = os.path.join(CSRC, "common.h")
    variants = [("", [], LIB)] + ([("_dev", ["-DDVLP_DEV"], DEV_LIB)] if dev else [])
    jobs, links = [], []
    for suffix, defs, lib in variants:
        objs = []
        for s in SOURCES:
            src = os.path.join(CSRC, s)
            obj = os.path.join(LIBDIR, s.replace(".hip", suffix + ".o"))
            objs.append(obj)
            if force or _stale(obj, [src, hdr]):
                jobs.append([hipcc, *FLAGS, *defs, "-c", src, "-o", obj])
        links.append((lib, objs))

    def run(cmd):
        if verbose:
            print(" ".join(cmd), flush=True)
        r = subprocess.run(cmd, capture_output=True, text=True)
        if r.returncode != 0:
            raise RuntimeError(f"hipcc failed: {' '.join(cmd)}\n{r.stdout}\n{r.stderr}")

    with ThreadPoolExecutor(max_workers=min(8, os.cpu_count() or 4)) as ex:
        list(ex.map(run, jobs))
    for lib, objs in links:
        if force or jobs or _stale(lib, objs):
            run([hipcc, "--offload-arch=gfx950", "-shared", "-fPIC", "-o", lib, *objs])
    return LIB


def build_stamp(verbose: bool = True) -> str:
    """Developer build for tools/p8_timeline.py: the same library with gemm.hip compiled under -DDVLP_STAMP (per-workgroup
    s_memrealtime stamps in the 256-row GEMM).  Never loaded by the product (``_lib.LIB_PATH`` points at libdemovlp_hip.so)."""
    build(verbose=verbose)
    hipcc = _hipcc()
    out = os.path.join(LIBDIR, "libdemovlp_hip_stamp.so")
    obj = os.path.join(LIBDIR, "gemm_stamp.o")
    src = os.path.join(CSRC, "gemm.hip")
    objs = [os.path.join(LIBDIR, s.replace(".hip", "_dev.o")) for s in SOURCES if s != "gemm.hip"] + [obj]
    if _stale(obj, [src, os.path.join(CSRC, "common.h")]):
        subprocess.run([hipcc, *FLAGS, "-DDVLP_STAMP", "-DDVLP_DEV", "-c", src, "-o", obj], check=True)
    if _stale(out, objs):
        subprocess.run([hipcc, "--offload-arch=gfx950", "-shared", "-fPIC", "-o", out, *objs], check=True)
    return out


if __name__ == "__main__":
    print(build_stamp() if "--stamp" in sys.argv else build(force="--force" in sys.argv))
