"""Build libgenpercept_hip.so / libgenpercept_hip_f16.so (gfx950 only) in-tree with hipcc.  No JIT cache: the .so files travel with the snapshot."""
from __future__ import annotations

import os
import shutil
import subprocess
import sys
from concurrent.futures import ThreadPoolExecutor

HERE = os.path.dirname(os.path.abspath(__file__))
CSRC = os.path.join(HERE, "csrc")
LIBDIR = os.path.join(HERE, "lib")
# one library per 16-bit element type (csrc/common.h): same sources, same C-ABI, GP_F16 selects IEEE fp16 instead of bf16
LIBS = {"bf16": os.path.join(LIBDIR, "libgenpercept_hip.so"), "fp16": os.path.join(LIBDIR, "libgenpercept_hip_f16.so")}
LIB = LIBS["bf16"]
SOURCES = ["igemm.hip", "conv_halo.hip", "pgemm.hip", "conv_few.hip", "conv_img.hip", "contract.hip", "norm.hip", "attention.hip", "elementwise.hip", "prepost.hip", "microbench.hip", "engine.hip"]
FLAGS = ["--offload-arch=gfx950", "-O3", "-std=c++17", "-fPIC", "-Wall", "-Wno-unused-function"]
FLAGS += os.environ.get("GENPERCEPT_HIPCC_FLAGS", "").split()  # e.g. -DGP_HALO_ABLATIONS=1 for the profiling variants of conv_halo.hip


def _hipcc() -> str:
    for cand in (shutil.which("hipcc"), "/opt/rocm/bin/hipcc"):
        if cand and os.path.exists(cand):
            return cand
    raise RuntimeError("hipcc not found (need ROCm >= 7.0 for gfx950)")


def _stale(out: str, deps) -> bool:
    if not os.path.exists(out):
        return True
    t = os.path.getmtime(out)
    return any(os.path.getmtime(d) > t for d in deps)


def build_library(force: bool = False, verbose: bool = True, precisions=("bf16", "fp16")) -> str:
    """Compile (if stale) and link libgenpercept_hip.so (bf16 elements) and libgenpercept_hip_f16.so (fp16 elements); returns the bf16 path."""
    os.makedirs(LIBDIR, exist_ok=True)
    hipcc = _hipcc()
    headers = [os.path.join(CSRC, h) for h in os.listdir(CSRC) if h.endswith(".h")]
    headers.append(os.path.join(os.path.dirname(HERE), "include", "genpercept_hip.h"))
    jobs, links = [], []
    for prec in precisions:
        objdir = os.path.join(LIBDIR, "obj_" + prec)
        os.makedirs(objdir, exist_ok=True)
        defs = ["-DGP_F16=1"] if prec == "fp16" else []
        objs, dirty = [], False
        for src in SOURCES:
            s = os.path.join(CSRC, src)
            o = os.path.join(objdir, src.replace(".hip", ".o"))
            objs.append(o)
            if force or _stale(o, [s] + headers):
                jobs.append([hipcc, *FLAGS, *defs, "-c", s, "-o", o])
                dirty = True
        if force or dirty or not os.path.exists(LIBS[prec]):
            links.append([hipcc, "--offload-arch=gfx950", "-shared", "-fPIC", *objs, "-o", LIBS[prec]])

    def run(cmd):
        if verbose:
            print(" ".join(cmd), flush=True)
        r = subprocess.run(cmd, capture_output=True, text=True)
        if r.returncode != 0:
            raise RuntimeError(f"hipcc failed:\n{' '.join(cmd)}\n{r.stdout}\n{r.stderr}")

    with ThreadPoolExecutor(max_workers=min(len(jobs), os.cpu_count() or 4) or 1) as ex:
        list(ex.map(run, jobs))
    for cmd in links:
        run(cmd)
    return LIB


ASAN_LIB = os.path.join(LIBDIR, "asan", "libgenpercept_hip_asan.so")


def asan_runtime() -> str:
    """clang's shared AddressSanitizer runtime of the ROCm toolchain (LD_PRELOAD it into the process that loads ASAN_LIB)."""
    import glob
    c = sorted(glob.glob("/opt/rocm/lib/llvm/lib/clang/*/lib/linux/libclang_rt.asan-x86_64.so"))
    if not c:
        raise RuntimeError("libclang_rt.asan-x86_64.so not found under /opt/rocm/lib/llvm")
    return c[-1]


def build_asan_host_library(verbose: bool = False) -> str:
    """SURVEY.md section 5 ("Race detection / sanitizers"): an `-fsanitize=address` HOST build of the C-ABI shim.  Only the host side is
    instrumented (`-Xarch_host -fsanitize=address`: the kernels are validated by goldens on the GPU, not by a sanitizer), -O1, same sources and the same header as the
    product library; tests/test_asan_host.py drives its host paths (argument checks, tensor registration and dtype conversion, struct layouts,
    error strings, log buffers) under the sanitizer.  Never loaded by the product (`engine.load_library` knows only LIBS)."""
    hipcc = _hipcc()
    objdir = os.path.join(LIBDIR, "asan")
    os.makedirs(objdir, exist_ok=True)
    headers = [os.path.join(CSRC, h) for h in os.listdir(CSRC) if h.endswith(".h")] + [os.path.join(os.path.dirname(HERE), "include", "genpercept_hip.h")]
    # (device code is compiled too -- unsanitised, -O1: the host objects reference their fat binary, and HIP registers it when the library loads)
    flags = ["--offload-arch=gfx950", "-O1", "-std=c++17", "-fPIC", "-Xarch_host", "-fsanitize=address", "-Xarch_host", "-fno-omit-frame-pointer",
             "-Xarch_host", "-g", "-w"]
    objs, jobs = [], []
    for src in SOURCES:
        s, o = os.path.join(CSRC, src), os.path.join(objdir, src.replace(".hip", ".o"))
        objs.append(o)
        if _stale(o, [s] + headers):
            jobs.append([hipcc, *flags, "-c", s, "-o", o])

    def run(cmd):
        if verbose:
            print(" ".join(cmd), flush=True)
        r = subprocess.run(cmd, capture_output=True, text=True)
        if r.returncode != 0:
            raise RuntimeError(f"hipcc failed:\n{' '.join(cmd)}\n{r.stdout}\n{r.stderr}")

    with ThreadPoolExecutor(max_workers=min(len(jobs), os.cpu_count() or 4) or 1) as ex:
        list(ex.map(run, jobs))
    if jobs or not os.path.exists(ASAN_LIB):
        run([hipcc, "--offload-arch=gfx950", "-shared", "-fPIC", "-fsanitize=address", "-shared-libsan", *objs, "-o", ASAN_LIB])
    return ASAN_LIB


if __name__ == "__main__":
    if "--asan" in sys.argv:
        print(build_asan_host_library(verbose=True))
    else:
        print(build_library(force="--force" in sys.argv))
