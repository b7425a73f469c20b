"""Builds libwhisper_amd.so (HIP kernels + C-ABI) in-tree for gfx950 with hipcc.

    python -m whisper_amd.build            # incremental
    python -m whisper_amd.build --force

hipcc cross-compiles without a GPU.  The .so lands next to this file so it travels with the source tree.
"""
from __future__ import annotations

import os
import subprocess
import sys
from concurrent.futures import ThreadPoolExecutor

HERE = os.path.dirname(os.path.abspath(__file__))
CSRC = os.path.join(HERE, "csrc")
OBJ = os.path.join(HERE, "csrc", "build")
LIB = os.path.join(HERE, "libwhisper_amd.so")
SOURCES = ["mel.hip", "gemm.hip", "attention.hip", "flash.hip", "xattn.hip", "elementwise.hip", "beam.hip", "engine.hip"]
FLAGS = ["--offload-arch=gfx950", "-O3", "-std=c++17", "-fPIC", "-ffp-contract=off", "-Wall", "-Wno-unused-function"]


def _stale(target: str, deps) -> bool:
    if not os.path.exists(target):
        return True
    t = os.path.getmtime(target)
    return any(os.path.getmtime(d) > t for d in deps)


def build_library(force: bool = False, verbose: bool = False) -> str:
    os.makedirs(OBJ, exist_ok=True)
    headers = [os.path.join(CSRC, "common.h"), os.path.join(HERE, "..", "include", "whisper_amd.h")]
    srcs = [s for s in SOURCES if os.path.exists(os.path.join(CSRC, s))]
    objs, jobs = [], []
    for s in srcs:
        src, obj = os.path.join(CSRC, s), os.path.join(OBJ, s.replace(".hip", ".o"))
        objs.append(obj)
        if force or _stale(obj, [src] + headers):
            jobs.append(["hipcc", *FLAGS, "-c", src, "-o", obj])

    def run(cmd):
        if verbose:
            print(" ".join(cmd), flush=True)
        r = subprocess.run(cmd, capture_output=True, text=True)
        if r.returncode != 0:
            raise RuntimeError(f"hipcc failed: {' '.join(cmd)}\n{r.stdout}\n{r.stderr}")
        if verbose and r.stderr.strip():
            print(r.stderr)

    with ThreadPoolExecutor(max_workers=min(6, max(1, len(jobs)))) as ex:
        list(ex.map(run, jobs))
    if jobs or force or _stale(LIB, objs):
        run(["hipcc", "--offload-arch=gfx950", "-shared", "-fPIC", "-o", LIB, *objs])
    return LIB


if __name__ == "__main__":
    print(build_library(force="--force" in sys.argv, verbose=True))
