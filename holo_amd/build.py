"""Builds libholo_spf_hip.so for gfx950 in-tree (holo_amd/libholo_spf_hip.so).

hipcc cross-compiles without a GPU; the built .so travels to the GPU box with the repo snapshot.
"""
from __future__ import annotations

import os
import shutil
import subprocess

HERE = os.path.dirname(os.path.abspath(__file__))
CSRC = os.path.join(HERE, "csrc")
LIB = os.path.join(HERE, "libholo_spf_hip.so")
SOURCES = ["spf_capi.hip", "hub_sort.hip"]
DEPS = ["spf_capi.hip", "spf_kernels.hip.h", "spf_repair.hip.h", "graph_build.hip.h", "spf_multi.hip.h", "hub_sort.hip", "hub_sort.h", os.path.join("..", "..", "include", "holo_spf_hip.h")]


def hipcc_path() -> str:
    for cand in (shutil.which("hipcc"), "/opt/rocm/bin/hipcc"):
        if cand and os.path.exists(cand):
            return cand
    raise RuntimeError("hipcc not found: cannot build libholo_spf_hip.so")


def needs_build() -> bool:
    if not os.path.exists(LIB):
        return True
    t = os.path.getmtime(LIB)
    return any(os.path.getmtime(os.path.join(CSRC, d)) > t for d in DEPS)


def build_lib(force: bool = False, verbose: bool = False) -> str:
    if not force and not needs_build():
        return LIB
    # -amdgpu-kernarg-preload-count: the first 16 kernel-argument dwords arrive in SGPRs at wave launch instead of through
    # a scalar load — one dependent round trip less at the head of every wave (measured: +1.2 % on the bench, more on the
    # latency-bound sparse sweeps); the compiler keeps a compatible prologue for firmware that does not preload
    flags = ["--offload-arch=gfx950", "-O3", "-std=c++17", "-fPIC", "-Wall", "-Wno-unused-function",
             "-mllvm", "-amdgpu-kernarg-preload-count=16"]
    flags += os.environ.get("HSPF_BUILD_DEFS", "").split()      # experiments: -DNAME=value
    # one object per source, compiled side by side (the engine and the rocPRIM sorts of the hub-mode graph build), then linked
    objs, procs = [], []
    for src in SOURCES:
        obj = os.path.join(CSRC, src + ".o")
        cmd = [hipcc_path()] + flags + ["-c", os.path.join(CSRC, src), "-o", obj]
        if verbose:
            print(" ".join(cmd))
        procs.append((cmd, subprocess.Popen(cmd, cwd=CSRC)))
        objs.append(obj)
    for cmd, p in procs:
        if p.wait() != 0:
            raise subprocess.CalledProcessError(p.returncode, cmd)
    link = [hipcc_path(), "--offload-arch=gfx950", "-shared", "-fPIC"] + objs + ["-ldl", "-pthread", "-o", LIB + ".tmp"]
    if verbose:
        print(" ".join(link))
    subprocess.check_call(link, cwd=CSRC)
    for obj in objs:
        os.remove(obj)
    os.replace(LIB + ".tmp", LIB)
    return LIB


if __name__ == "__main__":
    print(build_lib(force=True, verbose=True))
