"""Builds libspock_amd.so (the C-ABI library with the gfx950 kernels) in-tree with hipcc.

    hipcc --offload-arch=gfx950 -O3 -shared -fPIC csrc/*.cpp csrc/*.hip -o libspock_amd.so

hipcc cross-compiles without a GPU.  The object is git-ignored but travels to the GPU box."""
from __future__ import annotations

import os
import shutil
import subprocess

HERE = os.path.dirname(os.path.abspath(__file__))
CSRC = os.path.join(HERE, "csrc")
SOURCES = ["spock_kernels.hip", "spock_abi.cpp", "spock_host.cpp", "spock_rt_hip.cpp"]
HEADERS = ["spock_device.hpp", "spock_group.hpp", "spock_internal.hpp", "spock_rt.hpp", "spock_sp.hpp", "spock_team_local.hpp", os.path.join("..", "..", "include", "spock_amd.h")]
OUT = os.path.join(HERE, "libspock_amd.so")


def _hipcc() -> str:
    for cand in (shutil.which("hipcc"), "/opt/rocm/bin/hipcc"):
        if cand and os.path.exists(cand):
            return cand
    raise RuntimeError("hipcc not found (needed to build the gfx950 kernels)")


def needs_build() -> bool:
    if not os.path.exists(OUT):
        return True
    t = os.path.getmtime(OUT)
    deps = [os.path.join(CSRC, s) for s in SOURCES + HEADERS] + [os.path.abspath(__file__)]
    return any(os.path.getmtime(d) > t for d in deps)


def build(force: bool = False, verbose: bool = False) -> str:
    if not force and not needs_build():
        return OUT
    objs = []
    tmp = os.path.join(HERE, "build")
    os.makedirs(tmp, exist_ok=True)
    flags = ["--offload-arch=gfx950", "-O3", "-std=c++17", "-fPIC", "-Wall", "-Wno-unused-variable", "-Wno-cuda-compat"]
    for s in SOURCES:
        o = os.path.join(tmp, s + ".o")
        cmd = [_hipcc()] + flags + (["-x", "hip"] if s.endswith(".cpp") else []) + ["-c", os.path.join(CSRC, s), "-o", o]
        if verbose:
            print(" ".join(cmd))
        subprocess.check_call(cmd)
        objs.append(o)
    cmd = [_hipcc(), "--offload-arch=gfx950", "-shared", "-fPIC", "-o", OUT] + objs
    if verbose:
        print(" ".join(cmd))
    subprocess.check_call(cmd)
    return OUT


if __name__ == "__main__":
    print(build(force=True, verbose=True))
