"""Builds libspock_amd.so (the C-ABI library with the gfx950 kernels) in-tree with hipcc.

    hipcc --offload-arch=gfx950 -O3 -shared -fPIC csrc/*.cpp csrc/*.hip -o libspock_amd.so

hipcc cross-compiles without a GPU.  The object is git-ignored but travels to the GPU box."""
from __future__ import annotations

import os
import shutil
import subprocess

HERE = os.path.dirname(os.path.abspath(__file__))
CSRC = os.path.join(HERE, "csrc")
SOURCES = ["spock_kernels.hip", "spock_fkernels.hip", "spock_skernels.hip", "spock_ikernels.hip", "spock_lkernels.hip", "spock_rkernels.hip", "spock_bkernels.hip", "spock_abi.cpp", "spock_host.cpp", "spock_rt_hip.cpp"]
HEADERS = ["spock_flat_local.hpp", "spock_gpu_exec.hpp", "spock_device.hpp", "spock_group.hpp", "spock_igroup.hpp", "spock_ilevel.hpp", "spock_rgroup.hpp", "spock_build.hpp", "spock_shapes.hpp", "spock_internal.hpp", "spock_rt.hpp", "spock_sp.hpp", "spock_team_local.hpp", os.path.join("..", "..", "include", "spock_amd.h")]
_API = os.path.join("..", "..", "include", "spock_amd.h")
DEPS = {   # headers each translation unit includes
    "spock_skernels.hip": ["spock_gpu_exec.hpp", "spock_device.hpp", "spock_internal.hpp"],
    "spock_fkernels.hip": ["spock_flat_local.hpp", "spock_gpu_exec.hpp", "spock_device.hpp", "spock_team_local.hpp", "spock_internal.hpp"],
    "spock_kernels.hip": ["spock_gpu_exec.hpp", "spock_device.hpp", "spock_group.hpp", "spock_team_local.hpp", "spock_sp.hpp", "spock_internal.hpp", "spock_shapes.hpp"],
    "spock_ikernels.hip": ["spock_device.hpp", "spock_igroup.hpp", "spock_internal.hpp", "spock_shapes.hpp"],
    "spock_lkernels.hip": ["spock_device.hpp", "spock_igroup.hpp", "spock_ilevel.hpp", "spock_internal.hpp", "spock_shapes.hpp"],
    "spock_rkernels.hip": ["spock_device.hpp", "spock_igroup.hpp", "spock_ilevel.hpp", "spock_rgroup.hpp", "spock_internal.hpp", "spock_shapes.hpp"],
    "spock_bkernels.hip": ["spock_device.hpp", "spock_build.hpp", "spock_internal.hpp"],
    "spock_abi.cpp": ["spock_internal.hpp", "spock_rt.hpp", _API],
    "spock_host.cpp": ["spock_internal.hpp"],
    "spock_rt_hip.cpp": ["spock_rt.hpp"],
}
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


def build(force: bool = False, verbose: bool = False, out: str | None = None, tmp_name: str = "build") -> str:
    global OUT
    if out:
        OUT = out
    if not force and not needs_build():
        return OUT
    objs = []
    tmp = os.path.join(HERE, tmp_name)
    os.makedirs(tmp, exist_ok=True)
    flags = ["--offload-arch=gfx950", "-O3", "-std=c++17", "-fPIC", "-Wall", "-Wno-unused-variable", "-Wno-cuda-compat"]
    flags += os.environ.get("SPOCK_BUILD_DEFS", "").split()      # experiment builds (tools/): extra -D... flags
    kernel_only = os.environ.get("SPOCK_BUILD_KERNEL_FLAGS", "").split()   # flags for the kernel translation units (*.hip) alone
    def compile_one(s):
        o = os.path.join(tmp, s + ".o")
        cmd = [_hipcc()] + flags + (kernel_only if s.endswith(".hip") else []) + (["-x", "hip"] if s.endswith(".cpp") else []) + ["-c", os.path.join(CSRC, s), "-o", o]
        if verbose:
            print(" ".join(cmd))
        # recompile only what changed
        deps = [os.path.join(CSRC, s)] + [os.path.join(CSRC, h) for h in DEPS[s]]
        if force or not os.path.exists(o) or any(os.path.getmtime(d) > os.path.getmtime(o) for d in deps):
            subprocess.check_call(cmd)
        return o

    from concurrent.futures import ThreadPoolExecutor
    with ThreadPoolExecutor(max_workers=min(8, len(SOURCES))) as ex:   # the two kernel files take minutes each: side by side
        objs = list(ex.map(compile_one, SOURCES))
    cmd = [_hipcc(), "--offload-arch=gfx950", "-shared", "-fPIC", "-o", OUT] + objs
    if verbose:
        print(" ".join(cmd))
    subprocess.check_call(cmd)
    return OUT


OUT_F32 = os.path.join(HERE, "libspock_amd_f32.so")


def build_f32(force: bool = False, verbose: bool = False) -> str:
    """The fp32 variant (BASELINE config 5's tolerance study): the same sources with `real` = float and single-precision
    literals in the kernel translation units; compile-time shapes narrowed to the benchmark's (the run-time-dimension kernels
    cover everything else).  Same C ABI (double at the boundary)."""
    global OUT
    keep, keep_defs, keep_k = OUT, os.environ.get("SPOCK_BUILD_DEFS"), os.environ.get("SPOCK_BUILD_KERNEL_FLAGS")
    os.environ["SPOCK_BUILD_DEFS"] = "-DSPOCK_REAL=float -DSPOCK_STATIC_SHAPES(X)=X(5,2,2,5)X(5,5,2,5)"
    # single-precision literals in the KERNEL translation units only: the host build step (Riccati, eigen-decompositions, the step-size
    # defaults 0.99 / 3.28 / 10.0 and their thresholds in spock_host.cpp / spock_abi.cpp) stays double, so that the fp32 study
    # perturbs the device arithmetic and nothing else
    os.environ["SPOCK_BUILD_KERNEL_FLAGS"] = "-cl-single-precision-constant"
    try:
        return build(force=force, verbose=verbose, out=OUT_F32, tmp_name="build_f32")
    finally:
        OUT = keep
        for k, v in (("SPOCK_BUILD_DEFS", keep_defs), ("SPOCK_BUILD_KERNEL_FLAGS", keep_k)):
            if v is None:
                os.environ.pop(k, None)
            else:
                os.environ[k] = v


if __name__ == "__main__":
    print(build(force=True, verbose=True))
    print(build_f32(force=True, verbose=True))
