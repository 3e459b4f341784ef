"""Experiment builds of the element-major kernels: compile spock_lkernels.hip (the level-streaming kernels) for the (5, 2, 2, 5) shape only, with extra -D flags,
into exp/libspock_<name>.so (git-ignored, travels to the GPU box); the other objects come from the regular build.

    python tools/build_ivariant.py w5 -DSPOCK_ILV_LEVEL_WAVES=5          (NOLINK=1: compile only, print the register table)
"""
import os, subprocess, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import importlib.util
spec = importlib.util.spec_from_file_location("b", os.path.join(ROOT, "spock.jl_amd", "build.py")); b = importlib.util.module_from_spec(spec); spec.loader.exec_module(b)
name, extra = sys.argv[1], sys.argv[2:]
out_dir = os.path.join(ROOT, "exp"); os.makedirs(out_dir, exist_ok=True)
obj = os.path.join(out_dir, f"ikernels_{name}.o")
flags = ["--offload-arch=gfx950", "-O3", "-std=c++17", "-fPIC", "-Wno-unused-variable", "-Wno-cuda-compat",
         "-DSPOCK_STATIC_SHAPES(X)=X(5,2,2,5)"] + extra
SRC = os.environ.get("IVARIANT_SRC", "spock_lkernels.hip")    # IVARIANT_SRC=spock_ikernels.hip: the group kernels (crown) instead
subprocess.check_call([b._hipcc()] + flags + ["-c", os.path.join(b.CSRC, SRC), "-o", obj])
os.environ["KOBJ"] = obj
subprocess.call([sys.executable, os.path.join(ROOT, "tools", "kernel_resources.py"), "ilevel", "5, 2>"])
if not os.environ.get("NOLINK"):
    # host-side objects: compiled here when stale (seconds); the instance-major kernels come from the last regular build
    others = [os.path.join(b.HERE, "build", s_) for s_ in ("spock_kernels.hip.o", "spock_ikernels.hip.o", "spock_lkernels.hip.o", "spock_bkernels.hip.o") if s_ != SRC + ".o"]
    for src in ("spock_abi.cpp", "spock_host.cpp", "spock_rt_hip.cpp"):
        o = os.path.join(out_dir, "host_" + src + ".o")
        deps = [os.path.join(b.CSRC, src)] + [os.path.join(b.CSRC, h) for h in b.DEPS[src]]
        if not os.path.exists(o) or any(os.path.getmtime(d_) > os.path.getmtime(o) for d_ in deps):
            subprocess.check_call([b._hipcc()] + flags[:6] + ["-x", "hip", "-c", os.path.join(b.CSRC, src), "-o", o])
        others.append(o)
    out = os.path.join(out_dir, f"libspock_{name}.so")
    subprocess.check_call([b._hipcc(), "--offload-arch=gfx950", "-shared", "-fPIC", "-o", out, obj] + others)
    print(out)
