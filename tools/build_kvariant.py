"""Experiment builds of the instance-major kernels (spock_kernels.hip) with the run-time-dimension variants only, into
exp/libspock_<name>.so; the other objects come from the regular build.     python tools/build_kvariant.py mf20 [-D...]"""
import os, subprocess, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import importlib.util
spec = importlib.util.spec_from_file_location("b", os.path.join(ROOT, "spock.jl_amd", "build.py")); b = importlib.util.module_from_spec(spec); spec.loader.exec_module(b)
name, extra = sys.argv[1], sys.argv[2:]
out_dir = os.path.join(ROOT, "exp"); os.makedirs(out_dir, exist_ok=True)
obj = os.path.join(out_dir, f"kernels_{name}.o")
flags = ["--offload-arch=gfx950", "-O3", "-std=c++17", "-fPIC", "-Wno-unused-variable", "-Wno-cuda-compat", "-DSPOCK_STATIC_SHAPES(X)=X(5,2,2,5)"] + extra
subprocess.check_call([b._hipcc()] + flags + ["-c", os.path.join(b.CSRC, "spock_kernels.hip"), "-o", obj])
others = [os.path.join(b.HERE, "build", s) for s in ("spock_ikernels.hip.o", "spock_lkernels.hip.o", "spock_bkernels.hip.o", "spock_abi.cpp.o", "spock_host.cpp.o", "spock_rt_hip.cpp.o")]
out = os.path.join(out_dir, f"libspock_{name}.so")
subprocess.check_call([b._hipcc(), "--offload-arch=gfx950", "-shared", "-fPIC", "-o", out, obj] + others)
os.remove(obj)
print(out)
