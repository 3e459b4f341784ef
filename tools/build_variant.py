"""Experiment builds: compile the kernels with extra -D flags into exp/libspock_<name>.so (git-ignored, travels to the
GPU box).  Only the (5, 2, 2, 5) compile-time shape is kept, to build in about a minute.

    python tools/build_variant.py w4 -DSPOCK_WAVES_PER_EU=4
"""
import os, subprocess, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import importlib.util
spec = importlib.util.spec_from_file_location("b", os.path.join(ROOT, "spock.jl_amd", "build.py")); b = importlib.util.module_from_spec(spec); spec.loader.exec_module(b)
b.build()
name, extra = sys.argv[1], sys.argv[2:]
out_dir = os.path.join(ROOT, "exp"); os.makedirs(out_dir, exist_ok=True)
obj = os.path.join(out_dir, f"kernels_{name}.o")
flags = ["--offload-arch=gfx950", "-O3", "-std=c++17", "-fPIC", "-Wno-unused-variable", "-Wno-cuda-compat",
         "-DSPOCK_STATIC_SHAPES(X)=X(5,2,2,5)"] + extra
subprocess.check_call([b._hipcc()] + flags + ["-c", os.path.join(b.CSRC, "spock_kernels.hip"), "-o", obj])
others = [os.path.join(b.HERE, "build", s + ".o") for s in b.SOURCES if s != "spock_kernels.hip"]
out = os.path.join(out_dir, f"libspock_{name}.so")
subprocess.check_call([b._hipcc(), "--offload-arch=gfx950", "-shared", "-fPIC", "-o", out, obj] + others)
os.remove(obj)
print(out)
