"""Register / scratch / LDS usage of every kernel in the built library (reads the gfx950 code object's metadata).

    python tools/kernel_resources.py [filter-substring ...]
"""
import os, re, subprocess, sys, tempfile

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
LLVM = "/opt/rocm/lib/llvm/bin"


def main():
    k = os.environ.get("KOBJ", "spock_kernels.hip.o")
    obj = k if os.path.isabs(k) else os.path.join(ROOT, "spock.jl_amd", "build", k)
    d = tempfile.mkdtemp()
    fat, co = os.path.join(d, "fat.bin"), os.path.join(d, "k.co")
    subprocess.check_call([f"{LLVM}/llvm-objcopy", "--dump-section", f".hip_fatbin={fat}", obj])
    subprocess.check_call([f"{LLVM}/clang-offload-bundler", "--unbundle", "--type=o", f"--input={fat}",
                           "--targets=hipv4-amdgcn-amd-amdhsa--gfx950", f"--output={co}"])
    notes = subprocess.run([f"{LLVM}/llvm-readelf", "--notes", co], capture_output=True, text=True).stdout
    rows = []
    for blk in notes.split("- .agpr_count:")[1:]:
        g = lambda k: re.search(rf"\.{k}:\s+(\S+)", blk)
        name = g("name").group(1)
        dem = subprocess.run(["c++filt", name], capture_output=True, text=True).stdout.strip()
        dem = re.sub(r"spock::", "", dem).replace("void ", "")
        rows.append((dem, int(g("vgpr_count").group(1)), int(re.match(r"\s*(\d+)", blk).group(1)), int(g("sgpr_count").group(1)),
                     int(g("private_segment_fixed_size").group(1)), int(g("group_segment_fixed_size").group(1))))
    flt = sys.argv[1:]
    print(f"{'vgpr':>5} {'agpr':>5} {'sgpr':>5} {'scratch':>8} {'lds':>6}  kernel")
    for dem, vg, ag, sg, scr, lds in sorted(rows):
        if all(f in dem for f in flt):
            print(f"{vg:5d} {ag:5d} {sg:5d} {scr:8d} {lds:6d}  {dem[:120]}")


if __name__ == "__main__":
    main()
