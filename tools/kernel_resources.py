"""Registers / scratch / LDS of every kernel of the built HIP objects (read from the code objects' metadata notes).

    python tools/kernel_resources.py [substring]
"""
import os
import re
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
LIB = os.environ.get("MCN_HIP_LIB", os.path.join(ROOT, "modelcrowdnav_amd", "csrc", "libmcn_hip.so"))
LLVM = "/opt/rocm/lib/llvm/bin"


def code_objects(tmp):
    """One device code object per translation unit (each .o carries its own fat binary)."""
    csrc = os.path.dirname(LIB)
    for obj in sorted(f for f in os.listdir(csrc) if f.endswith(".o")):
        fat, co = os.path.join(tmp, obj + ".fat"), os.path.join(tmp, obj + ".co")
        r = subprocess.run([os.path.join(LLVM, "llvm-objcopy"), "--dump-section", ".hip_fatbin=" + fat,
                            os.path.join(csrc, obj)], stderr=subprocess.DEVNULL)
        if r.returncode:
            continue
        subprocess.run([os.path.join(LLVM, "clang-offload-bundler"), "--unbundle", "--type=o", "--input=" + fat,
                        "--targets=hipv4-amdgcn-amd-amdhsa--gfx950", "--output=" + co], check=True)
        yield co


def main():
    want = sys.argv[1] if len(sys.argv) > 1 else ""
    tmp = "/tmp/mcn_kres"
    os.makedirs(tmp, exist_ok=True)
    for co in code_objects(tmp):
        notes = subprocess.run([os.path.join(LLVM, "llvm-readelf"), "--notes", co], capture_output=True, text=True).stdout
        for blk in notes.split("- .agpr_count:")[1:]:
            name = re.search(r"\.name:\s+(\S+)", blk).group(1)
            dem = subprocess.run(["c++filt", name], capture_output=True, text=True).stdout.strip()
            if want not in dem:
                continue
            get = lambda k: int(re.search(r"\.%s:\s+(\d+)" % k, blk).group(1))
            print("%-90s vgpr %3d agpr %3d sgpr %3d scratch %4d lds %6d vspill %d" % (
                dem[:90], get("vgpr_count"), int(blk.split()[0]), get("sgpr_count"), get("private_segment_fixed_size"),
                get("group_segment_fixed_size"), get("vgpr_spill_count")))


if __name__ == "__main__":
    main()
