"""Scratch / spill / register report of every kernel in the built objects (tps_amd/csrc/_obj/*.o), read from the
code-object metadata -- no recompilation.   python tools/spill_report.py [substring filter] [--all]
Prints the kernels that use scratch; the last line counts them."""
import os
import re
import subprocess
import sys
import tempfile

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
OBJ = os.path.join(ROOT, "tps_amd", "csrc", "_obj")
LLVM = "/opt/rocm/lib/llvm/bin"
show_all = "--all" in sys.argv  # every matching kernel, not only those with scratch (adds the LDS bytes)
args = [a for a in sys.argv[1:] if a != "--all"]
flt = args[0] if args else ""
total = spilled = 0
with tempfile.TemporaryDirectory() as tmp:
    for o in sorted(os.listdir(OBJ)):
        if not o.endswith(".o"):
            continue
        co = os.path.join(tmp, o + ".co")
        r = subprocess.run([f"{LLVM}/clang-offload-bundler", "--unbundle", "--type=o", "--targets=hipv4-amdgcn-amd-amdhsa--gfx950",
                            f"--input={os.path.join(OBJ, o)}", f"--output={co}"], capture_output=True, text=True)
        if r.returncode != 0 or not os.path.exists(co):
            # the device code sits in the .hip_fatbin section of the host object
            fb = os.path.join(tmp, o + ".fatbin")
            subprocess.run([f"{LLVM}/llvm-objcopy", "--dump-section", f".hip_fatbin={fb}", os.path.join(OBJ, o)], check=True)
            subprocess.run([f"{LLVM}/clang-offload-bundler", "--unbundle", "--type=o", "--targets=hipv4-amdgcn-amd-amdhsa--gfx950",
                            f"--input={fb}", f"--output={co}"], check=True)
        notes = subprocess.run([f"{LLVM}/llvm-readelf", "--notes", co], capture_output=True, text=True).stdout
        for blk in notes.split("- .agpr_count")[1:]:
            name = re.search(r"\.name:\s+(\S+)", blk).group(1)
            if name.endswith(".kd") or flt not in name:
                continue
            get = lambda k: int(re.search(r"\.%s:\s+(\d+)" % k, blk).group(1))
            total += 1
            scratch, vs, ss = get("private_segment_fixed_size"), get("vgpr_spill_count"), get("sgpr_spill_count")
            if scratch or vs:
                spilled += 1
            if scratch or vs or show_all:
                dn = subprocess.run(["c++filt", name], capture_output=True, text=True).stdout.strip()
                dn = re.sub(r"\(.*", "", dn).replace("tpsrhs::", "").replace("void ", "")
                print(f"{o:24s} {dn:100s} vgpr {get('vgpr_count'):3d} scratch {scratch:4d} vgpr_spill {vs:3d} sgpr_spill {ss:3d}"
                      + (f" lds {get('group_segment_fixed_size'):6d}" if show_all else ""))
print(f"{spilled} of {total} kernels use scratch")
