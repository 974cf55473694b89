"""Scratch / spill census of the built kernels, read from the code-object metadata of tps_amd/csrc/_obj/*.o (no
recompilation).  Shared by tools/spill_report.py and tests/test_spill_allowlist.py."""
import os
import re
import subprocess
import tempfile

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
OBJ = os.path.join(ROOT, "tps_amd", "csrc", "_obj")
LLVM = "/opt/rocm/lib/llvm/bin"
TARGET = "hipv4-amdgcn-amd-amdhsa--gfx950"


def _code_object(obj, tmp):
    co = os.path.join(tmp, os.path.basename(obj) + ".co")
    r = subprocess.run([f"{LLVM}/clang-offload-bundler", "--unbundle", "--type=o", f"--targets={TARGET}", f"--input={obj}",
                        f"--output={co}"], capture_output=True, text=True)
    if r.returncode != 0 or not os.path.exists(co) or os.path.getsize(co) == 0:
        fb = co + ".fatbin"  # the device code sits in the .hip_fatbin section of the host object
        subprocess.run([f"{LLVM}/llvm-objcopy", "--dump-section", f".hip_fatbin={fb}", obj, co + ".discard.o"], check=True)  # (an explicit output: with one file name llvm-objcopy rewrites its INPUT in place -- and bumps the mtime the build compares)
        subprocess.run([f"{LLVM}/clang-offload-bundler", "--unbundle", "--type=o", f"--targets={TARGET}", f"--input={fb}",
                        f"--output={co}"], check=True)
    return co


def short_name(mangled):
    dn = subprocess.run(["c++filt", mangled], capture_output=True, text=True).stdout.strip()
    dn = re.sub(r"\(.*", "", dn).replace("tpsrhs::", "").replace("void ", "")
    return re.sub(r"\s+", "", dn)


def census(objdir=OBJ):
    """-> list of dicts {unit, kernel (demangled, no spaces), vgpr, agpr, sgpr, scratch, vgpr_spill, sgpr_spill, lds}"""
    out = []
    with tempfile.TemporaryDirectory() as tmp:
        for o in sorted(os.listdir(objdir)):
            if not o.endswith(".o"):
                continue
            notes = subprocess.run([f"{LLVM}/llvm-readelf", "--notes", _code_object(os.path.join(objdir, o), tmp)],
                                   capture_output=True, text=True).stdout
            names = []
            for blk in notes.split("- .agpr_count")[1:]:
                name = re.search(r"\.name:\s+(\S+)", blk).group(1)
                if name.endswith(".kd"):
                    continue
                get = lambda k: int(re.search(r"\.%s:\s+(\d+)" % k, blk).group(1))
                agpr = int(re.match(r":\s+(\d+)", blk).group(1))
                names.append(dict(unit=o, mangled=name, vgpr=get("vgpr_count"), agpr=agpr, sgpr=get("sgpr_count"),
                                  scratch=get("private_segment_fixed_size"), vgpr_spill=get("vgpr_spill_count"),
                                  sgpr_spill=get("sgpr_spill_count"), lds=get("group_segment_fixed_size")))
            dem = subprocess.run(["c++filt"], input="\n".join(k["mangled"] for k in names), capture_output=True, text=True).stdout.splitlines()
            for k, d in zip(names, dem):
                d = re.sub(r"\(.*", "", d).replace("tpsrhs::", "").replace("void ", "")
                k["kernel"] = re.sub(r"\s+", "", d)
            out.extend(names)
    return out
