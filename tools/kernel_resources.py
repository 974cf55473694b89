"""VGPR / scratch / occupancy of every kernel of a .hip file (hipcc -Rpass-analysis=kernel-resource-usage):
tools/kernel_resources.py tps_amd/csrc/plasma3d.hip [substring filter]"""
import re
import subprocess
import sys

src = sys.argv[1]
flt = sys.argv[2] if len(sys.argv) > 2 else ""
out = subprocess.run(["/opt/rocm/bin/hipcc", "--offload-arch=gfx950", "-O3", "-std=c++17", "-fPIC", "-c", src, "-o",
                      "/dev/null", "-Rpass-analysis=kernel-resource-usage"], capture_output=True, text=True).stderr
cur = None
rows = {}
for line in out.splitlines():
    m = re.search(r"Function Name: (\S+)", line)
    if m:
        cur = subprocess.run(["c++filt", m.group(1)], capture_output=True,
                             text=True).stdout.strip()
        cur = re.sub(r"\(.*", "", cur).replace("tpsrhs::", "").replace("void ", "")
        rows[cur] = {}
        continue
    for key, pat in (("vgpr", r" VGPRs: (\d+)"), ("agpr", r"AGPRs: (\d+)"), ("scratch", r"ScratchSize \[bytes/lane\]: (\d+)"),
                     ("occ", r"Occupancy \[waves/SIMD\]: (\d+)"), ("lds", r"LDS Size \[bytes/block\]: (\d+)"),
                     ("sgpr", r" SGPRs: (\d+)")):
        m = re.search(pat, line)
        if m and cur:
            rows[cur][key] = int(m.group(1))
for k, v in rows.items():
    if flt in k:
        print(f"{k:110s} vgpr {v.get('vgpr'):4d} agpr {v.get('agpr', 0):3d} scratch {v.get('scratch'):5d} "
              f"occ {v.get('occ'):2d} lds {v.get('lds'):6d}")
