"""Highest VGPR index used between consecutive s_barrier's of a kernel (hipcc -S output):
tools/vgpr_segments.py file.s k_flux [ILi3ELi3E]"""
import re
import sys

lines = open(sys.argv[1]).read().splitlines()
kern = sys.argv[2]
pat = sys.argv[3] if len(sys.argv) > 3 else "ILi3ELi3E"
starts = [(i, l.split(":")[0]) for i, l in enumerate(lines) if re.match(r"^_ZN6tpsrhs\S*:", l)]
for n, (i, name) in enumerate(starts):
    if kern not in name or pat not in name:
        continue
    j = starts[n + 1][0] if n + 1 < len(starts) else len(lines)
    body = lines[i:j]
    mx = []
    for l in body:
        regs = [int(x) for x in re.findall(r"\bv(\d+)\b", l)] + [int(b) for a, b in re.findall(r"v\[(\d+):(\d+)\]", l)]
        mx.append(max(regs) if regs else 0)
    s0 = 0
    for k, l in enumerate(body):
        if "s_barrier" in l or k == len(body) - 1:
            seg = body[s0:k + 1]
            nv = sum(1 for x in seg if re.match(r"\s+v_", x))
            nl = sum(1 for x in seg if re.match(r"\s+ds_", x))
            ng = sum(1 for x in seg if re.match(r"\s+global_", x))
            print(f"seg {s0:5d}-{k:5d} maxv {max(mx[s0:k+1] or [0]):4d} valu {nv:4d} lds {nl:4d} global {ng:3d}")
            s0 = k + 1
