"""Static instruction mix of the p=3 hex kernels from `hipcc -S` output (tools/isa_mix.py file.s)."""
import collections
import re
import sys

lines = open(sys.argv[1]).read().splitlines()
pat = sys.argv[2] if len(sys.argv) > 2 else "ILi3ELi3E"
starts = [(i, l.split(":")[0]) for i, l in enumerate(lines) if re.match(r"^_ZN6tpsrhs\S*:", l)]
for n, (i, name) in enumerate(starts):
    if pat not in name:
        continue
    j = starts[n + 1][0] if n + 1 < len(starts) else len(lines)
    c = collections.Counter()
    for line in lines[i:j]:
        t = line.strip().split()[0] if line.strip() else ""
        if re.match(r"^(ds_|global_|v_|s_|buffer_|scratch_|flat_)", t):
            c[t] += 1
    keep = {k: v for k, v in c.items() if k.startswith(("ds_", "global_", "flat_", "scratch_")) or k in
            ("s_waitcnt", "s_barrier", "v_fma_f64", "v_fmac_f64_e32", "v_mul_f64", "v_add_f64", "v_rcp_f64_e32",
             "v_rsq_f64_e32", "v_div_scale_f64", "v_sqrt_f64_e32")}
    print(name[10:44], "total", sum(c.values()), keep)
