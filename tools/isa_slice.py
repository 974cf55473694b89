"""Backward slice in a disassembled kernel (tools/kernel_isa.py --dump): who wrote the register (pair) read at a line?
    python tools/isa_slice.py <dump.s> <line (1-based)> <reg, e.g. v[44:45] or v12 or s[2:3]> [depth]
Straight-line search upwards (no control-flow reconstruction: read the result with the branches in mind)."""
import re
import sys


def regs(tok):
    tok = tok.strip().rstrip(",")
    tok = tok.lstrip("-|").rstrip("|")
    m = re.match(r"^([vsa])\[(\d+):(\d+)\]$", tok)
    if m:
        return {(m.group(1), i) for i in range(int(m.group(2)), int(m.group(3)) + 1)}
    m = re.match(r"^([vsa])(\d+)$", tok)
    if m:
        return {(m.group(1), int(m.group(2)))}
    return set()


def parse(line):
    body = line.split("//")[0].strip()
    parts = body.split(None, 1)
    op = parts[0]
    ops = [o.strip() for o in parts[1].split(",")] if len(parts) > 1 else []
    ops = [o.split()[0] if o else o for o in ops]
    return op, ops


def defs_uses(op, ops):
    if op.startswith(("ds_write", "global_store", "scratch_store", "buffer_store", "s_waitcnt", "s_cbranch", "s_branch", "s_barrier", "s_nop")):
        return set(), set().union(*[regs(o) for o in ops]) if ops else set()
    if op.startswith(("v_cmp", "s_cmp")) and not op.startswith("v_cmpx"):
        d = regs(ops[0]) if ops and not op.startswith("s_cmp") and op.endswith("_e64") else set()
        return d, set().union(*[regs(o) for o in ops[(1 if d else 0):]]) if ops else set()
    if op in ("v_writelane_b32", "v_fmac_f64_e32", "v_fmac_f32_e32", "v_mac_f32_e32") and ops:
        return regs(ops[0]), set().union(*[regs(o) for o in ops])
    if not ops:
        return set(), set()
    nd = 1
    if op.startswith(("ds_read2", "ds_read2st64")):
        nd = 1
    d = set().union(*[regs(o) for o in ops[:nd]])
    u = set().union(*[regs(o) for o in ops[nd:]]) if len(ops) > nd else set()
    return d, u


def main():
    path, line, reg = sys.argv[1], int(sys.argv[2]), sys.argv[3]
    depth = int(sys.argv[4]) if len(sys.argv) > 4 else 3
    L = open(path).read().splitlines()
    seen = set()

    def walk(idx, want, lvl):
        if lvl > depth or not want:
            return
        need = set(want)
        i = idx - 1
        while i >= 0 and need:
            op, ops = parse(L[i])
            d, u = defs_uses(op, ops)
            hit = d & need
            if hit:
                key = (i, tuple(sorted(hit)))
                print("  " * lvl + f"{i + 1}: {L[i].split('//')[0].strip()}")
                need -= hit
                if key not in seen:
                    seen.add(key)
                    srcs = u
                    if op in ("v_fmac_f64_e32",):
                        srcs = u  # accumulator included
                    walk(i, srcs, lvl + 1)
            i -= 1

    walk(line - 1, regs(reg), 0)


main()
