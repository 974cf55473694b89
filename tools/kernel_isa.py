"""Static instruction mix of one kernel of a built object: python tools/kernel_isa.py <object.o> <demangled substring> [--dump file]
(unbundles the gfx950 code object, disassembles it with llvm-objdump, counts instruction classes of the matching kernels)"""
import collections
import os
import re
import subprocess
import sys
import tempfile

LLVM = "/opt/rocm/lib/llvm/bin"


def code_object(obj, tmp):
    co = os.path.join(tmp, "k.co")
    r = subprocess.run([f"{LLVM}/clang-offload-bundler", "--unbundle", "--type=o", "--targets=hipv4-amdgcn-amd-amdhsa--gfx950",
                        f"--input={obj}", f"--output={co}"], capture_output=True, text=True)
    if r.returncode != 0 or not os.path.exists(co) or os.path.getsize(co) == 0:
        fb = os.path.join(tmp, "k.fatbin")
        subprocess.run([f"{LLVM}/llvm-objcopy", "--dump-section", f".hip_fatbin={fb}", obj, os.path.join(tmp, "discard.o")], check=True)  # (an explicit output: with one file name llvm-objcopy rewrites its INPUT in place -- and bumps the mtime the build compares)
        subprocess.run([f"{LLVM}/clang-offload-bundler", "--unbundle", "--type=o", "--targets=hipv4-amdgcn-amd-amdhsa--gfx950",
                        f"--input={fb}", f"--output={co}"], check=True)
    return co


def kernels(obj, pattern):
    with tempfile.TemporaryDirectory() as tmp:
        co = code_object(obj, tmp)
        dis = subprocess.run([f"{LLVM}/llvm-objdump", "-d", "--no-show-raw-insn", "-C", co], capture_output=True, text=True).stdout
    out, name, body = [], None, []
    for line in dis.splitlines():
        m = re.match(r"^[0-9a-f]+ <(.*)>:$", line)
        if m:
            if name is not None:
                out.append((name, body))
            name, body = m.group(1), []
        elif name is not None and line.strip():
            body.append(line.strip())
    if name is not None:
        out.append((name, body))
    return [(n, b) for n, b in out if pattern in n and not n.endswith(".kd")]


def mix(body):
    c = collections.Counter()
    for l in body:
        c[l.split()[0]] += 1
    return c


def summary(c):
    tot = sum(c.values())
    valu = sum(v for k, v in c.items() if k.startswith("v_"))
    f64 = sum(v for k, v in c.items() if re.match(r"v_(fma|mul|add|fmac|rcp|rsq|max|min|ldexp|frexp|rndne|cvt|cmp|div|trig|sqrt).*f64", k))
    lane = c["v_readlane_b32"] + c["v_writelane_b32"]
    mov = sum(v for k, v in c.items() if k.startswith(("v_mov_b32", "v_mov_b64", "v_accvgpr")))
    return dict(total=tot, valu=valu, f64=f64, readwritelane=lane, v_mov=mov, salu=sum(v for k, v in c.items() if k.startswith("s_")),
                s_load=sum(v for k, v in c.items() if k.startswith("s_load")), ds=sum(v for k, v in c.items() if k.startswith("ds_")),
                vmem=sum(v for k, v in c.items() if k.startswith(("global_", "flat_", "buffer_"))),
                scratch=sum(v for k, v in c.items() if k.startswith("scratch_")), waitcnt=c["s_waitcnt"])


if __name__ == "__main__":
    obj, pat = sys.argv[1], sys.argv[2]
    for name, body in kernels(obj, pat):
        print(name[:150])
        print("  ", summary(mix(body)))
        if "--dump" in sys.argv:
            open(sys.argv[sys.argv.index("--dump") + 1], "w").write("\n".join(body))
        if "--top" in sys.argv:
            for k, v in mix(body).most_common(40):
                print(f"     {k:28s} {v}")
