"""Registers / LDS / spills of the kernels in one object:  python tools/regs.py <object.o> [substring]"""
import os, shutil, sys, tempfile
sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
import spill_lib
tmp = tempfile.mkdtemp()
shutil.copy(sys.argv[1], os.path.join(tmp, "x.o"))
pat = sys.argv[2] if len(sys.argv) > 2 else ""
for k in spill_lib.census(tmp):
    if pat in k["kernel"]:
        print(k["kernel"][:90], {a: k[a] for a in ("vgpr", "agpr", "sgpr", "scratch", "vgpr_spill", "sgpr_spill", "lds")})
shutil.rmtree(tmp)
