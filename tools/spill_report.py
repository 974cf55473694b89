"""Scratch / spill / register report of every kernel in the built objects (tps_amd/csrc/_obj/*.o), read from the
code-object metadata -- no recompilation.
    python tools/spill_report.py [substring filter] [--all] [--write-allowlist]
Prints the kernels that use scratch (with --all: every matching kernel); the last line counts them.
--write-allowlist rewrites tests/golden/spill_allowlist.txt (see tests/test_spill_allowlist.py)."""
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
import spill_lib  # noqa: E402

show_all = "--all" in sys.argv
write = "--write-allowlist" in sys.argv
args = [a for a in sys.argv[1:] if not a.startswith("--")]
flt = args[0].replace(" ", "") if args else ""
cen = [k for k in spill_lib.census() if flt in k["kernel"]]
spilled = [k for k in cen if k["scratch"] or k["vgpr_spill"]]
for k in (cen if show_all else spilled):
    print(f"{k['unit']:24s} {k['kernel']:90s} vgpr {k['vgpr']:3d} agpr {k['agpr']:3d} scratch {k['scratch']:4d} "
          f"vgpr_spill {k['vgpr_spill']:3d} sgpr_spill {k['sgpr_spill']:3d}" + (f" lds {k['lds']:6d}" if show_all else ""))
print(f"{len(spilled)} of {len(cen)} kernels use scratch")
if write:
    path = os.path.join(spill_lib.ROOT, "tests", "golden", "spill_allowlist.txt")
    names = sorted({k["kernel"] for k in spilled})
    with open(path, "w") as f:
        f.write("# kernels that spill VGPRs or use scratch (tools/spill_report.py --write-allowlist); see tests/test_spill_allowlist.py\n")
        f.write("\n".join(names) + "\n")
    print(f"wrote {len(names)} names to {path}")
