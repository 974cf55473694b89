"""The search behind tests/test_mms_ternary_periodic.py (test/mms.ternary_2d.test: `ternary_2d_2t_periodic_ambipolar`, a
manufactured solution of the TPS team's MASA fork, whose form is not in the reference).  Runs members of the family
f = f0 + dfx gx(2 pi kfx (x / Lx -+ offset_fx)) + dfy gy(2 pi kfy (y / Ly -+ offset_fy)) through the oracle and prints the six
relative errors next to the reference's (the helpers are tests/mms_util.py::ternary_*).

    python tools/mms_ternary_periodic.py [form ...]     form = "cos-" | "sin+" | ... with per-field overrides "cos-|u=sc-|v=cs-"
                                                        (two letters: g of the x term, g of the y term; then the offset sign)
"""
import os
import sys

sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", "tests"))
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))

from mms_util import TERNARY_REF, ternary_run  # noqa: E402

if __name__ == "__main__":
    print("reference".ljust(34), " ".join("%.5e" % v for v in TERNARY_REF))
    for form in sys.argv[1:] or ["sin-", "cos-", "sin+", "cos+", "cos-|u=sc-", "cos-|v=cs-", "cos-|u=sc-|v=cs-"]:
        e = ternary_run(form, n_fine=int(os.environ.get("N_FINE", "40")), source=os.environ.get("SOURCE", "fine"))
        print(form.ljust(34), " ".join("%.5e" % v for v in e), flush=True)
