"""Dump the tabulated data the reference's own tests hold for this path into one fixture:
    test/inputs/rate-coefficients/*.h5   14 electron-impact rate coefficients of argon, /table {500, 2}
                                         (T [K], k_f); used by test/inputs/input.radDecay.ini:176-300
    test/inputs/rad-data/nec_sample.0.h5 net emission coefficient, /table {60, 2}; input.radDecay.ini:83-87
Run in the build container (needs /root/reference and the HDF5 command line tools):
    python tests/golden/tables/make_tables.py
The values are copied bit for bit (h5dump -b LE writes the raw little-endian doubles).  -> reference_tables.npz
"""
import glob
import os
import subprocess
import tempfile

import numpy as np

REF = "/root/reference/test/inputs"
H5DUMP = "/opt/conda/bin/h5dump"
out = {}
files = sorted(glob.glob(os.path.join(REF, "rate-coefficients", "*.h5"))) + [os.path.join(REF, "rad-data", "nec_sample.0.h5")]
for f in files:
    with tempfile.NamedTemporaryFile(suffix=".bin") as tmp:
        subprocess.run([H5DUMP, "-d", "/table", "-b", "LE", "-o", tmp.name, f], check=True, stdout=subprocess.DEVNULL)
        raw = np.fromfile(tmp.name, dtype="<f8")
    hdr = subprocess.run([H5DUMP, "-H", f], check=True, capture_output=True, text=True).stdout
    shape = tuple(int(v) for v in hdr.split("SIMPLE { (")[1].split(")")[0].split(","))
    assert shape[1] == 2 and raw.size == shape[0] * 2, (f, shape, raw.size)
    name = os.path.splitext(os.path.basename(f))[0].replace(".", "_")
    out[name] = raw.reshape(shape)
    t = out[name]
    assert np.all(np.diff(t[:, 0]) > 0), f
    print(f"{name:28s} {shape}  x in [{t[0, 0]:.6g}, {t[-1, 0]:.6g}]  f in [{t[:, 1].min():.3e}, {t[:, 1].max():.3e}]")
np.savez_compressed(os.path.join(os.path.dirname(os.path.abspath(__file__)), "reference_tables.npz"), **out)
