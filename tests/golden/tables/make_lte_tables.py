"""Dump the local-thermodynamic-equilibrium tables the reference's own unit test reads (test/test_lte_mixture.cpp:175-186)
into one fixture, as the one-dimensional temperature tables the reference's device build takes (`flow/lte/table_dim = 1`,
datasets "T_energy_R_c" and "T_mu_kappa_sigma" of src/M2ulPhyS.cpp:176-255 -- the HDF5 files of its production inputs,
lte-data/argon_thermo_1atm.h5 and argon_transport_1atm.h5, are git-LFS pointers in this checkout):
    test/inputs/argon_lte_thermo_table.dat       101 temperatures x 11 densities, columns T, rho, ., e, ., ., R, ., c, ., .
                                                 (the columns LteMixture reads: 0, 1, 3, 6, 8; src/lte_mixture.cpp:48-60)
    test/inputs/air_simple_transport_table.dat   200 temperatures x 150 densities, columns T, rho, mu, kappa, sigma
                                                 (src/lte_transport_properties.cpp:42-50)
A one-dimensional table = the slice of the two-dimensional one at one density (the first, 0.005 kg/m^3, and the second,
0.255 kg/m^3: the two densities of the unit test's spot checks); the transport table does not depend on the density.
Run in the build container (needs /root/reference):
    python tests/golden/tables/make_lte_tables.py      -> lte_tables.npz
The values are the decimal numbers of the files parsed by numpy (float64).
"""
import os

import numpy as np

REF = "/root/reference/test/inputs"


def read2d(path):
    with open(path) as f:
        nx, ny = (int(v) for v in f.readline().split())
        data = np.loadtxt(f)
    assert data.shape[0] == nx * ny, (path, data.shape, nx, ny)
    return nx, ny, data


out = {}
nT, nrho, th = read2d(os.path.join(REF, "argon_lte_thermo_table.dat"))
th = th.reshape(nrho, nT, -1)  # the temperature runs fastest
for j, tag in ((0, "rho0p005"), (1, "rho0p255")):
    s = th[j]
    assert np.all(s[:, 1] == s[0, 1]) and np.all(np.diff(s[:, 0]) > 0) and np.all(np.diff(s[:, 3]) > 0)
    out[f"thermo_{tag}"] = np.stack([s[:, 0], s[:, 3], s[:, 6], s[:, 8]], axis=1)  # T, e, R, c
    out[f"density_{tag}"] = np.array(s[0, 1])
    print(f"thermo_{tag}: rho = {s[0, 1]}, {nT} temperatures {s[0, 0]} .. {s[-1, 0]} K, e {s[0, 3]:.4e} .. {s[-1, 3]:.4e}")
nT2, nrho2, tr = read2d(os.path.join(REF, "air_simple_transport_table.dat"))
tr = tr.reshape(nrho2, nT2, -1)
assert np.all(tr[0, :, 2:] == tr[-1, :, 2:])  # no density dependence
s = tr[0]
out["transport"] = np.stack([s[:, 0], s[:, 2], s[:, 3], s[:, 4]], axis=1)  # T, mu, kappa, sigma
print(f"transport: {nT2} temperatures {s[0, 0]} .. {s[-1, 0]} K, mu {s[0, 2]:.4e} .. {s[-1, 2]:.4e}")
np.savez_compressed(os.path.join(os.path.dirname(os.path.abspath(__file__)), "lte_tables.npz"), **out)
