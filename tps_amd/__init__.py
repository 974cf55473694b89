"""tps_amd -- MI355X-native explicit DG right-hand side of the TPS compressible solver.

Only what the hot path needs: the HIP library behind the C ABI of ``include/tpsrhs.h``
(``tps_amd/csrc``), its ctypes mirror (:mod:`tps_amd.capi`), the Python image of the reference's
``RHSoperator`` interface (:mod:`tps_amd.rhs_operator`) and synthetic-input builders
(:mod:`tps_amd.meshgen`, :mod:`tps_amd.cases`).
"""
__version__ = "0.1.0"
