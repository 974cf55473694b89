"""TPS restart files (HDF5) <-> the operator's state vector, over ``libtpsrhs_io.so`` (``include/tpsrhs_io.h``).

The reference's ``M2ulPhyS::restart_files_hdf5`` (``src/io.cpp:195-262``) with the partitioned read / write of the
``/solution`` family (``src/io.cpp:43-193, 701-776``): attributes ``iteration``, ``time``, ``dt``, ``order``,
``dimension`` and one dataset of NDofs doubles per conserved variable.  Host arrays; the caller moves them to the GPU."""
from __future__ import annotations

import ctypes as C
import os

import numpy as np

LIB_PATH = os.path.join(os.path.dirname(os.path.abspath(__file__)), "csrc", "libtpsrhs_io.so")
EXPORTED_SYMBOLS = ["tpsrhs_restart_variable_names", "tpsrhs_restart_info_read", "tpsrhs_restart_read", "tpsrhs_restart_read_serial", "tpsrhs_restart_read_change_order", "tpsrhs_restart_write",
                    "tpsrhs_io_last_error"]
_LIB = None


class RestartInfo(C.Structure):
    _fields_ = [("iteration", C.c_int), ("time", C.c_double), ("dt", C.c_double), ("order", C.c_int), ("dimension", C.c_int),
                ("dofs_global", C.c_int64), ("ndofs", C.c_int64)]


def load():
    global _LIB
    if _LIB is None:
        if not os.path.exists(LIB_PATH):
            raise RuntimeError(f"{LIB_PATH} missing: run __graft_entry__.build()")
        lib = C.CDLL(LIB_PATH)
        lib.tpsrhs_io_last_error.restype = C.c_char_p
        lib.tpsrhs_restart_variable_names.argtypes = [C.c_int, C.c_int, C.POINTER(C.c_char_p), C.c_int, C.c_int, C.POINTER(C.c_char_p)]
        lib.tpsrhs_restart_info_read.argtypes = [C.c_char_p, C.POINTER(RestartInfo)]
        lib.tpsrhs_restart_read.argtypes = [C.c_char_p, C.c_int, C.c_int64, C.POINTER(C.c_char_p), C.c_int, C.c_void_p,
                                            C.POINTER(RestartInfo)]
        lib.tpsrhs_restart_read_serial.argtypes = [C.c_char_p, C.c_int, C.c_int64, C.c_int, C.c_void_p, C.POINTER(C.c_char_p), C.c_int,
                                                   C.c_void_p, C.POINTER(RestartInfo)]
        lib.tpsrhs_restart_read_change_order.argtypes = [C.c_char_p, C.c_int, C.c_int64, C.c_int, C.c_int, C.c_int, C.POINTER(C.c_char_p),
                                                         C.c_void_p, C.POINTER(RestartInfo)]
        lib.tpsrhs_restart_write.argtypes = [C.c_char_p, C.c_int, C.c_int64, C.POINTER(C.c_char_p), C.c_void_p, C.POINTER(RestartInfo)]
        _LIB = lib
    return _LIB


def variable_names(nvel, species_names=(), two_temperature=False):
    """dataset names of ``/solution`` in state-vector order (``src/M2ulPhyS.cpp:1825-1852``)"""
    lib = load()
    sp = (C.c_char_p * max(1, len(species_names)))(*[s.encode() for s in species_names])
    out = (C.c_char_p * 16)()
    n = lib.tpsrhs_restart_variable_names(int(nvel), len(species_names), sp, 1 if two_temperature else 0, 16, out)
    if n < 0:
        raise RuntimeError(lib.tpsrhs_io_last_error().decode())
    return [out[i].decode() for i in range(n)]


def _names(names):
    return (C.c_char_p * len(names))(*[s.encode() for s in names])


def read_info(path):
    info = RestartInfo()
    if load().tpsrhs_restart_info_read(str(path).encode(), C.byref(info)) != 0:
        raise RuntimeError(load().tpsrhs_io_last_error().decode())
    return info


def read(path, names, ndofs, order=-1):
    """-> (U (len(names), ndofs) float64, RestartInfo); ``order`` >= 0 must match the file's"""
    U = np.zeros((len(names), int(ndofs)))
    info = RestartInfo()
    if load().tpsrhs_restart_read(str(path).encode(), len(names), int(ndofs), _names(names), int(order),
                                  U.ctypes.data_as(C.c_void_p), C.byref(info)) != 0:
        raise RuntimeError(load().tpsrhs_io_last_error().decode())
    return U, info


def write(path, names, U, iteration=0, time=0.0, dt=0.0, order=1, dimension=3, dofs_global=-1):
    U = np.ascontiguousarray(U, dtype=np.float64)
    assert U.ndim == 2 and U.shape[0] == len(names)
    info = RestartInfo(int(iteration), float(time), float(dt), int(order), int(dimension), int(dofs_global), U.shape[1])
    if load().tpsrhs_restart_write(str(path).encode(), len(names), U.shape[1], _names(names), U.ctypes.data_as(C.c_void_p),
                                   C.byref(info)) != 0:
        raise RuntimeError(load().tpsrhs_io_last_error().decode())


def read_serial(path, names, global_elements, dofs_per_element, order=-1):
    """A rank's part of a SERIALISED restart file (one file for the unpartitioned mesh, ``src/io.cpp:104-172, 460-530``):
    ``global_elements[e]`` = id of local element e in the unpartitioned mesh.  -> (U (len(names), ne * dofs_per_element), info)"""
    ge = np.ascontiguousarray(global_elements, dtype=np.int64)
    U = np.zeros((len(names), ge.size * int(dofs_per_element)))
    info = RestartInfo()
    if load().tpsrhs_restart_read_serial(str(path).encode(), len(names), ge.size, int(dofs_per_element), ge.ctypes.data_as(C.c_void_p),
                                         _names(names), int(order), U.ctypes.data_as(C.c_void_p), C.byref(info)) != 0:
        raise RuntimeError(load().tpsrhs_io_last_error().decode())
    return U, info


def read_change_order(path, names, num_elements, dim, order, basis_type=0):
    """``io/restartMode = variableP`` (``src/io.cpp:174-193, 797-850``): a file of ANOTHER polynomial order, interpolated element
    by element to the nodes of ``order``.  -> (U (len(names), num_elements * (order+1)^dim), info with the file's order)"""
    U = np.zeros((len(names), int(num_elements) * (int(order) + 1) ** int(dim)))
    info = RestartInfo()
    if load().tpsrhs_restart_read_change_order(str(path).encode(), len(names), int(num_elements), int(dim), int(order), int(basis_type),
                                               _names(names), U.ctypes.data_as(C.c_void_p), C.byref(info)) != 0:
        raise RuntimeError(load().tpsrhs_io_last_error().decode())
    return U, info
