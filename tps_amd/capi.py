"""ctypes mirror of ``include/tpsrhs.h`` and the loader of the HIP library.

The structures are a field-for-field image of the C header; nothing here computes.  The product
library is ``tps_amd/csrc/libtpsrhs.so`` (built by ``__graft_entry__.build()``); loading fails
loudly when it is missing -- there is no CPU fallback.
"""
from __future__ import annotations

import ctypes as C
import os

import numpy as np

MAXDIM = 3
MAXSPECIES = 8
MAXEQUATIONS = MAXDIM + 2 + MAXSPECIES
MAXREACTIONS = 34
MAXCHEMPARAMS = 3
NUM_GASPARAMS = 4

# enums (values of src/dataStructures.hpp)
EULER, NS, NS_PASSIVE = 0, 1, 2
DRY_AIR, USER_DEFINED, LTE_FLUID = 0, 1, 2
ARGON_MINIMAL, ARGON_MIXTURE, CONSTANT = 0, 1, 2
ARRHENIUS, HOFFERTLIEN, TABULATED_RXN = 0, 1, 2
NONE_RAD, NET_EMISSION = 0, 1
SPECIES_MW, SPECIES_CHARGES, FORMATION_ENERGY, SPECIES_DEGENERACY = 0, 1, 2, 3
CLMB_ATT, CLMB_REP, AR_AR1P, AR_E, AR_AR, NONE_ARGCOLL = 0, 1, 2, 3, 4, 5
INLET, OUTLET, WALL = 0, 1, 2
SUB_DENS_VEL = 2
SUB_P = 0
INV, SLIP, VISC_ADIAB, VISC_ISOTH, VISC_GNRL = 0, 1, 2, 3, 4

STATUS = {0: "OK", 1: "INVALID_ARGUMENT", 2: "UNSUPPORTED", 3: "MESH", 4: "DEVICE", 5: "NO_DEVICE", 6: "HALO"}

_dp = C.POINTER(C.c_double)
_ip = C.POINTER(C.c_int)


class Mesh(C.Structure):
    _fields_ = [
        ("dim", C.c_int), ("num_vertices", C.c_int), ("num_elements", C.c_int),
        ("elem_vertices", _ip), ("elem_coords", _dp),
        ("num_bdr_faces", C.c_int), ("bdr_vertices", _ip), ("bdr_attributes", _ip),
        ("num_shared_faces", C.c_int), ("shared_vertices", _ip), ("shared_neighbor_rank", _ip),
    ]


class Disc(C.Structure):
    _fields_ = [("order", C.c_int), ("basis_type", C.c_int), ("int_rule_type", C.c_int),
                ("axisymmetric", C.c_int), ("use_bc_in_grad", C.c_int)]


class DryAir(C.Structure):
    _fields_ = [("specific_heat_ratio", C.c_double), ("gas_constant", C.c_double), ("visc_mult", C.c_double),
                ("bulk_visc_mult", C.c_double), ("sutherland_C1", C.c_double), ("sutherland_S0", C.c_double),
                ("sutherland_Pr", C.c_double)]


class PerfectMixture(C.Structure):
    _fields_ = [("num_species", C.c_int), ("is_electron_included", C.c_int), ("ambipolar", C.c_int),
                ("two_temperature", C.c_int), ("gas_params", C.c_double * (MAXSPECIES * NUM_GASPARAMS)),
                ("molar_cv", C.c_double * MAXSPECIES)]


class ConstantTransport(C.Structure):
    _fields_ = [("viscosity", C.c_double), ("bulk_viscosity", C.c_double), ("diffusivity", C.c_double * MAXSPECIES),
                ("thermal_conductivity", C.c_double), ("electron_thermal_conductivity", C.c_double),
                ("mt_freq", C.c_double * MAXSPECIES), ("electron_index", C.c_int)]


class GasTransport(C.Structure):
    _fields_ = [("neutral_index", C.c_int), ("ion_index", C.c_int), ("electron_index", C.c_int),
                ("third_order_k_electron", C.c_int), ("collision_index", C.c_int * (MAXSPECIES * MAXSPECIES)),
                ("multiply", C.c_int), ("flux_trns_multiplier", C.c_double * 4),
                ("spcs_trns_multiplier", C.c_double * 1), ("diff_mult", C.c_double), ("mobil_mult", C.c_double)]


class Table(C.Structure):
    _fields_ = [("n_data", C.c_int), ("x_data", _dp), ("f_data", _dp), ("x_log_scale", C.c_int),
                ("f_log_scale", C.c_int)]


class Chemistry(C.Structure):
    _fields_ = [("num_reactions", C.c_int), ("electron_index", C.c_int),
                ("reaction_energies", C.c_double * MAXREACTIONS), ("detailed_balance", C.c_int * MAXREACTIONS),
                ("reactant_stoich", C.c_int16 * (MAXSPECIES * MAXREACTIONS)),
                ("product_stoich", C.c_int16 * (MAXSPECIES * MAXREACTIONS)),
                ("reaction_models", C.c_int * MAXREACTIONS),
                ("equilibrium_constant_params", C.c_double * (MAXCHEMPARAMS * MAXREACTIONS)),
                ("rate_params", C.c_double * (MAXCHEMPARAMS * MAXREACTIONS)),
                ("rate_tables", Table * MAXREACTIONS), ("minimum_temperature", C.c_double)]


class Radiation(C.Structure):
    _fields_ = [("model", C.c_int), ("nec_table", Table)]


class Physics(C.Structure):
    _fields_ = [("eq_system", C.c_int), ("working_fluid", C.c_int), ("dry_air", DryAir), ("mixture", PerfectMixture),
                ("transport_model", C.c_int), ("constant_transport", ConstantTransport),
                ("gas_transport", GasTransport), ("chemistry", Chemistry), ("radiation", Radiation)]


class BC(C.Structure):
    _fields_ = [("attribute", C.c_int), ("category", C.c_int), ("type", C.c_int),
                ("data", C.c_double * (4 + MAXSPECIES))]


HALO_FN = C.CFUNCTYPE(C.c_int, C.c_void_p, C.c_int, C.c_void_p, C.c_void_p, C.c_int, _ip,
                      C.POINTER(C.c_int64), C.POINTER(C.c_int64), C.c_void_p)


class Runtime(C.Structure):
    _fields_ = [("device", C.c_int), ("stream", C.c_void_p), ("halo", HALO_FN), ("halo_ctx", C.c_void_p)]


# ------------------------------------------------------------------------------------------
def dry_air_physics(eq_system=NS, visc_mult=1.0, bulk_visc_mult=0.0, gamma=1.4, gas_constant=287.058) -> Physics:
    """[flow] fluid = dry_air with the reference's defaults (src/M2ulPhyS.cpp:2716-2718,2882-2883)."""
    ph = Physics()
    ph.eq_system = eq_system
    ph.working_fluid = DRY_AIR
    ph.dry_air = DryAir(gamma, gas_constant, visc_mult, bulk_visc_mult, 1.458e-6, 110.4, 0.71)
    return ph


def make_bc(attribute, category, bc_type, data=()) -> BC:
    bc = BC()
    bc.attribute, bc.category, bc.type = attribute, category, bc_type
    for i, v in enumerate(data):
        bc.data[i] = float(v)
    return bc


class MeshArgs:
    """Keeps the numpy arrays alive behind a ``tpsrhs_mesh``."""

    def __init__(self, hm):
        self.ev = np.ascontiguousarray(hm.elem_vertices, dtype=np.int32)
        self.ex = np.ascontiguousarray(hm.elem_coords, dtype=np.float64)
        self.bv = np.ascontiguousarray(hm.bdr_vertices, dtype=np.int32)
        self.ba = np.ascontiguousarray(hm.bdr_attributes, dtype=np.int32)
        m = Mesh()
        m.dim = hm.dim
        m.num_vertices = hm.num_vertices
        m.num_elements = hm.num_elements
        m.elem_vertices = self.ev.ctypes.data_as(_ip)
        m.elem_coords = self.ex.ctypes.data_as(_dp)
        m.num_bdr_faces = int(self.bv.shape[0])
        m.bdr_vertices = self.bv.ctypes.data_as(_ip)
        m.bdr_attributes = self.ba.ctypes.data_as(_ip)
        if hm.shared_vertices is not None and len(hm.shared_vertices):
            self.sv = np.ascontiguousarray(hm.shared_vertices, dtype=np.int32)
            self.sr = np.ascontiguousarray(hm.shared_neighbor_rank, dtype=np.int32)
            m.num_shared_faces = int(self.sv.shape[0])
            m.shared_vertices = self.sv.ctypes.data_as(_ip)
            m.shared_neighbor_rank = self.sr.ctypes.data_as(_ip)
        else:
            m.num_shared_faces = 0
        self.c = m


# ------------------------------------------------------------------------------------------
_LIB = None
LIB_PATH = os.environ.get("TPSRHS_LIB") or os.path.join(os.path.dirname(os.path.abspath(__file__)), "csrc", "libtpsrhs.so")


class LibraryMissing(RuntimeError):
    pass


def load():
    """Load ``libtpsrhs.so`` (HIP, gfx950).  Raises :class:`LibraryMissing` when it was not built."""
    global _LIB
    if _LIB is not None:
        return _LIB
    if not os.path.exists(LIB_PATH):
        raise LibraryMissing(
            f"{LIB_PATH} not found: build it with `python -c 'import __graft_entry__ as g; g.build()'`. "
            "tps_amd has no CPU fallback.")
    # torch ships its own libamdhip64; load it first so that this process holds ONE HIP runtime and
    # the library's stream/pointer arguments mean the same thing on both sides
    import torch  # noqa: F401

    lib = C.CDLL(LIB_PATH)
    vp = C.c_void_p
    lib.tpsrhs_create.restype = C.c_int
    lib.tpsrhs_create.argtypes = [C.POINTER(Mesh), C.POINTER(Disc), C.POINTER(Physics), C.c_int, C.POINTER(BC),
                                  C.POINTER(Runtime), C.POINTER(vp)]
    lib.tpsrhs_destroy.argtypes = [vp]
    lib.tpsrhs_mult.restype = C.c_int
    lib.tpsrhs_mult.argtypes = [vp, vp, vp, C.c_double, _dp]
    lib.tpsrhs_mult_host.restype = C.c_int
    lib.tpsrhs_mult_host.argtypes = [vp, vp, vp, C.c_double, _dp]
    lib.tpsrhs_update_gradients.argtypes = [vp, vp]
    lib.tpsrhs_get_primitives.argtypes = [vp, vp]
    lib.tpsrhs_get_gradients.argtypes = [vp, vp]
    lib.tpsrhs_height.restype = C.c_int64
    lib.tpsrhs_height.argtypes = [vp]
    lib.tpsrhs_num_dofs.restype = C.c_int64
    lib.tpsrhs_num_dofs.argtypes = [vp]
    lib.tpsrhs_num_equation.argtypes = [vp]
    lib.tpsrhs_enable_kernel_timing.argtypes = [vp, C.c_int]
    lib.tpsrhs_kernel_times.argtypes = [vp, C.c_int, C.POINTER(C.c_char_p), _dp]
    lib.tpsrhs_kernel_bytes.argtypes = [vp, C.c_int, C.POINTER(C.c_char_p), _dp]
    lib.tpsrhs_face_tables.restype = C.c_int
    lib.tpsrhs_face_tables.argtypes = [C.POINTER(Mesh), C.c_int, C.POINTER(BC), vp, vp, vp, vp]
    lib.tpsrhs_status_string.restype = C.c_char_p
    lib.tpsrhs_status_string.argtypes = [C.c_int]
    lib.tpsrhs_last_error.restype = C.c_char_p
    lib.tpsrhs_version.restype = C.c_char_p
    _LIB = lib
    return lib


EXPORTED_SYMBOLS = [
    "tpsrhs_create", "tpsrhs_destroy", "tpsrhs_mult", "tpsrhs_mult_host", "tpsrhs_update_gradients",
    "tpsrhs_get_primitives", "tpsrhs_get_gradients", "tpsrhs_height", "tpsrhs_num_dofs", "tpsrhs_num_equation",
    "tpsrhs_enable_kernel_timing", "tpsrhs_kernel_times", "tpsrhs_kernel_bytes", "tpsrhs_face_tables",
    "tpsrhs_status_string",
    "tpsrhs_last_error", "tpsrhs_version",
]


def face_tables(host_mesh, bcs=()):
    """Host-only call of ``tpsrhs_face_tables``: returns (face_nbr, face_orient, shared_slot, shared_orient)."""
    lib = load()
    ma = MeshArgs(host_mesh)
    nlf = 2 * host_mesh.dim
    ne = host_mesh.num_elements
    ns = ma.c.num_shared_faces
    fn = np.zeros(ne * nlf, dtype=np.int32)
    fo = np.zeros(ne * nlf, dtype=np.uint8)
    ss = np.zeros(max(ns, 1), dtype=np.int32)
    so = np.zeros(max(ns, 1), dtype=np.uint8)
    arr = (BC * max(1, len(bcs)))(*bcs)
    st = lib.tpsrhs_face_tables(C.byref(ma.c), len(bcs), arr, fn.ctypes.data, fo.ctypes.data, ss.ctypes.data,
                                so.ctypes.data)
    if st != 0:
        raise RuntimeError(f"tpsrhs_face_tables: {lib.tpsrhs_status_string(st).decode()}: "
                           f"{lib.tpsrhs_last_error().decode()}")
    return fn.reshape(ne, nlf), fo.reshape(ne, nlf), ss[:ns], so[:ns]
