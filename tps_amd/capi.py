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
SUB_DENS_VEL, SUB_DENS_VEL_NR, SUB_VEL_CONST_ENT = 2, 6, 7
SUB_DENS_VEL_FACE_X, SUB_DENS_VEL_FACE_Y, SUB_DENS_VEL_FACE_Z = 3, 4, 5
SUB_P, SUB_P_NR, SUB_MF_NR, SUB_MF_NR_PW = 0, 2, 3, 4
INV, SLIP, VISC_ADIAB, VISC_ISOTH, VISC_GNRL = 0, 1, 2, 3, 4
ADIAB, ISOTH, SHTH, NONE_THMCND = 0, 1, 2, 3  # ThermalCondition of viscous_general walls

STATUS = {0: "OK", 1: "INVALID_ARGUMENT", 2: "UNSUPPORTED", 3: "MESH", 4: "DEVICE", 5: "NO_DEVICE", 6: "HALO"}

_dp = C.POINTER(C.c_double)
_ip = C.POINTER(C.c_int)


class Mesh(C.Structure):
    _fields_ = [
        ("dim", C.c_int), ("num_vertices", C.c_int), ("num_elements", C.c_int),
        ("elem_vertices", _ip), ("elem_coords", _dp),
        ("num_bdr_faces", C.c_int), ("bdr_vertices", _ip), ("bdr_attributes", _ip),
        ("num_shared_faces", C.c_int), ("shared_vertices", _ip), ("shared_neighbor_rank", _ip),
        ("elem_size", _dp),
    ]


class Disc(C.Structure):
    _fields_ = [("order", C.c_int), ("basis_type", C.c_int), ("int_rule_type", C.c_int),
                ("axisymmetric", C.c_int), ("use_bc_in_grad", C.c_int), ("use_roe", C.c_int),
                ("ref_length", C.c_double)]


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


class MixingLength(C.Structure):  # mixingLengthTransportData, [flow/mixing-length]
    _fields_ = [("max_mixing_length", C.c_double), ("pr_ratio", C.c_double), ("lewis", C.c_double),
                ("bulk_multiplier", C.c_double)]


class Sgs(C.Structure):  # [flow] sgsModel / sgsModelConstant / sgsFloor
    _fields_ = [("model_type", C.c_int), ("model_const", C.c_double), ("model_floor", C.c_double)]


class ViscSponge(C.Structure):  # [viscosityMultiplierFunction]
    _fields_ = [("enabled", C.c_int), ("normal", C.c_double * 3), ("point", C.c_double * 3), ("width", C.c_double),
                ("ratio", C.c_double)]


SGS_NONE, SGS_SMAGORINSKY, SGS_SIGMA = 0, 1, 2
SPONGE_USERDEF, SPONGE_MIXEDOUT = 0, 1


class Lte(C.Structure):  # LteMixtureInput + the TableInputs of src/M2ulPhyS.cpp:176-255 (flow/lte/table_dim = 1)
    _fields_ = [("energy_table", Table), ("gas_constant_table", Table), ("sound_speed_table", Table),
                ("viscosity_table", Table), ("conductivity_table", Table), ("electric_conductivity_table", Table)]


class Physics(C.Structure):
    _fields_ = [("eq_system", C.c_int), ("working_fluid", C.c_int), ("dry_air", DryAir), ("mixture", PerfectMixture),
                ("transport_model", C.c_int), ("constant_transport", ConstantTransport),
                ("gas_transport", GasTransport), ("chemistry", Chemistry), ("radiation", Radiation),
                ("sgs", Sgs), ("visc_sponge", ViscSponge), ("lte", Lte)]


class BC(C.Structure):
    _fields_ = [("attribute", C.c_int), ("category", C.c_int), ("type", C.c_int),
                ("data", C.c_double * (4 + MAXSPECIES))]


HALO_FN = C.CFUNCTYPE(C.c_int, C.c_void_p, C.c_int, C.c_void_p, C.c_void_p, C.c_int, _ip,
                      C.POINTER(C.c_int64), C.POINTER(C.c_int64), C.c_void_p)


REDUCE_FN = C.CFUNCTYPE(C.c_int, C.c_void_p, C.c_void_p, C.c_int, C.c_int, C.c_void_p)
REDUCE_SUM, REDUCE_MIN = 0, 1


class Runtime(C.Structure):
    _fields_ = [("device", C.c_int), ("stream", C.c_void_p), ("halo", HALO_FN), ("halo_ctx", C.c_void_p),
                ("reduce", REDUCE_FN), ("reduce_ctx", C.c_void_p)]


MAXHEATSOURCES, MAXSPONGEZONES, MAXPASSIVESCALARS = 4, 2, 4
SPONGE_PLANAR, SPONGE_ANNULUS = 0, 1


class HeatSource(C.Structure):  # heatSourceData (src/dataStructures.hpp:528-535)
    _fields_ = [("value", C.c_double), ("radius", C.c_double), ("point1", C.c_double * 3), ("point2", C.c_double * 3)]


class SpongeZone(C.Structure):  # SpongeZoneData (src/dataStructures.hpp:260-287)
    _fields_ = [("type", C.c_int), ("solution_type", C.c_int), ("tol", C.c_double),
                ("normal", C.c_double * 3), ("point0", C.c_double * 3),
                ("point_init", C.c_double * 3), ("r1", C.c_double), ("r2", C.c_double), ("mult_factor", C.c_double),
                ("target_U", C.c_double * MAXEQUATIONS)]


class PassiveScalarData(C.Structure):  # passiveScalarData, [passiveScalar*] (src/M2ulPhyS.cpp:2855-2875)
    _fields_ = [("coords", C.c_double * 3), ("radius", C.c_double), ("value", C.c_double)]


class Forcing(C.Structure):
    _fields_ = [("has_pressure_gradient", C.c_int), ("pressure_gradient", C.c_double * 3),
                ("num_heat_sources", C.c_int), ("heat_sources", HeatSource * MAXHEATSOURCES),
                ("num_sponge_zones", C.c_int), ("sponge_zones", SpongeZone * MAXSPONGEZONES),
                ("num_passive_scalars", C.c_int), ("passive_scalars", PassiveScalarData * MAXPASSIVESCALARS)]


def make_forcing(pressure_gradient=None, heat_sources=(), sponge_zones=(), passive_scalars=()) -> Forcing:
    """heat_sources: dicts(value, radius, point1, point2); sponge_zones: dicts(type, normal, point0, point_init,
    target_U[, r1, r2, mult_factor]) -- the [heatSource*] / [spongezone*] input sections
    (src/M2ulPhyS.cpp:2752-2785, 3680-3755) with the target already in conserved variables; passive_scalars:
    dicts(xyz, radius, value), the [passiveScalar*] sections."""
    f = Forcing()
    f.num_passive_scalars = len(passive_scalars)
    for i, ps in enumerate(passive_scalars):
        for d in range(3):
            f.passive_scalars[i].coords[d] = float(ps["xyz"][d]) if d < len(ps["xyz"]) else 0.0
        f.passive_scalars[i].radius = float(ps["radius"])
        f.passive_scalars[i].value = float(ps["value"])
    if pressure_gradient is not None:
        f.has_pressure_gradient = 1
        for d, v in enumerate(pressure_gradient):
            f.pressure_gradient[d] = float(v)
    f.num_heat_sources = len(heat_sources)
    for i, h in enumerate(heat_sources):
        f.heat_sources[i].value = float(h["value"])
        f.heat_sources[i].radius = float(h["radius"])
        for d in range(3):
            f.heat_sources[i].point1[d] = float(h["point1"][d]) if d < len(h["point1"]) else 0.0
            f.heat_sources[i].point2[d] = float(h["point2"][d]) if d < len(h["point2"]) else 0.0
    f.num_sponge_zones = len(sponge_zones)
    for i, z in enumerate(sponge_zones):
        sz = f.sponge_zones[i]
        sz.type = int(z.get("type", SPONGE_PLANAR))
        sz.solution_type = int(z.get("solution_type", SPONGE_USERDEF))
        sz.tol = float(z.get("tol", 0.0))
        for name, key in (("normal", "normal"), ("point0", "point0"), ("point_init", "point_init")):
            for d in range(3):
                getattr(sz, name)[d] = float(z[key][d]) if d < len(z[key]) else 0.0
        sz.r1, sz.r2 = float(z.get("r1", 0.0)), float(z.get("r2", 0.0))
        sz.mult_factor = float(z.get("mult_factor", 1.0))
        for eq, v in enumerate(z["target_U"]):
            sz.target_U[eq] = float(v)
    return f


# ------------------------------------------------------------------------------------------
def dry_air_physics(eq_system=NS, visc_mult=1.0, bulk_visc_mult=0.0, gamma=1.4, gas_constant=287.058) -> Physics:
    """[flow] fluid = dry_air with the reference's defaults (src/M2ulPhyS.cpp:2716-2718,2882-2883)."""
    ph = Physics()
    ph.eq_system = eq_system
    ph.working_fluid = DRY_AIR
    ph.dry_air = DryAir(gamma, gas_constant, visc_mult, bulk_visc_mult, 1.458e-6, 110.4, 0.71)
    return ph


UNIVERSALGASCONSTANT = 8.3144598  # src/equation_of_state.hpp:55

_LTE_TABLES = None


def lte_tables():
    """The reference's own LTE tables (test/test_lte_mixture.cpp:175-186) as one-dimensional temperature tables:
    tests/golden/tables/lte_tables.npz, written by make_lte_tables.py next to it."""
    global _LTE_TABLES
    if _LTE_TABLES is None:
        import os

        path = os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tests", "golden", "tables", "lte_tables.npz")
        with np.load(path) as z:
            _LTE_TABLES = {k: np.array(z[k]) for k in z.files}
    return _LTE_TABLES


def lte_physics(eq_system=NS, density="rho0p005", radiation=False) -> Physics:
    """[flow] fluid = lte_table, lte/table_dim = 1 (test/inputs/plasma.lte1d.ini:30-34): argon thermodynamics at the
    density slice `density` of test/inputs/argon_lte_thermo_table.dat and the transport columns of
    test/inputs/air_simple_transport_table.dat -- the pair of files the reference's unit test combines ("inconsistent ...
    ok here", test/test_lte_mixture.cpp:178-182).  radiation: the reference's net-emission sample table."""
    t = lte_tables()
    th, tr = t["thermo_" + density], t["transport"]
    ph = Physics()
    ph.eq_system = eq_system
    ph.working_fluid = LTE_FLUID
    ph.dry_air = DryAir(1.4, 287.058, 1.0, 0.0, 1.458e-6, 110.4, 0.71)  # unused by the table gas
    keep = []
    ph.lte.energy_table = make_table(th[:, 0], th[:, 1], keep=keep)
    ph.lte.gas_constant_table = make_table(th[:, 0], th[:, 2], keep=keep)
    ph.lte.sound_speed_table = make_table(th[:, 0], th[:, 3], keep=keep)
    ph.lte.viscosity_table = make_table(tr[:, 0], tr[:, 1], keep=keep)
    ph.lte.conductivity_table = make_table(tr[:, 0], tr[:, 2], keep=keep)
    ph.lte.electric_conductivity_table = make_table(tr[:, 0], tr[:, 3], keep=keep)
    if radiation:
        nec = reference_table("nec_sample_0")
        ph.radiation.model = NET_EMISSION
        ph.radiation.nec_table = make_table(nec[:, 0], nec[:, 1], keep=keep)
    ph._keep = keep
    return ph


def make_table(x, f, x_log=False, f_log=False, keep=None) -> Table:
    """TableInput over two float64 arrays; `keep` (a list) receives the arrays so they outlive the call."""
    x = np.ascontiguousarray(x, dtype=np.float64)
    f = np.ascontiguousarray(f, dtype=np.float64)
    assert x.shape == f.shape and x.ndim == 1
    if keep is not None:
        keep += [x, f]
    return Table(len(x), x.ctypes.data_as(_dp), f.ctypes.data_as(_dp), int(x_log), int(f_log))


_REF_TABLES = None


def reference_table(name) -> np.ndarray:
    """A table the reference's own tests hold, (N, 2) = (abscissa, value), bit for bit: the 14 argon
    electron-impact rate coefficients of test/inputs/rate-coefficients/*.h5 ("Ionization", "3BdyRecomb_Ground",
    "StepIonization_Metastable", ...) and "nec_sample_0" = test/inputs/rad-data/nec_sample.0.h5.  Read from the
    fixture tests/golden/tables/reference_tables.npz (generator next to it)."""
    global _REF_TABLES
    if _REF_TABLES is None:
        root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
        _REF_TABLES = dict(np.load(os.path.join(root, "tests", "golden", "tables", "reference_tables.npz")))
    return _REF_TABLES[name]


def reference_nec_table(keep) -> "Table":
    """[plasma_models/radiation_model/net_emission] of test/inputs/input.radDecay.ini:81-87: nec_sample.0.h5,
    x_log = f_log = False, order 1"""
    t = reference_table("nec_sample_0")
    return make_table(t[:, 0], t[:, 1], False, False, keep)


# the tabulated electron-impact reactions of test/inputs/input.radDecay.ini:176-345 (x_log = f_log = False, order 1):
# reactants, products (by species name), table, reaction energy [J/mol]
RADDECAY_REACTIONS = [
    ({"ar": 1, "e": 1}, {"ion": 1, "e": 2}, "Ionization", 1520571.3883),
    ({"ion": 1, "e": 2}, {"ar": 1, "e": 1}, "3BdyRecomb_Ground", -1520571.3883),
    ({"m": 1, "e": 1}, {"ion": 1, "e": 2}, "StepIonization_Metastable", 403710.426440),
    ({"ion": 1, "e": 2}, {"m": 1, "e": 1}, "3BdyRecomb_Metastable", -403710.426440),
    ({"ar": 1, "e": 1}, {"m": 1, "e": 1}, "Excitation_Metastable", 1116860.96186),
    ({"m": 1, "e": 1}, {"ar": 1, "e": 1}, "DeExcitation_Metastable", -1116860.96186),
    ({"r": 1, "e": 1}, {"ion": 1, "e": 2}, "StepIonization_Resonant", 389703.996814),
    ({"ion": 1, "e": 2}, {"r": 1, "e": 1}, "3BdyRecomb_Resonant", -389703.996814),
    ({"ar": 1, "e": 1}, {"r": 1, "e": 1}, "Excitation_Resonant", 1130867.391486),
    ({"r": 1, "e": 1}, {"ar": 1, "e": 1}, "DeExcitation_Resonant", -1130867.391486),
    ({"p": 1, "e": 1}, {"ion": 1, "e": 2}, "StepIonization_4p", 250621.50241031335),
    ({"ion": 1, "e": 2}, {"p": 1, "e": 1}, "3BdyRecomb_4p", -250621.50241031335),
    ({"ar": 1, "e": 1}, {"p": 1, "e": 1}, "Excitation_4p", 1269949.8858896866),
    ({"p": 1, "e": 1}, {"ar": 1, "e": 1}, "DeExcitation_4p", -1269949.8858896866),
]


def argon_ternary_physics(eq_system=NS, two_temperature=False, transport=ARGON_MINIMAL, reactions="arrhenius",
                          third_order_ke=True, radiation=False, ambipolar=True) -> Physics:
    """[plasma_models] of the reference's argon inputs: species (Ar.+1, E, Ar) in mixture order -- active
    species first, electron second to last, neutral background last (src/M2ulPhyS.cpp:3114-3236);
    ambipolar; atoms/species values of test/inputs/argonMinimal.ini:70-97; the two Arrhenius reactions
    (ionisation by electron impact, three-body recombination) of
    test/inputs/lomach.torch.reacting.ini:185-212, stoichiometry permuted into mixture order."""
    ph = Physics()
    keep = []
    ph.eq_system = eq_system
    ph.working_fluid = USER_DEFINED
    m_ar, m_e = 39.948e-3, 5.4858e-7
    nsp = 3
    mx = ph.mixture
    mx.num_species, mx.is_electron_included, mx.ambipolar, mx.two_temperature = nsp, 1, int(ambipolar), int(two_temperature)
    mw = [m_ar - m_e, m_e, m_ar]
    charge = [1.0, -1.0, 0.0]
    eform = [1520.57e3, 0.0, 0.0]
    degen = [4.0, 2.0, 1.0]
    for sp in range(nsp):
        mx.gas_params[sp + SPECIES_MW * nsp] = mw[sp]
        mx.gas_params[sp + SPECIES_CHARGES * nsp] = charge[sp]
        mx.gas_params[sp + FORMATION_ENERGY * nsp] = eform[sp]
        mx.gas_params[sp + SPECIES_DEGENERACY * nsp] = degen[sp]
        mx.molar_cv[sp] = 1.5  # in units of R (perfect_mixture/constant_molar_cv)
    ph.transport_model = transport
    ct = ph.constant_transport  # values in the spirit of test/inputs/*constant transport* inputs
    ct.viscosity, ct.bulk_viscosity = 5.0e-5, 0.0
    ct.thermal_conductivity, ct.electron_thermal_conductivity = 0.05, 0.2
    for sp, (d, f) in enumerate([(3.0e-3, 1.0e9), (2.0e-1, 0.0), (2.5e-3, 4.0e8)]):
        ct.diffusivity[sp], ct.mt_freq[sp] = d, f
    ct.electron_index = 1
    gt = ph.gas_transport
    gt.neutral_index, gt.ion_index, gt.electron_index = 2, 0, 1
    # collision types of the species pairs, [i + j*nsp] for i <= j (src/M2ulPhyS.cpp identifyCollisionType)
    pair = {(0, 0): CLMB_REP, (0, 1): CLMB_ATT, (0, 2): AR_AR1P, (1, 1): CLMB_REP, (1, 2): AR_E, (2, 2): AR_AR}
    for (i, j), c in pair.items():
        gt.collision_index[i + j * nsp] = c
    gt.third_order_k_electron = int(third_order_ke)
    gt.multiply = 0
    for k in range(4):
        gt.flux_trns_multiplier[k] = 1.0
    gt.spcs_trns_multiplier[0] = gt.diff_mult = gt.mobil_mult = 1.0
    ch = ph.chemistry
    ch.electron_index = 1
    ch.minimum_temperature = 2000.0
    if reactions:
        ch.num_reactions = 2
        rxn = [  # reactants, products in mixture order (Ar+, E, Ar); A, b, E; reaction energy
            ((0, 1, 1), (1, 2, 0), (74072.331348, 1.511, 1176329.772504), 1520571.3883),
            ((1, 2, 0), (0, 1, 1), (34126.47475259143, 0.368, -377725.908714), -1520571.3883)]
        for r, (re_, pr, abe, en) in enumerate(rxn):
            ch.reaction_energies[r] = en
            ch.detailed_balance[r] = 0
            ch.reaction_models[r] = ARRHENIUS
            for sp in range(nsp):
                ch.reactant_stoich[sp + r * nsp] = re_[sp]
                ch.product_stoich[sp + r * nsp] = pr[sp]
            for k in range(3):
                ch.rate_params[k + r * MAXCHEMPARAMS] = abe[k]
        if reactions == "tabulated":  # reactions 1 and 2 of test/inputs/input.radDecay.ini:176-198, the reference's tables
            for r, name in enumerate(("Ionization", "3BdyRecomb_Ground")):
                t = reference_table(name)
                ch.reaction_models[r] = TABULATED_RXN
                ch.rate_tables[r] = make_table(t[:, 0], t[:, 1], False, False, keep)
        if reactions == "tabulated_loglog":  # LinearTable's logarithmic axes: the ionisation law sampled log-log
            T = np.geomspace(300.0, 5.0e4, 257)
            A, b, E = rxn[0][2]
            ch.reaction_models[0] = TABULATED_RXN
            ch.rate_tables[0] = make_table(T, A * T ** b * np.exp(-E / UNIVERSALGASCONSTANT / T) + 1e-300, True, True, keep)
        if reactions == "balance":  # recombination replaced by detailed balance of the ionisation
            ch.num_reactions = 1
            ch.detailed_balance[0] = 1
            for k, v in enumerate((2.9e22 / 6.0221409e23, 1.5, 182850.0)):
                ch.equilibrium_constant_params[k] = v
        if reactions == "hoffertlien":
            ch.reaction_models[0] = HOFFERTLIEN
            for k, v in enumerate((1.0e-2, 0.5, 1.85e-18)):
                ch.rate_params[k] = v
    if radiation:  # the reference's net-emission table
        ph.radiation.model = NET_EMISSION
        ph.radiation.nec_table = reference_nec_table(keep)
    ph._keep = keep
    return ph


def argon_levels_physics(levels=3, ambipolar=False, eq_system=NS, transport=CONSTANT, two_temperature=True, reactions=True,
                         radiation=False, third_order_ke=True) -> Physics:
    """Argon with `levels` (0..5) excited neutral levels: mixture order Ar.+1, [Ar_m, Ar_r, Ar_p, Ar_h1, Ar_h2][:levels], E, Ar.
    levels = 4, 5 (seven / eight species, the MAXSPECIES = 8 of the reference's device build): two further lumped
    neutral levels between Ar_p and the ion -- SYNTHETIC, the reference ships no compressible-solver input with more
    than six species; they exercise the species-count limit, not argon kinetics.
    levels = 3, not ambipolar: the mixture of the reference's torch input test/inputs/plasma.ini:200-275;
    levels = 2, not ambipolar: the five species of test/inputs/input.malamas.test.ini; the ambipolar variants drop
    the electron equation (four species: the count of test/inputs/perfectGas.argon.ini).  Transport: constant
    coefficients or the argon collision table (every neutral-neutral pair AR_AR, neutral-ion AR_AR1P,
    neutral-electron AR_E)."""
    ph = Physics()
    ph.eq_system = eq_system
    ph.working_fluid = USER_DEFINED
    m_ar, m_e = 39.948e-3, 5.48579908782496e-7
    nsp = 3 + levels
    ie, ib = nsp - 2, nsp - 1
    mx = ph.mixture
    mx.num_species, mx.is_electron_included, mx.ambipolar, mx.two_temperature = nsp, 1, int(ambipolar), int(two_temperature)
    lv = list(range(1, 1 + levels))  # mixture indices of the excited levels
    mw = [m_ar - m_e] + [m_ar] * levels + [m_e, m_ar]
    charge = [1.0] + [0.0] * levels + [-1.0, 0.0]
    eform = [1520571.3883] + [1116860.96186, 1130867.391486, 1269949.8858896866, 1361000.0, 1420000.0][:levels] + [0.0, 0.0]
    degen = [4.0] + [6.0, 6.0, 36.0, 60.0, 100.0][:levels] + [1.0, 1.0]
    for sp in range(nsp):
        mx.gas_params[sp + SPECIES_MW * nsp] = mw[sp]
        mx.gas_params[sp + SPECIES_CHARGES * nsp] = charge[sp]
        mx.gas_params[sp + FORMATION_ENERGY * nsp] = eform[sp]
        mx.gas_params[sp + SPECIES_DEGENERACY * nsp] = degen[sp]
        mx.molar_cv[sp] = 1.5
    ph.transport_model = transport
    ct = ph.constant_transport
    ct.viscosity, ct.bulk_viscosity = 5.0e-5, 1.0e-5
    ct.thermal_conductivity, ct.electron_thermal_conductivity = 0.05, 0.2
    df = [(3.0e-3, 1.0e9)] + [(2.6e-3, 3.0e8), (2.7e-3, 3.5e8), (2.8e-3, 3.2e8), (2.9e-3, 3.1e8), (3.1e-3, 2.9e8)][:levels] + \
         [(2.0e-1, 0.0), (2.5e-3, 4.0e8)]
    for sp, (d, f) in enumerate(df):
        ct.diffusivity[sp], ct.mt_freq[sp] = d, f
    ct.electron_index = ie
    gt = ph.gas_transport
    gt.neutral_index, gt.ion_index, gt.electron_index = ib, 0, ie
    kinds = ["ion"] + ["n"] * levels + ["e", "n"]
    rule = {("ion", "ion"): CLMB_REP, ("ion", "n"): AR_AR1P, ("ion", "e"): CLMB_ATT, ("n", "n"): AR_AR, ("n", "e"): AR_E,
            ("e", "e"): CLMB_REP, ("n", "ion"): AR_AR1P, ("e", "ion"): CLMB_ATT, ("e", "n"): AR_E}
    for i in range(nsp):
        for j in range(i, nsp):
            gt.collision_index[i + j * nsp] = rule[(kinds[i], kinds[j])]
    gt.third_order_k_electron = int(third_order_ke)
    gt.multiply = 0
    for k in range(4):
        gt.flux_trns_multiplier[k] = 1.0
    gt.spcs_trns_multiplier[0] = gt.diff_mult = gt.mobil_mult = 1.0
    ch = ph.chemistry
    ch.electron_index = ie
    ch.minimum_temperature = 2000.0
    keep = []
    names = ["ion"] + ["m", "r", "p", "h1", "h2"][:levels] + ["e", "ar"]
    if reactions == "tabulated":  # the tabulated reactions of test/inputs/input.radDecay.ini among the present species
        rxn = [r for r in RADDECAY_REACTIONS if all(k in names for k in list(r[0]) + list(r[1]))]
        ch.num_reactions = len(rxn)
        for r, (re_, pr, table, en) in enumerate(rxn):
            ch.reaction_energies[r] = en
            ch.detailed_balance[r] = 0
            ch.reaction_models[r] = TABULATED_RXN
            t = reference_table(table)
            ch.rate_tables[r] = make_table(t[:, 0], t[:, 1], False, False, keep)
            for sp in range(nsp):
                ch.reactant_stoich[sp + r * nsp] = re_.get(names[sp], 0)
                ch.product_stoich[sp + r * nsp] = pr.get(names[sp], 0)
    elif reactions:
        # species by name -> stoichiometry in mixture order; Arrhenius A, b, E; energy; detailed balance
        rxn = [({"e": 1, "ar": 1}, {"ion": 1, "e": 2}, (74072.331348, 1.511, 1176329.772504), 1520571.3883, 0),
               ({"e": 1, "ar": 1}, {"m": 1, "e": 1}, (2.1e4, 1.2, 9.1e5), 1116860.96186, 1),
               ({"m": 1, "e": 1}, {"ion": 1, "e": 2}, (5.6e5, 0.9, 3.3e5), 403710.42644, 0),
               ({"r": 1, "ar": 1}, {"m": 1, "ar": 1}, (3.0e2, 0.5, 2.0e4), -14006.429626, 0),
               ({"p": 1, "e": 1}, {"h1": 1, "e": 1}, (4.0e4, 0.8, 1.1e5), 91050.1141103134, 0),
               ({"h1": 1, "e": 1}, {"h2": 1, "e": 1}, (6.0e4, 0.7, 7.0e4), 59000.0, 0),
               ({"h2": 1, "e": 1}, {"ion": 1, "e": 2}, (9.0e5, 0.6, 1.2e5), 100571.3883, 0)]
        rxn = [r for r in rxn if all(k in names for k in list(r[0]) + list(r[1]))]
        ch.num_reactions = len(rxn)
        for r, (re_, pr, abe, en, db) in enumerate(rxn):
            ch.reaction_energies[r] = en
            ch.detailed_balance[r] = db
            ch.reaction_models[r] = ARRHENIUS
            for sp in range(nsp):
                ch.reactant_stoich[sp + r * nsp] = re_.get(names[sp], 0)
                ch.product_stoich[sp + r * nsp] = pr.get(names[sp], 0)
            for k in range(3):
                ch.rate_params[k + r * MAXCHEMPARAMS] = abe[k]
            if db:
                for k, v in enumerate((6.0, 0.0, 134330.0)):
                    ch.equilibrium_constant_params[k + r * MAXCHEMPARAMS] = v
    if radiation:  # the reference's net-emission table
        ph.radiation.model = NET_EMISSION
        ph.radiation.nec_table = reference_nec_table(keep)
    ph._keep = keep
    return ph


def argon_six_species_physics(eq_system=NS, transport=CONSTANT, two_temperature=True, reactions=True,
                              radiation=False, third_order_ke=True) -> Physics:
    """The mixture of the reference's torch input test/inputs/plasma.ini:200-275: Ar (background), E, Ar.+1 and
    the excited levels Ar_m, Ar_r, Ar_p; NOT ambipolar (the electron density has its own equation)."""
    return argon_levels_physics(3, False, eq_system, transport, two_temperature, reactions, radiation, third_order_ke)


def air_five_species_physics(eq_system=EULER) -> Physics:
    """[atoms], [species] of test/inputs/perfectGas.air.ini:96-134 as test/test_speed_of_sound.cpp:37-41 sets them
    up: CO2, Ar, O2, E, N2 in mixture order (N2 = background last, electron second to last), NOT ambipolar,
    single temperature, no transport / chemistry model (constant transport with zero coefficients)."""
    ph = Physics()
    ph.eq_system = eq_system
    ph.working_fluid = USER_DEFINED
    mC, mO, mAr, mN, mE = 12.011e-3, 15.999e-3, 39.948e-3, 14.0067e-3, 5.4858e-07
    mw = [mC + 2 * mO, mAr, 2 * mO, mE, 2 * mN]
    cv = [3.4230, 1.5, 2.5257, 1.5, 2.5017]
    nsp = 5
    mx = ph.mixture
    mx.num_species, mx.is_electron_included, mx.ambipolar, mx.two_temperature = nsp, 1, 0, 0
    for sp in range(nsp):
        mx.gas_params[sp + SPECIES_MW * nsp] = mw[sp]
        mx.gas_params[sp + SPECIES_CHARGES * nsp] = -1.0 if sp == 3 else 0.0
        mx.gas_params[sp + FORMATION_ENERGY * nsp] = 0.0
        mx.gas_params[sp + SPECIES_DEGENERACY * nsp] = 1.0
        mx.molar_cv[sp] = cv[sp]
    ph.transport_model = CONSTANT
    ph.constant_transport.electron_index = 3
    ph.chemistry.electron_index = 3
    ph.chemistry.num_reactions = 0
    ph._keep = []
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
        if getattr(hm, "elem_size", None) is not None:
            self.es = np.ascontiguousarray(hm.elem_size, dtype=np.float64)
            m.elem_size = self.es.ctypes.data_as(_dp)
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
    lib.tpsrhs_get_plasma_conductivity.argtypes = [vp, vp, vp]
    lib.tpsrhs_height.restype = C.c_int64
    lib.tpsrhs_height.argtypes = [vp]
    lib.tpsrhs_num_dofs.restype = C.c_int64
    lib.tpsrhs_num_dofs.argtypes = [vp]
    lib.tpsrhs_num_equation.argtypes = [vp]
    lib.tpsrhs_enable_kernel_timing.argtypes = [vp, C.c_int]
    lib.tpsrhs_kernel_times.argtypes = [vp, C.c_int, C.POINTER(C.c_char_p), _dp]
    lib.tpsrhs_mult_times.argtypes = [vp, C.c_int, _dp]
    lib.tpsrhs_eval_pointwise.argtypes = [vp, C.c_int, C.c_int64, vp, vp]
    lib.tpsrhs_table_eval.argtypes = [C.POINTER(Table), C.c_int64, vp, vp]
    lib.tpsrhs_math_eval.argtypes = [C.c_int, C.c_int64, vp, vp]
    lib.tpsrhs_kernel_bytes.argtypes = [vp, C.c_int, C.POINTER(C.c_char_p), _dp]
    lib.tpsrhs_rk4_step.argtypes = [vp, C.c_void_p, _dp, C.c_double, _dp, C.POINTER(C.c_int64)]
    lib.tpsrhs_advance.argtypes = [vp, C.c_void_p, _dp, _dp, C.c_int, C.c_int, C.c_double, C.c_double,
                                   C.POINTER(C.c_int64)]
    lib.tpsrhs_set_dt.argtypes = [vp, C.c_double]
    lib.tpsrhs_set_forcing.argtypes = [vp, C.POINTER(Forcing)]
    lib.tpsrhs_set_joule_heating.argtypes = [vp, C.c_void_p]
    lib.tpsrhs_set_mixing_length.argtypes = [vp, C.c_void_p, C.POINTER(MixingLength)]
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
    "tpsrhs_get_primitives", "tpsrhs_get_gradients", "tpsrhs_get_plasma_conductivity", "tpsrhs_height", "tpsrhs_num_dofs", "tpsrhs_num_equation",
    "tpsrhs_enable_kernel_timing", "tpsrhs_kernel_times", "tpsrhs_mult_times", "tpsrhs_kernel_bytes",
    "tpsrhs_eval_pointwise", "tpsrhs_table_eval", "tpsrhs_math_eval", "tpsrhs_face_tables",
    "tpsrhs_rk4_step", "tpsrhs_advance", "tpsrhs_set_dt", "tpsrhs_set_forcing", "tpsrhs_set_joule_heating", "tpsrhs_set_mixing_length", "tpsrhs_status_string",
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
