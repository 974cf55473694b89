/*
 * tpsrhs.h -- C ABI of the MI355X-native replacement for the explicit DG right-hand side of the
 * TPS compressible solver (M2ulPhyS): RHSoperator::Mult.
 *
 * Every entry point names the reference interface it replaces (paths relative to the pecos/tps
 * source tree).  Only plain C types cross this boundary: pointers, sizes, PODs.  No torch, no
 * MFEM, no C++ types.
 *
 * Conventions shared with the reference (part of the contract):
 *   - state vectors are "byNODES": U[n + eq * NDofs]            (src/M2ulPhyS.cpp:575-579)
 *   - gradients:  gradUp[n + eq*NDofs + d*num_equation*NDofs]   (src/rhs_operator.cpp:521)
 *   - DG L2 space: dof n = element * dofs_per_element + local,  local = i + j*(p+1) + k*(p+1)^2
 *     (lexicographic in the element's reference frame; MFEM L2_{Quadrilateral,Hexahedron}Element)
 *   - conserved state  [rho, rho u (nvel), rho E, rho Y_sp (active) ..., (rho e_e)]
 *     primitive state  [rho, u (nvel), T_h, n_sp (active) ..., (T_e)]   (src/equation_of_state.cpp:679-700)
 *   - element vertex order is MFEM's (quad: counter-clockwise; hex: bottom ccw then top ccw).
 */
#ifndef TPSRHS_H_
#define TPSRHS_H_

#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

/* ---- capacity limits: src/dataStructures.hpp:41-65 (gpudata::MAX*) ------------------------- */
#define TPSRHS_MAXDIM 3
#define TPSRHS_MAXSPECIES 8
#define TPSRHS_MAXEQUATIONS (TPSRHS_MAXDIM + 2 + TPSRHS_MAXSPECIES)
#define TPSRHS_MAXREACTIONS 34
#define TPSRHS_MAXCHEMPARAMS 3
#define TPSRHS_MAXTABLE 1000
#define TPSRHS_MAXORDER 5 /* MAXDOFS = 216 = hex p=5 */

/* ---- status codes (the reference uses assert/exit/MPI_Abort; src/equation_of_state.cpp:643-650) */
enum tpsrhs_status {
  TPSRHS_OK = 0,
  TPSRHS_ERR_INVALID_ARGUMENT = 1,
  TPSRHS_ERR_UNSUPPORTED = 2, /* a configuration outside the built hot-path scope */
  TPSRHS_ERR_MESH = 3,        /* inconsistent connectivity / boundary faces */
  TPSRHS_ERR_DEVICE = 4,      /* HIP runtime error */
  TPSRHS_ERR_NO_DEVICE = 5,   /* no gfx950 device visible: the library never falls back to CPU */
  TPSRHS_ERR_HALO = 6         /* halo-exchange callback failed */
};

/* ---- enums with the reference's names and values: src/dataStructures.hpp:67-196 -------------- */
enum tpsrhs_equations { TPSRHS_EULER = 0, TPSRHS_NS = 1, TPSRHS_NS_PASSIVE = 2 };
enum tpsrhs_working_fluid { TPSRHS_DRY_AIR = 0, TPSRHS_USER_DEFINED = 1, TPSRHS_LTE_FLUID = 2 };
enum tpsrhs_transport_model {
  TPSRHS_ARGON_MINIMAL = 0,
  TPSRHS_ARGON_MIXTURE = 1,
  TPSRHS_CONSTANT = 2,
  TPSRHS_LTE_TRANSPORT = 3,
  TPSRHS_MIXING_LENGTH = 4
};
enum tpsrhs_reaction_model {
  TPSRHS_ARRHENIUS = 0,
  TPSRHS_HOFFERTLIEN = 1,
  TPSRHS_TABULATED_RXN = 2,
  TPSRHS_GRIDFUNCTION_RXN = 3,
  TPSRHS_RADIATIVE_DECAY = 4
};
enum tpsrhs_radiation_model { TPSRHS_NONE_RAD = 0, TPSRHS_NET_EMISSION = 1 };
enum tpsrhs_gas_params {
  TPSRHS_SPECIES_MW = 0,
  TPSRHS_SPECIES_CHARGES = 1,
  TPSRHS_FORMATION_ENERGY = 2,
  TPSRHS_SPECIES_DEGENERACY = 3,
  TPSRHS_NUM_GASPARAMS = 4
};
enum tpsrhs_gas_coll {
  TPSRHS_CLMB_ATT = 0,
  TPSRHS_CLMB_REP = 1,
  TPSRHS_AR_AR1P = 2,
  TPSRHS_AR_E = 3,
  TPSRHS_AR_AR = 4,
  TPSRHS_NONE_ARGCOLL = 5
};
enum tpsrhs_bc_category { TPSRHS_INLET = 0, TPSRHS_OUTLET = 1, TPSRHS_WALL = 2 };
enum tpsrhs_inlet_type {
  TPSRHS_UNI_DENS_VEL = 0,
  TPSRHS_INTERPOLATE = 1,
  TPSRHS_SUB_DENS_VEL = 2,
  /* density and velocity RELATIVE TO THE INLET FACE (src/inletBC.cpp:453-464, 758-864; "subsonicFaceBasedX/Y/Z"): data =
   * {rho, U_normal (into the domain), U_tangent, unused, active species ...}; the face frame is the inward unit normal
   * made orthogonal to the global x / y / z axis, the axis itself, and their cross product.  3-D only (the reference's
   * frame has three components). */
  TPSRHS_SUB_DENS_VEL_FACE_X = 3,
  TPSRHS_SUB_DENS_VEL_FACE_Y = 4,
  TPSRHS_SUB_DENS_VEL_FACE_Z = 5,
  TPSRHS_SUB_DENS_VEL_NR = 6,   /* non-reflecting, density and velocity (src/inletBC.cpp:576-727) */
  TPSRHS_SUB_VEL_CONST_ENT = 7  /* non-reflecting, velocity, constant entropy (same routine, L2 = 0) */
};
enum tpsrhs_outlet_type {
  TPSRHS_SUB_P = 0,
  TPSRHS_SUB_P_NR = 2,     /* non-reflecting pressure outlet (src/outletBC.cpp:573-728) */
  TPSRHS_SUB_MF_NR = 3,    /* non-reflecting mass-flow outlet (src/outletBC.cpp:739-892) */
  TPSRHS_SUB_MF_NR_PW = 4  /* point-wise variant (src/outletBC.cpp:894-1027) */
};
enum tpsrhs_wall_type {
  TPSRHS_INV = 0,
  TPSRHS_SLIP = 1,
  TPSRHS_VISC_ADIAB = 2,
  TPSRHS_VISC_ISOTH = 3,
  TPSRHS_VISC_GNRL = 4
};
enum tpsrhs_thermal_condition { TPSRHS_ADIAB = 0, TPSRHS_ISOTH = 1, TPSRHS_SHTH = 2, TPSRHS_NONE_THMCND = 3 };
enum tpsrhs_basis_type { TPSRHS_BASIS_GAUSS_LEGENDRE = 0, TPSRHS_BASIS_GAUSS_LOBATTO = 1 };

/* ---- mesh: what the reference takes from mfem::ParMesh (src/M2ulPhyS.cpp:296-470) ------------ */
typedef struct tpsrhs_mesh {
  int dim;          /* 2 (quadrilaterals) or 3 (hexahedra) */
  int num_vertices; /* topological vertices (periodic images identified) */
  int num_elements;
  const int *elem_vertices;   /* [num_elements * 2^dim] topological vertex ids, MFEM order */
  const double *elem_coords;  /* [num_elements * 2^dim * dim] coordinates of each element's own
                                 vertices (order-1 discontinuous nodes: periodic meshes keep their
                                 geometry, as with mfem::Mesh::MakePeriodic) */
  int num_bdr_faces;
  const int *bdr_vertices;    /* [num_bdr_faces * 2^(dim-1)] */
  const int *bdr_attributes;  /* [num_bdr_faces], matched to tpsrhs_bc::attribute */
  /* faces shared with other ranks (mfem::ParMesh shared faces; src/rhs_operator.cpp:716-773).
   * Listed in the same order on both ranks of a pair, grouped by neighbour rank. */
  int num_shared_faces;
  const int *shared_vertices;       /* [num_shared_faces * 2^(dim-1)] local vertex ids, ordered by
                                       ascending GLOBAL vertex id (identical physical order on both
                                       sides) */
  const int *shared_neighbor_rank;  /* [num_shared_faces] */
  /* Optional: [num_elements] mfem::Mesh::GetElementSize(e, 1) (the smallest singular value of the Jacobian at the
   * element centre), the grid scale of the sub-grid scale models (src/rhs_operator.cpp:154,
   * src/face_integrator.cpp:253).  NULL: the library computes it from elem_coords. */
  const double *elem_size;
} tpsrhs_mesh;

/* ---- discretisation: [flow] order/basisType/integrationRule (src/M2ulPhyS.cpp:557-579,2670) -- */
typedef struct tpsrhs_disc {
  int order;             /* polynomial order p, 1..TPSRHS_MAXORDER */
  int basis_type;        /* tpsrhs_basis_type (reference basisType) */
  int int_rule_type;     /* 0 Gauss-Legendre, 1 Gauss-Lobatto (reference integrationRule) */
  int axisymmetric;      /* dim must be 2; nvel = 3 */
  int use_bc_in_grad;    /* boundaryConditions/useBCinGrad (src/M2ulPhyS.cpp:3480) */
  int use_roe;           /* flow/useRoe (src/M2ulPhyS.cpp:2676): RiemannSolverTPS::Eval_Roe on interior faces and
                          * inviscid walls (src/riemann_solver.cpp:66-72,117-206); the reference's formula is
                          * 2-D, single-species, not axisymmetric -- anything else: TPSRHS_ERR_UNSUPPORTED */
  double ref_length;     /* flow/refLength (src/M2ulPhyS.cpp:2677), relaxation length of the non-reflecting
                          * boundary conditions; 0 = the reference's default 1.0 */
} tpsrhs_disc;

/* ---- physics parameter blocks: the PODs of src/dataStructures.hpp:537-729 -------------------- */
typedef struct tpsrhs_dry_air { /* DryAirInput + SutherlandData + visc multipliers */
  double specific_heat_ratio;   /* 1.4     (src/M2ulPhyS.cpp:2882) */
  double gas_constant;          /* 287.058 (src/M2ulPhyS.cpp:2883) */
  double visc_mult;             /* flow/viscosityMultiplier */
  double bulk_visc_mult;        /* flow/bulkViscosityMultiplier */
  double sutherland_C1;         /* 1.458e-6 */
  double sutherland_S0;         /* 110.4 */
  double sutherland_Pr;         /* 0.71 */
} tpsrhs_dry_air;

typedef struct tpsrhs_perfect_mixture { /* PerfectMixtureInput */
  int num_species;
  int is_electron_included;
  int ambipolar;
  int two_temperature;
  double gas_params[TPSRHS_MAXSPECIES * TPSRHS_NUM_GASPARAMS]; /* [sp + param * num_species] */
  double molar_cv[TPSRHS_MAXSPECIES];                          /* J/(mol K) */
} tpsrhs_perfect_mixture;

typedef struct tpsrhs_constant_transport { /* constantTransportData */
  double viscosity;
  double bulk_viscosity;
  double diffusivity[TPSRHS_MAXSPECIES];
  double thermal_conductivity;
  double electron_thermal_conductivity;
  double mt_freq[TPSRHS_MAXSPECIES];
  int electron_index;
} tpsrhs_constant_transport;

typedef struct tpsrhs_gas_transport { /* GasTransportInput */
  int neutral_index;
  int ion_index;
  int electron_index;
  int third_order_k_electron;
  int collision_index[TPSRHS_MAXSPECIES * TPSRHS_MAXSPECIES]; /* tpsrhs_gas_coll, [i + j*nsp] */
  int multiply;
  double flux_trns_multiplier[4];
  double spcs_trns_multiplier[1];
  double diff_mult;
  double mobil_mult;
} tpsrhs_gas_transport;

typedef struct tpsrhs_table { /* TableInput (order 1 interpolation only, src/table.cpp:52-110) */
  int n_data;
  const double *x_data;
  const double *f_data;
  int x_log_scale;
  int f_log_scale;
} tpsrhs_table;

typedef struct tpsrhs_chemistry { /* ChemistryInput */
  int num_reactions;
  int electron_index;
  double reaction_energies[TPSRHS_MAXREACTIONS];
  int detailed_balance[TPSRHS_MAXREACTIONS];
  int16_t reactant_stoich[TPSRHS_MAXSPECIES * TPSRHS_MAXREACTIONS]; /* [sp + r * num_species] */
  int16_t product_stoich[TPSRHS_MAXSPECIES * TPSRHS_MAXREACTIONS];
  int reaction_models[TPSRHS_MAXREACTIONS];                          /* tpsrhs_reaction_model */
  double equilibrium_constant_params[TPSRHS_MAXCHEMPARAMS * TPSRHS_MAXREACTIONS];
  double rate_params[TPSRHS_MAXCHEMPARAMS * TPSRHS_MAXREACTIONS];    /* A, b, E of the model */
  tpsrhs_table rate_tables[TPSRHS_MAXREACTIONS];                     /* TABULATED_RXN only */
  double minimum_temperature;
} tpsrhs_chemistry;

typedef struct tpsrhs_radiation { /* RadiationInput */
  int model; /* tpsrhs_radiation_model */
  tpsrhs_table nec_table;
} tpsrhs_radiation;

/* Fluxes: sub-grid scale model ([flow] sgsModel / sgsModelConstant / sgsFloor, src/M2ulPhyS.cpp:2689-2699,
 * src/fluxes.cpp:513-665) and the planar viscous sponge ([viscosityMultiplierFunction], src/M2ulPhyS.cpp:2788-2808,
 * src/fluxes.cpp:669-688).  The sub-grid scale models: dry air, 3-D (the reference's strain tensor indexes three
 * directions).  The viscous sponge: dry air planar 2-D and 3-D, and every 2-D formulation with the heavy kernel
 * interface -- axisymmetric dry air and table gas, the mixtures planar and axisymmetric (there it also scales the
 * diffusion velocities of the active species, src/fluxes.cpp:240-245).  Gauss-Legendre pair; anything else:
 * TPSRHS_ERR_UNSUPPORTED. */
enum tpsrhs_sgs_model { TPSRHS_SGS_NONE = 0, TPSRHS_SGS_SMAGORINSKY = 1, TPSRHS_SGS_SIGMA = 2 };
typedef struct tpsrhs_sgs {
  int model_type;      /* tpsrhs_sgs_model */
  double model_const;  /* 0: the reference's default, 0.12 (Smagorinsky) / 0.135 (sigma) */
  double model_floor;  /* sgsFloor */
} tpsrhs_sgs;
typedef struct tpsrhs_visc_sponge { /* viscositySpongeData (src/M2ulPhyS.cpp:583-600) */
  int enabled;
  double normal[3], point[3]; /* the normal is normalised, as the constructor of every CPU build of the reference does
                               * (src/fluxes.cpp:77-90; its device constructor :98-125 does not) */
  double width, ratio;
} tpsrhs_visc_sponge;

/* WorkingFluid::LTE_FLUID: the local-thermodynamic-equilibrium table gas, LteMixture + LteTransport
 * (src/lte_mixture.cpp, src/lte_transport_properties.cpp) with ONE-DIMENSIONAL tables in the temperature
 * (`flow/lte/table_dim = 1`, the variant of the reference's device build, src/M2ulPhyS.cpp:164-255: the columns of its
 * "T_energy_R_c" and "T_mu_kappa_sigma" datasets; linear scales).  One species, num_equation = nvel + 2; the inverse
 * table T(e) is the energy table with abscissae and values swapped, as the reference builds it (:193-200), so the
 * energies must increase with the temperature.  The two-dimensional (T, rho) tables of the reference's CPU build
 * interpolate with GSL (third party, `flow/lte/table_dim = 2`) and are not built.  The radiation sink of SourceTerm
 * (src/source_term.cpp:207-209) is tpsrhs_physics::radiation; the electric conductivity table feeds the
 * plasma-conductivity side output of SourceTerm (:196): tpsrhs_get_plasma_conductivity.  Every table must have linear
 * scales (x_log_scale = f_log_scale = 0, as the reference hard-codes): TPSRHS_ERR_INVALID_ARGUMENT otherwise.
 * Built for the axisymmetric formulation (the reference's LTE inputs, test/inputs/plasma.lte1d.ini), Gauss-Legendre
 * pair; anything else: TPSRHS_ERR_UNSUPPORTED. */
typedef struct tpsrhs_lte { /* LteMixtureInput + the TableInputs of src/M2ulPhyS.cpp:176-255 */
  tpsrhs_table energy_table;                /* T -> specific internal energy e [J/kg] */
  tpsrhs_table gas_constant_table;          /* T -> mixture gas constant R [J/(kg K)], p = rho R T */
  tpsrhs_table sound_speed_table;           /* T -> c [m/s] */
  tpsrhs_table viscosity_table;             /* T -> mu [Pa s] */
  tpsrhs_table conductivity_table;          /* T -> kappa [W/(m K)] */
  tpsrhs_table electric_conductivity_table; /* T -> sigma [S/m] */
} tpsrhs_lte;

typedef struct tpsrhs_physics {
  int eq_system;        /* tpsrhs_equations */
  int working_fluid;    /* tpsrhs_working_fluid */
  tpsrhs_dry_air dry_air;
  tpsrhs_perfect_mixture mixture;
  int transport_model;  /* tpsrhs_transport_model (USER_DEFINED fluids) */
  tpsrhs_constant_transport constant_transport;
  tpsrhs_gas_transport gas_transport;
  tpsrhs_chemistry chemistry;
  tpsrhs_radiation radiation;
  tpsrhs_sgs sgs;
  tpsrhs_visc_sponge visc_sponge;
  tpsrhs_lte lte;       /* LTE_FLUID only */
} tpsrhs_physics;

/* ---- boundary conditions: [boundaryConditions/...] (src/M2ulPhyS.cpp:3480-3700) --------------- */
typedef struct tpsrhs_bc {
  int attribute; /* mesh boundary attribute ("patch") */
  int category;  /* tpsrhs_bc_category */
  int type;      /* tpsrhs_inlet_type | tpsrhs_outlet_type | tpsrhs_wall_type */
  /* inlet SUB_DENS_VEL: rho, u, v, w, then active species (src/inletBC.cpp:729-757)
   * outlet SUB_P:       p                                  (src/outletBC.cpp:731-737)
   * non-reflecting types (perfect gas only, as the reference's characteristic algebra):
   *   inlet SUB_DENS_VEL_NR / SUB_VEL_CONST_ENT: rho, u, v, w;  outlet SUB_P_NR: p;  SUB_MF_NR(_PW): mass flow;
   *   for all of them data[4..6] = the patch tangent `tangent1` the reference takes from its first boundary
   *   face (src/outletBC.cpp:160-172; all zero: the library takes an edge of its own first face of the patch --
   *   the boundary term does not depend on the choice for a planar patch) and data[7] = the total patch area
   *   `area_` over all ranks (src/outletBC.cpp:329-343; mass-flow types only).
   * wall INV, SLIP, VISC_ADIAB: no data                    (src/wallBC.cpp:277-469)
   * wall VISC_ISOTH:    T_wall                             (src/wallBC.cpp:96-111)
   * wall VISC_GNRL:     T_h, T_e, heavy thermal condition, electron thermal condition (WallData,
   *                     src/dataStructures.hpp:564-570; tpsrhs_thermal_condition)  (src/wallBC.cpp:112-148) */
  double data[4 + TPSRHS_MAXSPECIES];
} tpsrhs_bc;

/* ---- runtime: device, stream and the halo-exchange hook -------------------------------------- */
/* Called twice per Mult on a partitioned mesh, where the reference posts MPI_Isend/Irecv of
 * neighbour-element data (src/rhs_operator.cpp:775-831).  `send` and `recv` are DEVICE buffers of
 * doubles; segment r of each (offsets[r] .. offsets[r+1]) goes to / comes from neighbor_ranks[r].
 * The callback must enqueue or complete the exchange so that `recv` is valid for work submitted to
 * `stream` after it returns (RCCL send/recv on that stream, or a blocking GPU-aware MPI call).
 * phase 0: U/Up face traces, phase 1: viscous normal-flux traces.  Return 0 on success. */
typedef int (*tpsrhs_halo_fn)(void *ctx, int phase, const double *send, double *recv,
                              int num_neighbors, const int *neighbor_ranks,
                              const int64_t *send_offsets, const int64_t *recv_offsets, void *stream);

/* Reduction over the ranks of the job (`op`: tpsrhs_reduce_op), in place, of `count` doubles in DEVICE memory,
 * ordered on `stream`.  SUM: the MPI_Allreduce of the boundary means of the non-reflecting inlet/outlet conditions
 * (src/outletBC.cpp:533-540, src/inletBC.cpp:548-555; one call per Mult for all such patches together --
 * ranks without faces on a patch contribute zeros, so the job-wide sum equals the reference's
 * per-patch communicator), and of the plane sums of a mixed-out sponge zone (src/forcing_terms.cpp:732-735).
 * MIN: the MPI_Allreduce of the time step in tpsrhs_advance (src/M2ulPhyS.cpp:2013-2016).  Return 0 on success.
 * `stream` may be NULL: that is the legacy default stream, and the implementation must order its work THERE (a
 * torch-based hook maps it to torch.cuda.default_stream -- torch.cuda.ExternalStream(0) is a different stream). */
enum tpsrhs_reduce_op { TPSRHS_REDUCE_SUM = 0, TPSRHS_REDUCE_MIN = 1 };
typedef int (*tpsrhs_reduce_fn)(void *ctx, double *values, int count, int op, void *stream);

typedef struct tpsrhs_runtime {
  int device;            /* HIP device ordinal (reference: rank % numGpusPerRank, src/tps.cpp:196) */
  void *stream;          /* hipStream_t for all work of this operator, NULL = default stream */
  tpsrhs_halo_fn halo;   /* required when mesh.num_shared_faces > 0 */
  void *halo_ctx;
  tpsrhs_reduce_fn reduce; /* required when mesh.num_shared_faces > 0 and a non-reflecting patch exists, a mixed-out
                            * sponge zone is set, or tpsrhs_advance runs with a variable time step */
  void *reduce_ctx;
} tpsrhs_runtime;

typedef struct tpsrhs_operator *tpsrhs_handle;

/* Replaces the RHSoperator constructor and everything it precomputes (Me_inv, Ke, face tables:
 * src/rhs_operator.cpp:39-322, src/gradients.cpp:84-133, src/M2ulPhyS.cpp:816-1486). */
int tpsrhs_create(const tpsrhs_mesh *mesh, const tpsrhs_disc *disc, const tpsrhs_physics *physics,
                  int num_bcs, const tpsrhs_bc *bcs, const tpsrhs_runtime *runtime,
                  tpsrhs_handle *out);

/* RHSoperator::~RHSoperator (src/rhs_operator.cpp:324-341). */
int tpsrhs_destroy(tpsrhs_handle h);

/* RHSoperator::Mult(const Vector &x, Vector &y) const  (src/rhs_operator.hpp:157,
 * src/rhs_operator.cpp:343-464).  x, y: DEVICE pointers, num_equation*NDofs doubles, byNODES.
 * `time` is TimeDependentOperator::GetTime() (src/rhs_operator.cpp:459).  When max_char_speed is
 * non-NULL the rank-local maximum of |u|+c over the nodes (src/rhs_operator.cpp:549-553) is copied
 * to that HOST address (this synchronises the stream; pass NULL to stay asynchronous; the
 * MPI_Allreduce(MAX) of src/rhs_operator.cpp:558 is the caller's). */
int tpsrhs_mult(tpsrhs_handle h, const double *x, double *y, double time, double *max_char_speed);

/* Same with HOST vectors (copies over PCIe; for the MFEM adapter when Vectors live on the host). */
int tpsrhs_mult_host(tpsrhs_handle h, const double *x, double *y, double time,
                     double *max_char_speed);

/* RHSoperator::updatePrimitives + updateGradients (src/rhs_operator.cpp:623-713): refresh the
 * operator-owned Up / gradUp from x (device pointer). */
int tpsrhs_update_gradients(tpsrhs_handle h, const double *x);

/* Up / gradUp grid functions Mult refreshes as a side effect (src/rhs_operator.hpp:74-77):
 * copy the operator-owned device arrays to a DEVICE buffer (neq*NDofs / dim*neq*NDofs doubles). */
int tpsrhs_get_primitives(tpsrhs_handle h, double *up_out);
int tpsrhs_get_gradients(tpsrhs_handle h, double *gradup_out);
/* SourceTerm's side output (src/source_term.cpp:125-199): the `plasma_conductivity_` grid function that the EM solver of
 * the cycle-avg-joule-coupled runs reads between flow steps (the other half of that coupling is
 * tpsrhs_set_joule_heating).  sigma at every node of the state `x` (DEVICE, the layout of tpsrhs_mult; species rows clamped
 * at zero as SourceTerm does): the SrcTrns::ELECTRIC_CONDUCTIVITY of the transport model's source properties --
 * mixtures: computeMixtureElectricConductivity(mobility, n) * MOLARELECTRONCHARGE (src/transport_properties.cpp:421-428,
 * src/gas_transport.cpp:725-739, 1455-1463); table gas: max(sigma(T), 1) (src/lte_transport_properties.cpp:109-126).
 * `sigma_out`: NDofs doubles, DEVICE memory, ordered on the operator's stream.  TPSRHS_ERR_UNSUPPORTED where the
 * reference does not store it: dry air (no SourceTerm) and reacting mixtures that are not ambipolar (:178-197). */
int tpsrhs_get_plasma_conductivity(tpsrhs_handle h, const double *x, double *sigma_out);

/* A->Height() (src/rhs_operator.cpp:49), vfes->GetNDofs(), num_equation. */
/* Point-wise closures of the gas model on the device, for n conserved states U[eq * n + i] (device pointers,
 * synchronous): the public GasMixture methods the reference's unit tests call directly --
 * GetPrimitivesFromConservatives (src/equation_of_state.cpp:321-335,679-700), ComputePressure (:605-628 of the
 * header, :1044-1062), ComputeSpeedOfSound (:337-348,1405-1432; test/test_speed_of_sound.cpp:84-93),
 * ComputeMaxCharSpeed (:278-292,1359-1373).  out: [neq][n] for the primitives, [n] otherwise. */
enum tpsrhs_point_quantity {
  TPSRHS_POINT_PRIMITIVES = 0,
  TPSRHS_POINT_PRESSURE = 1,
  TPSRHS_POINT_SOUND_SPEED = 2,
  TPSRHS_POINT_MAX_CHAR_SPEED = 3
};
int tpsrhs_eval_pointwise(tpsrhs_handle h, int quantity, int64_t n, const double *U, double *out);

/* TableInterpolator::eval of a LinearTable (src/table.cpp:52-110; test/test_table.cpp:104-121) on the device
 * for n abscissae (x, f: device pointers; the table itself is host data as in tpsrhs_chemistry).  Synchronous. */
int tpsrhs_table_eval(const tpsrhs_table *table, int64_t n, const double *x, double *f);

/* Diagnostic: the elementary functions the device closures are built on (tps_amd/csrc/fastmath.hpp; they replace
 * the libm calls of the reference's point physics -- pow / exp / log of src/collision_integrals.cpp:53-201,
 * src/reaction.cpp:41-83 -- and its divisions / sqrt), evaluated on the device for n arguments (device pointers,
 * synchronous), so that their accuracy is a tested number: 0 exp, 1 exp without the range check, 2 log,
 * 3 log of a positive finite argument, 4 reciprocal, 5 sqrt, 6 reciprocal sqrt. */
int tpsrhs_math_eval(int function, int64_t n, const double *x, double *y);

int64_t tpsrhs_height(tpsrhs_handle h);
int64_t tpsrhs_num_dofs(tpsrhs_handle h);
int tpsrhs_num_equation(tpsrhs_handle h);

/* Per-kernel device time, averaged over the tpsrhs_mult calls since timing was enabled (at most the
 * last 128), measured with hipEvents on the operator's stream (enabling adds event records only;
 * reading synchronises the stream).  names[i] are static strings.  Returns the number of kernels
 * written (<= capacity).  With a halo callback the k_gradient / k_flux intervals include it. */
int tpsrhs_enable_kernel_timing(tpsrhs_handle h, int enable);
int tpsrhs_kernel_times(tpsrhs_handle h, int capacity, const char **names, double *milliseconds);
/* Device time of each recorded tpsrhs_mult call, first kernel start to last kernel end (the same event sets):
 * what the median of SURVEY 8(d) is taken over.  Returns the number of calls written (<= capacity, <= 128). */
int tpsrhs_mult_times(tpsrhs_handle h, int capacity, double *milliseconds);

/* Algorithmic HBM bytes one tpsrhs_mult moves per kernel (DESIGN.md "bytes per unit"), matching
 * the order of tpsrhs_kernel_times. */
int tpsrhs_kernel_bytes(tpsrhs_handle h, int capacity, const char **names, double *bytes);

/* ---- next row of the scope table: one explicit RK4 step on the device ----
 * Replaces M2ulPhyS::solveStep's `timeIntegrator->Step(*U, time, dt); Check_NAN(); Check_Undershoot();`
 * (src/M2ulPhyS.cpp:2004-2008) for the RK4 integrator (time-integrator type 4, src/M2ulPhyS.cpp:721-739):
 * the four stages of MFEM's RK4Solver::Step [third party: MFEM >= 4.4, linalg/ode.cpp] with the stage
 * combinations fused into one streaming kernel per stage, the NaN census of Check_NAN
 * (src/M2ulPhyS.cpp:2463-2524) and, for USER_DEFINED fluids, the species clamp of Check_Undershoot
 * (:2526-2548) fused into the last one.  x: device vector, updated in place; *time += dt.
 * max_char_speed (may be NULL): value left by the last stage's Mult, which the reference turns into the
 * next dt (src/M2ulPhyS.cpp:2013-2016).  nan_count (may be NULL): number of NaN entries of the new x. */
int tpsrhs_rk4_step(tpsrhs_handle h, double *x, double *time, double dt, double *max_char_speed, int64_t *nan_count);

/* The time loop on the device: `num_steps` times M2ulPhyS::solveStep (src/M2ulPhyS.cpp:2004-2019) --
 *   timeIntegrator->Step(*U, time, dt); Check_NAN(); Check_Undershoot();
 *   if (!constant_dt) dt = MPI_MIN over ranks of CFL * hmin / max_char_speed / dim;
 * -- with dt, the time and the NaN census kept in device memory, so that nothing returns to the host between
 * steps (one synchronisation at the end).  x: device vector, updated in place; *time and *dt: in/out (dt of
 * the NEXT step on return); hmin: the reference's minimum element size (mesh->GetElementSize(i, 1),
 * src/M2ulPhyS.cpp:757-761 [third party: MFEM], passed in by the adapter); nan_count (may be NULL): NaN entries
 * seen by the censuses of all steps (the reference exits at the first). */
int tpsrhs_advance(tpsrhs_handle h, double *x, double *time, double *dt, int num_steps, int constant_dt, double cfl,
                   double hmin, int64_t *nan_count);

/* The time step the non-reflecting boundary conditions integrate their boundary state with: the reference's
 * BoundaryCondition holds a reference to M2ulPhyS::dt (src/BoundaryCondition.hpp:54) and advances `boundaryU`
 * by dt in EVERY Mult (src/outletBC.cpp:712-724).  tpsrhs_rk4_step sets it itself. */
int tpsrhs_set_dt(tpsrhs_handle h, double dt);

/* ---- next row of the scope table: the other ForcingTerms of RHSoperator ----
 * RHSoperator appends these to its `forcing` array (src/rhs_operator.cpp:101-166) and adds them to y
 * after the inverse mass (src/rhs_operator.cpp:451-461).  SourceTerm and AxisymmetricSource follow from
 * tpsrhs_physics / tpsrhs_disc; the ones below are configured here, after tpsrhs_create. */
#define TPSRHS_MAXHEATSOURCES 4
#define TPSRHS_MAXSPONGEZONES 2
#define TPSRHS_MAXPASSIVESCALARS 4

typedef struct tpsrhs_heat_source { /* heatSourceData, type "cylinder" (src/dataStructures.hpp:528-535) */
  double value;                      /* added to y[(dim+1)*NDofs + node] (src/forcing_terms.cpp:923-936) */
  double radius, point1[3], point2[3];
} tpsrhs_heat_source;

enum tpsrhs_sponge_type { TPSRHS_SPONGE_PLANAR = 0, TPSRHS_SPONGE_ANNULUS = 1 };  /* SpongeZoneType */
enum tpsrhs_sponge_solution { TPSRHS_SPONGE_USERDEF = 0, TPSRHS_SPONGE_MIXEDOUT = 1 }; /* SpongeZoneSolution */

typedef struct tpsrhs_sponge_zone { /* SpongeZoneData (src/dataStructures.hpp:260-287) */
  int type;                          /* tpsrhs_sponge_type */
  int solution_type;                 /* tpsrhs_sponge_solution.  MIXEDOUT (src/forcing_terms.cpp:713-743): at every
                                      * Mult the target is the mixed-out state of the mean convective normal flux over
                                      * the nodes within `tol` of the plane through point_init (planar zone) / of the
                                      * cylinder of radius r1 (annulus), summed over the ranks with runtime.reduce;
                                      * target_U is ignored.  Dry air, planar 2-D / 3-D; else TPSRHS_ERR_UNSUPPORTED */
  double tol;                        /* MIXEDOUT: node search tolerance (spongezone/tolerance) */
  double normal[3], point0[3], point_init[3]; /* normal is normalised by the library (forcing_terms.cpp:528-532) */
  double r1, r2;                     /* annulus radii */
  double mult_factor;
  /* SpongeZone::targetU, the CONSERVED target state the constructor derives from targetUp with the
   * mixture's modifyEnergyForPressure (src/forcing_terms.cpp:486-517); the adapter copies it from there.
   * Its speed of sound (forcing_terms.cpp:653-656) is evaluated by the library. */
  double target_U[TPSRHS_MAXEQUATIONS];
} tpsrhs_sponge_zone;

/* PassiveScalar ([passiveScalars], src/M2ulPhyS.cpp:2855-2875; src/forcing_terms.cpp:768-870): at the nodes closer than
 * `radius` to `coords` (the first `dim` entries), y[(num_equation-1) NDofs + node] -= |u| (Up_last - rho value) / radius
 * with |u| over the `dim` velocity components and Up_last the LAST primitive variable -- whatever the equation system
 * (the reference appends the term whenever such an entry exists: test/inputs/argonMinimal.ini:118-124 relaxes the last
 * species of the ternary plasma with it). */
typedef struct tpsrhs_passive_scalar { /* passiveScalarData (src/dataStructures.hpp:519-526) */
  double coords[3];
  double radius;
  double value;
} tpsrhs_passive_scalar;

typedef struct tpsrhs_forcing {
  int has_pressure_gradient;         /* config.thereIsForcing(): ConstantPressureGradient, forcing_terms.cpp:115-171 */
  double pressure_gradient[3];
  int num_heat_sources;              /* enabled HeatSource entries only */
  tpsrhs_heat_source heat_sources[TPSRHS_MAXHEATSOURCES];
  int num_sponge_zones;              /* SpongeZone */
  tpsrhs_sponge_zone sponge_zones[TPSRHS_MAXSPONGEZONES];
  int num_passive_scalars;           /* PassiveScalar */
  tpsrhs_passive_scalar passive_scalars[TPSRHS_MAXPASSIVESCALARS];
} tpsrhs_forcing;

/* Replaces the forcing.Append(...) calls of the RHSoperator constructor for ConstantPressureGradient, PassiveScalar,
 * SpongeZone and HeatSource.  NULL removes them. */
int tpsrhs_set_forcing(tpsrhs_handle h, const tpsrhs_forcing *forcing);

/* MixingLengthTransport (src/mixing_length_transport.cpp:44-131; [flow] useMixingLength, flow/mixing-length/...,
 * src/M2ulPhyS.cpp:265-277, 2701-2708): an algebraic eddy viscosity rho l^2 |S| with l = min(0.41 d, max_mixing_length)
 * on top of the molecular transport of the flux (viscosity, bulk viscosity, heavy-species conductivity; not the
 * species diffusivities, not the source terms), d = the wall-distance grid function `distance_` of the reference:
 * `distance` is its DEVICE array (NDofs doubles, owned by the caller, read at every Mult).  NULL switches the model off.
 * Built for the 2-D kernels (planar and axisymmetric mixtures, axisymmetric dry air: the formulations of the reference's
 * inputs that use it, test/inputs/plasma.ini:48, pipe.axisym.mix.ini:19); 3-D and planar dry air:
 * TPSRHS_ERR_UNSUPPORTED. */
typedef struct tpsrhs_mixing_length { /* mixingLengthTransportData (src/dataStructures.hpp:548-554) */
  double max_mixing_length; /* 0 turns the model off, as in the reference */
  double pr_ratio;          /* flow/mixing-length/Pr_ratio (`Prt_`), default 1 */
  double lewis;             /* `Let_`: read by the reference, not used by its flux transport */
  double bulk_multiplier;   /* bulk eddy viscosity = bulk_multiplier * mu_t */
} tpsrhs_mixing_length;
int tpsrhs_set_mixing_length(tpsrhs_handle h, const double *distance, const tpsrhs_mixing_length *params);

/* JouleHeating (src/forcing_terms.cpp:443-471): `joule_heating` is the DEVICE array of the
 * `joule_heating_` grid function (NDofs doubles, owned by the caller, read at every Mult; the EM solver
 * refreshes it between steps).  Positive entries are added to the total-energy equation and, for a
 * two-temperature mixture, to the electron-energy equation.  NULL disables the term. */
int tpsrhs_set_joule_heating(tpsrhs_handle h, const double *joule_heating);

/* Host-only view of the face topology tpsrhs_create derives from a mesh (the role of the
 * indirection arrays of src/M2ulPhyS.cpp:816-1486); touches no device.  Outputs (caller-allocated):
 *   face_nbr[ne*2*dim]   >= 0: trace slot (element*2*dim + local face) of the neighbour, slots
 *                        >= ne*2*dim are halo slots in shared-face order; < 0: -(bc index + 1)
 *   face_orient[ne*2*dim] 3-bit code (swap | flip_a<<1 | flip_b<<2) from my face frame to the
 *                        neighbour's (interior) or to the canonical frame (shared)
 *   shared_slot[nshared], shared_orient[nshared]: own slot / frame code of every shared face */
int tpsrhs_face_tables(const tpsrhs_mesh *mesh, int num_bcs, const tpsrhs_bc *bcs, int32_t *face_nbr,
                       uint8_t *face_orient, int32_t *shared_slot, uint8_t *shared_orient);

const char *tpsrhs_status_string(int status);
const char *tpsrhs_last_error(void); /* thread-local text of the last failure */
const char *tpsrhs_version(void);

#ifdef __cplusplus
}
#endif
#endif /* TPSRHS_H_ */
