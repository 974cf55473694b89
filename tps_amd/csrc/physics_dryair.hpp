// Device point-wise physics, dry air (gamma-law gas, Sutherland viscosity).
//
// Flattened restatement of the reference's virtual hierarchy for WorkingFluid::DRY_AIR:
//   DryAir                 src/equation_of_state.cpp:146-412, src/equation_of_state.hpp:605-628
//   DryAirTransport        src/transport_properties.cpp:205-266
//   Fluxes                 src/fluxes.cpp:135-170 (convective), :178-335 (viscous), :344-505 (boundary)
//   RiemannSolverTPS       src/riemann_solver.cpp:53-115 (Lax-Friedrichs)
//   Inlet/Outlet/WallBC    src/inletBC.cpp:729-757, src/outletBC.cpp:731-737, src/wallBC.cpp:277-510
// The pressure / temperature / viscosity of a state are computed once per point and reused
// (the reference recomputes them in every call, src/fluxes.hpp:59-64).
#ifndef TPSRHS_PHYSICS_DRYAIR_HPP_
#define TPSRHS_PHYSICS_DRYAIR_HPP_

#include <hip/hip_runtime.h>

#include "../../include/tpsrhs.h"
#include "fastmath.hpp"

namespace tpsrhs {

#ifndef TPSRHS_MINW_GRAD
#define TPSRHS_MINW_GRAD 1
#endif
#ifndef TPSRHS_MINW_FLUX
#define TPSRHS_MINW_FLUX 3  // <= 168 VGPRs: 3 waves per SIMD (the allocator otherwise lands on 170)
#endif

struct BcDev {
  int category, type;
  double data[4 + TPSRHS_MAXSPECIES];
};
constexpr int MAXBC = 16;

struct DryAirParams {
  double gamma, Rg, inv_Rg, visc_mult, bulk_mult, C1, S0, cp_div_pr;
  int eq_system;  // tpsrhs_equations
  int use_bc_in_grad;
  int num_bcs;
  int use_roe;  // flow/useRoe: Roe flux on interior faces and inviscid walls (2-D only)
  // non-reflecting inlet / outlet types: time step of the boundary-state update, relaxation length, and the
  // boundary states the Riemann solver sees in this Mult ([face ordinal][q][eq]; swapped by the host)
  double nr_dt, ref_length;
  const double *bstate;
  const int *nr_ordinal;  // [face slot] -> ordinal among the faces of non-reflecting patches
  // sub-grid scale model and viscous sponge of Fluxes (src/fluxes.cpp:223-246; the LES flavour of the kernels)
  int sgs_type;  // 0 none, 1 Smagorinsky, 2 sigma
  int vs_enabled;
  double sgs_const, sgs_floor;
  double vs_n[3], vs_p[3], vs_width, vs_ratio;
  const double *elem_delta;  // [ne] Mesh::GetElementSize(e, 1) / order (src/rhs_operator.cpp:154)
  BcDev bc[MAXBC];
};

// MixingLengthTransport (src/mixing_length_transport.cpp:62-131): the wall-distance grid function and the model's
// constants (tpsrhs_set_mixing_length); `distance` NULL = off.  Read by the 2-D kernels.
struct MixLenDev {
  const double *distance;  // [NDofs] device, owned by the caller
  double lmax, prt, bulk;
};
// The viscous sponge of Fluxes ([viscosityMultiplierFunction], src/fluxes.cpp:232-246, 669-688) for the 2-D kernels with
// the heavy interface (axisymmetric dry air / table gas, the mixtures planar and axisymmetric): the plane, by value in
// MeshDev; the 3-D / planar dry-air kernels carry theirs in DryAirParams (the LES flavour).
struct VsDev {
  int enabled;
  double n[2], p[2], width, ratio;
};
__device__ inline double visc_sponge_weight_2d(const VsDev &v, const double *X) {  // viscSpongePlanar
  const double factor = fmax(v.ratio, 1.0);
  const double dist = (X[0] - v.p[0]) * v.n[0] + (X[1] - v.p[1]) * v.n[1];
  double wgt = 0.5 * (tanh(dist / v.width - 2.0) + 1.0);
  wgt *= (factor - 1.0);
  wgt += 1.0;
  return wgt;
}
struct EddyCtx {  // what a closure needs of both at one point: the model's constants, the (interpolated) distance, the sponge weight
  const MixLenDev *ml;  // the mixing-length model's block (wave-uniform: the kernel's argument), NULL = off
  double dist;
  // Viscous sponge: base of the block's LDS array of weights (wave-uniform), one entry per lane, or NULL (none).  The
  // weight is needed AFTER the transport closure; carried in a register across it, it is the 257th and 258th VGPR of the
  // six-species two-temperature k_gradient (one wave per SIMD instead of two: torch6 k_gradient 0.49 -> 0.67 ms,
  // measured), so the kernels park it in the lane's LDS word and the closure reads it where it scales its coefficients.
  const double *vsw;
};
__device__ inline EddyCtx eddy_off() { return EddyCtx{nullptr, 0.0, nullptr}; }
__device__ inline EddyCtx eddy_at(const MixLenDev &ml, double dist) { return EddyCtx{ml.distance != nullptr ? &ml : nullptr, dist, nullptr}; }
__device__ inline double sponge_weight(const EddyCtx &ec) { return ec.vsw ? ec.vsw[threadIdx.x] : 1.0; }
// eddy viscosity rho l^2 |S| added to the molecular viscosity / raw bulk viscosity / heavy conductivity of a point
// (g[eq + d*NEQ]: the primitive gradient; NVEL = 3 with DIM = 2: the axisymmetric strain terms)
template <int DIM, int NVEL, int NEQ>
__device__ inline void add_mixing_length(const EddyCtx &ec, const double *U, const double *g, double radius, double &visc,
                                         double &bulk, double &k) {
  if (!ec.ml) return;
  const double cp_over_pr = k / visc;
  const double rho = U[0];
  double S = 0.0;
#pragma unroll
  for (int i = 0; i < DIM; i++)
#pragma unroll
    for (int j = 0; j < DIM; j++) {
      const double Sij = 0.5 * (g[(1 + i) + j * NEQ] + g[(1 + j) + i * NEQ]);
      S += 2 * Sij * Sij;
    }
  if constexpr (NVEL != DIM) {
    const double ur = U[1] / rho, ut = U[3] / rho;
    double Szx = 0.5 * g[3 + 0 * NEQ];
    if (radius > 0) Szx -= 0.5 * ut / radius;
    const double Szy = 0.5 * g[3 + 1 * NEQ];
    double Szz = 0.0;
    if (radius > 0) Szz += ur / radius;
    S += 2 * (2 * Szx * Szx + 2 * Szy * Szy + Szz * Szz);
  }
  S = sqrt(S);
  double l = 0.41 * ec.dist;
  if (l > ec.ml->lmax) l = ec.ml->lmax;
  const double mut = rho * l * l * S;
  visc += mut;
  bulk += ec.ml->bulk * mut;
  k += mut * cp_over_pr * ec.ml->prt;
}

// Where a point sits, for the closures that depend on it (LES flavour): grid scale of its element and position
struct PointCtx {
  double delta;
  double X[3];
};

// Slip wall in 2-D (computeSlipWallFlux, src/wallBC.cpp:326-428): the reference mirrors the velocity component
// along its first frame vector in the frame (n, t) where its "arbitrary tangent" t is, for dim = 2, NOT orthogonal
// to n (previous_dir coincides with dir there); the ghost momentum g solves  n.g = -n.v,  t.g = t.v  in that
// skewed frame.  In 3-D the frame is orthonormal and the result is the plain mirror state of the inviscid wall.
__device__ inline void slip_ghost_momentum_2d(const double *n, const double *U, double *Ug) {
  const double sml = 1.0e-15;
  const double nn = sqrt(fmax(n[0] * n[0] + n[1] * n[1], sml));
  const double u[2] = {n[0] / nn, n[1] / nn};
  const int dir = (fabs(u[1]) >= fabs(u[0])) ? 1 : 0, nd = 1 - dir;
  double t[2];
  t[nd] = 1.0;
  t[dir] = (u[dir] * -1.0 + u[nd] * 1.0) * (-1.0 / u[dir]);
  const double tm = fmax(sqrt(t[0] * t[0] + t[1] * t[1]), sml);
  t[0] /= tm;
  t[1] /= tm;
  const double v[2] = {U[1] / U[0], U[2] / U[0]};
  const double a = -(u[0] * v[0] + u[1] * v[1]), b = t[0] * v[0] + t[1] * v[1];
  const double det = u[0] * t[1] - u[1] * t[0];
  Ug[1] = U[0] * (t[1] * a - u[1] * b) / det;
  Ug[2] = U[0] * (-t[0] * a + u[0] * b) / det;
}

// Inlet "relative to the face" (SUB_DENS_VEL_FACE_X/Y/Z, src/inletBC.cpp:758-864): momentum of the prescribed state in
// global coordinates.  Face frame: the inward unit normal minus its component along the global axis `axis` (NOT
// renormalised, as in the reference), tangent1 = that x axis (with the reference's signs), tangent2 = the axis; the
// prescribed momentum (rho Un, rho Ut, 0) is given in that frame and comes back through the inverse of M = rows
// (normal, tangent1, tangent2) [MFEM CalcInverse of a 3 x 3: adjugate / determinant].
__device__ inline void face_inlet_momentum(const double *n, int axis, double rho, double Un, double Ut, double *mom) {
  double mod = 0.0;
  for (int d = 0; d < 3; d++) mod += n[d] * n[d];
  const double sc = -1.0 / sqrt(mod);
  double un[3] = {n[0] * sc, n[1] * sc, n[2] * sc};
  double t2[3] = {0.0, 0.0, 0.0};
  t2[axis] = 1.0;
  const double tn = un[axis];  // (t . n) / |t|^2 with |t| = 1
  un[axis] -= tn;
  const double t1[3] = {+(un[1] * t2[2] - un[2] * t2[1]), -(un[0] * t2[2] - un[2] * t2[0]), +(un[0] * t2[1] - un[1] * t2[0])};
  const double M[9] = {un[0], un[1], un[2], t1[0], t1[1], t1[2], t2[0], t2[1], t2[2]};  // row-major
  const double c00 = M[4] * M[8] - M[5] * M[7], c01 = M[5] * M[6] - M[3] * M[8], c02 = M[3] * M[7] - M[4] * M[6];
  const double det = M[0] * c00 + M[1] * c01 + M[2] * c02;
  const double inv[9] = {c00, M[2] * M[7] - M[1] * M[8], M[1] * M[5] - M[2] * M[4],
                         c01, M[0] * M[8] - M[2] * M[6], M[2] * M[3] - M[0] * M[5],
                         c02, M[1] * M[6] - M[0] * M[7], M[0] * M[4] - M[1] * M[3]};
  const double mn[3] = {rho * Un, rho * Ut, 0.0};
  for (int i = 0; i < 3; i++) mom[i] = (inv[3 * i] * mn[0] + inv[3 * i + 1] * mn[1] + inv[3 * i + 2] * mn[2]) / det;
}
__host__ __device__ inline bool is_face_inlet(int category, int type) {
  return category == TPSRHS_INLET && type >= TPSRHS_SUB_DENS_VEL_FACE_X && type <= TPSRHS_SUB_DENS_VEL_FACE_Z;
}

__host__ __device__ inline bool is_non_reflecting(int category, int type) {
  return (category == TPSRHS_INLET && (type == TPSRHS_SUB_DENS_VEL_NR || type == TPSRHS_SUB_VEL_CONST_ENT)) ||
         (category == TPSRHS_OUTLET && (type == TPSRHS_SUB_P_NR || type == TPSRHS_SUB_MF_NR || type == TPSRHS_SUB_MF_NR_PW));
}

// NR_: instantiate the non-reflecting inlet / outlet types.  A flavour of its own, picked only when such a
// patch exists: the extra ghost-state path costs the hot k_flux registers (measured +5 % at cfg2).
// LES_: the sub-grid scale models and the planar viscous sponge, evaluated wherever the molecular transport is
// (a flavour of its own for the same reason).
template <int DIM_, bool NR_ = false, bool LES_ = false>
struct DryAirPhys {
  static constexpr bool LES = LES_;
  // the parameter block of the dry-air kernels travels by value in the kernel-argument segment (the non-reflecting
  // flavour changes it per Mult); see PlasmaPhys::KArg for the other arrangement
  typedef DryAirParams KArg;
  typedef const DryAirParams &PRef;
  typedef const BcDev &BcRef;
  __device__ static inline PRef pref(const KArg &k) { return k; }
  __device__ static inline PRef relaunder(PRef p) { return p; }
  static constexpr int DIM = DIM_;
  static constexpr int NVEL = DIM_;
  static constexpr int NEQ = DIM_ + 2;
  static constexpr int NACTIVE = 0;
  static constexpr bool HAS_SOURCE = false;
  static constexpr bool TWO_TEMPERATURE = false;
  static constexpr int MAX_ORDER = 5;
  static constexpr bool HAS_NR_BC = NR_;  // non-reflecting inlet / outlet types (perfect gas only, as in the reference)
  static constexpr bool AXISYM = false;
  static constexpr bool VISC_USES_GRAD_RHO = false;  // Newtonian stress + Fourier flux: grad u and grad T only
  static constexpr bool HEAVY = false;  // light point physics: inlined at every face pass
  static constexpr bool TWO_STEP = false;  // no state-only closure worth separating from the gradient terms
  static constexpr bool LEAN_TRACE = false;
  static constexpr bool LAUNDER_FLUX = false;  // k_flux re-fetches the parameter image where its face term starts (the table gas)
  struct FluxCoef {};
  static constexpr int MINW_GRAD = TPSRHS_MINW_GRAD, MINW_FLUX = TPSRHS_MINW_FLUX;  // launch-bound waves per SIMD
  // k_gradient of the p = 3 hex (one wave per element, 10 KB of LDS).  Round 2 capped it at 128 VGPRs = four waves per
  // SIMD (natural allocation 133 = three): 0.403 -> 0.382 ms on the builder's box, but with 4 spilled VGPRs + 84 B of
  // scratch, and the driver's box measured it slower (0.423 -> 0.453 ms).  Round 3: no bench kernel may spill
  // (tests/test_spill_allowlist.py); the cap is a build switch for A/B runs only.  Round 4: the kernel reaches 128 registers
  // on its own (k_gradient issues the neighbour records pair by pair, kernels.hpp `LATE`): four waves without the cap.
#ifndef TPSRHS_DRY_GRAD_WAVES
#define TPSRHS_DRY_GRAD_WAVES 3
#endif
  static constexpr int minw_grad(int dim, int p, int nc) {
    return (dim == 3 && p == 3 && !nc && !LES_) ? TPSRHS_DRY_GRAD_WAVES : MINW_GRAD;
  }
  typedef DryAirParams Params;

  // One reciprocal of the density per state; everything else multiplies by it (the reference
  // divides by state[0] in every routine; the results differ in the last bit only).
  struct State {
    double ir;         // 1/rho
    double k;          // |rho u|^2 / rho
    double p;          // pressure
    double vel[NVEL];  // u
  };
  __device__ static inline State make_state(const Params &p, const double *U) {
    State s;
    s.ir = fast_rcp(U[0]);
    double m2 = 0.0;
#pragma unroll
    for (int d = 0; d < NVEL; d++) {
      m2 += U[1 + d] * U[1 + d];
      s.vel[d] = U[1 + d] * s.ir;
    }
    s.k = m2 * s.ir;
    s.p = (p.gamma - 1.0) * (U[1 + NVEL] - 0.5 * s.k);
    return s;
  }

  __device__ static inline double pressure(const Params &p, const double *U) { return make_state(p, U).p; }

  // GetPrimitivesFromConservatives, src/equation_of_state.cpp:321-335
  __device__ static inline void prim(const Params &p, const double *U, double *Up) {
    const State s = make_state(p, U);
    Up[0] = U[0];
#pragma unroll
    for (int d = 0; d < NVEL; d++) Up[1 + d] = s.vel[d];
    Up[1 + NVEL] = s.p * p.inv_Rg * s.ir;
  }

  __device__ static inline void clamp_species(double *) {}

  // ComputeMaxCharSpeed, src/equation_of_state.cpp:278-292
  __device__ static inline double max_char_speed(const Params &p, const double *U, const State &s) {
    return fast_sqrt(s.k * s.ir) + fast_sqrt(p.gamma * s.p * s.ir);
  }
  __device__ static inline double max_char_speed(const Params &p, const double *U) {
    return max_char_speed(p, U, make_state(p, U));
  }
  // DryAir::ComputeSpeedOfSound, src/equation_of_state.cpp:337-348
  __device__ static inline double sound_speed(const Params &p, const double *U) {
    const State s = make_state(p, U);
    return sqrt(p.gamma * s.p * s.ir);
  }

  // F(U).n, src/fluxes.cpp:135-170 contracted with n as in RiemannSolverTPS::ComputeFluxDotN
  __device__ static inline void conv_flux_n(const Params &p, const double *U, const State &s, const double *n,
                                            double *Fn) {
    double un = 0.0;
#pragma unroll
    for (int d = 0; d < DIM; d++) un += s.vel[d] * n[d];
    Fn[0] = U[0] * un;
#pragma unroll
    for (int i = 0; i < NVEL; i++) Fn[1 + i] = U[1 + i] * un + (i < DIM ? s.p * n[i] : 0.0);
    Fn[1 + NVEL] = un * (U[1 + NVEL] + s.p);
  }

  // full convective flux tensor F[eq + d*NEQ]
  __device__ static inline void conv_flux(const Params &p, const double *U, const State &s, double *F) {
    const double H = U[1 + NVEL] + s.p;
#pragma unroll
    for (int d = 0; d < DIM; d++) {
      F[0 + d * NEQ] = U[1 + d];
#pragma unroll
      for (int i = 0; i < NVEL; i++) F[1 + i + d * NEQ] = U[1 + i] * s.vel[d];
      F[1 + d + d * NEQ] += s.p;
      F[1 + NVEL + d * NEQ] = s.vel[d] * H;
    }
  }

  // Lax-Friedrichs flux with the area-weighted normal, src/riemann_solver.cpp:89-115
  __device__ static inline void lax_friedrichs(const Params &p, const double *U1, const double *U2, const double *n,
                                               double *F) {
    const State s1 = make_state(p, U1), s2 = make_state(p, U2);
    const double lam = fmax(max_char_speed(p, U1, s1), max_char_speed(p, U2, s2));
    double f1[NEQ], f2[NEQ];
    conv_flux_n(p, U1, s1, n, f1);
    conv_flux_n(p, U2, s2, n, f2);
    double nm = 0.0;
#pragma unroll
    for (int d = 0; d < DIM; d++) nm += n[d] * n[d];
    const double hl = 0.5 * lam * fast_sqrt(nm);
#pragma unroll
    for (int eq = 0; eq < NEQ; eq++) F[eq] = 0.5 * (f1[eq] + f2[eq]) - hl * (U2[eq] - U1[eq]);
  }

  // RiemannSolverTPS::Eval_Roe, src/riemann_solver.cpp:117-206 (Roe, Lohner): the reference's formula knows two
  // velocity components and gamma - 1 = 0.4
  __device__ static inline void roe(const Params &p, const double *U1, const double *U2, const double *n, double *F) {
    static_assert(DIM == 2, "Eval_Roe is 2-D");
    const State s1 = make_state(p, U1), s2 = make_state(p, U2);
    const double normag = sqrt(n[0] * n[0] + n[1] * n[1]);
    const double un[2] = {n[0] / normag, n[1] / normag};
    double f1[NEQ], f2[NEQ], mean[NEQ];
    conv_flux_n(p, U1, s1, un, f1);
    conv_flux_n(p, U2, s2, un, f2);
#pragma unroll
    for (int eq = 0; eq < NEQ; eq++) mean[eq] = f1[eq] + f2[eq];
    const double sr1 = sqrt(U1[0]), sr2 = sqrt(U2[0]);
    const double r = sqrt(U1[0] * U2[0]);
    const double vel[2] = {(U1[1] / sr1 + U2[1] / sr2) / (sr1 + sr2), (U1[2] / sr1 + U2[2] / sr2) / (sr1 + sr2)};
    const double qk = vel[0] * un[0] + vel[1] * un[1];
    const double H = ((U1[3] + s1.p) / sr1 + (U2[3] + s2.p) / sr2) / (sr1 + sr2);
    const double v2h = 0.5 * (vel[0] * vel[0] + vel[1] * vel[1]);
    const double a2 = 0.4 * (H - v2h);
    const double a = sqrt(a2);
    double l0 = qk;
    if (fabs(l0) < 1e-4) l0 = 1e-4;
    const double dP = s2.p - s1.p;
    const double dU = U2[1] / U2[0] - U1[1] / U1[0];
    const double dV = U2[2] / U2[0] - U1[2] / U1[0];
    const double dQ = dU * un[0] + dV * un[1];
    const double drho = U2[0] - U1[0] - dP / a2;
    double DF1[4] = {drho, vel[0] * drho, vel[1] * drho, v2h * drho};
    DF1[1] += r * (dU - un[0] * dQ);
    DF1[2] += r * (dV - un[1] * dQ);
    DF1[3] += r * (vel[0] * dU + vel[1] * dV - qk * dQ);
    const double c4 = fabs(qk + a) * (dP + r * a * dQ) * 0.5 / a2;
    const double c5 = fabs(qk - a) * (dP - r * a * dQ) * 0.5 / a2;
    const double DF4[4] = {c4, (vel[0] + un[0] * a) * c4, (vel[1] + un[1] * a) * c4, (H + qk * a) * c4};
    const double DF5[4] = {c5, (vel[0] - un[0] * a) * c5, (vel[1] - un[1] * a) * c5, (H - qk * a) * c5};
#pragma unroll
    for (int i = 0; i < 4; i++) F[i] = (mean[i] - (DF1[i] * fabs(l0) + DF4[i] + DF5[i])) * 0.5 * normag;
  }
  // RiemannSolverTPS::Eval (src/riemann_solver.cpp:66-72): interior faces ask with LF = false ...
  __device__ static inline void riemann(const Params &p, const double *U1, const double *U2, const double *n, double *F) {
    if constexpr (DIM == 2) {
      if (p.use_roe) {
        roe(p, U1, U2, n, F);
        return;
      }
    }
    lax_friedrichs(p, U1, U2, n, F);
  }
  // ... and so does the inviscid wall (src/wallBC.cpp:301); every other boundary type passes LF = true
  __device__ static inline void riemann_bc(const Params &p, const BcDev &bc, const double *U1, const double *Ug,
                                           const double *n, double *F) {
    if constexpr (DIM == 2) {
      if (p.use_roe && bc.category == TPSRHS_WALL && (bc.type == TPSRHS_INV || bc.type == TPSRHS_SLIP)) {
        roe(p, U1, Ug, n, F);
        return;
      }
    }
    lax_friedrichs(p, U1, Ug, n, F);
  }

  struct Transport {
    double visc, bulk, k;  // bulk already has -2/3 visc applied
  };
  // DryAirTransport::ComputeFluxMolecularTransport, src/transport_properties.cpp:224-266
  __device__ static inline Transport transport(const Params &p, const State &s) {
    const double T = s.p * p.inv_Rg * s.ir;
    Transport t;
    t.visc = p.C1 * p.visc_mult * (T * fast_sqrt(T)) * fast_rcp(T + p.S0);
    t.bulk = p.bulk_mult * t.visc - 2.0 / 3.0 * t.visc;
    t.k = p.cp_div_pr * t.visc;
    return t;
  }
  // Fluxes::sgsSmag, src/fluxes.cpp:513-537 (three velocity components)
  __device__ static inline double sgs_smagorinsky(const Params &p, double rho, const double *g, double delta) {
    static_assert(DIM == 3, "the reference's strain tensor indexes three directions");
    const double S0 = g[1 + 0 * NEQ], S1 = g[2 + 1 * NEQ], S2 = g[3 + 2 * NEQ];
    const double S3 = 0.5 * (g[1 + 1 * NEQ] + g[2 + 0 * NEQ]);
    const double S4 = 0.5 * (g[1 + 2 * NEQ] + g[3 + 0 * NEQ]);
    const double S5 = 0.5 * (g[2 + 2 * NEQ] + g[3 + 1 * NEQ]);
    double Smag = S0 * S0;
    Smag += S1 * S1;
    Smag += S2 * S2;
    Smag += 2.0 * S3 * S3;
    Smag += 2.0 * S4 * S4;
    Smag += 2.0 * S5 * S5;
    Smag = sqrt(2.0 * Smag);
    const double d_model = p.sgs_const * fmax(delta - p.sgs_floor, 0.0);
    return rho * d_model * d_model * Smag;
  }
  // Fluxes::sgsSigma, src/fluxes.cpp:543-665, the branch without LAPACK (eigenvalues of g^T g, its 12-digit pi)
  __device__ static inline double sgs_sigma(const Params &p, double rho, const double *g, double delta) {
    static_assert(DIM == 3, "three singular values");
    const double sml = 1.0e-12, pi = 3.14159265359, onethird = 1. / 3.;
    const double d_model = fmax(delta - p.sgs_floor, sml);
    double Q[3][3];
#pragma unroll
    for (int i = 0; i < 3; i++)
#pragma unroll
      for (int j = 0; j < 3; j++) {
        double q = 0.0;
#pragma unroll
        for (int k = 0; k < 3; k++) q += g[k + 1 + i * NEQ] * g[k + 1 + j * NEQ];
        Q[i][j] = q;
      }
    const double d4 = pow(d_model, 4);
#pragma unroll
    for (int i = 0; i < 3; i++)
#pragma unroll
      for (int j = 0; j < 3; j++) Q[i][j] *= d4;
    const double p1 = Q[0][1] * Q[0][1] + Q[0][2] * Q[0][2] + Q[1][2] * Q[1][2];
    const double q = onethird * (Q[0][0] + Q[1][1] + Q[2][2]);
    const double p2 = (Q[0][0] - q) * (Q[0][0] - q) + (Q[1][1] - q) * (Q[1][1] - q) + (Q[2][2] - q) * (Q[2][2] - q) + 2.0 * p1;
    const double pp = sqrt(fmax(p2, 0.0) / 6.0);
    double B[3][3];
#pragma unroll
    for (int i = 0; i < 3; i++)
#pragma unroll
      for (int j = 0; j < 3; j++) B[i][j] = (Q[i][j] - (i == j ? q : 0.0)) * (1.0 / fmax(pp, sml));
    const double detB = B[0][0] * (B[1][1] * B[2][2] - B[2][1] * B[1][2]) - B[0][1] * (B[1][0] * B[2][2] - B[2][0] * B[1][2]) +
                        B[0][2] * (B[1][0] * B[2][1] - B[2][0] * B[1][1]);
    const double r = 0.5 * detB;
    double phi;
    if (r <= -1.0)
      phi = onethird * pi;
    else if (r >= 1.0)
      phi = 0.0;
    else
      phi = onethird * acos(r);
    double ev[3];
    ev[0] = q + 2.0 * pp * cos(phi);
    ev[2] = q + 2.0 * pp * cos(phi + (2.0 * onethird * pi));
    ev[1] = 3.0 * q - ev[0] - ev[2];
    const double s0 = sqrt(fmax(ev[0], sml)), s1 = sqrt(fmax(ev[1], sml)), s2 = sqrt(fmax(ev[2], sml));
    double mu = s2 * (s0 - s1) * (s1 - s2);
    mu = fmax(mu, 0.0);
    mu /= (s0 * s0);
    mu *= (p.sgs_const * p.sgs_const);
    mu *= rho;
    if (mu != mu) mu = 0.0;
    return mu;
  }
  // Fluxes::viscSpongePlanar, src/fluxes.cpp:669-688 (the normal as given, not normalised)
  __device__ static inline double visc_sponge_weight(const Params &p, const double *X) {
    const double factor = fmax(p.vs_ratio, 1.0);
    double dist = 0.0;
#pragma unroll
    for (int d = 0; d < DIM; d++) dist += (X[d] - p.vs_p[d]) * p.vs_n[d];
    double wgt = 0.5 * (tanh(dist / p.vs_width - 2.0) + 1.0);
    wgt *= (factor - 1.0);
    wgt += 1.0;
    return wgt;
  }
  // the transport of a point: molecular, then the sub-grid scale model and the viscous sponge (src/fluxes.cpp:216-246)
  __device__ static inline Transport transport(const Params &p, const State &s, const double *U, const double *g,
                                               const PointCtx *c) {
    Transport t = transport(p, s);
    if constexpr (LES_) {
      if (p.sgs_type > 0) {
        if constexpr (DIM == 3) {
          const double pr_cp = t.visc / t.k;
          const double mu_sgs = (p.sgs_type == 1) ? sgs_smagorinsky(p, U[0], g, c->delta) : sgs_sigma(p, U[0], g, c->delta);
          t.bulk *= (1.0 + mu_sgs / t.visc);
          t.visc += mu_sgs;
          t.k += (mu_sgs / pr_cp);
        }
      }
      if (p.vs_enabled) {
        const double wgt = visc_sponge_weight(p, c->X);
        t.visc *= wgt;
        t.bulk *= wgt;
        t.k *= wgt;
      }
    }
    return t;
  }

  // ComputeViscousFluxes, src/fluxes.cpp:178-335; g[eq + d*NEQ] = d(Up_eq)/dx_d; F[eq + d*NEQ]
  __device__ static inline void visc_flux(const Params &p, const double *U, const State &s, const double *g,
                                          double *F, const PointCtx *c = nullptr) {
#pragma unroll
    for (int i = 0; i < NEQ * DIM; i++) F[i] = 0.0;
    if (p.eq_system == TPSRHS_EULER) return;
    const Transport t = transport(p, s, U, g, c);
    double divV = 0.0;
#pragma unroll
    for (int i = 0; i < DIM; i++) divV += g[(1 + i) + i * NEQ];
    double stress[DIM * DIM];
#pragma unroll
    for (int i = 0; i < DIM; i++)
#pragma unroll
      for (int j = 0; j < DIM; j++) stress[i + j * DIM] = t.visc * (g[(1 + j) + i * NEQ] + g[(1 + i) + j * NEQ]);
#pragma unroll
    for (int i = 0; i < DIM; i++) stress[i + i * DIM] += t.bulk * divV;
#pragma unroll
    for (int i = 0; i < DIM; i++)
#pragma unroll
      for (int j = 0; j < DIM; j++) F[(1 + i) + j * NEQ] = stress[i + j * DIM];
#pragma unroll
    for (int i = 0; i < DIM; i++) {
      double vt = 0.0;
#pragma unroll
      for (int j = 0; j < DIM; j++) vt += stress[i + j * DIM] * s.vel[j];
      F[(1 + NVEL) + i * NEQ] = vt + t.k * g[(1 + NVEL) + i * NEQ];
    }
  }
  __device__ static inline void visc_flux(const Params &p, const double *U, const double *g, double *F,
                                          const PointCtx *c = nullptr) {
    visc_flux(p, U, make_state(p, U), g, F, c);
  }

  // F_c - F_v as one tensor F[eq + d*NEQ] (src/rhs_operator.cpp:532-541)
  __device__ static inline void total_flux(const Params &p, const double *U, const State &s, const double *g,
                                           double *F, const PointCtx *c = nullptr) {
    const double H = U[1 + NVEL] + s.p;
#pragma unroll
    for (int d = 0; d < DIM; d++) {
      F[0 + d * NEQ] = U[1 + d];
#pragma unroll
      for (int i = 0; i < NVEL; i++) F[1 + i + d * NEQ] = U[1 + i] * s.vel[d];
      F[1 + d + d * NEQ] += s.p;
      F[1 + NVEL + d * NEQ] = s.vel[d] * H;
    }
    if (p.eq_system == TPSRHS_EULER) return;
    const Transport t = transport(p, s, U, g, c);
    double divV = 0.0;
#pragma unroll
    for (int i = 0; i < DIM; i++) divV += g[(1 + i) + i * NEQ];
#pragma unroll
    for (int i = 0; i < DIM; i++) {
      double vt = 0.0;
#pragma unroll
      for (int j = 0; j < DIM; j++) {
        double st = t.visc * (g[(1 + j) + i * NEQ] + g[(1 + i) + j * NEQ]);
        if (i == j) st += t.bulk * divV;
        F[(1 + j) + i * NEQ] -= st;  // stress is symmetric: tau_ij enters F[1+j][i]
        vt += st * s.vel[j];
      }
      F[(1 + NVEL) + i * NEQ] -= vt + t.k * g[(1 + NVEL) + i * NEQ];
    }
  }

  // F_v(U, g) . n without forming the tensor
  __device__ static inline void visc_flux_n(const Params &p, const double *U, const double *g, const double *n,
                                            double *Fn, const PointCtx *c = nullptr) {
#pragma unroll
    for (int eq = 0; eq < NEQ; eq++) Fn[eq] = 0.0;
    if (p.eq_system == TPSRHS_EULER) return;
    const State s = make_state(p, U);
    const Transport t = transport(p, s, U, g, c);
    double divV = 0.0;
#pragma unroll
    for (int i = 0; i < DIM; i++) divV += g[(1 + i) + i * NEQ];
    double e = 0.0, qn = 0.0;
#pragma unroll
    for (int i = 0; i < DIM; i++) {
      double sn = 0.0;  // (stress . n)_i
#pragma unroll
      for (int j = 0; j < DIM; j++) sn += (g[(1 + j) + i * NEQ] + g[(1 + i) + j * NEQ]) * n[j];
      sn = t.visc * sn + t.bulk * divV * n[i];
      Fn[1 + i] = sn;
      e += sn * s.vel[i];
      qn += g[(1 + NVEL) + i * NEQ] * n[i];
    }
    Fn[1 + NVEL] = e + t.k * qn;
  }

  // ComputeBdrViscousFluxes with zero prescribed species flux and, when `adiabatic`, zero heat
  // flux (src/fluxes.cpp:344-505 with the bcFlux_ of src/wallBC.cpp:86-111); nu = unit normal
  __device__ static inline void bdr_visc_flux(const Params &p, const double *Uw, const double *g, const double *nu,
                                              bool adiabatic, double *Fn, const PointCtx *c = nullptr) {
#pragma unroll
    for (int eq = 0; eq < NEQ; eq++) Fn[eq] = 0.0;
    if (p.eq_system == TPSRHS_EULER) return;
    const State sw = make_state(p, Uw);
    const Transport t = transport(p, sw, Uw, g, c);
    double divV = 0.0;
#pragma unroll
    for (int i = 0; i < DIM; i++) divV += g[(1 + i) + i * NEQ];
    double sn[DIM];  // stress . n
#pragma unroll
    for (int i = 0; i < DIM; i++) {
      double s = 0.0;
#pragma unroll
      for (int j = 0; j < DIM; j++) {
        double st = t.visc * (g[(1 + j) + i * NEQ] + g[(1 + i) + j * NEQ]);
        if (i == j) st += t.bulk * divV;
        s += st * nu[j];
      }
      sn[i] = s;
    }
    double q = 0.0;  // heat flux (standard sign): -k dT/dn
    if (!adiabatic) {
#pragma unroll
      for (int d = 0; d < DIM; d++) q -= t.k * g[(1 + NVEL) + d * NEQ] * nu[d];
    }
    double e = -q;
#pragma unroll
    for (int d = 0; d < NVEL; d++) {
      Fn[1 + d] = sn[d];
      e += sn[d] * sw.vel[d];
    }
    Fn[1 + NVEL] = e;
  }

  // ---- non-reflecting inlet (src/inletBC.cpp:576-727) and outlets (src/outletBC.cpp:573-728, 739-892,
  // 894-1027): characteristic estimate of dU/dt at a boundary point from the patch mean of the primitives, the
  // normal gradient and the target; `state2` (the boundary state the Riemann solver sees) advanced by dt -> newU.
  __device__ static inline void nr_update(const Params &p, const BcDev &bc, double dt, const double *meanUp, const double *n,
                                          const double *U, const double *g, const double *state2, double *newU) {
    const bool inlet = bc.category == TPSRHS_INLET;
    double un[DIM], t1[DIM], t2[3] = {0.0, 0.0, 0.0};
    {
      double mod = 0.0;
#pragma unroll
      for (int d = 0; d < DIM; d++) mod += n[d] * n[d];
      const double sc = (inlet ? -1.0 : 1.0) / sqrt(mod);  // inlet: pointing into the domain
#pragma unroll
      for (int d = 0; d < DIM; d++) {
        un[d] = n[d] * sc;
        t1[d] = bc.data[4 + d];
      }
    }
    double meanVel[3] = {0.0, 0.0, 0.0};
#pragma unroll
    for (int d = 0; d < DIM; d++) {
      meanVel[0] += un[d] * meanUp[d + 1];
      meanVel[1] += t1[d] * meanUp[d + 1];
    }
    if constexpr (DIM == 3) {
      t2[0] = un[1] * t1[2] - un[2] * t1[1];
      t2[1] = un[2] * t1[0] - un[0] * t1[2];
      t2[2] = un[0] * t1[1] - un[1] * t1[0];
#pragma unroll
      for (int d = 0; d < 3; d++) meanVel[2] += t2[d] * meanUp[d + 1];
    }
    double ng[NEQ];
#pragma unroll
    for (int eq = 0; eq < NEQ; eq++) {
      ng[eq] = 0.0;
#pragma unroll
      for (int d = 0; d < DIM; d++) ng[eq] += un[d] * g[eq + d * NEQ];
    }
    const State s = make_state(p, U);
    const double T = s.p * p.inv_Rg * s.ir;
    const double dpdn = p.Rg * (T * ng[0] + U[0] * ng[NVEL + 1]);  // ComputePressureDerivative, equation_of_state.cpp:350-359
    const double meanP = p.Rg * meanUp[0] * meanUp[NVEL + 1];
    const double c = sqrt(p.gamma * p.Rg * meanUp[NVEL + 1]);
    double meanK = 0.0;
#pragma unroll
    for (int d = 0; d < DIM; d++) meanK += meanUp[1 + d] * meanUp[1 + d];
    meanK *= 0.5;
    const double sigma = c / p.ref_length;
    double L1, L2, L3 = 0.0, L4 = 0.0, L5;
    if (inlet) {
      double dv[DIM];
#pragma unroll
      for (int d = 0; d < DIM; d++) dv[d] = meanUp[1 + d] - bc.data[1 + d];
      L1 = 0.0;
#pragma unroll
      for (int d = 0; d < DIM; d++) L1 += un[d] * ng[1 + d];
      L1 = dpdn - meanUp[0] * c * L1;
      L1 *= meanVel[0] - c;
      L5 = 0.0;
#pragma unroll
      for (int d = 0; d < DIM; d++) L5 += dv[d] * un[d];
      L5 *= sigma * 2.0 * meanUp[0] * c;
#pragma unroll
      for (int d = 0; d < DIM; d++) L3 += dv[d] * t1[d];
      L3 *= sigma;
      if constexpr (DIM == 3) {
#pragma unroll
        for (int d = 0; d < 3; d++) L4 += dv[d] * t2[d];
        L4 *= sigma;
      }
      L2 = sigma * c * c * (meanUp[0] - bc.data[0]) - 0.5 * L5;
      if (bc.type == TPSRHS_SUB_VEL_CONST_ENT) L2 = 0.0;
    } else {
      L2 = (c * c * ng[0] - dpdn) * meanVel[0];
#pragma unroll
      for (int d = 0; d < DIM; d++) L3 += t1[d] * ng[1 + d];
      L3 *= meanVel[0];
      if constexpr (DIM == 3) {
#pragma unroll
        for (int d = 0; d < 3; d++) L4 += t2[d] * ng[1 + d];
        L4 *= meanVel[0];
      }
      L5 = 0.0;
#pragma unroll
      for (int d = 0; d < DIM; d++) L5 += un[d] * ng[1 + d];
      L5 = dpdn + meanUp[0] * c * L5;
      L5 *= meanVel[0] + c;
      if (bc.type == TPSRHS_SUB_P_NR) {
        L1 = sigma * (meanP - bc.data[0]);
      } else {
        double vn = meanVel[0];
        if (bc.type == TPSRHS_SUB_MF_NR_PW) {
          vn = 0.0;
#pragma unroll
          for (int d = 0; d < DIM; d++) vn += U[1 + d] * un[d];
          vn /= U[0];
        }
        L1 = -sigma * (vn - bc.data[0] / meanUp[0] / bc.data[7]);
        L1 *= meanUp[0] * c;
      }
    }
    const double d1 = (L2 + 0.5 * (L5 + L1)) / c / c;
    const double d2 = 0.5 * (L5 - L1) / meanUp[0] / c;
    const double d5 = 0.5 * (L5 + L1);
    double f[NEQ];
    f[0] = d1;
    f[1] = meanVel[0] * d1 + meanUp[0] * d2;
    f[2] = meanVel[1] * d1 + meanUp[0] * L3;
    if constexpr (DIM == 3) f[3] = meanVel[2] * d1 + meanUp[0] * L4;
    f[1 + DIM] = meanUp[0] * meanVel[0] * d2;
    f[1 + DIM] += meanUp[0] * meanVel[1] * L3;
    if constexpr (DIM == 3) f[1 + DIM] += meanUp[0] * meanVel[2] * L4;
    f[1 + DIM] += meanK * d1 + d5 / (p.gamma - 1.0);
    // boundary state in the (normal, tangent) frame, advanced, and back: momX = M^-1 momN, rows of M = un, t1, t2
    double sn[NEQ];
#pragma unroll
    for (int eq = 0; eq < NEQ; eq++) sn[eq] = state2[eq];
#pragma unroll
    for (int d = 0; d < DIM; d++) sn[1 + d] = 0.0;
#pragma unroll
    for (int d = 0; d < DIM; d++) {
      sn[1] += state2[1 + d] * un[d];
      sn[2] += state2[1 + d] * t1[d];
      if constexpr (DIM == 3) sn[3] += state2[1 + d] * t2[d];
    }
#pragma unroll
    for (int eq = 0; eq < NEQ; eq++) newU[eq] = sn[eq] - dt * f[eq];
    if constexpr (DIM == 2) {
      const double det = un[0] * t1[1] - un[1] * t1[0];
      const double m0 = newU[1], m1 = newU[2];
      newU[1] = (t1[1] * m0 - un[1] * m1) / det;
      newU[2] = (-t1[0] * m0 + un[0] * m1) / det;
    } else {
      const double M[9] = {un[0], un[1], un[2], t1[0], t1[1], t1[2], t2[0], t2[1], t2[2]};  // row-major
      const double c00 = M[4] * M[8] - M[5] * M[7], c01 = M[5] * M[6] - M[3] * M[8], c02 = M[3] * M[7] - M[4] * M[6];
      const double det = M[0] * c00 + M[1] * c01 + M[2] * c02;
      const double inv[9] = {c00, M[2] * M[7] - M[1] * M[8], M[1] * M[5] - M[2] * M[4],
                             c01, M[0] * M[8] - M[2] * M[6], M[2] * M[3] - M[0] * M[5],
                             c02, M[1] * M[6] - M[0] * M[7], M[0] * M[4] - M[1] * M[3]};
      const double m0 = newU[1], m1 = newU[2], m2 = newU[3];
#pragma unroll
      for (int i = 0; i < 3; i++) newU[1 + i] = (inv[3 * i] * m0 + inv[3 * i + 1] * m1 + inv[3 * i + 2] * m2) / det;
    }
  }
  // GetConservativesFromPrimitives, src/equation_of_state.cpp:294-319 (initial boundary state)
  __device__ static inline void cons(const Params &p, const double *Up, double *U) {
    U[0] = Up[0];
    double k = 0.0;
#pragma unroll
    for (int d = 0; d < NVEL; d++) {
      U[1 + d] = Up[0] * Up[1 + d];
      k += Up[1 + d] * Up[1 + d];
    }
    U[1 + NVEL] = p.Rg * Up[0] * Up[1 + NVEL] / (p.gamma - 1.0) + 0.5 * Up[0] * k;
  }

  // DryAir::computeConservedStateFromConvectiveFlux, src/equation_of_state.cpp:414-444 (mixed-out sponge target)
  static constexpr bool HAS_MIXED_OUT = true;
  __device__ static inline void state_from_mean_flux(const Params &p, const double *mf, const double *n, double *U) {
    const double gamma = p.gamma;
    double temp = 0.0;
#pragma unroll
    for (int d = 0; d < DIM; d++) temp += mf[1 + d] * n[d];
    const double A = 1. - 2. * gamma / (gamma - 1.);
    const double B = 2 * temp / (gamma - 1.);
    double Cc = -2. * mf[0] * mf[1 + NVEL];
#pragma unroll
    for (int d = 0; d < NVEL; d++) Cc += mf[1 + d] * mf[1 + d];
    const double pr = (-B - sqrt(B * B - 4. * A * Cc)) / (2. * A);
    double Up[NEQ];
    Up[0] = mf[0] * mf[0] / (temp - pr);
    Up[1 + NVEL] = pr / (p.Rg * Up[0]);
#pragma unroll
    for (int d = 0; d < NVEL; d++) Up[1 + d] = (mf[1 + d] - pr * n[d]) / mf[0];
    cons(p, Up, U);
  }

  // ---- boundary conditions ------------------------------------------------------------------
  // ghost (second) state handed to the Riemann solver
  // (`bs`: this face point's record of the non-reflecting boundary state, or NULL)
  __device__ static inline void bc_ghost(const Params &p, const BcDev &bc, const double *U, const double *n,
                                         double *Ug, const double *bs = nullptr) {
#pragma unroll
    for (int eq = 0; eq < NEQ; eq++) Ug[eq] = U[eq];
    if (NR_ && is_non_reflecting(bc.category, bc.type)) {  // state2 = boundaryU[bdrN], src/outletBC.cpp:693-727
#pragma unroll
      for (int eq = 0; eq < NEQ; eq++) Ug[eq] = bs[eq];
    } else if (bc.category == TPSRHS_INLET) {  // SUB_DENS_VEL, src/inletBC.cpp:729-757
      const double pres = pressure(p, U);
      Ug[0] = bc.data[0];
      double k = 0.0;
      if constexpr (DIM == 3) {
        if (is_face_inlet(bc.category, bc.type)) {  // src/inletBC.cpp:758-864: tmpU = 2 state2 - stateIn in the momentum rows
          double mom[3];
          face_inlet_momentum(n, bc.type - TPSRHS_SUB_DENS_VEL_FACE_X, bc.data[0], bc.data[1], bc.data[2], mom);
#pragma unroll
          for (int d = 0; d < 3; d++) {
            Ug[1 + d] = 2.0 * mom[d] - U[1 + d];
            k += Ug[1 + d] * Ug[1 + d];
          }
          k *= 0.5 / Ug[0];
          Ug[1 + NVEL] = pres / (p.gamma - 1.0) + k;
          return;
        }
      }
#pragma unroll
      for (int d = 0; d < NVEL; d++) {
        Ug[1 + d] = bc.data[0] * bc.data[1 + d];
        k += Ug[1 + d] * Ug[1 + d];
      }
      k *= 0.5 / Ug[0];
      Ug[1 + NVEL] = pres / (p.gamma - 1.0) + k;  // modifyEnergyForPressure, :402-411
    } else if (bc.category == TPSRHS_OUTLET) {  // SUB_P, src/outletBC.cpp:731-737
      double k = 0.0;
#pragma unroll
      for (int d = 0; d < NVEL; d++) k += U[1 + d] * U[1 + d];
      k *= 0.5 / U[0];
      Ug[1 + NVEL] = bc.data[0] / (p.gamma - 1.0) + k;
    } else if (bc.type == TPSRHS_INV || bc.type == TPSRHS_SLIP) {  // mirrored normal momentum, src/wallBC.cpp:277-301
      // (SLIP, :326-428, builds the same mirror state through a wall-aligned frame)
      double nm = 0.0;
#pragma unroll
      for (int d = 0; d < DIM; d++) nm += n[d] * n[d];
      nm = sqrt(nm);
      double vn = 0.0;
#pragma unroll
      for (int d = 0; d < DIM; d++) vn += (U[1 + d] / U[0]) * (n[d] / nm);
#pragma unroll
      for (int d = 0; d < DIM; d++) Ug[1 + d] = U[0] * (U[1 + d] / U[0] - 2.0 * vn * (n[d] / nm));
      if (DIM == 2 && bc.type == TPSRHS_SLIP) slip_ghost_momentum_2d(n, U, Ug);
    } else if (bc.type == TPSRHS_VISC_ADIAB) {  // computeStagnationState, :367-378
      const double pres = pressure(p, U);
#pragma unroll
      for (int d = 0; d < NVEL; d++) Ug[1 + d] = 0.0;
      Ug[1 + NVEL] = pres / (p.gamma - 1.0);
    } else {  // VISC_ISOTH, src/wallBC.cpp:471-485
      if (p.use_bc_in_grad) {
#pragma unroll
        for (int d = 0; d < NVEL; d++) Ug[1 + d] = -U[1 + d];
      } else {
#pragma unroll
        for (int d = 0; d < NVEL; d++) Ug[1 + d] = 0.0;
        Ug[1 + NVEL] = p.Rg / (p.gamma - 1.0) * U[0] * bc.data[0];  // computeStagnantStateWithTemp, :380-387
      }
    }
  }

  // additive viscous part of the boundary flux: what the wall routines subtract from bdrFlux
  // after the Riemann solve (src/wallBC.cpp:303-319, 439-468, 487-509); zero for inlet/outlet
  __device__ static inline void bc_visc_term(const Params &p, const BcDev &bc, const double *U, const double *g,
                                             const double *n, double *out, const PointCtx *c = nullptr) {
#pragma unroll
    for (int eq = 0; eq < NEQ; eq++) out[eq] = 0.0;
    if (bc.category != TPSRHS_WALL || p.eq_system == TPSRHS_EULER || bc.type == TPSRHS_SLIP) return;  // slip: Riemann flux only
    double nm = 0.0;
#pragma unroll
    for (int d = 0; d < DIM; d++) nm += n[d] * n[d];
    nm = sqrt(nm);
    double fin[NEQ];
    visc_flux_n(p, U, g, n, fin, c);
    double fw[NEQ];
    if (bc.type == TPSRHS_INV) {
      double Ug[NEQ];
      bc_ghost(p, bc, U, n, Ug);
      visc_flux_n(p, Ug, g, n, fw, c);
    } else {
      double Uw[NEQ], nu[DIM];
#pragma unroll
      for (int d = 0; d < DIM; d++) nu[d] = n[d] / nm;
#pragma unroll
      for (int eq = 0; eq < NEQ; eq++) Uw[eq] = U[eq];
#pragma unroll
      for (int d = 0; d < NVEL; d++) Uw[1 + d] = 0.0;
      if (bc.type == TPSRHS_VISC_ADIAB) {
        Uw[1 + NVEL] = pressure(p, U) / (p.gamma - 1.0);
        bdr_visc_flux(p, Uw, g, nu, true, fw, c);
      } else {
        Uw[1 + NVEL] = p.Rg / (p.gamma - 1.0) * U[0] * bc.data[0];
        bdr_visc_flux(p, Uw, g, nu, false, fw, c);
      }
#pragma unroll
      for (int eq = 0; eq < NEQ; eq++) fw[eq] *= nm;
    }
#pragma unroll
    for (int eq = 1; eq < NEQ; eq++) out[eq] = -0.5 * fw[eq] - 0.5 * fin[eq];
  }

  // ghost primitive state of the gradient jump (useBCinGrad), src/wallBC.cpp:241-266
  __device__ static inline void bc_grad_prim(const Params &p, const BcDev &bc, const double *Up, double *UpB) {
#pragma unroll
    for (int eq = 0; eq < NEQ; eq++) UpB[eq] = Up[eq];
    if (p.use_bc_in_grad && bc.category == TPSRHS_WALL && bc.type == TPSRHS_VISC_ISOTH) {
#pragma unroll
      for (int d = 0; d < NVEL; d++) UpB[1 + d] = 0.0;
      UpB[1 + NVEL] = bc.data[0];
    }
  }

  __device__ static inline void source(const Params &, const double *, const double *, const double *, double *) {}
  __device__ static inline double electric_conductivity(const Params &, const double *) { return 0.0; }  // no SourceTerm for dry air
};

}  // namespace tpsrhs
#endif
