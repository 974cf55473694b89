// Device point-wise physics: dry air in the AXISYMMETRIC formulation (dim 2, velocity components
// r, z, theta).  Same closures as physics_dryair.hpp plus the 1/r terms; kept apart from the tuned
// planar / 3-D class, with the "heavy" kernel interface (visc_trace, total_flux with a radius,
// axisym_source).
//   DryAir             src/equation_of_state.cpp:150-412
//   DryAirTransport    src/transport_properties.cpp:205-276
//   Fluxes             src/fluxes.cpp:135-505 (axisymmetric branches :286-313,441-466)
//   AxisymmetricSource src/forcing_terms.cpp:293-380
//   boundary ghosts    src/inletBC.cpp:729-757, src/outletBC.cpp:731-737, src/wallBC.cpp:277-510
#ifndef TPSRHS_PHYSICS_DRYAIR_AXISYM_HPP_
#define TPSRHS_PHYSICS_DRYAIR_AXISYM_HPP_

#include <type_traits>

#include "physics_dryair.hpp"
#include "physics_plasma.hpp"  // TableDev, table_eval (LinearTable on the device)

namespace tpsrhs {

// ---- WorkingFluid::LTE_FLUID with one-dimensional tables (flow/lte/table_dim = 1: the variant of the reference's
// device build, src/M2ulPhyS.cpp:164-255): one species, the state (rho, rho u, rho E); thermodynamics and transport
// are LinearTables in the temperature.
//   LteMixture    src/lte_mixture.cpp:76-470          LteTransport  src/lte_transport_properties.cpp:60-140
// The parameter block extends the dry-air one (boundary conditions, switches); the gas constants of the latter are unused.
struct LteParams : DryAirParams {
  TableDev tab_e, tab_R, tab_c, tab_T;    // e(T), R(T), c(T) and the inverse T(e) (the energy table swapped, src/M2ulPhyS.cpp:193-200)
  TableDev tab_mu, tab_k, tab_sigma;      // mu(T), kappa(T), sigma(T)
  TableDev tab_nec;                       // net emission coefficient (src/radiation.hpp:54-69), when `radiation`
  int radiation;
  // Search aids (the intervals they find are LinearTable::findInterval's, verified against the abscissae):
  //  * the inverse table has no uniform axis, and the reference's bisection is 7 dependent loads per state: `ehint[j]` is
  //    the interval of the left edge of the j-th of `nhint` uniform bins over the energy axis, the search walks on from there;
  //  * e, R, c (one file, one temperature column) and mu, kappa share their abscissae: one search serves the tables
  //    evaluated at the same temperature.
  const int *ehint;
  int nhint, thermo_same_grid, trans_same_grid;
  double e0, inv_de;
};
// LinearTable::eval / eval_x (src/table.cpp:80-113) in a known interval
template <class TD>
__device__ inline double table_at(const TD &t, int idx, double xe) {
  const double xt = t.x_log ? flog(xe) : xe;
  double ft = t.a[idx] + t.b[idx] * xt;
  if (t.f_log) ft = fexp(ft);
  return ft;
}
template <class TD>
__device__ inline double table_slope_at(const TD &t, int idx, double xe) {
  const double xt = t.x_log ? flog(xe) : xe;
  double ft_x = t.b[idx] * (t.x_log ? 1.0 / xe : 1.0);
  if (t.f_log) ft_x *= fexp(t.a[idx] + t.b[idx] * xt);
  return ft_x;
}
template <class TD>
__device__ inline double table_eval_x(const TD &t, double xe) { return table_slope_at(t, table_interval(t, xe), xe); }
// findInterval of the inverse table T(e) through the bins
template <class LP>
__device__ inline int lte_energy_interval(const LP &p, double e) {
  const auto &t = p.tab_T;
  if (p.ehint) {
    const double fj = fmin(fmax((e - p.e0) * p.inv_de, 0.0), static_cast<double>(p.nhint - 1));  // (NaN -> 0)
    int g = p.ehint[static_cast<int>(fj)];
    while (g < t.n - 2 && e > t.x[g + 1]) g++;
    if ((g == 0 || e > t.x[g]) && (g == t.n - 2 || !(e > t.x[g + 1]))) return g;
  }
  return table_interval(t, e);
}
// LteMixture::ComputeTemperatureInternal, src/lte_mixture.cpp:161-218: Newton on e(T) = energy from the inverse table
// (the reference asserts convergence; a state that does not converge returns NaN here and is caught like any other).
// `it`: the interval of the result in the temperature grid of the thermodynamic tables.
template <class LP>
__device__ inline double lte_temperature(const LP &p, double energy, int &it) {
  double T = table_at(p.tab_T, lte_energy_interval(p, energy), energy);
  it = table_interval(p.tab_e, T);
  double res = energy - table_at(p.tab_e, it, T);
  const double res0 = fabs(res);
  const double atol = 1e-18, rtol = 1e-12, dT_atol = 1e-12, dT_rtol = 1e-8;
  bool converged = (fabs(res) < atol) || (fabs(res) / fabs(res0) < rtol);
  int niter = 0;
  while (!converged && niter < 20) {
    const double dedT = table_slope_at(p.tab_e, it, T);
    const double dT = res / dedT;
    T += dT;
    it = table_interval(p.tab_e, T);
    res = energy - table_at(p.tab_e, it, T);
    converged = (fabs(res) < atol) || (fabs(res) / res0 < rtol) || (fabs(dT) < dT_atol) || (fabs(dT) / T < dT_rtol);
    niter++;
  }
  return converged ? T : __builtin_nan("");
}
// a thermodynamic table other than e(T) at a temperature whose interval in e's grid is known
template <class LP, class TD>
__device__ inline double lte_thermo_at(const LP &p, const TD &t, int it, double T) {
  return p.thermo_same_grid ? table_at(t, it, T) : table_eval(t, T);
}
// LteMixture::ComputeTemperatureFromDensityPressure, src/lte_mixture.cpp:236-296: Newton on p = rho R(T) T
// (the reference goes on with the last iterate when the iteration has not converged)
template <class LP>
__device__ inline double lte_temperature_rho_p(const LP &p, double rho, double pres, int &it) {
  double T = pres / (rho * 208.);
  it = table_interval(p.tab_R, T);
  double R = table_at(p.tab_R, it, T);
  double res = pres - rho * R * T;
  const double res0 = fabs(res);
  const double atol = 1e-18, rtol = 1e-12, dT_atol = 1e-12, dT_rtol = 1e-8;
  bool converged = (fabs(res) < atol) || (fabs(res) / fabs(res0) < rtol);
  int niter = 0;
  while (!converged && niter < 20) {
    const double R_T = table_slope_at(p.tab_R, it, T);
    const double dpdT = rho * R + rho * R_T * T;
    const double dT = res / dpdT;
    T += dT;
    it = table_interval(p.tab_R, T);
    R = table_at(p.tab_R, it, T);
    res = pres - rho * R * T;
    converged = (fabs(res) < atol) || (fabs(res) / res0 < rtol) || (fabs(dT) < dT_atol) || (fabs(dT) / T < dT_rtol);
    niter++;
  }
  return T;
}

// LTE_ = false: dry air (gamma law, Sutherland); true: the table gas
template <bool LTE_>
struct GasAxiPhys {
  static constexpr int DIM = 2, NVEL = 3, NEQ = 5, NACTIVE = 0, ITH = 4;
  static constexpr bool HAS_SOURCE = true, AXISYM = true, HEAVY = true, TWO_TEMPERATURE = false, HAS_NR_BC = false;
  static constexpr bool TWO_STEP = false;
  static constexpr bool LEAN_TRACE = false;
  static constexpr bool LAUNDER_FLUX = LTE_;  // k_flux re-fetches the parameter image where its face term starts (the table gas)
  struct FluxCoef {};
  static constexpr bool VISC_USES_GRAD_RHO = false;
  static constexpr int MAX_ORDER = 4;
  static constexpr int MINW_GRAD = 1, MINW_FLUX = 2;
  static constexpr int minw_grad(int, int, int) { return MINW_GRAD; }
  static constexpr bool LES = false;  // sub-grid scale models / viscous sponge: dry air, planar and 3-D
  static constexpr bool HAS_MIXED_OUT = false;  // mixed-out sponge target: dry air, planar / 3-D
  static constexpr bool LTE = LTE_;
  typedef std::conditional_t<LTE_, LteParams, DryAirParams> Params;
  // The table gas reads its parameter block -- eight table records, 70-180 spilled SGPRs when it travelled by value in the
  // kernel-argument segment (round 3) -- from a device image through the CONSTANT address space, like the plasma kernels
  // (PlasmaPhys::KArg); dry air keeps the by-value block.
  typedef std::conditional_t<LTE_, const Params *, Params> KArg;
  typedef std::conditional_t<LTE_, const Params __attribute__((address_space(4))) &, const Params &> PRef;
  typedef std::conditional_t<LTE_, const BcDev __attribute__((address_space(4))) &, const BcDev &> BcRef;
  __device__ static inline PRef pref(const KArg &k) {
    if constexpr (LTE_)
      return *(const Params __attribute__((address_space(4))) *)k;
    else
      return k;
  }
  __device__ static inline PRef relaunder(PRef p) {
    if constexpr (LTE_) {  // as PlasmaPhys::relaunder: loads through the result cannot be hoisted above this point
      const unsigned long long a = reinterpret_cast<unsigned long long>(&p);
      unsigned lo = __builtin_amdgcn_readfirstlane(static_cast<unsigned>(a));
      unsigned hi = __builtin_amdgcn_readfirstlane(static_cast<unsigned>(a >> 32));
      asm volatile("" : "+s"(lo), "+s"(hi));
      return *reinterpret_cast<const Params __attribute__((address_space(4))) *>((static_cast<unsigned long long>(hi) << 32) | lo);
    } else {
      return p;
    }
  }
  struct State {
    double ir, k, p;
    double vel[NVEL];
    double T;  // table gas only
    int it;    // ... and the interval of T in the grid of the thermodynamic tables
  };
  __device__ static inline State make_state(PRef p, const double *U) {
    State s;
    s.ir = fast_rcp(U[0]);
    double m2 = 0.0;
#pragma unroll
    for (int d = 0; d < NVEL; d++) {
      m2 += U[1 + d] * U[1 + d];
      s.vel[d] = U[1 + d] * s.ir;
    }
    s.k = m2 * s.ir;
    if constexpr (LTE_) {  // LteMixture::ComputePressure, src/lte_mixture.cpp:119-131
      s.T = lte_temperature(p, (U[ITH] - 0.5 * s.k) / U[0], s.it);
      s.p = U[0] * lte_thermo_at(p, p.tab_R, s.it, s.T) * s.T;
    } else {
      s.p = (p.gamma - 1.0) * (U[ITH] - 0.5 * s.k);
    }
    return s;
  }
  // temperature of a state; rho e of a density at a pressure (modifyEnergyForPressure) / at a temperature
  // (computeStagnantStateWithTemp); speed of sound
  __device__ static inline double temperature(PRef p, const State &s) {
    if constexpr (LTE_)
      return s.T;
    else
      return s.p * p.inv_Rg * s.ir;
  }
  __device__ static inline double rho_e_at_pressure(PRef p, double rho, double pres) {
    if constexpr (LTE_)
    {
      int it;  // (the interval is R's: e shares it when the tables share their grid)
      const double T = lte_temperature_rho_p(p, rho, pres, it);
      return rho * (p.thermo_same_grid ? table_at(p.tab_e, it, T) : table_eval(p.tab_e, T));  // src/lte_mixture.cpp:448-467
    }
    else
      return pres / (p.gamma - 1.0);
  }
  __device__ static inline double rho_e_at_temperature(PRef p, double rho, double T) {
    if constexpr (LTE_)
      return rho * table_eval(p.tab_e, T);  // src/lte_mixture.cpp:424-441
    else
      return p.Rg / (p.gamma - 1.0) * rho * T;
  }
  __device__ static inline double sound(PRef p, const State &s) {
    if constexpr (LTE_)
      return lte_thermo_at(p, p.tab_c, s.it, s.T);  // src/lte_mixture.cpp:357-372
    else
      return fast_sqrt(p.gamma * s.p * s.ir);
  }
  // computeStagnationState: DryAir's (src/equation_of_state.cpp:367-378) rebuilds rho e from the pressure; the table gas
  // inherits GasMixture's (:100-113), total minus bulk kinetic energy
  __device__ static inline double stagnation_energy(PRef p, const double *U, const State &s) {
    if constexpr (LTE_)
      return U[ITH] - 0.5 * s.k;
    else
      return s.p / (p.gamma - 1.0);
  }
  __device__ static inline void prim(PRef p, const double *U, double *Up) {
    const State s = make_state(p, U);
    Up[0] = U[0];
#pragma unroll
    for (int d = 0; d < NVEL; d++) Up[1 + d] = s.vel[d];
    Up[ITH] = temperature(p, s);
  }
  __device__ static inline void clamp_species(double *) {}
  __device__ static inline double max_char_speed(PRef p, const double *, const State &s) {
    return fast_sqrt(s.k * s.ir) + sound(p, s);
  }
  __device__ static inline double max_char_speed(PRef p, const double *U) {
    return max_char_speed(p, U, make_state(p, U));
  }
  __device__ static inline double pressure(PRef p, const double *U) { return make_state(p, U).p; }
  __device__ static inline double sound_speed(PRef p, const double *U) {  // src/equation_of_state.cpp:337-348
    const State s = make_state(p, U);
    if constexpr (LTE_)
      return sound(p, s);
    else
      return sqrt(p.gamma * s.p * s.ir);
  }
  __device__ static inline void conv_flux_n(const double *U, const State &s, const double *n, double *Fn) {
    const double un = s.vel[0] * n[0] + s.vel[1] * n[1];
    Fn[0] = U[0] * un;
    Fn[1] = U[1] * un + s.p * n[0];
    Fn[2] = U[2] * un + s.p * n[1];
    Fn[3] = U[3] * un;
    Fn[ITH] = un * (U[ITH] + s.p);
  }
  __device__ static inline void lax_friedrichs(PRef p, const double *U1, const double *U2, const double *n,
                                               double *F) {
    const State s1 = make_state(p, U1), s2 = make_state(p, U2);
    const double lam = fmax(max_char_speed(p, U1, s1), max_char_speed(p, U2, s2));
    double f1[NEQ], f2[NEQ];
    conv_flux_n(U1, s1, n, f1);
    conv_flux_n(U2, s2, n, f2);
    const double hl = 0.5 * lam * fast_sqrt(n[0] * n[0] + n[1] * n[1]);
#pragma unroll
    for (int eq = 0; eq < NEQ; eq++) F[eq] = 0.5 * (f1[eq] + f2[eq]) - hl * (U2[eq] - U1[eq]);
  }
  // RiemannSolverTPS::Eval: Lax-Friedrichs only here (Eval_Roe is 2-D single-species, not axisymmetric)
  __device__ static inline void riemann(PRef p, const double *U1, const double *U2, const double *n, double *F) {
    lax_friedrichs(p, U1, U2, n, F);
  }
  __device__ static inline void riemann_bc(PRef p, BcRef, const double *U1, const double *Ug,
                                           const double *n, double *F) {
    lax_friedrichs(p, U1, Ug, n, F);
  }
  // Sutherland viscosity, bulk viscosity and conductivity at the temperature of a conserved state
  __device__ static inline void transport(PRef p, const State &s, double &visc, double &bulk, double &k) {
    if constexpr (LTE_) {  // LteTransport::ComputeFluxMolecularTransport, src/lte_transport_properties.cpp:84-107
      const int im = table_interval(p.tab_mu, s.T);
      visc = table_at(p.tab_mu, im, s.T);
      bulk = 0.0;
      k = p.trans_same_grid ? table_at(p.tab_k, im, s.T) : table_eval(p.tab_k, s.T);
    } else {
      const double T = s.p * p.inv_Rg * s.ir;
      visc = p.C1 * p.visc_mult * T * fast_sqrt(T) / (T + p.S0);
      bulk = p.bulk_mult * visc;
      k = p.cp_div_pr * visc;
    }
  }
  // Fv(U, g) . n with the axisymmetric stresses; `zero_heat` drops the conduction term (adiabatic wall)
  __device__ static inline void visc_normal_flux(PRef p, const double *U, const double *g, const double *n,
                                                 double radius, bool zero_heat, double *Fn, const EddyCtx &ec = eddy_off()) {
    const State s = make_state(p, U);
    double visc, bulkv, k;
    transport(p, s, visc, bulkv, k);
    visc_normal_flux_of(s, visc, bulkv, k, U, g, n, radius, zero_heat, Fn, ec);
  }
  // ... of a state whose closure (for the table gas: a Newton inversion of e(T) and the transport tables) is known
  __device__ static inline void visc_normal_flux_of(const State &s, double visc, double bulkv, double k, const double *U,
                                                    const double *g, const double *n, double radius, bool zero_heat, double *Fn,
                                                    const EddyCtx &ec = eddy_off()) {
    add_mixing_length<DIM, NVEL, NEQ>(ec, U, g, radius, visc, bulkv, k);
    double bulk = bulkv - 2. / 3. * visc;
    const double vsw = sponge_weight(ec);  // viscous sponge, src/fluxes.cpp:232-238 (after the -2/3 mu of the bulk viscosity)
    visc *= vsw;
    bulk *= vsw;
    k *= vsw;
    double divV = g[1 + 0 * NEQ] + g[2 + 1 * NEQ];
    if (radius > 0) divV += s.vel[0] / radius;
    double e = 0.0;
    Fn[0] = 0.0;
#pragma unroll
    for (int i = 0; i < DIM; i++) {
      double sn = 0.0;
#pragma unroll
      for (int j = 0; j < DIM; j++) {
        double st = visc * (g[(1 + j) + i * NEQ] + g[(1 + i) + j * NEQ]);
        if (i == j) st += bulk * divV;
        sn += st * n[j];
      }
      Fn[1 + i] = sn;
      e += sn * s.vel[i];
    }
    double ttr = g[3 + 0 * NEQ];
    if (radius > 0) ttr -= s.vel[2] / radius;
    const double tn = visc * (ttr * n[0] + g[3 + 1 * NEQ] * n[1]);
    Fn[3] = tn;
    e += tn * s.vel[2];
    if (!zero_heat) e += k * (g[ITH + 0 * NEQ] * n[0] + g[ITH + 1 * NEQ] * n[1]);
    Fn[ITH] = e;
  }
  __device__ static inline void total_flux(PRef p, const double *U, const State &s, const double *g,
                                           double radius, double *F, const EddyCtx &ec = eddy_off()) {
    const double H = U[ITH] + s.p;
#pragma unroll
    for (int d = 0; d < DIM; d++) {
      F[0 + d * NEQ] = U[1 + d];
#pragma unroll
      for (int i = 0; i < NVEL; i++) F[1 + i + d * NEQ] = U[1 + i] * s.vel[d] + (i == d ? s.p : 0.0);
      F[ITH + d * NEQ] = s.vel[d] * H;
    }
    if (p.eq_system == TPSRHS_EULER) return;
    // F_c - F_v: the viscous flux along each coordinate direction is its normal flux with n = e_d; the state's closure
    // (`s`: the caller's) and the transport coefficients are evaluated ONCE, not per direction (round 4)
    double visc, bulkv, k;
    transport(p, s, visc, bulkv, k);
#pragma unroll
    for (int d = 0; d < DIM; d++) {
      const double nd[DIM] = {d == 0 ? 1.0 : 0.0, d == 1 ? 1.0 : 0.0};
      double fv[NEQ];
      visc_normal_flux_of(s, visc, bulkv, k, U, g, nd, radius, false, fv, ec);
#pragma unroll
      for (int eq = 0; eq < NEQ; eq++) F[eq + d * NEQ] -= fv[eq];
    }
  }
  __device__ static inline void bc_ghost(PRef p, BcRef bc, const double *U, const double *n,
                                         double *Ug, const double * = nullptr) {
#pragma unroll
    for (int eq = 0; eq < NEQ; eq++) Ug[eq] = U[eq];
    const State s = make_state(p, U);
    if (bc.category == TPSRHS_INLET) {  // modifyEnergyForPressure with the interior pressure
      Ug[0] = bc.data[0];
      double ke = 0.0;
#pragma unroll
      for (int d = 0; d < NVEL; d++) {
        Ug[1 + d] = bc.data[0] * bc.data[1 + d];
        ke += 0.5 * Ug[1 + d] * Ug[1 + d] / Ug[0];
      }
      Ug[ITH] = rho_e_at_pressure(p, Ug[0], s.p) + ke;
    } else if (bc.category == TPSRHS_OUTLET) {
      Ug[ITH] = rho_e_at_pressure(p, U[0], bc.data[0]) + 0.5 * s.k;
    } else if (bc.type == TPSRHS_INV || bc.type == TPSRHS_SLIP) {
      const double nm = sqrt(n[0] * n[0] + n[1] * n[1]);
      const double vn = s.vel[0] * (n[0] / nm) + s.vel[1] * (n[1] / nm);
#pragma unroll
      for (int d = 0; d < DIM; d++) Ug[1 + d] = U[0] * (s.vel[d] - 2.0 * vn * (n[d] / nm));
      if (bc.type == TPSRHS_SLIP) slip_ghost_momentum_2d(n, U, Ug);
    } else if (bc.type == TPSRHS_VISC_ADIAB) {  // computeStagnationState, src/equation_of_state.cpp:367-378
#pragma unroll
      for (int d = 0; d < NVEL; d++) Ug[1 + d] = 0.0;
      Ug[ITH] = stagnation_energy(p, U, s);
    } else {  // VISC_ISOTH
      if (p.use_bc_in_grad) {
#pragma unroll
        for (int d = 0; d < NVEL; d++) Ug[1 + d] = -U[1 + d];
      } else {
#pragma unroll
        for (int d = 0; d < NVEL; d++) Ug[1 + d] = 0.0;
        Ug[ITH] = rho_e_at_temperature(p, U[0], bc.data[0]);
      }
    }
  }
  // the viscous trace of one face quadrature point (see PlasmaPhys::visc_trace)
  __device__ static inline void visc_trace(PRef p, int nb, const double *U, const double *g, const double *n,
                                           double radius, double *fn, const EddyCtx &ec = eddy_off()) {
#pragma unroll
    for (int eq = 0; eq < NEQ; eq++) fn[eq] = 0.0;
    if (p.eq_system == TPSRHS_EULER) return;
    if (nb >= 0) {
      visc_normal_flux(p, U, g, n, radius, false, fn, ec);
      return;
    }
    BcRef bc = p.bc[-nb - 1];
    if (bc.category != TPSRHS_WALL || bc.type == TPSRHS_SLIP) return;  // slip wall: Riemann flux only (src/wallBC.cpp:326-428)
    // the wall routines hand the flux class distance 0 (viscous walls, src/wallBC.cpp:441-536) or the interpolated
    // distance (inviscid wall, :309-313)
    EddyCtx wec = ec;
    if (bc.type != TPSRHS_INV) wec.dist = 0.0;
    double Uw[NEQ], f[NEQ];
    bool adiabatic = false;
    if (bc.type == TPSRHS_INV) {
      bc_ghost(p, bc, U, n, Uw);
    } else {
      const State s = make_state(p, U);
#pragma unroll
      for (int eq = 0; eq < NEQ; eq++) Uw[eq] = U[eq];
#pragma unroll
      for (int d = 0; d < NVEL; d++) Uw[1 + d] = 0.0;
      if (bc.type == TPSRHS_VISC_ADIAB) {
        Uw[ITH] = stagnation_energy(p, U, s);
        adiabatic = true;
      } else {
        Uw[ITH] = rho_e_at_temperature(p, U[0], bc.data[0]);
      }
    }
    visc_normal_flux(p, Uw, g, n, radius, adiabatic, f, wec);
#pragma unroll
    for (int eq = 1; eq < NEQ; eq++) fn[eq] = -0.5 * f[eq];
    visc_normal_flux(p, U, g, n, radius, false, f, wec);
#pragma unroll
    for (int eq = 1; eq < NEQ; eq++) fn[eq] -= 0.5 * f[eq];
  }
  __device__ static inline void bc_grad_prim(PRef p, BcRef bc, const double *Up, double *UpB) {
#pragma unroll
    for (int eq = 0; eq < NEQ; eq++) UpB[eq] = Up[eq];
    if (p.use_bc_in_grad && bc.category == TPSRHS_WALL && bc.type == TPSRHS_VISC_ISOTH) {
#pragma unroll
      for (int d = 0; d < NVEL; d++) UpB[1 + d] = 0.0;
      UpB[ITH] = bc.data[0];
    }
  }
  // SourceTerm::updateTerms (src/source_term.cpp:62-256) for one species: only the radiation sink is left
  __device__ static inline void source(PRef p, const double *, const double *Up, const double *, double *src) {
#pragma unroll
    for (int eq = 0; eq < NEQ; eq++) src[eq] = 0.0;
    if constexpr (LTE_) {
      if (p.radiation == TPSRHS_NET_EMISSION) src[ITH] += -4.0 * kPi * table_eval(p.tab_nec, Up[ITH]);  // src/radiation.hpp:68
    }
  }
  // SrcTrns::ELECTRIC_CONDUCTIVITY: LteTransport::ComputeSourceMolecularTransport, src/lte_transport_properties.cpp:109-126
  __device__ static inline double electric_conductivity(PRef p, const double *U) {
    if constexpr (LTE_) {
      const double sigma = table_eval(p.tab_sigma, make_state(p, U).T);
      return sigma < 1.0 ? 1.0 : sigma;
    } else {
      return 0.0;
    }
  }
  __device__ static inline void axisym_source(PRef p, const double *U, const double *Up, const double *g,
                                              double radius, double *src) {
    const double rho = Up[0], ur = Up[1], ut = Up[3];
    double pres;
    if constexpr (LTE_)
      pres = Up[0] * table_eval(p.tab_R, Up[ITH]) * Up[ITH];  // LteMixture::ComputePressureFromPrimitives, src/lte_mixture.cpp:138-147
    else
      pres = p.Rg * Up[0] * Up[ITH];  // DryAir::ComputePressureFromPrimitives
    double tau_tt = 0.0, tau_tr = 0.0;
    if (p.eq_system != TPSRHS_EULER) {
      double visc, bulkv, k;
      if constexpr (LTE_) {  // LteTransport::GetViscosities, src/lte_transport_properties.cpp:128-140: at the primitive T
        visc = table_eval(p.tab_mu, Up[ITH]);
        bulkv = 0.0;
      } else {
        transport(p, make_state(p, U), visc, bulkv, k);  // GetViscosities, src/transport_properties.cpp:268-276
      }
      const double bulk = bulkv - 2. / 3. * visc;
      double divV = g[1 + 0 * NEQ] + g[2 + 1 * NEQ];
      if (radius > 0) divV += ur / radius;
      tau_tt = (radius > 0) ? 2.0 * ur / radius * visc : 0.0;
      tau_tt += bulk * divV;
      tau_tr = g[3 + 0 * NEQ];
      if (radius > 0) tau_tr -= ut / radius;
      tau_tr *= visc;
    }
    src[1] += (pres + rho * ut * ut - tau_tt) / radius;
    src[3] += (-rho * ur * ut + tau_tr) / radius;
  }
};
typedef GasAxiPhys<false> DryAirAxiPhys;
typedef GasAxiPhys<true> LteAxiPhys;

}  // namespace tpsrhs
#endif
