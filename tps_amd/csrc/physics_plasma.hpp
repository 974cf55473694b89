// Device point-wise physics, multi-species plasma (WorkingFluid::USER_DEFINED, PerfectMixture).
//
// Flattened restatement of the reference's virtual hierarchy; compile-time policy = (NVEL, NSP,
// ambipolar, two-temperature, transport model), so that every loop is unrolled and the species
// arrays live in registers.  Number densities, temperatures and pressure of a state are computed
// ONCE per point and shared by the Riemann solver, the fluxes, the transport and the sources (the
// reference recomputes them in every call).
//   PerfectMixture        src/equation_of_state.cpp:478-1942
//   ConstantTransport     src/transport_properties.cpp:303-449
//   GasMinimalTransport   src/gas_transport.cpp:43-830 (argon ternary), collision fits
//                         src/collision_integrals.cpp:53-201
//   Fluxes                src/fluxes.cpp:135-505 ; RiemannSolverTPS src/riemann_solver.cpp:53-115
//   Chemistry / Reaction  src/chemistry.cpp:161-299, src/reaction.cpp:41-83, src/table.cpp:52-110
//   SourceTerm            src/source_term.cpp:107-251 ; NetEmission src/radiation.hpp:68
//   boundary ghosts       src/inletBC.cpp:729-757, src/outletBC.cpp:731-737, src/wallBC.cpp:277-510
#ifndef TPSRHS_PHYSICS_PLASMA_HPP_
#define TPSRHS_PHYSICS_PLASMA_HPP_

#include <hip/hip_runtime.h>

#include "physics_dryair.hpp"

namespace tpsrhs {

// x/y as x * (1/y) with the refined hardware reciprocal in the point physics below (<= 2 ulp; the IEEE
// division sequence costs ~10 FP64 instructions, and the transport closures divide ~90 times per point)
#pragma clang fp reciprocal(on)

constexpr double kRgas = 8.3144598;  // src/equation_of_state.hpp:55-67
constexpr double kAvogadro = 6.0221409e+23;
constexpr double kBoltz = kRgas / kAvogadro;
constexpr double kEps0 = 8.8541878128e-12;
constexpr double kQe = 1.60218e-19;
constexpr double kMolarQe = kQe * kAvogadro;
constexpr double kPi = 3.14159265358979323846;
constexpr double kXeps = 1.0e-30;

struct TableDev {  // LinearTable with the interval coefficients precomputed on the host
  int n, x_log, f_log, pad;
  const double *x, *a, *b;
  double x0, inv_dx;  // uniformly spaced abscissae (the reference's rate tables are): 1 / spacing, else 0
};
struct ChemDev {  // ChemistryInput, device image (lives in a device buffer; uniform reads)
  int num_reactions, electron_index;
  double min_temperature;
  double energy[TPSRHS_MAXREACTIONS];
  double rate[TPSRHS_MAXCHEMPARAMS * TPSRHS_MAXREACTIONS];
  double keq[TPSRHS_MAXCHEMPARAMS * TPSRHS_MAXREACTIONS];
  signed char reactant[TPSRHS_MAXSPECIES * TPSRHS_MAXREACTIONS];
  signed char product[TPSRHS_MAXSPECIES * TPSRHS_MAXREACTIONS];
  signed char model[TPSRHS_MAXREACTIONS];
  signed char detailed_balance[TPSRHS_MAXREACTIONS];
  TableDev table[TPSRHS_MAXREACTIONS];
  // table_share[r] = the first reaction whose rate table has the same abscissae as reaction r's (r itself if none): the
  // interval found for one of them at a temperature serves the others at the same temperature (the 14 rate tables of the
  // reference's test/inputs/rate-coefficients share one 500-point grid; the host compares the abscissae value by value)
  signed char table_share[TPSRHS_MAXREACTIONS];
  int radiation;  // tpsrhs_radiation_model
  TableDev nec;
};

constexpr int PLASMA_MAXBC = 8;
template <int NSP>
struct PlasmaParams {
  double mw[NSP], charge[NSP], eform[NSP], cv[NSP], cp[NSP];
  // derived on the host (fill_plasma_params): reciprocal molar masses; per-particle masses m = mw / N_A, their
  // roots, k_f / m; sqrt(reduced mass of a pair) / d_fc of the binary diffusivities (src/gas_transport.cpp:291-310)
  double imw[NSP], mwp[NSP], sq_mwp[NSP], kf_imwp[NSP], sq_muw_idfc[NSP * NSP];
  // every other product of constants the closures use (a uniform FP64 expression left in a kernel is computed by
  // the vector ALU, hoisted out of the point loops and then kept in -- or spilled from -- a VGPR pair):
  //   v_f sqrt(m_sp); v_f k_f / sqrt(m_e) and sqrt(2) times that; (q_e / k_B) Z_sp; R / mw_sp; 1 / cv_e
  double vf_sq_mwp[NSP], ke_fac, ke_fac3, qkb_charge[NSP], rg_imw[NSP], icv_e;
  // constant transport
  double c_visc, c_bulk, c_k, c_ke, c_diff[NSP], c_mtfreq[NSP];
  int c_eidx;
  // argon ternary
  int third_order, multiply;
  int coll[NSP * NSP];  // tpsrhs_gas_coll of the species pairs [i + j*NSP], i <= j (argon mixture transport)
  // which collision types occur (bit = 1 << tpsrhs_gas_coll): among the (sp, sp) pairs of the heavy species, among the
  // (heavy, electron) pairs, among the pairs of two different heavy species -- filled by the host
  int cmask_diag, cmask_e, cmask_h;
  double mult_flux[4], mult_spcs, mult_diff, mult_mobil;
  const ChemDev *chem;
  int eq_system, use_bc_in_grad, num_bcs, axisymmetric;
  BcDev bc[PLASMA_MAXBC];
};

__device__ inline double ipow(double x, int k) {  // pow(x, small non-negative integer), pow(0,0) = 1
  double r = 1.0;
  for (int i = 0; i < k; i++) r *= x;
  return r;
}
// LinearTable::findInterval (src/table.cpp:52-78): the interval idx with x[idx] < xe <= x[idx+1], clamped to the
// table (lower_bound, then first - 1).  Bisection as in the reference, or -- on a uniformly spaced table -- the
// interval computed from the spacing and corrected / verified against the abscissae, which selects the same
// interval with 2-3 dependent loads instead of 9-10.  (TD: a TableDev in the generic or in the CONSTANT address space --
// the records of the table gas live in its parameter image.)
template <class TD>
__device__ inline int table_interval(const TD &t, double xe) {
  if (t.inv_dx > 0.0) {
    int g = static_cast<int>((xe - t.x0) * t.inv_dx);
    g = max(0, min(t.n - 2, g));
    if (g > 0 && !(xe > t.x[g])) g--;
    if (g < t.n - 2 && xe > t.x[g + 1]) g++;
    if ((g == 0 || xe > t.x[g]) && (g == t.n - 2 || !(xe > t.x[g + 1]))) return g;
  }
  int count = t.n, first = 0;
  while (count > 0) {
    int it = first;
    const int step = count / 2;
    it += step;
    if (xe > t.x[it]) {
      first = ++it;
      count -= step + 1;
    } else {
      count = step;
    }
  }
  first = max(1, min(t.n - 1, first));
  return first - 1;
}
template <class TD>
__device__ inline double table_eval(const TD &t, double xe) {  // src/table.cpp:80-101
  const int idx = table_interval(t, xe);
  const double xt = t.x_log ? flog(xe) : xe;
  double ft = t.a[idx] + t.b[idx] * xt;
  if (t.f_log) ft = fexp(ft);
  return ft;
}

namespace coll {  // collision-integral fits, src/collision_integrals.cpp
// c0 log(1 + c1 Tp^c2)^c3 / Tp^2 with the powers taken through exp/log of the argument's logarithm,
// which the fits of one point share (the reference calls pow twice per fit; the results agree to a few
// ulp, far inside the stated tolerance), and exp / log from fastmath.hpp: 2 x (23 + 33) FP64 instructions
// per fit where two pow() calls of the device library are 450
struct Arg {
  double ln, inv2;  // log(Tp), 1/Tp^2
};
__device__ inline Arg arg(double Tp) {
  Arg a;
  a.ln = flog(Tp);
  const double r = fast_rcp(Tp);
  a.inv2 = r * r;
  return a;
}
__device__ inline double cfit(double c0, double c1, double c2, double c3, const Arg &a) {
  // 1 + c1 Tp^c2 >= 1: the inner logarithm needs no special cases (a NaN passes through); its value L is > 0
  // unless Tp^c2 underflowed, where the fit is 0 (c3 > 0) -- so the outer logarithm needs none either.  The two
  // exponents are small multiples of logarithms of finite positive numbers: the unchecked exponential.
  const double L = flog_pos(1.0 + c1 * fexp<false>(c2 * a.ln));
  const double v = c0 * fexp<false>(c3 * flog_pos(L)) * a.inv2;
  return (L > 0.0) ? v : ((L == 0.0) ? 0.0 : v);
}
// N fits of one argument in lock step (fexp_n / flog_pos_n): the same operations per fit as cfit, interleaved
template <int N>
__device__ inline void cfit_n(const double (&c)[N][4], const Arg &a, double (&out)[N]) {
  double t[N], E[N], w[N], L[N], lL[N], u[N], P[N];
#pragma unroll
  for (int i = 0; i < N; i++) t[i] = c[i][2] * a.ln;
  fexp_n<N>(t, E);
#pragma unroll
  for (int i = 0; i < N; i++) w[i] = 1.0 + c[i][1] * E[i];
  flog_pos_n<N>(w, L);
  flog_pos_n<N>(L, lL);
#pragma unroll
  for (int i = 0; i < N; i++) u[i] = c[i][3] * lL[i];
  fexp_n<N>(u, P);
#pragma unroll
  for (int i = 0; i < N; i++) {
    const double v = c[i][0] * P[i] * a.inv2;
    out[i] = (L[i] > 0.0) ? v : ((L[i] == 0.0) ? 0.0 : v);
  }
}
__device__ inline double att11(const Arg &a) { return cfit(0.2150, 5.2194, 1.0472, 1.2435, a); }
__device__ inline double att12(const Arg &a) { return cfit(0.0991, 7.4684, 1.0155, 1.1536, a); }
__device__ inline double att13(const Arg &a) { return cfit(0.0616, 7.8271, 0.9452, 1.1105, a); }
__device__ inline double att14(const Arg &a) { return cfit(0.0308, 13.9567, 0.9511, 1.1803, a); }
__device__ inline double att15(const Arg &a) { return cfit(0.0232, 13.7888, 0.9148, 1.1532, a); }
__device__ inline double rep22(const Arg &a) { return cfit(0.4128, 1.2436, 1.1830, 1.0123, a); }
__device__ inline double rep23(const Arg &a) { return cfit(0.2203, 1.8832, 1.2059, 0.9851, a); }
__device__ inline double rep24(const Arg &a) { return cfit(0.1323, 2.7248, 1.2129, 0.9847, a); }
// Coulomb fits by (attractive 0 / repulsive 1, l, r): c0, c1, c2, c3 (src/collision_integrals.cpp:53-115);
// rows the reference does not have are zero
__constant__ static double c_coulomb[2][2][5][4] = {
    {{{0.2150, 5.2194, 1.0472, 1.2435}, {0.0991, 7.4684, 1.0155, 1.1536}, {0.0616, 7.8271, 0.9452, 1.1105},
      {0.0308, 13.9567, 0.9511, 1.1803}, {0.0232, 13.7888, 0.9148, 1.1532}},
     {{0, 0, 0, 0}, {0.2423, 4.6796, 1.3290, 1.1279}, {0.1221, 8.7542, 1.3875, 1.1110}, {0.0619, 18.2538, 1.4341, 1.1618},
      {0, 0, 0, 0}}},
    {{{0.3904, 0.9100, 1.1025, 1.0544}, {0.1547, 1.6597, 1.1725, 0.9792}, {0.0814, 2.5815, 1.1948, 0.9570},
      {0.0683, 1.9774, 1.2033, 0.8264}, {0.0346, 4.5177, 1.2132, 0.9294}},
     {{0, 0, 0, 0}, {0.4128, 1.2436, 1.1830, 1.0123}, {0.2203, 1.8832, 1.2059, 0.9851}, {0.1323, 2.7248, 1.2129, 0.9847},
      {0, 0, 0, 0}}}};
__device__ inline double ArAr11(double lnT) { return 2.2910e-18 * fexp(-0.3032 * lnT); }
__device__ inline double ArAr22(double T) { return 1.7e-18 * fast_rsqrt(fast_sqrt(T)); }  // T^-0.25
__device__ inline double iArAr22(double sqrtT) { return fast_sqrt(sqrtT) * (1.0 / 1.7e-18); }  // 1 / ArAr22, from sqrt(T)
__device__ inline double ArAr1P11(double lnT) { return 4.574321e-18 * fexp(-0.1805 * lnT); }
__device__ inline double eAr1r(int r, double logT) {
  const double C[5][9] = {
      {6.36254140e-18, 1.84835040e-18, -5.87727093e-18, 3.20023027e-18, -8.50509054e-19, 1.28163820e-19,
       -1.11712910e-20, 5.25649382e-22, -1.03296658e-23},
      {1.91338172e-17, 5.45418129e-18, -1.78361685e-17, 9.75657946e-18, -2.61115722e-18, 3.98310268e-19,
       -3.53503678e-20, 1.70375066e-21, -3.45211955e-23},
      {3.04685398e-17, 8.39750994e-18, -2.88132528e-17, 1.60147037e-17, -4.34837891e-18, 6.73136845e-19,
       -6.06704580e-20, 2.97216168e-21, -6.12760944e-23},
      {3.90777949e-17, 1.04696956e-17, -3.73774204e-17, 2.10610498e-17, -5.79029566e-18, 9.07573157e-19,
       -8.28466766e-20, 4.11188110e-21, -8.59225098e-23},
      {4.41333290e-17, 1.15696010e-17, -4.25651305e-17, 2.42442440e-17, -6.73359258e-18, 1.06641697e-18,
       -9.83933863e-20, 4.93775812e-21, -1.04362372e-22}};
  // summed in the reference's order (the terms cancel to 1e-3 of their size: the order is part of the result)
  double fit = C[r - 1][0] * fast_rcp(logT), pw = 1.0;
#pragma unroll
  for (int k = 1; k < 9; k++) {
    fit += C[r - 1][k] * pw;
    pw *= logT;
  }
  return fit;
}
// eAr1r(r, logT) for r = 1 .. NR in lock step: per r the reference's summation order, the NR sums interleaved
template <int NR>
__device__ inline void eAr1r_n(double logT, double (&fit)[NR]) {
  static_assert(NR <= 5, "five fits");
  constexpr double C[5][9] = {
      {6.36254140e-18, 1.84835040e-18, -5.87727093e-18, 3.20023027e-18, -8.50509054e-19, 1.28163820e-19,
       -1.11712910e-20, 5.25649382e-22, -1.03296658e-23},
      {1.91338172e-17, 5.45418129e-18, -1.78361685e-17, 9.75657946e-18, -2.61115722e-18, 3.98310268e-19,
       -3.53503678e-20, 1.70375066e-21, -3.45211955e-23},
      {3.04685398e-17, 8.39750994e-18, -2.88132528e-17, 1.60147037e-17, -4.34837891e-18, 6.73136845e-19,
       -6.06704580e-20, 2.97216168e-21, -6.12760944e-23},
      {3.90777949e-17, 1.04696956e-17, -3.73774204e-17, 2.10610498e-17, -5.79029566e-18, 9.07573157e-19,
       -8.28466766e-20, 4.11188110e-21, -8.59225098e-23},
      {4.41333290e-17, 1.15696010e-17, -4.25651305e-17, 2.42442440e-17, -6.73359258e-18, 1.06641697e-18,
       -9.83933863e-20, 4.93775812e-21, -1.04362372e-22}};
  const double il = fast_rcp(logT);
#pragma unroll
  for (int r = 0; r < NR; r++) fit[r] = C[r][0] * il;
  double pw = 1.0;
#pragma unroll
  for (int k = 1; k < 9; k++) {
#pragma unroll
    for (int r = 0; r < NR; r++) fit[r] += C[r][k] * pw;
    pw *= logT;
  }
}
}  // namespace coll

#ifndef TPSRHS_PLASMA_MINW_FLUX
#define TPSRHS_PLASMA_MINW_FLUX 2
#endif
#ifndef TPSRHS_PLASMA_MINW_GRAD3
#define TPSRHS_PLASMA_MINW_GRAD3 2
#endif
#ifndef TPSRHS_PLASMA_MINW_GRAD_LEAN
#define TPSRHS_PLASMA_MINW_GRAD_LEAN 3
#endif
enum { TRANSPORT_CONSTANT = 0, TRANSPORT_ARGON_MINIMAL = 1, TRANSPORT_ARGON_MIXTURE = 2 };

template <int DIM_, int NVEL_, int NSP_, bool AMBI, bool TWOT, int TRANSPORT>
struct PlasmaPhys {
  static constexpr int DIM = DIM_, NVEL = NVEL_, NSP = NSP_;
  static constexpr int NACTIVE = AMBI ? NSP_ - 2 : NSP_ - 1;
  static constexpr int NEQ = NVEL_ + 2 + NACTIVE + (TWOT ? 1 : 0);
  static constexpr int IE = NSP_ - 2, IB = NSP_ - 1;  // electron, background
  static constexpr int ITH = NVEL_ + 1, ITE = NEQ - 1;
  static constexpr bool HAS_SOURCE = true;
  static constexpr bool TWO_TEMPERATURE = TWOT;
  static constexpr int MAX_ORDER = 4;
  static constexpr bool HAS_NR_BC = false;
  static constexpr bool LES = false;  // sub-grid scale models / viscous sponge: dry air, planar and 3-D  // the reference's non-reflecting conditions are perfect-gas algebra
  static constexpr bool HAS_MIXED_OUT = false;  // mixed-out sponge target: dry air, planar / 3-D
  static constexpr bool VISC_USES_GRAD_RHO = true;  // mole-fraction gradients need grad(rho)
  static constexpr bool AXISYM = NVEL_ > DIM_;  // dim 2 with (r, z, theta) velocity components
  // waves per SIMD asked of the allocator.  More than three species (9-13 equations): one wave per SIMD, 512
  // registers -- at two the sweeps spilled 150-340 VGPRs to scratch, and two such instantiations returned
  // scheduling-dependent wrong results (DESIGN.md "spilled instantiations")
  static constexpr int MINW_GRAD = (NSP_ > 3) ? 1 : TPSRHS_PLASMA_MINW_GRAD3;
  // The 3-D face kernel of the ternary mixtures carries the closure of a face point in its LEAN form (ViscLean below):
  // 12-18 values across the gradient interpolation instead of 38, the nodal gradient re-read from the L2 instead of held
  // in LDS -- 168 registers and 12 KB, THREE waves per SIMD (round 4; the collocated p <= 3 hexes, one wave per block).
  static constexpr bool LEAN_TRACE = (NSP_ == 3) && (DIM_ == 3);
  static constexpr bool LAUNDER_FLUX = false;  // k_flux re-fetches the parameter image where its face term starts (the table gas)
  static constexpr int minw_grad(int dim, int p, int nc) {
    return (LEAN_TRACE && dim == 3 && !nc && p <= 3) ? TPSRHS_PLASMA_MINW_GRAD_LEAN : MINW_GRAD;
  }
  // (2-D / axisymmetric sweeps of four to six species stay at two: 20-200 spilled VGPRs, the state every round-1 and
  // round-2 sweep of the randomised parity driver ran in, and 20 % faster at torch6)
  static constexpr int MINW_FLUX = (NSP_ > 6 || (NSP_ > 3 && DIM_ == 3)) ? 1 : TPSRHS_PLASMA_MINW_FLUX;
  typedef PlasmaParams<NSP_> Params;
  // How the kernels see the parameter block.  It lives in a device buffer and is read through the CONSTANT address
  // space: every access is a scalar load (s_load) of a wave-uniform address.  Passed by value in the kernel-argument
  // segment (round 1-2) the same loads were hoisted to the top of the kernel by the loop-invariant code motion -- some
  // 70 SGPRs of masses, charges, reduced-mass factors ... live across every loop -- and came back lane by lane from
  // spill VGPRs: 972 v_readlane / v_writelane of 4 581 VALU instructions in the reacting k_gradient.  `relaunder`
  // passes the pointer through an empty asm statement: loads through the result cannot be moved above it, so a
  // closure placed after it loads what it needs where it needs it, with SMEM instructions that cost no VALU issue.
  typedef const Params *KArg;
  typedef const Params __attribute__((address_space(4))) &PRef;
  typedef const BcDev __attribute__((address_space(4))) &BcRef;
  __device__ static inline PRef pref(KArg k) { return *(const Params __attribute__((address_space(4))) *)k; }
  __device__ static inline PRef relaunder(PRef p) {
    // (inside divergent control flow the compiler may hold the -- uniform -- address in VGPRs: readfirstlane brings
    // it back to scalar registers, and folds away when it already is there)
    const unsigned long long a = reinterpret_cast<unsigned long long>(&p);
    unsigned lo = __builtin_amdgcn_readfirstlane(static_cast<unsigned>(a));
    unsigned hi = __builtin_amdgcn_readfirstlane(static_cast<unsigned>(a >> 32));
    asm volatile("" : "+s"(lo), "+s"(hi));
    return *reinterpret_cast<const Params __attribute__((address_space(4))) *>((static_cast<unsigned long long>(hi) << 32) | lo);
  }
  struct Transport {};

  // everything the closures need from one conserved state
  struct State {
    double ir, k, p, pe, Th, Te, c;  // 1/rho, |rho u|^2/rho, pressures, temperatures, sound speed
    double vel[NVEL];
    double n[NSP];
  };

  __device__ static inline void number_densities(PRef p, const double *U, double *n) {  // :947-961
#pragma unroll
    for (int sp = 0; sp < NSP; sp++) n[sp] = 0.0;
#pragma unroll
    for (int sp = 0; sp < NACTIVE; sp++) n[sp] = U[NVEL + 2 + sp] * p.imw[sp];
    double rhoB = U[0];
#pragma unroll
    for (int sp = 0; sp < NACTIVE; sp++) rhoB -= p.mw[sp] * n[sp];
    if (AMBI) {
      double ne = 0.0;
#pragma unroll
      for (int sp = 0; sp < NACTIVE; sp++) ne += p.charge[sp] * n[sp];
      ne = fmax(ne, 0.0);
      n[IE] = ne;
      rhoB -= ne * p.mw[IE];
    }
    n[IB] = rhoB * p.imw[IB];
  }
  __device__ static inline double heavies_cv(PRef p, const double *n) {  // :576-584
    double c = 0.0;
#pragma unroll
    for (int sp = 0; sp < NACTIVE; sp++)
      if (sp != IE) c += n[sp] * p.cv[sp];
    return c + n[IB] * p.cv[IB];
  }
  __device__ static inline double heavies_n(const double *n) {
    double nh = 0.0;
#pragma unroll
    for (int sp = 0; sp < NACTIVE; sp++)
      if (sp != IE) nh += n[sp];
    return nh + n[IB];
  }
  __device__ static inline State make_state(PRef p, const double *U) {
    State s;
    number_densities(p, U, s.n);
    s.ir = fast_rcp(U[0]);
    double m2 = 0.0;
#pragma unroll
    for (int d = 0; d < NVEL; d++) {
      m2 += U[1 + d] * U[1 + d];
      s.vel[d] = U[1 + d] * s.ir;
    }
    s.k = m2 * s.ir;
    // computeTemperaturesBase, :1141-1172
    const double chv = heavies_cv(p, s.n);
    double ctot = chv;
    if (!TWOT) ctot += s.n[IE] * p.cv[IE];
    double e = U[ITH];
#pragma unroll
    for (int sp = 0; sp < NSP - 2; sp++) e -= s.n[sp] * p.eform[sp];
    double Th = -0.5 * s.k + e;
    if (TWOT) Th -= U[ITE];
    s.Th = Th * fast_rcp(ctot);
    s.Te = TWOT ? U[ITE] * fast_rcp(s.n[IE] * p.cv[IE]) : s.Th;
    // computePressureBase, :1044-1062
    const double nh = heavies_n(s.n);
    s.pe = s.n[IE] * kRgas * s.Te;
    s.p = kRgas * (nh * s.Th + s.n[IE] * s.Te);
    // speed of sound: heavies' heat ratio, :1311-1340
    const double gamma = 1.0 + nh * kRgas * fast_rcp(chv);
    s.c = fast_sqrt(gamma * s.p * s.ir);
    return s;
  }
  __device__ static inline double pressure(PRef p, const double *U) { return make_state(p, U).p; }

  // GetPrimitivesFromConservatives, :679-700
  __device__ static inline void prim(PRef p, const double *U, double *Up) {
    const State s = make_state(p, U);
    Up[0] = U[0];
#pragma unroll
    for (int d = 0; d < NVEL; d++) Up[1 + d] = s.vel[d];
    Up[ITH] = s.Th;
#pragma unroll
    for (int sp = 0; sp < NACTIVE; sp++) Up[NVEL + 2 + sp] = s.n[sp];
    if (TWOT) Up[ITE] = s.Te;
  }
  // species densities are clamped to >= 0 wherever a state is interpolated (src/face_integrator.cpp:297-302)
  __device__ static inline void clamp_species(double *U) {
#pragma unroll
    for (int sp = 0; sp < NACTIVE; sp++) U[NVEL + 2 + sp] = fmax(U[NVEL + 2 + sp], 0.0);
  }
  __device__ static inline double max_char_speed(PRef, const double *, const State &s) {  // :1359-1373
    return fast_sqrt(s.k * s.ir) + s.c;
  }
  __device__ static inline double max_char_speed(PRef p, const double *U) {
    return max_char_speed(p, U, make_state(p, U));
  }
  __device__ static inline double sound_speed(PRef p, const double *U) {  // :1405-1432
    return make_state(p, U).c;
  }

  // F(U).n, src/fluxes.cpp:135-170
  __device__ static inline void conv_flux_n(PRef p, const double *U, const State &s, const double *n,
                                            double *Fn) {
    double un = 0.0;
#pragma unroll
    for (int d = 0; d < DIM; d++) un += s.vel[d] * n[d];
    Fn[0] = U[0] * un;
#pragma unroll
    for (int i = 0; i < NVEL; i++) Fn[1 + i] = U[1 + i] * un + (i < DIM ? s.p * n[i] : 0.0);
    Fn[ITH] = un * (U[ITH] + s.p);
#pragma unroll
    for (int sp = 0; sp < NACTIVE; sp++) Fn[NVEL + 2 + sp] = U[NVEL + 2 + sp] * un;
    if (TWOT) Fn[ITE] = (U[ITE] + s.pe) * un;
  }
  __device__ static inline void lax_friedrichs(PRef p, const double *U1, const double *U2, const double *n,
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

  // RiemannSolverTPS::Eval: Lax-Friedrichs only here (Eval_Roe is 2-D single-species, not axisymmetric)
  __device__ static inline void riemann(PRef p, const double *U1, const double *U2, const double *n, double *F) {
    lax_friedrichs(p, U1, U2, n, F);
  }
  __device__ static inline void riemann_bc(PRef p, BcRef, const double *U1, const double *Ug,
                                           const double *n, double *F) {
    lax_friedrichs(p, U1, Ug, n, F);
  }
  // ---- species primitives and mole-fraction gradient --------------------------------------
  struct Species {
    double X[NSP], Y[NSP], n[NSP], ntot, intot;
  };
  // computeSpeciesPrimitives (:882-927): its own number densities (electrons not clamped, background
  // from the mass-fraction remainder), kept apart from computeNumberDensities as in the reference
  __device__ static inline Species species(PRef p, const double *U) {
    Species q;
    const double ir = fast_rcp(U[0]);
    double n = 0.0, ne = 0.0, Yb = 1.0;
#pragma unroll
    for (int sp = 0; sp < NSP; sp++) q.n[sp] = 0.0;
#pragma unroll
    for (int sp = 0; sp < NACTIVE; sp++) {
      q.n[sp] = U[NVEL + 2 + sp] * p.imw[sp];
      n += q.n[sp];
      if (AMBI) ne += p.charge[sp] * q.n[sp];
      q.Y[sp] = U[NVEL + 2 + sp] * ir;
      Yb -= q.Y[sp];
    }
    if (AMBI) {
      q.n[IE] = ne;
      n += ne;
      q.Y[IE] = ne * p.mw[IE] * ir;
      Yb -= q.Y[IE];
    }
    q.Y[IB] = Yb;
    q.n[IB] = Yb * U[0] * p.imw[IB];
    n += q.n[IB];
    q.ntot = n;
    q.intot = fast_rcp(n);
#pragma unroll
    for (int sp = 0; sp < NSP; sp++) q.X[sp] = q.n[sp] * q.intot;
    return q;
  }
  // ---- transport ----------------------------------------------------------------------------
  struct Trans {
    double visc, bulk, k, ke;
    double V[NSP * DIM];  // diffusion velocities [sp + d*NSP]
    double n[NSP];        // number densities of computeSpeciesPrimitives
  };
  // The state-only part of ComputeFluxTransportProperties: everything the diffusion velocities need besides
  // the gradient.  The diffusion velocity is the same linear map of grad X in every direction, so the face
  // kernels apply it to the NORMAL derivative alone (diffusion_velocity below).
  struct TCoef {
    double visc, bulk, k, ke;
    double dX[NSP], mob[NSP], imho;        // D_sp / (X_sp + eps), mobilities, 1 / (sum mob n Z + eps)
    double Y[NSP], n[NSP], intot;          // computeSpeciesPrimitives: mass fractions, number densities, 1 / n
  };
  // species positions of the argon ternary mixture are fixed by the mixture ordering (electron
  // second to last, neutral background last); tpsrhs_create checks the tpsrhs_gas_transport indices
  static constexpr int I_E = IE, I_N = IB, I_ION = 0;
  struct Debye {
    double circle;
    coll::Arg e, h;  // nondimensional electron / heavy temperatures
  };
  // iTe, iTh = 1 / T_e, 1 / T_h
  __device__ static inline Debye debye(const double *n, double Th, double Te, double iTh, double iTe) {
    const double dfac = kBoltz * kEps0 / kQe / kQe;
    const double nOverT = (n[I_E] + kXeps) * iTe + (n[I_ION] + kXeps) * iTh;
    const double len2 = (dfac / kAvogadro) * fast_rcp(nOverT);
    const double length = fast_sqrt(len2);
    Debye d;
    d.circle = kPi * len2;
    const double f = length * (4.0 * kPi * dfac);
    d.e = coll::arg(f * Te);
    d.h = TWOT ? coll::arg(f * Th) : d.e;
    return d;
  }
  // sTe = sqrt(T_e), c_ke = v_f k_f sqrt(2 / m_e); Q2 = Q_ee^(2,2..4), QI = Q_ei^(1,1..5), QN = Q_en^(1,1..5)
  __device__ static inline double third_order_ke(const double *X, const double *Q2, const double *QI, const double *QN,
                                                 double sTe, double c_ke) {  // :400-489
    auto L11ea = [](const double *Q) { return 6.25 * Q[0] - 15. * Q[1] + 12. * Q[2]; };
    auto L12ea = [](const double *Q) { return 10.9375 * Q[0] - 39.375 * Q[1] + 57. * Q[2] - 30. * Q[3]; };
    auto L22ea = [](const double *Q) {
      return 19.140625 * Q[0] - 91.875 * Q[1] + 199.5 * Q[2] - 210. * Q[3] + 90. * Q[4];
    };
    constexpr double s2 = 1.4142135623730951;  // sqrt(2)
    // one collision partner at a time (electron, ion, neutral), each group reduced to its three sums
    double L11 = s2 * X[I_E] * Q2[0];
    double L12 = s2 * X[I_E] * (1.75 * Q2[0] - 2.0 * Q2[1]);
    double L22 = s2 * X[I_E] * (4.8125 * Q2[0] - 7.0 * Q2[1] + 5. * Q2[2]);
    L11 += X[I_ION] * L11ea(QI);
    L12 += X[I_ION] * L12ea(QI);
    L22 += X[I_ION] * L22ea(QI);
    L11 += X[I_N] * L11ea(QN);
    L12 += X[I_N] * L12ea(QN);
    L22 += X[I_N] * L22ea(QN);
    return c_ke * sTe * X[I_E] * fast_rcp(L11 - L12 * L12 * fast_rcp(L22));
  }

  // collisionInputs of the mixture transport (src/gas_transport.cpp:185-204): the Debye length sums
  // Z^2 n / T_e over all species
  struct MixColl {
    double circle, Th, lnTe, lnTh;
    coll::Arg e, h;
  };
  __device__ static inline MixColl mix_inputs(PRef p, const double *n, double Th, double Te) {
    const double dfac = kBoltz * kEps0 / kQe / kQe;
    double nOverT = 0.0;
    const double iTe = fast_rcp(Te);
#pragma unroll
    for (int sp = 0; sp < NSP; sp++) nOverT += (n[sp] + kXeps) * iTe * p.charge[sp] * p.charge[sp];
    const double len2 = (dfac / kAvogadro) * fast_rcp(nOverT);
    const double length = fast_sqrt(len2);
    MixColl c;
    c.circle = kPi * len2;
    const double f = length * (4.0 * kPi * dfac);
    c.e = coll::arg(f * Te);
    c.h = TWOT ? coll::arg(f * Th) : c.e;
    c.Th = Th;
    c.lnTe = flog(Te);
    c.lnTh = TWOT ? flog(Th) : c.lnTe;
    return c;
  }
  // GasMixtureTransport::collisionIntegral (src/gas_transport.cpp:995-1283), argon types, by collision TYPE.
  //   CLMB_ATT / CLMB_REP: pi lambda_D^2 x the Coulomb fit (l, r) at the nondimensional temperature of the pair (the
  //   electron's if it takes part); AR_E: the e-Ar polynomial (1, r); AR_AR1P: Ar-Ar+ (1,1); AR_AR: Ar-Ar (1,1) / (2,2).
  // tpsrhs_create has checked that every (pair, l, r) the transport asks for exists.
  // A point has one set of collision inputs, so collisionIntegral(i, j, l, r) takes at most one value per (type, l, r,
  // electron pair or not): the values of the types that occur (`mask`, wave-uniform, from the host) are computed ONCE
  // per point and every species pair selects its own with its uniform type -- the same functions of the same arguments
  // as the reference's pair-by-pair calls.  (Rounds 1-3 inlined a five-way type branch per pair: the e-Ar polynomial
  // once per neutral species, the attractive Coulomb fit once per ion; a 7-species k_flux was 23 000 instructions with
  // 900 branches.)
  struct CollTab {
    double att, rep, ar1p, are, arar;  // by tpsrhs_gas_coll: CLMB_ATT, CLMB_REP, AR_AR1P, AR_E, AR_AR
  };
  template <int L, int R>
  __device__ static inline CollTab coll_table(int mask, bool with_e, const MixColl &c) {
    CollTab t;
    t.att = t.rep = t.ar1p = t.are = t.arar = 0.0;
    const coll::Arg &a = with_e ? c.e : c.h;
    const double ln = with_e ? c.lnTe : c.lnTh;
    if (mask & (1 << TPSRHS_CLMB_ATT)) {
      const double *k = coll::c_coulomb[0][L - 1][R - 1];
      t.att = c.circle * coll::cfit(k[0], k[1], k[2], k[3], a);
    }
    if (mask & (1 << TPSRHS_CLMB_REP)) {
      const double *k = coll::c_coulomb[1][L - 1][R - 1];
      t.rep = c.circle * coll::cfit(k[0], k[1], k[2], k[3], a);
    }
    if (mask & (1 << TPSRHS_AR_AR1P)) t.ar1p = coll::ArAr1P11(ln);
    if (mask & (1 << TPSRHS_AR_E)) t.are = coll::eAr1r(R, ln);
    if (mask & (1 << TPSRHS_AR_AR)) t.arar = (L == 1) ? coll::ArAr11(ln) : coll::ArAr22(c.Th);
    return t;
  }
  // the table entry of the pair (i, j): a chain of selects on the wave-uniform type.  The five entries are read
  // unconditionally first: written as `type == X ? t.x : v` the conditional loads are merged by the optimiser into ONE
  // load with a selected offset, and the table then lives in scratch memory instead of registers.
  __device__ static inline double coll_pick(PRef p, int i, int j, const CollTab &t) {
    const int a = i < j ? i : j, b = i < j ? j : i;
    const int type = p.coll[a + b * NSP];
    const double q_att = t.att, q_rep = t.rep, q_ar1p = t.ar1p, q_are = t.are, q_arar = t.arar;
    double v = q_arar;
    v = (type == TPSRHS_AR_E) ? q_are : v;
    v = (type == TPSRHS_AR_AR1P) ? q_ar1p : v;
    v = (type == TPSRHS_CLMB_REP) ? q_rep : v;
    v = (type == TPSRHS_CLMB_ATT) ? q_att : v;
    return v;
  }

  // ComputeFluxTransportProperties of the selected model; E-field = 0 (src/fluxes.cpp:200-201).
  // `diffusion` = false skips the diffusion velocities (walls prescribe zero species fluxes).
  //
  // Algebra of the collision-integral models, arranged for few FP64 divisions (the reference's order of
  // operations divides ~90 times per point; each division is 10 instructions, a product with a reciprocal 1):
  //   sqrt(m T) = sqrt(m) sqrt(T), sqrt(T / mu) = sqrt(T) / sqrt(mu) with the mass factors from the host;
  //   1 / D_ij = n Q_ij sqrt(mu_ij) / (d_fc sqrt(T)) directly (CurtissHirschfelder only ever divides by D_ij);
  //   one reciprocal each of T_e, T_h, n, X_sp + eps, shared by everything that divides by them.
  // Results differ from the reference's order by rounding (a few ulp).
  __device__ static inline void transport_coeffs(PRef p, const double *U, double Th, double Te, bool diffusion,
                                                 TCoef &t) {
    const Species q = species(p, U);
#pragma unroll
    for (int sp = 0; sp < NSP; sp++) {
      t.n[sp] = q.n[sp];
      t.Y[sp] = q.Y[sp];
      t.dX[sp] = t.mob[sp] = 0.0;
    }
    t.intot = q.intot;
    t.imho = 0.0;
    double diff[NSP] = {}, mob[NSP] = {};  // (the multiplier block below reads them even when `diffusion` is off)
    const double iTe = fast_rcp(Te), iTh = TWOT ? fast_rcp(Th) : iTe;
    if (TRANSPORT == TRANSPORT_CONSTANT) {
      t.visc = p.c_visc;
      t.bulk = p.c_bulk;
      t.k = p.c_k;
      t.ke = p.c_ke;
#pragma unroll
      for (int sp = 0; sp < NSP; sp++) {
        diff[sp] = p.c_diff[sp];
        const double itemp = (sp == p.c_eidx) ? iTe : iTh;
        mob[sp] = p.qkb_charge[sp] * itemp * diff[sp];
      }
    } else if (TRANSPORT == TRANSPORT_ARGON_MIXTURE) {  // GasMixtureTransport, src/gas_transport.cpp:1285-1407
      const MixColl c = mix_inputs(p, q.n, Th, Te);
      const double sTe = fast_sqrt(Te), sTh = TWOT ? fast_sqrt(Th) : sTe;
      t.visc = t.bulk = t.k = 0.0;
      {
        const CollTab d22 = coll_table<2, 2>(p.cmask_diag, false, c);  // (sp, sp) of the heavy species
#pragma unroll
        for (int sp = 0; sp < NSP; sp++) {
          if (sp == IE) continue;
          const double sv = p.vf_sq_mwp[sp] * sTh * fast_rcp(coll_pick(p, sp, sp, d22));
          t.visc += q.X[sp] * sv;
          t.k += q.X[sp] * (sv * p.kf_imwp[sp]);
        }
      }
      const double ke_fac = p.ke_fac * sTe * q.X[IE];  // v_f k_f sqrt(T_e / m_e) X_e
      constexpr int EE = 1 << TPSRHS_CLMB_REP;         // (e, e): tpsrhs_create admits the repulsive Coulomb type only
      if (p.third_order) {  // :1388-1407
        constexpr double s2 = 1.4142135623730951;
        const double Q22 = coll_table<2, 2>(EE, true, c).rep, Q23 = coll_table<2, 3>(EE, true, c).rep,
                     Q24 = coll_table<2, 4>(EE, true, c).rep;
        double L11 = s2 * q.X[IE] * Q22;
        double L12 = s2 * q.X[IE] * (1.75 * Q22 - 2.0 * Q23);
        double L22 = s2 * q.X[IE] * (4.8125 * Q22 - 7.0 * Q23 + 5. * Q24);
        const CollTab e1 = coll_table<1, 1>(p.cmask_e, true, c), e2 = coll_table<1, 2>(p.cmask_e, true, c),
                      e3 = coll_table<1, 3>(p.cmask_e, true, c), e4 = coll_table<1, 4>(p.cmask_e, true, c),
                      e5 = coll_table<1, 5>(p.cmask_e, true, c);
#pragma unroll
        for (int sp = 0; sp < NSP; sp++) {
          if (sp == IE) continue;
          const double Q1[5] = {coll_pick(p, sp, IE, e1), coll_pick(p, sp, IE, e2), coll_pick(p, sp, IE, e3), coll_pick(p, sp, IE, e4),
                                coll_pick(p, sp, IE, e5)};
          L11 += q.X[sp] * (6.25 * Q1[0] - 15. * Q1[1] + 12. * Q1[2]);
          L12 += q.X[sp] * (10.9375 * Q1[0] - 39.375 * Q1[1] + 57. * Q1[2] - 30. * Q1[3]);
          L22 += q.X[sp] * (19.140625 * Q1[0] - 91.875 * Q1[1] + 199.5 * Q1[2] - 210. * Q1[3] + 90. * Q1[4]);
        }
        t.ke = s2 * ke_fac * fast_rcp(L11 - L12 * L12 * fast_rcp(L22));
      } else {
        t.ke = ke_fac * fast_rcp(coll_table<2, 2>(EE, true, c).rep);
      }
      if (diffusion) {
        double ibd[NSP * NSP];  // 1 / D_ij
#pragma unroll
        for (int i = 0; i < NSP * NSP; i++) ibd[i] = 0.0;
        const double nrsTe = q.ntot * (sTe * iTe), nrsTh = TWOT ? q.ntot * (sTh * iTh) : nrsTe;  // n / sqrt(T)
        // (1,1): the pairs with the electron at the electron temperature, the heavy pairs at the heavy-species one
        const CollTab pe = coll_table<1, 1>(p.cmask_e, true, c), ph = coll_table<1, 1>(p.cmask_h, false, c);
#pragma unroll
        for (int i = 0; i < NSP - 1; i++)
#pragma unroll
          for (int j = i + 1; j < NSP; j++)
            ibd[i + j * NSP] = ibd[j + i * NSP] = ((i == IE || j == IE) ? nrsTe : nrsTh) * p.sq_muw_idfc[i + j * NSP] *
                                                  coll_pick(p, i, j, (i == IE || j == IE) ? pe : ph);
#pragma unroll
        for (int i = 0; i < NSP; i++) {
          double a = 0.0;
#pragma unroll
          for (int j = 0; j < NSP; j++)
            if (i != j) a += (q.X[j] + kXeps) * ibd[i + j * NSP];
          diff[i] = (1.0 - q.Y[i]) * fast_rcp(a);
          mob[i] = p.qkb_charge[i] * ((i == IE) ? iTe : iTh) * diff[i];
        }
      }
      if (p.multiply) {
        t.visc *= p.mult_flux[0];
        t.bulk *= p.mult_flux[1];
        t.k *= p.mult_flux[2];
        t.ke *= p.mult_flux[3];
#pragma unroll
        for (int sp = 0; sp < NSP; sp++) {
          diff[sp] *= p.mult_diff;
          mob[sp] *= p.mult_mobil;
        }
      }
    } else {  // GasMinimalTransport::ComputeFluxTransportProperties, src/gas_transport.cpp:206-398
      const double sTe = fast_sqrt(Te), sTh = TWOT ? fast_sqrt(Th) : sTe;
      const Debye d = debye(q.n, Th, Te, iTh, iTe);
      const double lnTe = flog(Te), lnTh = TWOT ? flog(Th) : lnTe;
      double QeAr, Qatt, rep22h;
      if (p.third_order) {
        // the eight Coulomb fits of the electron temperature and the five e-Ar polynomials, four / five at a time in
        // lock step (cfit_n, eAr1r_n: the same operations per fit, interleaved -- independent chains for the VALU)
        constexpr double cA[4][4] = {{0.2150, 5.2194, 1.0472, 1.2435},   // att11
                                     {0.4128, 1.2436, 1.1830, 1.0123},   // rep22
                                     {0.2203, 1.8832, 1.2059, 0.9851},   // rep23
                                     {0.1323, 2.7248, 1.2129, 0.9847}};  // rep24
        constexpr double cB[4][4] = {{0.0991, 7.4684, 1.0155, 1.1536},    // att12
                                     {0.0616, 7.8271, 0.9452, 1.1105},    // att13
                                     {0.0308, 13.9567, 0.9511, 1.1803},   // att14
                                     {0.0232, 13.7888, 0.9148, 1.1532}};  // att15
        double A[4], B[4], QN[5];
        coll::cfit_n<4>(cA, d.e, A);
        coll::cfit_n<4>(cB, d.e, B);
        coll::eAr1r_n<5>(lnTe, QN);
        Qatt = A[0] * d.circle;
        QeAr = QN[0];
        const double Q2[3] = {d.circle * A[1], d.circle * A[2], d.circle * A[3]};
        const double QI[5] = {Qatt, d.circle * B[0], d.circle * B[1], d.circle * B[2], d.circle * B[3]};
        rep22h = TWOT ? coll::rep22(d.h) * d.circle : Q2[0];
        t.ke = third_order_ke(q.X, Q2, QI, QN, sTe, p.ke_fac3);
      } else {
        QeAr = coll::eAr1r(1, lnTe);
        Qatt = coll::att11(d.e) * d.circle;
        rep22h = coll::rep22(d.h) * d.circle;
        t.ke = p.ke_fac * sTe * q.X[I_E] * fast_rcp(TWOT ? coll::rep22(d.e) * d.circle : rep22h);
      }
      const double sv_ion = p.vf_sq_mwp[I_ION] * sTh * fast_rcp(rep22h);
      const double sv_n = p.vf_sq_mwp[I_N] * sTh * coll::iArAr22(sTh);
      t.bulk = 0.0;
      t.visc = q.X[I_ION] * sv_ion + q.X[I_N] * sv_n;
      t.k = q.X[I_ION] * (sv_ion * p.kf_imwp[I_ION]) + q.X[I_N] * (sv_n * p.kf_imwp[I_N]);
      __builtin_amdgcn_sched_barrier(0);  // the collision integrals of k_e are dead here: keep it that way
      if (diffusion) {
        const double nrsTe = q.ntot * (sTe * iTe), nrsTh = TWOT ? q.ntot * (sTh * iTh) : nrsTe;  // n / sqrt(T)
        double ibd[NSP * NSP];  // 1 / D_ij (:291-310)
#pragma unroll
        for (int i = 0; i < NSP * NSP; i++) ibd[i] = 0.0;
        ibd[I_E + I_N * NSP] = ibd[I_N + I_E * NSP] = nrsTe * p.sq_muw_idfc[I_E + I_N * NSP] * QeAr;
        ibd[I_N + I_ION * NSP] = ibd[I_ION + I_N * NSP] = nrsTh * p.sq_muw_idfc[I_N + I_ION * NSP] * coll::ArAr1P11(lnTh);
        ibd[I_E + I_ION * NSP] = ibd[I_ION + I_E * NSP] = nrsTe * p.sq_muw_idfc[I_E + I_ION * NSP] * Qatt;
        // CurtissHirschfelder, src/transport_properties.cpp:188-201
#pragma unroll
        for (int i = 0; i < NSP; i++) {
          double a = 0.0;
#pragma unroll
          for (int j = 0; j < NSP; j++)
            if (i != j) a += (q.X[j] + kXeps) * ibd[i + j * NSP];
          diff[i] = (1.0 - q.Y[i]) * fast_rcp(a);
          mob[i] = p.qkb_charge[i] * ((i == I_E) ? iTe : iTh) * diff[i];
        }
      }
      if (p.multiply) {
        t.visc *= p.mult_flux[0];
        t.bulk *= p.mult_flux[1];
        t.k *= p.mult_flux[2];
        t.ke *= p.mult_flux[3];
#pragma unroll
        for (int sp = 0; sp < NSP; sp++) {
          diff[sp] *= p.mult_diff;
          mob[sp] *= p.mult_mobil;
        }
      }
    }
    if (!diffusion) return;
    double mho = 0.0;
#pragma unroll
    for (int sp = 0; sp < NSP; sp++) {
      mho += mob[sp] * q.n[sp] * p.charge[sp];
      t.dX[sp] = diff[sp] * fast_rcp(q.X[sp] + kXeps);
      t.mob[sp] = mob[sp];
    }
    t.imho = AMBI ? fast_rcp(mho + kXeps) : 0.0;
  }
  // Diffusion velocities of one direction: -D grad X / X, ambipolar field, mass-flux correction
  // (src/transport_properties.cpp:59-136).  gs[eq] = derivative of the primitives in that direction -- a
  // Cartesian direction, or the normal derivative sum_d n_d dUp/dx_d (the map is linear and the same for every
  // direction, so V . n comes out of one application).  Only the density and species rows are read.
  __device__ static inline void diffusion_velocity(PRef p, const TCoef &t, const double *gs, double *V) {
    // ComputeMoleFractionGradient, src/equation_of_state.cpp:1534-1592
    double ne = 0.0, nb = gs[0], nt = 0.0;
#pragma unroll
    for (int sp = 0; sp < NACTIVE; sp++) {
      const double gsp = gs[NVEL + 2 + sp];
      if (AMBI) ne += gsp * p.charge[sp];
      nb -= gsp * p.mw[sp];
      nt += gsp;
    }
    if (AMBI) nb -= p.mw[IE] * ne;
    nb *= p.imw[IB];
    if (AMBI) nt += ne;
    nt += nb;
    const double in = t.intot;
    double gX[NSP];
#pragma unroll
    for (int sp = 0; sp < NACTIVE; sp++) gX[sp] = gs[NVEL + 2 + sp] * in - t.n[sp] * in * in * nt;
    if (AMBI) gX[IE] = ne * in - t.n[IE] * in * in * nt;
    gX[IB] = nb * in - t.n[IB] * in * in * nt;
#pragma unroll
    for (int sp = 0; sp < NSP; sp++) V[sp] = -t.dX[sp] * gX[sp];
    if (AMBI) {
      double ambE = 0.0;
#pragma unroll
      for (int sp = 0; sp < NSP; sp++) ambE -= V[sp] * t.n[sp] * p.charge[sp];
      ambE *= t.imho;
#pragma unroll
      for (int sp = 0; sp < NSP; sp++) V[sp] += t.mob[sp] * ambE;
    }
    double Vc = 0.0;
#pragma unroll
    for (int sp = 0; sp < NSP; sp++) Vc += t.Y[sp] * V[sp];
#pragma unroll
    for (int sp = 0; sp < NSP; sp++) V[sp] -= Vc;
  }
  // SrcTrns::ELECTRIC_CONDUCTIVITY of ComputeSourceTransportProperties: computeMixtureElectricConductivity(mobility, n) *
  // MOLARELECTRONCHARGE with the mobilities (q_e / k_B) Z_sp D_sp / T of the model's diffusivities -- the same construction
  // as in the flux transport, multipliers included (src/transport_properties.cpp:421-428, src/gas_transport.cpp:725-739,
  // 1455-1463).  What SourceTerm stores in plasma_conductivity_ (src/source_term.cpp:184-199).
  __device__ static inline double electric_conductivity(PRef p, const double *U) {
    const State s = make_state(p, U);
    TCoef c;
    transport_coeffs(p, U, s.Th, s.Te, true, c);
    double mho = 0.0;
#pragma unroll
    for (int sp = 0; sp < NSP; sp++) mho += c.mob[sp] * c.n[sp] * p.charge[sp];
    return mho * kMolarQe;
  }
  // ComputeFluxTransportProperties: coefficients + the diffusion velocities of every direction
  __device__ static inline void transport(PRef p, const double *U, double Th, double Te, const double *g,
                                          bool diffusion, Trans &t) {
    TCoef c;
    transport_coeffs(p, U, Th, Te, diffusion, c);
    t.visc = c.visc;
    t.bulk = c.bulk;
    t.k = c.k;
    t.ke = c.ke;
#pragma unroll
    for (int sp = 0; sp < NSP; sp++) t.n[sp] = c.n[sp];
#pragma unroll
    for (int k = 0; k < NSP * DIM; k++) t.V[k] = 0.0;
    if (!diffusion) return;
#pragma unroll
    for (int d = 0; d < DIM; d++) {
      double gs[NEQ], V[NSP];
#pragma unroll
      for (int eq = 0; eq < NEQ; eq++) gs[eq] = g[eq + d * NEQ];
      diffusion_velocity(p, c, gs, V);
#pragma unroll
      for (int sp = 0; sp < NSP; sp++) t.V[sp + d * NSP] = V[sp];
    }
  }
  // what SourceTerm uses of ComputeSourceTransportProperties (src/gas_transport.cpp:592-773,
  // src/transport_properties.cpp:392-449): the number densities of computeSpeciesPrimitives and the
  // electron momentum-transfer frequencies.  (The reference also evaluates the diffusion velocities and
  // the electric conductivity there; no term of the hot path reads them.)
  __device__ static inline void source_props(PRef p, const double *U, double Th, double Te, double *n,
                                             double *mtfreq) {
    const Species q = species(p, U);
#pragma unroll
    for (int sp = 0; sp < NSP; sp++) {
      n[sp] = q.n[sp];
      mtfreq[sp] = 0.0;
    }
    if (!TWOT) return;  // the frequencies enter the electron-energy exchange only
    if (TRANSPORT == TRANSPORT_CONSTANT) {
#pragma unroll
      for (int sp = 0; sp < NSP; sp++) mtfreq[sp] = p.c_mtfreq[sp];
    } else if (TRANSPORT == TRANSPORT_ARGON_MIXTURE) {  // src/gas_transport.cpp:1445-1459
      const double mff = 4. / 3. * kAvogadro * sqrt(8. * kBoltz / kPi);
      const MixColl c = mix_inputs(p, q.n, Th, Te);
      const double vth = (mff / (15. / 4. * kBoltz * 5. / 16. * sqrt(kPi * kBoltz))) * p.ke_fac * fast_sqrt(Te);  // mff sqrt(T_e / m_e)
      const CollTab pe = coll_table<1, 1>(p.cmask_e, true, c);
#pragma unroll
      for (int sp = 0; sp < NSP; sp++) {
        if (sp == IE) continue;
        mtfreq[sp] = vth * q.n[sp] * coll_pick(p, sp, IE, pe);
        if (p.multiply) mtfreq[sp] *= p.mult_spcs;
      }
    } else {
      const double mff = 4. / 3. * kAvogadro * sqrt(8. * kBoltz / kPi);
      const Debye d = debye(q.n, Th, Te, fast_rcp(Th), fast_rcp(Te));
      const double QeAr = coll::eAr1r(1, flog(Te)), Qatt = coll::att11(d.e) * d.circle;
      const double vth = (mff / (15. / 4. * kBoltz * 5. / 16. * sqrt(kPi * kBoltz))) * p.ke_fac * fast_sqrt(Te);  // mff sqrt(T_e / m_e)
      mtfreq[I_ION] = vth * q.n[I_ION] * Qatt;
      mtfreq[I_N] = vth * q.n[I_N] * QeAr;
      if (p.multiply) {
#pragma unroll
        for (int sp = 0; sp < NSP; sp++) mtfreq[sp] *= p.mult_spcs;
      }
    }
  }
  __device__ static inline void enthalpies(PRef p, const State &s, double *h) {  // :1192-1207
#pragma unroll
    for (int sp = 0; sp < NSP; sp++) h[sp] = s.n[sp] * (p.cp[sp] * ((sp == IE) ? s.Te : s.Th) + p.eform[sp]);
  }

  // ComputeViscousFluxes, src/fluxes.cpp:178-335 (3-D / planar 2-D part); Fv[eq + d*NEQ]
  // (the state-only closure `c` -- transport_coeffs -- is evaluated by the caller, before it touches the gradient)
  __device__ static inline void visc_flux(PRef p, const double *U, const State &s, const TCoef &c,
                                          const double *g, double radius, double *Fv, const EddyCtx &ec = eddy_off()) {
#pragma unroll
    for (int i = 0; i < NEQ * DIM; i++) Fv[i] = 0.0;
    if (p.eq_system == TPSRHS_EULER) return;
    struct {
      double visc, bulk, k, ke, V[NSP * DIM];
    } t;
    t.visc = c.visc;
    t.bulk = c.bulk;
    t.k = c.k;
    t.ke = c.ke;
    if constexpr (DIM == 2) add_mixing_length<DIM, NVEL, NEQ>(ec, U, g, radius, t.visc, t.bulk, t.k);
#pragma unroll
    for (int d = 0; d < DIM; d++) {
      double gs[NEQ], V[NSP];
#pragma unroll
      for (int eq = 0; eq < NEQ; eq++) gs[eq] = g[eq + d * NEQ];
      diffusion_velocity(p, c, gs, V);
#pragma unroll
      for (int sp = 0; sp < NSP; sp++) t.V[sp + d * NSP] = V[sp];
    }
    double h[NSP];
    enthalpies(p, s, h);
    double bulk = t.bulk - 2. / 3. * t.visc;
    double k = t.k;
    if constexpr (DIM == 2) {  // viscous sponge, src/fluxes.cpp:232-246: mu, mu_b - 2/3 mu, k_h and the ACTIVE species' velocities
      const double vsw = sponge_weight(ec);
      t.visc *= vsw;
      bulk *= vsw;
      k *= vsw;
#pragma unroll
      for (int d = 0; d < DIM; d++)
#pragma unroll
        for (int sp = 0; sp < NACTIVE; sp++) t.V[sp + d * NSP] *= vsw;
    }
    if (TWOT) {
#pragma unroll
      for (int d = 0; d < DIM; d++) {
        const double qe = t.ke * g[ITE + d * NEQ];
        Fv[ITH + d * NEQ] += qe;
        Fv[ITE + d * NEQ] += qe - h[IE] * t.V[IE + d * NSP];
      }
    } else {
      k += t.ke;
    }
    double divV = 0.0;
#pragma unroll
    for (int i = 0; i < DIM; i++) divV += g[(1 + i) + i * NEQ];
    double tau_t[DIM] = {};  // axisymmetric: tau_{theta r}, tau_{theta z} (src/fluxes.cpp:300-313)
    if (AXISYM) {
      if (radius > 0) divV += s.vel[0] / radius;
      double ttr = g[3 + 0 * NEQ];
      if (radius > 0) ttr -= s.vel[2] / radius;
      tau_t[0] = ttr * t.visc;
      tau_t[1] = t.visc * g[3 + 1 * NEQ];
    }
#pragma unroll
    for (int i = 0; i < DIM; i++) {
      double vt = 0.0;
#pragma unroll
      for (int j = 0; j < DIM; j++) {
        double st = t.visc * (g[(1 + j) + i * NEQ] + g[(1 + i) + j * NEQ]);
        if (i == j) st += bulk * divV;
        Fv[(1 + j) + i * NEQ] = st;
        vt += st * s.vel[j];
      }
      if (AXISYM) {
        Fv[(1 + 2) + i * NEQ] = tau_t[i];
        vt += s.vel[2] * tau_t[i];
      }
      double e = vt + k * g[ITH + i * NEQ];
#pragma unroll
      for (int sp = 0; sp < NSP; sp++) e -= h[sp] * t.V[sp + i * NSP];
      Fv[ITH + i * NEQ] += e;
#pragma unroll
      for (int sp = 0; sp < NACTIVE; sp++) Fv[(NVEL + 2 + sp) + i * NEQ] = -U[NVEL + 2 + sp] * t.V[sp + i * NSP];
    }
  }
  // closure of the nodal flux: everything of ComputeFluxTransportProperties that depends on the state alone
  typedef TCoef FluxCoef;
  __device__ static inline void flux_coeffs(PRef p, const double *U, const State &s, FluxCoef &c) {
    if (p.eq_system == TPSRHS_EULER) return;
    transport_coeffs(p, U, s.Th, s.Te, true, c);
  }
  __device__ static inline void total_flux(PRef p, const double *U, const State &s, const FluxCoef &c,
                                           const double *g, double radius, double *F, const EddyCtx &ec = eddy_off()) {
    double Fv[NEQ * DIM];
    visc_flux(p, U, s, c, g, radius, Fv, ec);
    const double H = U[ITH] + s.p;
#pragma unroll
    for (int d = 0; d < DIM; d++) {
      F[0 + d * NEQ] = U[1 + d] - Fv[0 + d * NEQ];
#pragma unroll
      for (int i = 0; i < NVEL; i++) F[1 + i + d * NEQ] = U[1 + i] * s.vel[d] + (i == d ? s.p : 0.0) - Fv[1 + i + d * NEQ];
      F[ITH + d * NEQ] = s.vel[d] * H - Fv[ITH + d * NEQ];
#pragma unroll
      for (int sp = 0; sp < NACTIVE; sp++)
        F[(NVEL + 2 + sp) + d * NEQ] = U[NVEL + 2 + sp] * s.vel[d] - Fv[(NVEL + 2 + sp) + d * NEQ];
      if (TWOT) F[ITE + d * NEQ] = (U[ITE] + s.pe) * s.vel[d] - Fv[ITE + d * NEQ];
    }
  }
  // Normal viscous flux Fv(U, g) . n of ComputeViscousFluxes, or -- with the wall prescriptions of
  // ComputeBdrViscousFluxes (src/fluxes.cpp:344-505) -- the same with zero species diffusion fluxes
  // (`zero_species`) and zero heat fluxes (`zero_heat`).  The reference evaluates the boundary variant
  // with the unit normal and rescales by |n|; the flux is linear in n, so n is used directly.
  // Wall prescriptions of BoundaryViscousFluxData (src/dataStructures.hpp:572-590): normal diffusion
  // velocities of all species, heavy-species heat flux, electron heat flux -- per unit area (the caller
  // passes |n|).
  struct WallFlux {
    bool species, heavy_heat, electron_heat;
    double Vn[NSP], hf, ef, nm;
  };
  __device__ static inline WallFlux no_prescription() {
    WallFlux w;
    w.species = w.heavy_heat = w.electron_heat = false;
#pragma unroll
    for (int sp = 0; sp < NSP; sp++) w.Vn[sp] = 0.0;
    w.hf = w.ef = 0.0;
    w.nm = 1.0;
    return w;
  }
  __device__ static inline void visc_normal_flux(PRef p, const double *U, const double *g, const double *n,
                                                 double radius, const WallFlux &w, double *Fn, const EddyCtx &ec = eddy_off()) {
    const State s = make_state(p, U);
    Trans t;
    transport(p, U, s.Th, s.Te, g, !w.species, t);
    if constexpr (DIM == 2) add_mixing_length<DIM, NVEL, NEQ>(ec, U, g, radius, t.visc, t.bulk, t.k);
    double h[NSP], Vn[NSP];
    enthalpies(p, s, h);
    const double vsw = (DIM == 2) ? sponge_weight(ec) : 1.0;
#pragma unroll
    for (int sp = 0; sp < NSP; sp++) {
      double a = 0.0;
#pragma unroll
      for (int d = 0; d < DIM; d++) a += t.V[sp + d * NSP] * n[d];
      if (DIM == 2 && sp < NACTIVE) a *= vsw;  // viscous sponge (src/fluxes.cpp:395-408), before the wall prescriptions
      Vn[sp] = w.species ? w.Vn[sp] * w.nm : a;
    }
    double bulk = t.bulk - 2. / 3. * t.visc;
    if constexpr (DIM == 2) {
      t.visc *= vsw;
      bulk *= vsw;
      t.k *= vsw;
    }
    double divV = 0.0;
#pragma unroll
    for (int i = 0; i < DIM; i++) divV += g[(1 + i) + i * NEQ];
    if (AXISYM && radius > 0) divV += s.vel[0] / radius;
    double e = 0.0;
    Fn[0] = 0.0;
#pragma unroll
    for (int i = 0; i < DIM; i++) {
      double sn = 0.0;
#pragma unroll
      for (int j = 0; j < DIM; j++) {
        double st = t.visc * (g[(1 + j) + i * NEQ] + g[(1 + i) + j * NEQ]);
        if (i == j) st += bulk * divV;
        sn += st * n[j];
      }
      Fn[1 + i] = sn;
      e += sn * s.vel[i];
    }
    if (AXISYM) {  // tau_{theta r} n_r + tau_{theta z} n_z, src/fluxes.cpp:300-313,457-466
      double ttr = g[3 + 0 * NEQ];
      if (radius > 0) ttr -= s.vel[2] / radius;
      const double tn = t.visc * (ttr * n[0] + g[3 + 1 * NEQ] * n[1]);
      Fn[1 + 2] = tn;
      e += tn * s.vel[2];
    }
    // heat fluxes in the reference's "primitive flux" sense (src/fluxes.cpp:468-482):
    // HF = -k grad T_h . n + sum_{heavy} h V_n,  EF = -k_e grad T_e . n + h_e V_n,e  (two-temperature)
    double HF = 0.0, EF = 0.0;
    if (w.heavy_heat) {
      HF = w.hf * w.nm;
    } else {
      const double k = TWOT ? t.k : t.k + t.ke;
#pragma unroll
      for (int d = 0; d < DIM; d++) HF -= k * g[ITH + d * NEQ] * n[d];
#pragma unroll
      for (int sp = 0; sp < NSP; sp++)
        if (!(TWOT && sp == IE)) HF += h[sp] * Vn[sp];
    }
    if (TWOT) {
      if (w.electron_heat) {
        EF = w.ef * w.nm;
      } else {
#pragma unroll
        for (int d = 0; d < DIM; d++) EF -= t.ke * g[ITE + d * NEQ] * n[d];
        EF += h[IE] * Vn[IE];
      }
    }
    Fn[ITH] = e - HF - EF;
#pragma unroll
    for (int sp = 0; sp < NACTIVE; sp++) Fn[NVEL + 2 + sp] = -U[NVEL + 2 + sp] * Vn[sp];
    if (TWOT) Fn[ITE] = -EF;
  }
  // GetConservativesFromPrimitives, src/equation_of_state.cpp:744-783
  __device__ static inline void cons(PRef p, const double *Up, double *U) {
    U[0] = Up[0];
    double ke = 0.0;
#pragma unroll
    for (int d = 0; d < NVEL; d++) {
      U[1 + d] = Up[1 + d] * Up[0];
      ke += Up[1 + d] * Up[1 + d];
    }
    ke *= 0.5 * Up[0];
    double n[NSP], rhoB = Up[0], ne = 0.0;
#pragma unroll
    for (int sp = 0; sp < NSP; sp++) n[sp] = 0.0;
#pragma unroll
    for (int sp = 0; sp < NACTIVE; sp++) {
      n[sp] = Up[NVEL + 2 + sp];
      U[NVEL + 2 + sp] = n[sp] * p.mw[sp];
      rhoB -= p.mw[sp] * n[sp];
      if (AMBI) ne += p.charge[sp] * n[sp];
    }
    if (AMBI) {
      ne = fmax(ne, 0.0);
      rhoB -= ne * p.mw[IE];
      n[IE] = ne;
    }
    n[IB] = rhoB * p.imw[IB];
    double ctot = heavies_cv(p, n);
    if (!TWOT) ctot += n[IE] * p.cv[IE];
    double e = ke + ctot * Up[ITH];
    if (TWOT) {
      U[ITE] = n[IE] * p.cv[IE] * Up[ITE];
      e += U[ITE];
    }
#pragma unroll
    for (int sp = 0; sp < NSP - 2; sp++) e += Up[NVEL + 2 + sp] * p.eform[sp];
    U[ITH] = e;
  }
  // VISC_GNRL wall state: modifyStateFromPrimitive with no slip and the prescribed temperatures
  // (src/wallBC.cpp:112-148, src/equation_of_state.cpp:131-140); data = {T_h, T_e, heavy cond, electron cond}
  __device__ static inline void general_wall_state(PRef p, BcRef bc, const double *U, double *Uw) {
    double up[NEQ];
    prim(p, U, up);
#pragma unroll
    for (int d = 0; d < NVEL; d++) up[1 + d] = 0.0;
    if (static_cast<int>(bc.data[2]) == TPSRHS_ISOTH) up[ITH] = bc.data[0];
    if (static_cast<int>(bc.data[3]) == TPSRHS_ISOTH) up[NEQ - 1] = bc.data[1];
    cons(p, up, Uw);
  }
  // computeSheathBdrFlux, src/equation_of_state.cpp:1909-1942: Bohm velocities of the positive ions,
  // the electron and background fluxes that balance them, the electron energy flux through the sheath
  __device__ static inline void sheath(PRef p, const double *Uw, WallFlux &w) {
    const State s = make_state(p, Uw);
#pragma unroll
    for (int sp = 0; sp < NSP; sp++) w.Vn[sp] = 0.0;
#pragma unroll
    for (int sp = 0; sp < NSP; sp++) {
      if (sp == IE || sp == IB) continue;  // the electron is negative, the background neutral
      const double Z = p.charge[sp];
      if (Z > 0.0) {
        const double VB = fast_sqrt((s.Th + Z * s.Te) * p.rg_imw[sp]);
        w.Vn[sp] = VB;
        w.Vn[IE] += Z * s.n[sp] * VB;
        w.Vn[IB] -= p.mw[sp] * s.n[sp] * VB;
      }
    }
    w.Vn[IE] /= s.n[IE];
    w.Vn[IB] -= p.mw[IE] * s.n[IE] * w.Vn[IE];
    w.Vn[IB] /= p.mw[IB] * s.n[IB];
    if (TWOT) {
      const double vTe = fast_sqrt((8.0 / kPi) * s.Te * p.rg_imw[IE]);
      const double gam = -log(4.0 / vTe * w.Vn[IE]);
      w.ef = w.Vn[IE] * (gam + 2.0) * s.n[IE] * kRgas * s.Te;
    }
  }

  // ---- boundary conditions ------------------------------------------------------------------
  __device__ static inline void stagnant_with_temp(PRef p, const double *U, double T, double *Uw) {  // :1596-1620
    double n[NSP];
    number_densities(p, U, n);
#pragma unroll
    for (int eq = 0; eq < NEQ; eq++) Uw[eq] = U[eq];
#pragma unroll
    for (int d = 0; d < NVEL; d++) Uw[1 + d] = 0.0;
    const double Ue = n[IE] * p.cv[IE] * T;
    double e = heavies_cv(p, n) * T + Ue;
    if (TWOT) Uw[ITE] = Ue;
#pragma unroll
    for (int sp = 0; sp < NSP - 2; sp++) e += n[sp] * p.eform[sp];
    Uw[ITH] = e;
  }
  __device__ static inline void energy_for_pressure(PRef p, const double *Uin, double pres, bool modE,
                                                    double *Uo) {  // :1698-1742
    double n[NSP];
    number_densities(p, Uin, n);
    double pe = 0.0, nsum = 0.0;
    if (TWOT && !modE) pe = n[IE] * kRgas * (Uin[ITE] / (n[IE] + kXeps) * p.icv_e);
#pragma unroll
    for (int sp = 0; sp < NSP; sp++)
      if (!(TWOT && !modE && sp == IE)) nsum += n[sp];
    const double Th = (pres - pe) / (nsum * kRgas);
    double rE = heavies_cv(p, n) * Th;
    double ee;
    if (TWOT)
      ee = modE ? n[IE] * p.cv[IE] * Th : Uin[ITE];
    else
      ee = n[IE] * p.cv[IE] * Th;
    rE += ee;
    double ke = 0.0;
#pragma unroll
    for (int d = 0; d < NVEL; d++) ke += 0.5 * Uin[1 + d] * Uin[1 + d] / Uin[0];
    rE += ke;
#pragma unroll
    for (int sp = 0; sp < NSP - 2; sp++) rE += n[sp] * p.eform[sp];
#pragma unroll
    for (int eq = 0; eq < NEQ; eq++) Uo[eq] = Uin[eq];
    if (TWOT) Uo[ITE] = ee;
    Uo[ITH] = rE;
  }
  __device__ static inline void bc_ghost(PRef p, BcRef bc, const double *U, const double *n,
                                         double *Ug, const double * = nullptr) {
#pragma unroll
    for (int eq = 0; eq < NEQ; eq++) Ug[eq] = U[eq];
    if (bc.category == TPSRHS_INLET) {  // src/inletBC.cpp:729-757
      const double pres = pressure(p, U);
      double s2[NEQ];
#pragma unroll
      for (int eq = 0; eq < NEQ; eq++) s2[eq] = U[eq];
      s2[0] = bc.data[0];
#pragma unroll
      for (int d = 0; d < NVEL; d++) s2[1 + d] = bc.data[0] * bc.data[1 + d];
      if constexpr (DIM == 3) {
        if (is_face_inlet(bc.category, bc.type)) {  // velocity relative to the face, mirrored momentum (src/inletBC.cpp:758-864)
          double mom[3];
          face_inlet_momentum(n, bc.type - TPSRHS_SUB_DENS_VEL_FACE_X, bc.data[0], bc.data[1], bc.data[2], mom);
#pragma unroll
          for (int d = 0; d < 3; d++) s2[1 + d] = 2.0 * mom[d] - U[1 + d];
        }
      }
#pragma unroll
      for (int sp = 0; sp < NACTIVE; sp++) s2[NVEL + 2 + sp] = bc.data[4 + sp];
      energy_for_pressure(p, s2, pres, true, Ug);
    } else if (bc.category == TPSRHS_OUTLET) {  // src/outletBC.cpp:731-737
      energy_for_pressure(p, U, bc.data[0], false, Ug);
    } else if (bc.type == TPSRHS_INV || bc.type == TPSRHS_SLIP) {  // src/wallBC.cpp:277-301 (SLIP :326-428: same mirror state)
      double nm = 0.0;
#pragma unroll
      for (int d = 0; d < DIM; d++) nm += n[d] * n[d];
      nm = fast_sqrt(nm);
      double vn = 0.0;
#pragma unroll
      for (int d = 0; d < DIM; d++) vn += (U[1 + d] / U[0]) * (n[d] / nm);
#pragma unroll
      for (int d = 0; d < DIM; d++) Ug[1 + d] = U[0] * (U[1 + d] / U[0] - 2.0 * vn * (n[d] / nm));
      if (DIM == 2 && bc.type == TPSRHS_SLIP) slip_ghost_momentum_2d(n, U, Ug);
    } else if (bc.type == TPSRHS_VISC_ADIAB) {  // GasMixture::computeStagnationState, :100-115
      double ke = 0.0;
#pragma unroll
      for (int d = 0; d < NVEL; d++) {
        ke += 0.5 * U[1 + d] * U[1 + d] / U[0];
        Ug[1 + d] = 0.0;
      }
      Ug[ITH] = U[ITH] - ke;
    } else if (bc.type == TPSRHS_VISC_GNRL) {  // src/wallBC.cpp:514-518
      general_wall_state(p, bc, U, Ug);
    } else {  // VISC_ISOTH, src/wallBC.cpp:471-485
      if (p.use_bc_in_grad) {
#pragma unroll
        for (int d = 0; d < NVEL; d++) Ug[1 + d] = -U[1 + d];
      } else {
        stagnant_with_temp(p, U, bc.data[0], Ug);
      }
    }
  }
  // State and wall prescriptions of pass `pass` of the viscous trace of a face point: pass 0 = the interior
  // state, pass 1 (wall faces) = the wall-side state of the wall type (src/wallBC.cpp:277-543)
  __device__ static inline void visc_pass_state(PRef p, int nb, int pass, const double *U, const double *n,
                                                double *Us, WallFlux &w) {
#pragma unroll
    for (int eq = 0; eq < NEQ; eq++) Us[eq] = U[eq];
    w = no_prescription();
    if (pass == 1) {  // the wall-side state
      BcRef wbc = p.bc[-nb - 1];
      const int type = wbc.type;
      const double twall = wbc.data[0];
      double ke = 0.0;
#pragma unroll
      for (int d = 0; d < NVEL; d++) ke += 0.5 * U[1 + d] * U[1 + d] / U[0];
      if (type == TPSRHS_INV) {
        double nm = 0.0, vn = 0.0;
#pragma unroll
        for (int d = 0; d < DIM; d++) nm += n[d] * n[d];
        nm = fast_sqrt(nm);
#pragma unroll
        for (int d = 0; d < DIM; d++) vn += (U[1 + d] / U[0]) * (n[d] / nm);
#pragma unroll
        for (int d = 0; d < DIM; d++) Us[1 + d] = U[0] * (U[1 + d] / U[0] - 2.0 * vn * (n[d] / nm));
      } else if (type == TPSRHS_VISC_ADIAB) {
#pragma unroll
        for (int d = 0; d < NVEL; d++) Us[1 + d] = 0.0;
        Us[ITH] = U[ITH] - ke;
        w.species = w.heavy_heat = w.electron_heat = true;
      } else if (type == TPSRHS_VISC_ISOTH) {
        stagnant_with_temp(p, U, twall, Us);
        w.species = true;
      } else {  // VISC_GNRL, src/wallBC.cpp:512-543
        BcRef bc = p.bc[-nb - 1];
        general_wall_state(p, bc, U, Us);
        const int hc = static_cast<int>(bc.data[2]), ec = static_cast<int>(bc.data[3]);
        w.species = true;
        w.heavy_heat = (hc == TPSRHS_ADIAB);
        w.electron_heat = TWOT && (ec == TPSRHS_ADIAB || ec == TPSRHS_SHTH);
        double nm = 0.0;
#pragma unroll
        for (int d = 0; d < DIM; d++) nm += n[d] * n[d];
        w.nm = fast_sqrt(nm);
        if (ec == TPSRHS_SHTH) sheath(p, Us, w);
      }
    }
  }
  // passes of the viscous trace of a face point: 0 = no viscous term (Euler; inlets, outlets and slip walls add
  // the Riemann flux only), 1 = interior face, 2 = wall face (interior state, then wall-side state)
  __device__ static inline int visc_passes(PRef p, int nb) {
    if (p.eq_system == TPSRHS_EULER) return 0;
    if (nb >= 0) return 1;
    BcRef bc = p.bc[-nb - 1];
    return (bc.category != TPSRHS_WALL || bc.type == TPSRHS_SLIP) ? 0 : 2;
  }
  // ---- the same trace in two steps (3-D face kernel): the state-only closure first -- the transcendental-heavy
  // part, evaluated while only the NEQ interpolated state values are live --, then the flux from the velocity
  // gradient and the NORMAL derivatives of the scalar primitives (the heat and diffusion fluxes are linear in
  // grad(.) . n), which the kernel interpolates afterwards.
  struct ViscCoef {
    TCoef t;
    double vel[NVEL], h[NSP];
  };
  __device__ static inline void visc_point_coeffs(PRef p, const double *U, bool diffusion, ViscCoef &c) {
    const State s = make_state(p, U);
    transport_coeffs(p, U, s.Th, s.Te, diffusion, c.t);
    enthalpies(p, s, c.h);
#pragma unroll
    for (int d = 0; d < NVEL; d++) c.vel[d] = s.vel[d];
  }
  // gv[i + j*DIM] = d u_i / d x_j; gn[eq] = sum_d n_d dUp_eq/dx_d (read for the scalar rows only).
  // Same terms as visc_normal_flux (ComputeViscousFluxes . n / ComputeBdrViscousFluxes, src/fluxes.cpp:178-505).
  __device__ static inline void visc_normal_flux_n(PRef p, const double *U, const ViscCoef &c, const double *gv,
                                                   const double *gn, const double *n, const WallFlux &w, double *Fn) {
    static_assert(!AXISYM, "planar / 3-D form");
    double Vn[NSP];
#pragma unroll
    for (int sp = 0; sp < NSP; sp++) Vn[sp] = 0.0;
    if (!w.species) diffusion_velocity(p, c.t, gn, Vn);
#pragma unroll
    for (int sp = 0; sp < NSP; sp++)
      if (w.species) Vn[sp] = w.Vn[sp] * w.nm;
    const double bulk = c.t.bulk - 2. / 3. * c.t.visc;
    double divV = 0.0;
#pragma unroll
    for (int i = 0; i < DIM; i++) divV += gv[i + i * DIM];
    double e = 0.0;
    Fn[0] = 0.0;
#pragma unroll
    for (int i = 0; i < DIM; i++) {
      double sn = 0.0;
#pragma unroll
      for (int j = 0; j < DIM; j++) {
        double st = c.t.visc * (gv[j + i * DIM] + gv[i + j * DIM]);
        if (i == j) st += bulk * divV;
        sn += st * n[j];
      }
      Fn[1 + i] = sn;
      e += sn * c.vel[i];
    }
    double HF = 0.0, EF = 0.0;
    if (w.heavy_heat) {
      HF = w.hf * w.nm;
    } else {
      const double k = TWOT ? c.t.k : c.t.k + c.t.ke;
      HF -= k * gn[ITH];
#pragma unroll
      for (int sp = 0; sp < NSP; sp++)
        if (!(TWOT && sp == IE)) HF += c.h[sp] * Vn[sp];
    }
    if (TWOT) {
      if (w.electron_heat) {
        EF = w.ef * w.nm;
      } else {
        EF -= c.t.ke * gn[ITE];
        EF += c.h[IE] * Vn[IE];
      }
    }
    Fn[ITH] = e - HF - EF;
#pragma unroll
    for (int sp = 0; sp < NACTIVE; sp++) Fn[NVEL + 2 + sp] = -U[NVEL + 2 + sp] * Vn[sp];
    if (TWOT) Fn[ITE] = -EF;
  }
  // ---- the LEAN form of the two-step trace.  What the flux needs of the closure after the gradient has been brought in is
  // little: mu, mu_b - 2/3 mu, k, the velocity -- and the heat carried by diffusion and the species fluxes, which are LINEAR
  // in the normal derivatives of the rows the diffusion velocities read (density and active species; diffusion_velocity is
  // a homogeneous linear map of them).  So the closure contracts that map with the enthalpies and species densities once:
  // (1 + NACTIVE) x (1 + NACTIVE [+ 1 two-temperature]) coefficients -- 4 for the ambipolar single-temperature ternary
  // mixture -- replace D/X, mobilities, mass fractions, number densities, enthalpies and the state (38 values -> 12).  The
  // wall prescriptions (WallFlux) become the constant terms.  Same terms as visc_normal_flux_n, summed in another order
  // (rounding).
  static constexpr int NIN = 1 + NACTIVE;  // inputs of the diffusion map: d(rho)/dn, d(n_sp)/dn of the active species
  struct ViscLean {
    double mu, bulkp, kq, ke;                      // kq multiplies dT_h/dn (k_h, + k_e single-temperature); ke: dT_e/dn
    double vel[NVEL];
    double hf0, ef0, sf0[NACTIVE];                 // prescribed (wall) parts of the heavy / electron heat flux and species rows
    double hfc[NIN], efc[NIN], sfc[NACTIVE][NIN];  // the same three as linear maps of the inputs
  };
  __device__ static inline void visc_point_lean(PRef p, const double *U, const WallFlux &w, ViscLean &L) {
    const State s = make_state(p, U);
    TCoef t;
    transport_coeffs(p, U, s.Th, s.Te, !w.species, t);
    double h[NSP];
    enthalpies(p, s, h);
    L.mu = t.visc;
    L.bulkp = t.bulk - 2. / 3. * t.visc;
#pragma unroll
    for (int d = 0; d < NVEL; d++) L.vel[d] = s.vel[d];
    L.kq = w.heavy_heat ? 0.0 : (TWOT ? t.k : t.k + t.ke);
    L.hf0 = w.heavy_heat ? w.hf * w.nm : 0.0;
    L.ke = (TWOT && !w.electron_heat) ? t.ke : 0.0;
    L.ef0 = (TWOT && w.electron_heat) ? w.ef * w.nm : 0.0;
#pragma unroll
    for (int sp = 0; sp < NACTIVE; sp++) L.sf0[sp] = 0.0;
#pragma unroll
    for (int j = 0; j < NIN; j++) {
      L.hfc[j] = L.efc[j] = 0.0;
#pragma unroll
      for (int sp = 0; sp < NACTIVE; sp++) L.sfc[sp][j] = 0.0;
    }
    if (w.species) {  // prescribed normal diffusion velocities (per unit area)
      double Vn[NSP];
#pragma unroll
      for (int sp = 0; sp < NSP; sp++) Vn[sp] = w.Vn[sp] * w.nm;
      if (!w.heavy_heat) {
#pragma unroll
        for (int sp = 0; sp < NSP; sp++)
          if (!(TWOT && sp == IE)) L.hf0 += h[sp] * Vn[sp];
      }
      if (TWOT && !w.electron_heat) L.ef0 += h[IE] * Vn[IE];
#pragma unroll
      for (int sp = 0; sp < NACTIVE; sp++) L.sf0[sp] = -U[NVEL + 2 + sp] * Vn[sp];
    } else {
#pragma unroll
      for (int j = 0; j < NIN; j++) {
        double gs[NEQ], V[NSP];
#pragma unroll
        for (int eq = 0; eq < NEQ; eq++) gs[eq] = 0.0;
        gs[j == 0 ? 0 : NVEL + 2 + (j - 1)] = 1.0;
        diffusion_velocity(p, t, gs, V);
        if (!w.heavy_heat) {
          double a = 0.0;
#pragma unroll
          for (int sp = 0; sp < NSP; sp++)
            if (!(TWOT && sp == IE)) a += h[sp] * V[sp];
          L.hfc[j] = a;
        }
        if (TWOT && !w.electron_heat) L.efc[j] = h[IE] * V[IE];
#pragma unroll
        for (int sp = 0; sp < NACTIVE; sp++) L.sfc[sp][j] = -U[NVEL + 2 + sp] * V[sp];
      }
    }
  }
  // gv[i + j*DIM] = d u_i / d x_j; gn[eq] = sum_d n_d dUp_eq/dx_d (scalar rows)
  __device__ static inline void visc_normal_flux_lean(const ViscLean &L, const double *gv, const double *gn, const double *n,
                                                      double *Fn) {
    static_assert(!AXISYM, "planar / 3-D form");
    double divV = 0.0;
#pragma unroll
    for (int i = 0; i < DIM; i++) divV += gv[i + i * DIM];
    double e = 0.0;
    Fn[0] = 0.0;
#pragma unroll
    for (int i = 0; i < DIM; i++) {
      double sn = 0.0;
#pragma unroll
      for (int j = 0; j < DIM; j++) {
        double st = L.mu * (gv[j + i * DIM] + gv[i + j * DIM]);
        if (i == j) st += L.bulkp * divV;
        sn += st * n[j];
      }
      Fn[1 + i] = sn;
      e += sn * L.vel[i];
    }
    double gin[NIN];
    gin[0] = gn[0];
#pragma unroll
    for (int sp = 0; sp < NACTIVE; sp++) gin[1 + sp] = gn[NVEL + 2 + sp];
    double HF = L.hf0 - L.kq * gn[ITH], EF = TWOT ? L.ef0 - L.ke * gn[ITE] : 0.0;
#pragma unroll
    for (int j = 0; j < NIN; j++) {
      HF += L.hfc[j] * gin[j];
      if (TWOT) EF += L.efc[j] * gin[j];
    }
    Fn[ITH] = e - HF - EF;
#pragma unroll
    for (int sp = 0; sp < NACTIVE; sp++) {
      double f = L.sf0[sp];
#pragma unroll
      for (int j = 0; j < NIN; j++) f += L.sfc[sp][j] * gin[j];
      Fn[NVEL + 2 + sp] = f;
    }
    if (TWOT) Fn[ITE] = -EF;
  }
  // The viscous trace of one face quadrature point.  Interior face (nb >= 0): Fv(U, g) . n.  Boundary
  // face: the complete additive viscous boundary term -1/2 (Fv_wall + Fv_in) . n of the wall types
  // (src/wallBC.cpp:302-320,448-468,492-510), zero for inlets and outlets.  One transport evaluation
  // site, looped: the kernels carry a single copy of the transport code.
  static constexpr bool HEAVY = true;
  // the closures split into a state-only part (collision integrals) and terms linear in the gradient: the kernels
  // evaluate the first before they bring the gradient into registers (flux_coeffs / visc_point_coeffs)
  static constexpr bool TWO_STEP = true;
  __device__ static inline void visc_trace(PRef p, int nb, const double *U, const double *g, const double *n,
                                           double radius, double *fn, const EddyCtx &ec = eddy_off()) {
#pragma unroll
    for (int eq = 0; eq < NEQ; eq++) fn[eq] = 0.0;
    const int npass = visc_passes(p, nb);
    // mixing length: the wall routines hand the flux class distance 0 (viscous walls, src/wallBC.cpp:441-536) or the
    // interpolated distance (inviscid wall, :309-313)
    EddyCtx wec = ec;
    if (DIM == 2 && nb < 0 && p.bc[nb < 0 ? -nb - 1 : 0].type != TPSRHS_INV) wec.dist = 0.0;
#pragma clang loop unroll(disable)
    for (int pass = 0; pass < npass; pass++) {
      PRef q = relaunder(p);  // the parameter loads of the closure stay inside the pass
      double Us[NEQ];
      WallFlux w;
      visc_pass_state(q, nb, pass, U, n, Us, w);
      double f[NEQ];
      visc_normal_flux(q, Us, g, n, radius, w, f, wec);
      if (nb >= 0) {
#pragma unroll
        for (int eq = 0; eq < NEQ; eq++) fn[eq] = f[eq];
      } else {
#pragma unroll
        for (int eq = 1; eq < NEQ; eq++) fn[eq] -= 0.5 * f[eq];
      }
    }
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

  // GetViscosities of the selected transport (src/transport_properties.cpp:440-449,
  // src/gas_transport.cpp:775-822): shear and bulk viscosity only
  __device__ static inline void viscosities(PRef p, const double *U, double Th, double Te, double &visc,
                                            double &bulk) {
    if (TRANSPORT == TRANSPORT_CONSTANT) {
      visc = p.c_visc;
      bulk = p.c_bulk;
      return;
    }
    const Species q = species(p, U);
    const double vf = 5. / 16. * sqrt(kPi * kBoltz);
    if (TRANSPORT == TRANSPORT_ARGON_MIXTURE) {  // src/gas_transport.cpp:1499-1535
      const MixColl c = mix_inputs(p, q.n, Th, Te);
      visc = 0.0;
      const CollTab d22 = coll_table<2, 2>(p.cmask_diag, false, c);
#pragma unroll
      for (int sp = 0; sp < NSP; sp++)
        if (sp != IE) visc += q.X[sp] * (p.vf_sq_mwp[sp] * fast_sqrt(Th) * fast_rcp(coll_pick(p, sp, sp, d22)));
      bulk = 0.0;
      if (p.multiply) {
        visc *= p.mult_flux[0];
        bulk *= p.mult_flux[1];
      }
      return;
    }
    const Debye d = debye(q.n, Th, Te, fast_rcp(Th), fast_rcp(Te));
    const double sTh = fast_sqrt(Th);
    const double sv_ion = p.vf_sq_mwp[I_ION] * sTh * fast_rcp(coll::rep22(d.h) * d.circle);
    const double sv_n = p.vf_sq_mwp[I_N] * sTh * coll::iArAr22(sTh);
    visc = q.X[I_ION] * sv_ion + q.X[I_N] * sv_n;
    bulk = 0.0;
    if (p.multiply) {
      visc *= p.mult_flux[0];
      bulk *= p.mult_flux[1];
    }
  }
  // AxisymmetricSource::updateTerms at one node (src/forcing_terms.cpp:293-380): the 1/r terms of the
  // radial and azimuthal momentum equations
  __device__ static inline void axisym_source(PRef p, const double *Uin, const double *Upin, const double *g,
                                              double radius, double *src) {
    double U[NEQ], Up[NEQ];
#pragma unroll
    for (int eq = 0; eq < NEQ; eq++) {
      U[eq] = Uin[eq];
      Up[eq] = Upin[eq];
    }
#pragma unroll
    for (int sp = 0; sp < NACTIVE; sp++) {
      const int eq = 3 + 2 + sp;  // src/forcing_terms.cpp:314 (nvel = 3 here)
      if (eq < NEQ) {
        U[eq] = fmax(U[eq], 0.0);
        Up[eq] = fmax(Up[eq], 0.0);
      }
    }
    const double rho = Up[0], ur = Up[1], ut = Up[3];
    // ComputePressureFromPrimitives, src/equation_of_state.cpp:988-1010
    double ne = 0.0, rhoB = Up[0], nh = 0.0;
#pragma unroll
    for (int sp = 0; sp < NACTIVE; sp++) {
      const double nsp = Up[NVEL + 2 + sp];
      if (AMBI) ne += p.charge[sp] * nsp;
      rhoB -= p.mw[sp] * nsp;
      if (sp != IE) nh += nsp;
    }
    if (AMBI) {
      ne = fmax(ne, 0.0);
      rhoB -= ne * p.mw[IE];
    } else {
      ne = Up[NVEL + 2 + IE];
    }
    nh += rhoB * p.imw[IB];
    const double Th = Up[ITH], Te = TWOT ? Up[ITE] : Up[ITH];
    const double pres = kRgas * (nh * Th + ne * Te);
    double tau_tt = 0.0, tau_tr = 0.0;
    if (p.eq_system != TPSRHS_EULER) {
      double visc, bulkv;
      viscosities(p, U, Th, Te, visc, bulkv);
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

  // ---- SourceTerm::updateTerms at one node, src/source_term.cpp:107-251 ------------------------
  __device__ static inline void source(PRef p, const double *Uin, const double *Upin, const double *g,
                                       double *src) {
    double U[NEQ], Up[NEQ];
#pragma unroll
    for (int eq = 0; eq < NEQ; eq++) {
      U[eq] = Uin[eq];
      Up[eq] = Upin[eq];
      src[eq] = 0.0;
    }
#pragma unroll
    for (int sp = 0; sp < NACTIVE; sp++) {
      const int eq = 3 + 2 + sp;  // hard-coded nvel = 3 in the reference (src/source_term.cpp:129)
      if (eq < NEQ) {
        U[eq] = fmax(U[eq], 0.0);
        Up[eq] = fmax(Up[eq], 0.0);
      }
    }
    // temperatures come from the nodal primitives, number densities from the (clamped) state
    const double Th = Up[ITH], Te = TWOT ? Up[ITE] : Up[ITH];
    struct {
      double n[NSP], mtfreq[NSP];
    } t;
    source_props(p, U, Th, Te, t.n, t.mtfreq);
    const ChemDev &c = *p.chem;
    double e_loss = 0.0;  // sum of reaction energy x progress rate over the electron-impact reactions
    if (c.num_reactions > 0) {
      const double Thl = fmax(Th, c.min_temperature), Tel = fmax(Te, c.min_temperature);
      double creation[NSP];
#pragma unroll
      for (int sp = 0; sp < NSP; sp++) creation[sp] = 0.0;
      int tab_root = -1, tab_idx = 0;  // the last table interval found: (grid, temperature) -> interval
      double tab_temp = 0.0;
      for (int r = 0; r < c.num_reactions; r++) {
        const bool el = (c.electron_index < 0) ? false : (c.reactant[c.electron_index + r * NSP] != 0);
        const double temp = el ? Tel : Thl;
        const double A = c.rate[0 + r * 3], b = c.rate[1 + r * 3], E = c.rate[2 + r * 3];
        double kf;
        if (c.model[r] == TPSRHS_ARRHENIUS) {  // A T^b exp(-E/RT) in one exponential
          kf = A * fexp(b * flog(temp) - E / kRgas / temp);
        } else if (c.model[r] == TPSRHS_HOFFERTLIEN) {
          const double tf = E / kBoltz / temp;
          kf = A * (tf + 2.0) * fexp(b * flog(temp) - tf);
        } else {  // LinearTable::eval (src/table.cpp:80-101); the interval search shared between tables of one grid
          const TableDev &tb = c.table[r];
          const int root = c.table_share[r];
          int ti;
          if (root == tab_root && temp == tab_temp) {
            ti = tab_idx;
          } else {
            ti = table_interval(tb, temp);
            tab_root = root;
            tab_temp = temp;
            tab_idx = ti;
          }
          const double xt = tb.x_log ? flog(temp) : temp;
          kf = tb.a[ti] + tb.b[ti] * xt;
          if (tb.f_log) kf = fexp(kf);
        }
        double rate = 1.0;
#pragma unroll
        for (int sp = 0; sp < NSP; sp++) rate *= ipow(t.n[sp], c.reactant[sp + r * NSP]);
        if (c.detailed_balance[r]) {
          const double kc = c.keq[0 + r * 3] * fexp(c.keq[1 + r * 3] * flog(temp) - c.keq[2 + r * 3] / temp);
          double bwd = 1.0;
#pragma unroll
          for (int sp = 0; sp < NSP; sp++) bwd *= ipow(t.n[sp], c.product[sp + r * NSP]);
          rate -= bwd / kc;
        }
        const double progress = kf * rate;
        if (TWOT && el) e_loss += c.energy[r] * progress;
#pragma unroll
        for (int sp = 0; sp < NSP; sp++) creation[sp] += progress * (c.product[sp + r * NSP] - c.reactant[sp + r * NSP]);
      }
#pragma unroll
      for (int sp = 0; sp < NACTIVE; sp++) src[NVEL + 2 + sp] += creation[sp] * p.mw[sp];
    }
    if (c.radiation == TPSRHS_NET_EMISSION) src[ITH] += -4.0 * kPi * table_eval(c.nec, Th);
    if (TWOT) {
      src[ITE] -= e_loss;
      // u . grad p_e, src/equation_of_state.cpp:1847-1870
#pragma unroll
      for (int d = 0; d < DIM; d++) {
        double neg = 0.0;
        if (AMBI) {
#pragma unroll
          for (int sp = 0; sp < NACTIVE; sp++) neg += g[(NVEL + 2 + sp) + d * NEQ] * p.charge[sp];
        } else {
          neg = g[(NVEL + NSP) + d * NEQ];
        }
        src[ITE] += (neg * Te + t.n[IE] * g[ITE + d * NEQ]) * kRgas * Up[1 + d];
      }
      const double me = p.mw[IE], ne = t.n[IE];
#pragma unroll
      for (int sp = 0; sp < NSP; sp++) {
        if (sp == IE) continue;
        const double ms = p.mw[sp];
        double e = 1.5 * kRgas * (Te - Th);
        e *= 2.0 * me * ms / (ms + me) / (ms + me) * ne * t.mtfreq[sp];
        src[ITE] -= e;
      }
    }
  }
};

}  // namespace tpsrhs
#endif
