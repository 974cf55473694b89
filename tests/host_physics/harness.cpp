// TEST INFRASTRUCTURE (tests/test_host_sanitize.py): the DEVICE point physics of the plasma kernels -- the very headers of
// tps_amd/csrc, compiled for the host through tests/host_physics/hip/hip_runtime.h -- run under AddressSanitizer and
// UndefinedBehaviorSanitizer on the states of a test case: state closure, transport, nodal flux, sources, viscous
// traces of interior and boundary faces with every pass, Lax-Friedrichs flux, boundary ghost states.  What the GPU
// cannot tell us (no GPU sanitizer on this pool): an out-of-bounds index into a private array, a shift or signed
// overflow, a read of an uninitialised bool / enum.  The parameter blocks are filled by the library's own host code
// (plasma_params_host.hpp).  One generated copy of the headers differs from the product's in ONE line: the empty
// `asm volatile` of PlasmaPhys::relaunder (an AMDGPU register constraint) is dropped by the build recipe.
#include <cstdio>
#include <vector>

#include "../../include/tpsrhs.h"
#include "plasma_params_host.hpp"

using namespace tpsrhs;

namespace {
std::vector<std::vector<double>> g_tables;  // homes of the table coefficients (host memory here)

template <class PH>
int run(const tpsrhs_disc *disc, const tpsrhs_physics *phys, int nbc, const tpsrhs_bc *bcs, long n, const double *U,
        const double *G, const double *N, double *F, double *S, double *FN) {
  constexpr int NEQ = PH::NEQ, DIM = PH::DIM;
  typename PH::Params prm;
  ChemDev chem;
  fill_plasma_params<PH::NSP>(prm, chem, disc, phys, nbc, bcs, [](const tpsrhs_table &t) {
    g_tables.emplace_back();
    TableDev td = table_coeffs(t, g_tables.back());
    td.x = g_tables.back().data();
    td.a = td.x + td.n;
    td.b = td.x + 2 * td.n;
    return td;
  });
  prm.chem = &chem;
  typename PH::PRef p = PH::pref(&prm);
  long bad = 0;
  for (long i = 0; i < n; i++) {
    double u[NEQ], g[NEQ * DIM], nrm[DIM];
    for (int eq = 0; eq < NEQ; eq++) u[eq] = U[eq * n + i];
    for (int k = 0; k < NEQ * DIM; k++) g[k] = G[k * n + i];
    for (int d = 0; d < DIM; d++) nrm[d] = N[i * DIM + d];
    double uc[NEQ];
    for (int eq = 0; eq < NEQ; eq++) uc[eq] = u[eq];
    PH::clamp_species(uc);
    const typename PH::State st = PH::make_state(p, uc);
    const double speed = PH::max_char_speed(p, uc, st);
    typename PH::FluxCoef fc;
    PH::flux_coeffs(p, uc, st, fc);
    double up[NEQ], src[NEQ], f[NEQ * DIM];
    PH::prim(p, u, up);
    PH::source(p, u, up, g, src);
    const double radius = PH::AXISYM ? 0.5 + 0.01 * (i % 7) : -1.0;
    if constexpr (PH::AXISYM) PH::axisym_source(p, u, up, g, radius, src);
    PH::total_flux(p, uc, st, fc, g, radius, f);
    for (int k = 0; k < NEQ * DIM; k++) F[k * n + i] = f[k];
    for (int eq = 0; eq < NEQ; eq++) S[eq * n + i] = src[eq];
    // viscous traces: interior face, then every boundary attribute (walls take two passes)
    double fn[NEQ], acc[NEQ];
    for (int eq = 0; eq < NEQ; eq++) acc[eq] = 0.0;
    for (int nb = 0; nb >= -nbc; nb--) {
      PH::visc_trace(p, nb, uc, g, nrm, radius, fn);
      for (int eq = 0; eq < NEQ; eq++) acc[eq] += fn[eq];
      if constexpr (!PH::AXISYM && DIM == 3) {  // the two-step form of the 3-D face kernel
        const int np = PH::visc_passes(p, nb);
        for (int pass = 0; pass < np; pass++) {
          double Us[NEQ], gv[DIM * DIM], gn[NEQ], f2[NEQ];
          typename PH::WallFlux w;
          typename PH::ViscCoef cf;
          PH::visc_pass_state(p, nb, pass, uc, nrm, Us, w);
          PH::visc_point_coeffs(p, Us, !w.species, cf);
          for (int eq = 0; eq < NEQ; eq++) {
            gn[eq] = 0.0;
            for (int d = 0; d < DIM; d++) gn[eq] += nrm[d] * g[eq + d * NEQ];
          }
          for (int a = 0; a < DIM; a++)
            for (int b = 0; b < DIM; b++) gv[a + b * DIM] = g[(1 + a) + b * NEQ];
          PH::visc_normal_flux_n(p, Us, cf, gv, gn, nrm, w, f2);
          for (int eq = 0; eq < NEQ; eq++) acc[eq] += 1e-3 * f2[eq];
        }
      }
      if (nb < 0) {  // ghost state + Riemann flux of the boundary attribute
        double ug[NEQ], fh[NEQ];
        PH::bc_ghost(p, p.bc[-nb - 1], uc, nrm, ug);
        PH::riemann_bc(p, p.bc[-nb - 1], uc, ug, nrm, fh);
        for (int eq = 0; eq < NEQ; eq++) acc[eq] += 1e-6 * fh[eq];
        double upb[NEQ];
        PH::bc_grad_prim(p, p.bc[-nb - 1], up, upb);
        acc[0] += 1e-12 * upb[NEQ - 1];
      }
    }
    {  // interior Riemann flux against a neighbouring state
      double u2[NEQ], fh[NEQ];
      for (int eq = 0; eq < NEQ; eq++) u2[eq] = uc[eq] * (1.0 + 1e-3 * ((eq + i) % 3));
      PH::riemann(p, uc, u2, nrm, fh);
      for (int eq = 0; eq < NEQ; eq++) acc[eq] += 1e-6 * fh[eq];
    }
    acc[0] += 1e-12 * (speed + PH::electric_conductivity(p, uc));
    for (int eq = 0; eq < NEQ; eq++) {
      FN[eq * n + i] = acc[eq];
      if (!std::isfinite(acc[eq]) || !std::isfinite(src[eq])) bad++;
    }
    for (int k = 0; k < NEQ * DIM; k++)
      if (!std::isfinite(f[k])) bad++;
  }
  return bad == 0 ? 0 : 2;
}
}  // namespace

// geometry: 3 = 3-D, 2 = planar 2-D, 1 = axisymmetric.  Returns 0 fine, 1 instantiation not in this harness, 2 non-finite output.
extern "C" int hostphys_run(int geometry, int nsp, int ambi, int two_t, int transport, const tpsrhs_disc *disc,
                            const tpsrhs_physics *phys, int nbc, const tpsrhs_bc *bcs, long n, const double *U, const double *G,
                            const double *N, double *F, double *S, double *FN) {
  g_tables.clear();
  g_tables.reserve(64);
#define CASE(GEO, DIM, NVEL, NSP, AMBI, TWOT, TR)                                                                    \
  if (geometry == GEO && nsp == NSP && (ambi != 0) == AMBI && (two_t != 0) == TWOT && transport == TR)                \
    return run<PlasmaPhys<DIM, NVEL, NSP, AMBI, TWOT, TR>>(disc, phys, nbc, bcs, n, U, G, N, F, S, FN);
  CASE(3, 3, 3, 3, true, false, TRANSPORT_ARGON_MINIMAL)   // the metric's workload
  CASE(3, 3, 3, 3, false, true, TRANSPORT_ARGON_MINIMAL)
  CASE(3, 3, 3, 3, true, true, TRANSPORT_ARGON_MIXTURE)
  CASE(3, 3, 3, 7, false, false, TRANSPORT_ARGON_MIXTURE)  // the instantiation of DESIGN.md section 5
  CASE(3, 3, 3, 4, true, true, TRANSPORT_ARGON_MIXTURE)    // its four-species sibling
  CASE(3, 3, 3, 6, false, true, TRANSPORT_ARGON_MIXTURE)
  CASE(3, 3, 3, 8, true, true, TRANSPORT_CONSTANT)
  CASE(2, 2, 2, 3, true, true, TRANSPORT_CONSTANT)
  CASE(2, 2, 2, 5, false, false, TRANSPORT_ARGON_MIXTURE)
  CASE(1, 2, 3, 3, true, true, TRANSPORT_ARGON_MINIMAL)    // cfg5
  CASE(1, 2, 3, 6, false, true, TRANSPORT_ARGON_MIXTURE)   // torch6
#undef CASE
  return 1;
}

#ifdef HOSTPHYS_MAIN
// stand-alone driver for MemorySanitizer (which must own the whole process): reads the cases that run_cases.py wrote with
// HOSTPHYS_DUMP=<file> -- [header ints | disc | physics | bcs | U | G | N] per case, PODs as raw bytes (the cases carry no
// table pointers) -- and checks that every output is initialised.
#include <sanitizer/msan_interface.h>
int main(int argc, char **argv) {
  if (argc < 2) return 64;
  FILE *f = std::fopen(argv[1], "rb");
  if (!f) return 65;
  int ncase = 0;
  for (;;) {
    long hdr[8];
    if (std::fread(hdr, sizeof(long), 8, f) != 8) break;
    const int geometry = hdr[0], nsp = hdr[1], ambi = hdr[2], two_t = hdr[3], tr = hdr[4], nbc = hdr[5], neq = hdr[7];
    const long n = hdr[6];
    const int dim = geometry == 3 ? 3 : 2;
    tpsrhs_disc disc;
    tpsrhs_physics phys;
    std::vector<tpsrhs_bc> bcs(nbc);
    if (std::fread(&disc, sizeof disc, 1, f) != 1 || std::fread(&phys, sizeof phys, 1, f) != 1 ||
        std::fread(bcs.data(), sizeof(tpsrhs_bc), nbc, f) != static_cast<size_t>(nbc))
      return 66;
    std::vector<double> U(neq * n), G(dim * neq * n), N(n * dim), F(dim * neq * n), S(neq * n), FN(neq * n);
    if (std::fread(U.data(), 8, U.size(), f) != U.size() || std::fread(G.data(), 8, G.size(), f) != G.size() ||
        std::fread(N.data(), 8, N.size(), f) != N.size())
      return 67;
    const int rc = hostphys_run(geometry, nsp, ambi, two_t, tr, &disc, &phys, nbc, bcs.data(), n, U.data(), G.data(), N.data(), F.data(),
                                S.data(), FN.data());
    if (rc != 0) {
      std::printf("case %d: hostphys_run returned %d\n", ncase, rc);
      return 2;
    }
    __msan_check_mem_is_initialized(F.data(), F.size() * 8);
    __msan_check_mem_is_initialized(S.data(), S.size() * 8);
    __msan_check_mem_is_initialized(FN.data(), FN.size() * 8);
    ncase++;
  }
  std::printf("MSAN CLEAN: %d cases\n", ncase);
  return 0;
}
#endif
