// libtpsrhs.so -- implementation of include/tpsrhs.h for gfx950 (MI355X).
//
// Host side: builds the face topology and the 1-D operator tables, keeps every field resident in
// HBM, and enqueues the three sweeps of kernels.hpp on one HIP stream per operator.  There is no
// CPU fallback: without a HIP device tpsrhs_create returns TPSRHS_ERR_NO_DEVICE.
#include "operator.hpp"
#include "physics_dryair.hpp"
#include "physics_dryair_axisym.hpp"
#include "physics_plasma.hpp"
#include "plasma_params_host.hpp"

#include <dlfcn.h>

#include <cmath>
#include <map>
#include <mutex>
#include <new>
#include <thread>

// The plasma kernel families (plasma_family.hpp) are shared objects of their own, libtpsrhs_<unit>.so next to this library,
// loaded when an operator of that family is created: load_family below.
void pick_dryair_axisym(tpsrhs_operator *op);
void pick_lte_axisym(tpsrhs_operator *op);
void pick_dryair_les(tpsrhs_operator *op);

static thread_local std::string g_last_error;

namespace {

// One shared object per plasma kernel family -- (geometry, species count, ambipolar or not, orders 1..3 or 4..5):
// libtpsrhs_plasma_<geo>_n<species>[a][_hi].so in the directory of this library, or in a directory of the colon-separated
// TPSRHS_FAMILY_PATH (development: A/B builds of one family).  dlopen'ed at the first operator that needs it and kept
// for the life of the process (the operator holds pointers to its launch functions).  The entry point is C and
// returns a status: no exception crosses the boundary.  A missing family is a loud error, never another code path.
typedef int (*family_pick_fn)(tpsrhs_operator *, int, int, char *, int);
family_pick_fn load_family(const std::string &unit) {
  static std::mutex mu;
  static std::map<std::string, family_pick_fn> loaded;
  std::lock_guard<std::mutex> lock(mu);
  auto it = loaded.find(unit);
  if (it != loaded.end()) return it->second;
  std::vector<std::string> dirs;
  if (const char *env = std::getenv("TPSRHS_FAMILY_PATH")) {
    std::string e(env);
    size_t pos = 0;
    while (pos <= e.size()) {
      const size_t c = e.find(':', pos);
      const std::string d = e.substr(pos, c == std::string::npos ? std::string::npos : c - pos);
      if (!d.empty()) dirs.push_back(d);
      if (c == std::string::npos) break;
      pos = c + 1;
    }
  }
  Dl_info info;
  if (dladdr(reinterpret_cast<const void *>(&load_family), &info) && info.dli_fname) {
    const std::string self(info.dli_fname);
    const size_t slash = self.rfind('/');
    dirs.push_back(slash == std::string::npos ? std::string(".") : self.substr(0, slash));
  }
  std::string tried;
  for (const std::string &d : dirs) {
    const std::string path = d + "/libtpsrhs_" + unit + ".so";
    if (void *h = dlopen(path.c_str(), RTLD_NOW | RTLD_LOCAL)) {
      const std::string sym = "pick_" + unit;
      auto fn = reinterpret_cast<family_pick_fn>(dlsym(h, sym.c_str()));
      if (!fn) throw std::runtime_error(path + " does not export " + sym);
      loaded[unit] = fn;
      return fn;
    }
    const char *why = dlerror();
    tried += "\n  " + path + ": " + (why ? why : "not loadable");
  }
  throw Unsupported("kernel family " + unit + " is not available (run __graft_entry__.build()):" + tried);
}

// LinearTable::LinearTable (src/table.cpp:39-50): interval coefficients (plasma_params_host.hpp), uploaded with the abscissae
TableDev upload_table(tpsrhs_operator *op, const tpsrhs_table &t) {
  std::vector<double> buf;
  TableDev td = table_coeffs(t, buf);
  double *d = dev_upload(buf);
  op->d_extra.push_back(d);
  td.x = d;
  td.a = d + td.n;
  td.b = d + 2 * td.n;
  return td;
}

// the parameter block of the plasma kernels (plasma_params_host.hpp) into the operator, the chemistry block to the device
template <int NSP>
void fill_plasma_params(tpsrhs_operator *op, const tpsrhs_disc *disc, const tpsrhs_physics *phys, int num_bcs,
                        const tpsrhs_bc *bcs) {
  static_assert(sizeof(PlasmaParams<NSP>) <= sizeof(op->params), "parameter block too large");
  PlasmaParams<NSP> &p = *new (op->params) PlasmaParams<NSP>;
  std::unique_ptr<ChemDev> c(new ChemDev);
  try {
    tpsrhs::fill_plasma_params<NSP>(p, *c, disc, phys, num_bcs, bcs, [op](const tpsrhs_table &t) { return upload_table(op, t); });
  } catch (const UnsupportedConfig &e) {
    throw Unsupported(e.what());
  }
  ChemDev *dc = dev_alloc<ChemDev>(1);
  HIP_CHECK(hipMemcpy(dc, c.get(), sizeof(ChemDev), hipMemcpyHostToDevice));
  op->d_chem = dc;
  p.chem = dc;
}

// mfem::Mesh::GetElementSize(e, 1): the smallest singular value of the Jacobian at the element centre
// (src/rhs_operator.cpp:154, src/face_integrator.cpp:253).  sqrt of the smallest eigenvalue of J^T J by cyclic
// Jacobi rotations.  verts: [ne][2^dim][dim] lexicographic corners.
std::vector<double> element_sizes(const std::vector<double> &verts, int dim, int ne) {
  std::vector<double> h(ne);
  const int nv = 1 << dim;
  for (int e = 0; e < ne; e++) {
    const double *V = &verts[static_cast<size_t>(e) * nv * dim];
    double J[3][3] = {{0, 0, 0}, {0, 0, 0}, {0, 0, 0}};  // J[i][a] = d x_i / d xi_a at the centre
    for (int v = 0; v < nv; v++)
      for (int a = 0; a < dim; a++) {
        const double sgn = ((v >> a) & 1) ? 1.0 : -1.0;
        for (int i = 0; i < dim; i++) J[i][a] += sgn * V[v * dim + i] / (nv / 2);
      }
    double A[3][3];
    for (int a = 0; a < dim; a++)
      for (int b = 0; b < dim; b++) {
        double t = 0.0;
        for (int i = 0; i < dim; i++) t += J[i][a] * J[i][b];
        A[a][b] = t;
      }
    for (int sweep = 0; sweep < 30; sweep++) {
      double off = 0.0;
      for (int a = 0; a < dim; a++)
        for (int b = a + 1; b < dim; b++) off += A[a][b] * A[a][b];
      if (off == 0.0) break;
      for (int a = 0; a < dim; a++)
        for (int b = a + 1; b < dim; b++) {
          if (A[a][b] == 0.0) continue;
          const double theta = (A[b][b] - A[a][a]) / (2.0 * A[a][b]);
          const double t = (theta >= 0.0 ? 1.0 : -1.0) / (std::fabs(theta) + std::sqrt(theta * theta + 1.0));
          const double c = 1.0 / std::sqrt(t * t + 1.0), sn = t * c;
          for (int k = 0; k < dim; k++) {  // A <- A R
            const double aka = A[k][a], akb = A[k][b];
            A[k][a] = c * aka - sn * akb;
            A[k][b] = sn * aka + c * akb;
          }
          for (int k = 0; k < dim; k++) {  // A <- R^T A
            const double aak = A[a][k], abk = A[b][k];
            A[a][k] = c * aak - sn * abk;
            A[b][k] = sn * aak + c * abk;
          }
        }
    }
    double lmin = A[0][0];
    for (int a = 1; a < dim; a++) lmin = std::min(lmin, A[a][a]);
    h[e] = std::sqrt(std::max(lmin, 0.0));
  }
  return h;
}

// Inverse element mass matrices of the non-collocated pair: M_jk = sum_q w_q det J(q) phi_j(q) phi_k(q) with the
// order-2p rule of the same family (MassIntegrator, src/rhs_operator.cpp:179-187), inverted per element
// (Cholesky; the reference calls DenseMatrix::Invert, :205-212) -- the role of Me_inv, built once, on the host,
// on all cores.  verts: [ne][2^dim][dim] lexicographic corners.
std::vector<double> inverse_mass_matrices(const std::vector<double> &verts, int dim, int order, int ne) {
  const Tables1D t = make_tables(order, dim, 1, 1);
  const int n1 = order + 1, qv = rule_points(1, 2 * order);
  const int npe = (dim == 3) ? n1 * n1 * n1 : n1 * n1, nq = (dim == 3) ? qv * qv * qv : qv * qv, nv = 1 << dim;
  // Phi[q][j]: tensor basis at the tensor rule
  std::vector<double> Phi(static_cast<size_t>(nq) * npe), wq(nq), xq(static_cast<size_t>(nq) * dim);
  for (int q = 0; q < nq; q++) {
    int qi[3] = {q % qv, (q / qv) % qv, q / (qv * qv)};
    wq[q] = 1.0;
    for (int d = 0; d < dim; d++) {
      wq[q] *= t.wv[qi[d]];
      xq[static_cast<size_t>(q) * dim + d] = t.xv[qi[d]];
    }
    for (int j = 0; j < npe; j++) {
      int ji[3] = {j % n1, (j / n1) % n1, j / (n1 * n1)};
      double v = 1.0;
      for (int d = 0; d < dim; d++) v *= t.Bv[qi[d] * n1 + ji[d]];
      Phi[static_cast<size_t>(q) * npe + j] = v;
    }
  }
  std::vector<double> out(static_cast<size_t>(ne) * npe * npe);
  auto work = [&](int e_begin, int e_end) {
    std::vector<double> M(static_cast<size_t>(npe) * npe), Wp(static_cast<size_t>(nq) * npe), L(static_cast<size_t>(npe) * npe);
    for (int e = e_begin; e < e_end; e++) {
      const double *V = &verts[static_cast<size_t>(e) * nv * dim];
      for (int q = 0; q < nq; q++) {  // det J at the point (multilinear map of the 2^dim corners)
        const double *xi = &xq[static_cast<size_t>(q) * dim];
        double J[9] = {0};
        for (int c = 0; c < nv; c++) {
          for (int m2 = 0; m2 < dim; m2++) {
            double g = 1.0;  // d/dxi_m2 of the corner's shape function
            for (int d = 0; d < dim; d++) {
              const int bit = (c >> d) & 1;
              if (d == m2)
                g *= bit ? 1.0 : -1.0;
              else
                g *= bit ? xi[d] : 1.0 - xi[d];
            }
            for (int i = 0; i < dim; i++) J[i + m2 * dim] += g * V[c * dim + i];
          }
        }
        const double det = (dim == 2) ? J[0] * J[3] - J[2] * J[1]
                                      : J[0] * (J[4] * J[8] - J[7] * J[5]) - J[3] * (J[1] * J[8] - J[7] * J[2]) +
                                            J[6] * (J[1] * J[5] - J[4] * J[2]);
        const double wd = wq[q] * det;
        for (int j = 0; j < npe; j++) Wp[static_cast<size_t>(q) * npe + j] = wd * Phi[static_cast<size_t>(q) * npe + j];
      }
      std::fill(M.begin(), M.end(), 0.0);
      for (int q = 0; q < nq; q++)
        for (int j = 0; j < npe; j++) {
          const double pj = Phi[static_cast<size_t>(q) * npe + j];
          const double *wrow = &Wp[static_cast<size_t>(q) * npe];
          double *mrow = &M[static_cast<size_t>(j) * npe];
          for (int k = 0; k < npe; k++) mrow[k] += pj * wrow[k];
        }
      // Cholesky M = L L^T, then M^-1 = L^-T L^-1
      std::fill(L.begin(), L.end(), 0.0);
      for (int j = 0; j < npe; j++) {
        double d = M[static_cast<size_t>(j) * npe + j];
        for (int k = 0; k < j; k++) d -= L[static_cast<size_t>(j) * npe + k] * L[static_cast<size_t>(j) * npe + k];
        if (!(d > 0.0)) throw std::runtime_error("element mass matrix is not positive definite (inverted element?)");
        const double ljj = std::sqrt(d);
        L[static_cast<size_t>(j) * npe + j] = ljj;
        for (int i = j + 1; i < npe; i++) {
          double v = M[static_cast<size_t>(i) * npe + j];
          for (int k = 0; k < j; k++) v -= L[static_cast<size_t>(i) * npe + k] * L[static_cast<size_t>(j) * npe + k];
          L[static_cast<size_t>(i) * npe + j] = v / ljj;
        }
      }
      // Linv (lower) in M's storage
      std::fill(M.begin(), M.end(), 0.0);
      for (int j = 0; j < npe; j++) {
        M[static_cast<size_t>(j) * npe + j] = 1.0 / L[static_cast<size_t>(j) * npe + j];
        for (int i = j + 1; i < npe; i++) {
          double v = 0.0;
          for (int k = j; k < i; k++) v -= L[static_cast<size_t>(i) * npe + k] * M[static_cast<size_t>(k) * npe + j];
          M[static_cast<size_t>(i) * npe + j] = v / L[static_cast<size_t>(i) * npe + i];
        }
      }
      double *o = &out[static_cast<size_t>(e) * npe * npe];
      for (int i = 0; i < npe; i++)
        for (int j = 0; j <= i; j++) {
          double v = 0.0;
          for (int k = i; k < npe; k++) v += M[static_cast<size_t>(k) * npe + i] * M[static_cast<size_t>(k) * npe + j];
          o[static_cast<size_t>(i) * npe + j] = o[static_cast<size_t>(j) * npe + i] = v;
        }
    }
  };
  const int nthreads = std::max(1, std::min<int>(static_cast<int>(std::thread::hardware_concurrency()), std::min(32, ne / 64 + 1)));
  std::vector<std::thread> pool;
  std::vector<std::exception_ptr> errs(nthreads);
  for (int t2 = 0; t2 < nthreads; t2++) {
    const int b = static_cast<int>(static_cast<int64_t>(ne) * t2 / nthreads), en = static_cast<int>(static_cast<int64_t>(ne) * (t2 + 1) / nthreads);
    pool.emplace_back([&, b, en, t2] {
      try {
        work(b, en);
      } catch (...) {
        errs[t2] = std::current_exception();
      }
    });
  }
  for (auto &th : pool) th.join();
  for (auto &e : errs)
    if (e) std::rethrow_exception(e);
  return out;
}

void setup(tpsrhs_operator *op, const tpsrhs_mesh *mesh, const tpsrhs_disc *disc, const tpsrhs_physics *phys,
           int num_bcs, const tpsrhs_bc *bcs, const tpsrhs_runtime *rt) {
  // basisType / integrationRule (src/M2ulPhyS.cpp:557-572): the collocated Gauss-Legendre pair (0, 0) of the
  // cylinder / wedge / torch inputs, or the Gauss-Lobatto pair (1, 1), the reference's defaults (:2671-2672)
  if (disc->basis_type == TPSRHS_BASIS_GAUSS_LEGENDRE && disc->int_rule_type == 0)
    op->nc = 0;
  else if (disc->basis_type == TPSRHS_BASIS_GAUSS_LOBATTO && disc->int_rule_type == 1)
    op->nc = 1;
  else
    throw Unsupported("basisType / integrationRule: built pairs are (0, 0) Gauss-Legendre and (1, 1) Gauss-Lobatto");
  if (op->nc && disc->axisymmetric) throw Unsupported("the Gauss-Lobatto pair is built for the planar 2-D and the 3-D formulation");
  const bool plasma = phys->working_fluid == TPSRHS_USER_DEFINED;
  if (disc->axisymmetric && mesh->dim != 2) throw std::invalid_argument("the axisymmetric formulation needs a 2-D mesh");
  const bool lte = phys->working_fluid == TPSRHS_LTE_FLUID;
  if (phys->working_fluid != TPSRHS_DRY_AIR && !plasma && !lte) throw std::invalid_argument("unknown working_fluid");
  if (lte && (!disc->axisymmetric || op->nc))
    throw Unsupported("WorkingFluid::LTE_FLUID (one-dimensional tables): built for the axisymmetric formulation, Gauss-Legendre pair");
  if (disc->use_roe && (mesh->dim != 2 || disc->axisymmetric || plasma || lte))
    throw Unsupported("flow/useRoe: Eval_Roe of the reference is 2-D, single-species, not axisymmetric "
                      "(src/riemann_solver.cpp:117-206)");
  if (phys->eq_system != TPSRHS_EULER && phys->eq_system != TPSRHS_NS) throw Unsupported("NS_PASSIVE is out of scope");
  if (num_bcs > (plasma ? PLASMA_MAXBC : MAXBC)) throw Unsupported("too many boundary conditions");
  for (int i = 0; i < num_bcs; i++) {
    const tpsrhs_bc &b = bcs[i];
    const bool nr = is_non_reflecting(b.category, b.type);
    if (nr && (plasma || lte || disc->axisymmetric))
      throw Unsupported("non-reflecting inlet/outlet types: perfect gas (dry air), not axisymmetric -- the reference's "
                        "characteristic algebra (src/outletBC.cpp:573-1027)");
    if (nr && (b.type == TPSRHS_SUB_MF_NR || b.type == TPSRHS_SUB_MF_NR_PW) && !(b.data[7] > 0.0))
      throw std::invalid_argument("mass-flow outlet: data[7] must hold the total patch area");
    const bool face_inlet = is_face_inlet(b.category, b.type);
    if (face_inlet && (mesh->dim != 3 || disc->axisymmetric))
      throw Unsupported("face-relative inlets (subsonicFaceBasedX/Y/Z): 3-D (the reference's face frame has three components)");
    const bool ok = nr || face_inlet || (b.category == TPSRHS_INLET && b.type == TPSRHS_SUB_DENS_VEL) ||
                    (b.category == TPSRHS_OUTLET && b.type == TPSRHS_SUB_P) ||
                    (b.category == TPSRHS_WALL &&
                     (b.type == TPSRHS_INV || b.type == TPSRHS_SLIP || b.type == TPSRHS_VISC_ADIAB || b.type == TPSRHS_VISC_ISOTH ||
                      (b.type == TPSRHS_VISC_GNRL && plasma)));
    if (!ok) throw Unsupported("boundary condition type outside the hot-path scope (attribute " + std::to_string(b.attribute) + ")");
    if (b.category == TPSRHS_WALL && b.type == TPSRHS_VISC_GNRL) {
      // the combinations the reference's input parser lets through (src/M2ulPhyS.cpp:3515-3582)
      const int hc = static_cast<int>(b.data[2]), ec = static_cast<int>(b.data[3]);
      const bool two_t = phys->mixture.two_temperature != 0;
      const bool hvy_ok = (hc == TPSRHS_ISOTH || hc == TPSRHS_ADIAB);
      const bool elec_ok = two_t ? (ec == TPSRHS_ISOTH || ec == TPSRHS_ADIAB || ec == TPSRHS_SHTH) : (ec == TPSRHS_SHTH);
      if (!hvy_ok) throw std::invalid_argument("viscous_general wall: heavy thermal condition must be isothermal or adiabatic");
      if (!elec_ok)
        throw std::invalid_argument("viscous_general wall: electron thermal condition not understood "
                                    "(single-temperature plasmas accept the sheath condition only)");
      if (ec == TPSRHS_SHTH && !phys->mixture.ambipolar)
        throw std::invalid_argument("viscous_general wall: plasma must be ambipolar for the sheath condition");
    }
  }
  op->dim = mesh->dim;
  op->order = disc->order;
  op->nvel = disc->axisymmetric ? 3 : op->dim;
  op->neq = op->nvel + 2;
  if (plasma) {
    const tpsrhs_perfect_mixture &mx = phys->mixture;
    if (!mx.is_electron_included) throw Unsupported("USER_DEFINED fluids without electrons are not built");
    if (mx.num_species < 3 || mx.num_species > TPSRHS_MAXSPECIES)
      throw Unsupported("USER_DEFINED fluids: 3 to 8 species (electron, background and 1 to 6 others)");
    if (mx.num_species > 7 && phys->transport_model == TPSRHS_ARGON_MIXTURE)  // the reference asserts, src/gas_transport.cpp:905-911
      throw Unsupported("argon_mixture transport supports at most 7 species (Ar, Ar.+1, Ar_m, Ar_r, Ar_p, Ar_h, E)");
    if (mx.num_species != 3 && phys->transport_model == TPSRHS_ARGON_MINIMAL)
      throw Unsupported("argon_minimal transport is the ternary (Ar, Ar.+1, E) model");
    if (phys->transport_model != TPSRHS_CONSTANT && phys->transport_model != TPSRHS_ARGON_MINIMAL &&
        phys->transport_model != TPSRHS_ARGON_MIXTURE)
      throw Unsupported("transport model outside the built scope (constant, argon_minimal, argon_mixture)");
    op->neq = op->nvel + 2 + (mx.ambipolar ? mx.num_species - 2 : mx.num_species - 1) + (mx.two_temperature ? 1 : 0);
  }
  op->phys = *phys;

  int ndev = 0;
  if (hipGetDeviceCount(&ndev) != hipSuccess || ndev == 0) {
    struct NoDevice : std::runtime_error {
      NoDevice() : std::runtime_error("no HIP device visible; tpsrhs has no CPU path") {}
    };
    throw NoDevice();
  }
  op->device = rt ? rt->device : 0;
  HIP_CHECK(hipSetDevice(op->device));
  op->stream = rt ? static_cast<hipStream_t>(rt->stream) : nullptr;
  if (const char *env = std::getenv("TPSRHS_SWEEP_ALT")) op->sweep_alt = env[0] != '0';  // (A/B switch; default on)
  if (const char *env = std::getenv("TPSRHS_FUSE_TRACES")) op->fuse_traces = env[0] != '0';  // (A/B switch; default on)
  op->halo = rt ? rt->halo : nullptr;
  op->halo_ctx = rt ? rt->halo_ctx : nullptr;
  op->reduce = rt ? rt->reduce : nullptr;
  op->reduce_ctx = rt ? rt->reduce_ctx : nullptr;

  op->topo = build_topology(*mesh, num_bcs, bcs);
  const Topology &tp = op->topo;
  if (tp.num_shared > 0 && !op->halo) throw std::invalid_argument("mesh has shared faces but runtime.halo is NULL");
  op->ne = tp.ne;
  op->nfaces = tp.nfaces;
  const int n1 = op->order + 1, q1 = rule_points(op->nc, (op->dim - 1) + 2 * op->order);
  op->nf = (op->dim == 3) ? n1 * n1 : n1;
  op->nq = (op->dim == 3) ? q1 * q1 : q1;
  const int npe = (op->dim == 3) ? n1 * n1 * n1 : n1 * n1;
  op->ndofs = static_cast<int64_t>(op->ne) * npe;

  // Viscous sponge of the 2-D kernels with the heavy interface (the mixtures planar and axisymmetric, axisymmetric dry air
  // and table gas): the plane travels in MeshDev, the closures scale their coefficients (src/fluxes.cpp:232-246)
  const bool heavy2d = op->dim == 2 && (plasma || disc->axisymmetric);
  if ((plasma || lte || disc->axisymmetric) && phys->sgs.model_type != TPSRHS_SGS_NONE)
    throw Unsupported("sub-grid scale models: built for dry air in 3-D");
  if (phys->visc_sponge.enabled && (plasma || lte || disc->axisymmetric)) {
    if (!heavy2d || op->nc)
      throw Unsupported("viscous sponge for mixtures: built for the planar 2-D and the axisymmetric formulation, Gauss-Legendre pair");
    const tpsrhs_visc_sponge &v = phys->visc_sponge;
    if (!(v.width > 0.0)) throw std::invalid_argument("visc_sponge.width must be positive");
    const double nmag = std::sqrt(v.normal[0] * v.normal[0] + v.normal[1] * v.normal[1]);  // normalised: src/fluxes.cpp:77-90
    if (!(nmag > 0.0)) throw std::invalid_argument("visc_sponge.normal is zero");
    op->vs2d.enabled = 1;
    for (int k = 0; k < 2; k++) {
      op->vs2d.n[k] = v.normal[k] / nmag;
      op->vs2d.p[k] = v.point[k];
    }
    op->vs2d.width = v.width;
    op->vs2d.ratio = v.ratio;
  }
  if (plasma) {
    const int nsp = phys->mixture.num_species;
    const bool ambi = phys->mixture.ambipolar != 0, two_t = phys->mixture.two_temperature != 0;
    switch (nsp) {
      case 3: fill_plasma_params<3>(op, disc, phys, num_bcs, bcs); break;
      case 4: fill_plasma_params<4>(op, disc, phys, num_bcs, bcs); break;
      case 5: fill_plasma_params<5>(op, disc, phys, num_bcs, bcs); break;
      case 6: fill_plasma_params<6>(op, disc, phys, num_bcs, bcs); break;
      case 7: fill_plasma_params<7>(op, disc, phys, num_bcs, bcs); break;
      default: fill_plasma_params<8>(op, disc, phys, num_bcs, bcs); break;  // MAXSPECIES of the reference's device build
    }
    const int tr = (phys->transport_model == TPSRHS_CONSTANT)
                       ? TRANSPORT_CONSTANT
                       : (phys->transport_model == TPSRHS_ARGON_MINIMAL ? TRANSPORT_ARGON_MINIMAL : TRANSPORT_ARGON_MIXTURE);
    // one shared object per (geometry, species count, ambipolar) family, and a second one for the polynomial orders
    // 4 and 5: every species count up to MAXSPECIES = 8 (MAXEQUATIONS = 13) of the reference's device build
    // (src/dataStructures.hpp:41-65), ambipolar or not
    const char *geo = (op->dim == 3) ? "3d" : (disc->axisymmetric ? "axi" : "2d");
    const std::string unit = std::string("plasma_") + geo + "_n" + std::to_string(nsp) + (ambi ? "a" : "") + ((op->order >= 4 && !op->nc) ? "_hi" : "");
    char msg[512] = {0};
    const int rc = load_family(unit)(op, two_t ? 1 : 0, tr, msg, static_cast<int>(sizeof msg));
    if (rc == TPSRHS_ERR_UNSUPPORTED) throw Unsupported(msg);
    if (rc == TPSRHS_ERR_INVALID_ARGUMENT) throw std::invalid_argument(msg);
    if (rc != TPSRHS_OK) throw DeviceError(msg);
  } else {
    // (the table gas extends the dry-air block: boundary conditions and switches are shared)
    static_assert(sizeof(LteParams) <= sizeof(op->params), "parameter block");
    LteParams *lp = lte ? new (op->params) LteParams : nullptr;
    if (lp) std::memset(static_cast<void *>(lp), 0, sizeof(LteParams));
    DryAirParams &d = lp ? *static_cast<DryAirParams *>(lp) : *new (op->params) DryAirParams;
    if (!lp) std::memset(&d, 0, sizeof(d));
    if (lp) {
      const tpsrhs_lte &in = phys->lte;
      // LinearTable with xLogScale = fLogScale = false, as the reference hard-codes for these tables (src/M2ulPhyS.cpp:176-255);
      // the inverse table T(e) below and its search aid assume linear axes
      for (const tpsrhs_table *t : {&in.energy_table, &in.gas_constant_table, &in.sound_speed_table, &in.viscosity_table,
                                    &in.conductivity_table, &in.electric_conductivity_table})
        if (t->x_log_scale || t->f_log_scale) throw std::invalid_argument("lte tables: linear scales only (x_log_scale = f_log_scale = 0)");
      lp->tab_e = upload_table(op, in.energy_table);
      lp->tab_R = upload_table(op, in.gas_constant_table);
      lp->tab_c = upload_table(op, in.sound_speed_table);
      for (int k = 1; k < in.energy_table.n_data; k++)
        if (!(in.energy_table.f_data[k] > in.energy_table.f_data[k - 1]))
          throw std::invalid_argument("lte.energy_table: e(T) must increase (the inverse table T(e) is its transpose)");
      tpsrhs_table rev = in.energy_table;  // "Construct e -> T table from T -> e", src/M2ulPhyS.cpp:193-200
      rev.x_data = in.energy_table.f_data;
      rev.f_data = in.energy_table.x_data;
      lp->tab_T = upload_table(op, rev);
      {  // search aid of the inverse table: the interval of the left edge of uniform bins over the energy axis
        const int N = in.energy_table.n_data, nh = 4 * (N - 1);
        const double *ex = in.energy_table.f_data;
        const double de = (ex[N - 1] - ex[0]) / nh;
        std::vector<int> hint(nh);
        int idx = 0;
        for (int j = 0; j < nh; j++) {
          const double left = ex[0] + j * de;
          while (idx < N - 2 && ex[idx + 1] < left) idx++;
          hint[j] = idx;
        }
        int *dh = dev_upload(hint);
        op->d_extra.push_back(dh);
        lp->ehint = dh;
        lp->nhint = nh;
        lp->e0 = ex[0];
        lp->inv_de = 1.0 / de;
      }
      auto same_grid = [](const tpsrhs_table &a, const tpsrhs_table &b) {
        if (a.n_data != b.n_data || a.x_log_scale != b.x_log_scale) return 0;
        for (int k = 0; k < a.n_data; k++)
          if (a.x_data[k] != b.x_data[k]) return 0;
        return 1;
      };
      lp->thermo_same_grid = same_grid(in.energy_table, in.gas_constant_table) && same_grid(in.energy_table, in.sound_speed_table);
      lp->trans_same_grid = same_grid(in.viscosity_table, in.conductivity_table);
      lp->tab_mu = upload_table(op, in.viscosity_table);
      lp->tab_k = upload_table(op, in.conductivity_table);
      lp->tab_sigma = upload_table(op, in.electric_conductivity_table);
      lp->radiation = phys->radiation.model;
      if (lp->radiation == TPSRHS_NET_EMISSION) lp->tab_nec = upload_table(op, phys->radiation.nec_table);
    }
    d.gamma = phys->dry_air.specific_heat_ratio;
    d.Rg = phys->dry_air.gas_constant;
    d.inv_Rg = 1.0 / d.Rg;
    d.visc_mult = phys->dry_air.visc_mult;
    d.bulk_mult = phys->dry_air.bulk_visc_mult;
    d.C1 = phys->dry_air.sutherland_C1;
    d.S0 = phys->dry_air.sutherland_S0;
    d.cp_div_pr = d.gamma * d.Rg / (phys->dry_air.sutherland_Pr * (d.gamma - 1.0));
    d.eq_system = phys->eq_system;
    d.use_bc_in_grad = disc->use_bc_in_grad;
    d.use_roe = disc->use_roe ? 1 : 0;
    d.ref_length = disc->ref_length > 0.0 ? disc->ref_length : 1.0;  // config.refLength default, run_configuration.cpp:66
    d.num_bcs = num_bcs;
    for (int i = 0; i < num_bcs; i++) {
      d.bc[i].category = bcs[i].category;
      d.bc[i].type = bcs[i].type;
      for (int k = 0; k < 4 + TPSRHS_MAXSPECIES; k++) d.bc[i].data[k] = bcs[i].data[k];
    }
    bool any_nr = false;
    for (int i = 0; i < num_bcs; i++) any_nr = any_nr || is_non_reflecting(bcs[i].category, bcs[i].type);
    // Fluxes: sub-grid scale model and viscous sponge (src/fluxes.cpp:223-246) -> the LES flavour of the kernels
    const bool les = !heavy2d && (phys->sgs.model_type != TPSRHS_SGS_NONE || phys->visc_sponge.enabled);
    if (les) {
      if (phys->sgs.model_type < 0 || phys->sgs.model_type > TPSRHS_SGS_SIGMA)
        throw std::invalid_argument("unknown sgs.model_type");
      if (disc->axisymmetric || op->nc || any_nr)
        throw Unsupported("sub-grid scale model / viscous sponge: built for dry air, planar 2-D and 3-D, Gauss-Legendre pair, "
                          "reflecting boundary types");
      if (phys->sgs.model_type != TPSRHS_SGS_NONE && op->dim != 3)
        throw Unsupported("sub-grid scale models need dim == 3 (the reference's strain tensor indexes three directions)");
      if (phys->visc_sponge.enabled && !(phys->visc_sponge.width > 0.0))
        throw std::invalid_argument("visc_sponge.width must be positive");
      d.sgs_type = phys->sgs.model_type;
      d.sgs_const = phys->sgs.model_const > 0.0 ? phys->sgs.model_const  // defaults of src/M2ulPhyS.cpp:2693-2698
                                                : (d.sgs_type == TPSRHS_SGS_SMAGORINSKY ? 0.12 : (d.sgs_type == TPSRHS_SGS_SIGMA ? 0.135 : 0.0));
      d.sgs_floor = phys->sgs.model_floor;
      d.vs_enabled = phys->visc_sponge.enabled ? 1 : 0;
      // "ensure normal is actually a unit normal": the constructor every CPU build of the reference uses, and its host
      // fluxClass, normalise vsd_.n (src/fluxes.cpp:77-90, src/M2ulPhyS.cpp:612-616); the device constructor
      // (:98-125) copies it as given.  The CPU path is this library's referent.
      double nmag = 0.0;
      for (int k = 0; k < op->dim; k++) nmag += phys->visc_sponge.normal[k] * phys->visc_sponge.normal[k];
      nmag = std::sqrt(nmag);
      if (phys->visc_sponge.enabled && !(nmag > 0.0)) throw std::invalid_argument("visc_sponge.normal is zero");
      for (int k = 0; k < 3; k++) {
        d.vs_n[k] = (k < op->dim && nmag > 0.0) ? phys->visc_sponge.normal[k] / nmag : 0.0;  // vsd.n / vsd.p: dim entries, the rest 0
        d.vs_p[k] = (k < op->dim) ? phys->visc_sponge.point[k] : 0.0;
      }
      d.vs_width = phys->visc_sponge.enabled ? phys->visc_sponge.width : 1.0;
      d.vs_ratio = phys->visc_sponge.enabled ? phys->visc_sponge.ratio : 1.0;
      std::vector<double> delta = mesh->elem_size ? std::vector<double>(mesh->elem_size, mesh->elem_size + tp.ne)
                                                  : element_sizes(tp.verts, op->dim, tp.ne);
      for (double &v : delta) v /= op->order;  // elSize = GetElementSize(e, 1) / order, src/rhs_operator.cpp:154
      double *dd = dev_upload(delta);
      op->d_extra.push_back(dd);
      d.elem_delta = dd;
      pick_dryair_les(op);
    } else if (lte)
      pick_lte_axisym(op);
    else if (disc->axisymmetric)
      pick_dryair_axisym(op);
    else if (op->dim == 3)
      any_nr ? pick_order<3, DryAirPhys<3, true>>(op) : pick_order<3, DryAirPhys<3>>(op);
    else
      any_nr ? pick_order<2, DryAirPhys<2, true>>(op) : pick_order<2, DryAirPhys<2>>(op);
  }

  op->d_verts = dev_upload(tp.verts);
  if (op->nc) op->d_minv = dev_upload(inverse_mass_matrices(tp.verts, op->dim, op->order, op->ne));
  {
    std::vector<int2> fi(tp.face_nbr.size());
    for (size_t i = 0; i < fi.size(); i++) fi[i] = make_int2(tp.face_nbr[i], tp.face_orient[i]);
    op->d_face_info = dev_upload(fi);
  }
  if (!plasma && !disc->axisymmetric) {  // faces of the non-reflecting patches, in slot order
    std::vector<int2> nrf;
    std::vector<int> ordinal(tp.face_nbr.size(), -1);
    bool any_nr = false;
    for (int i = 0; i < num_bcs; i++) any_nr = any_nr || is_non_reflecting(bcs[i].category, bcs[i].type);
    if (any_nr) {
      DryAirParams &d = *reinterpret_cast<DryAirParams *>(op->params);
      for (size_t slot = 0; slot < tp.face_nbr.size(); slot++) {
        const int nb = tp.face_nbr[slot];
        if (nb >= 0) continue;
        const int b = -nb - 1;
        if (!is_non_reflecting(bcs[b].category, bcs[b].type)) continue;
        ordinal[slot] = static_cast<int>(nrf.size());
        nrf.push_back(make_int2(static_cast<int>(slot), b));
        // tangent1 of the patch: the caller's (the reference takes it from its first boundary face,
        // src/outletBC.cpp:160-172) or an edge of our first face of the patch, orthogonalised to nothing --
        // as there, the patch is assumed planar
        double *t1 = &d.bc[b].data[4];
        if (t1[0] == 0.0 && t1[1] == 0.0 && t1[2] == 0.0) {
          const int e = static_cast<int>(slot) / op->nfaces, lf = static_cast<int>(slot) % op->nfaces, D = lf >> 1, sd = lf & 1;
          const int nv = 1 << op->dim, a = (op->dim == 2) ? 1 - D : (D == 0 ? 1 : 0);
          const int c0 = sd << D, c1 = c0 | (1 << a);
          double mod = 0.0, t[3] = {0, 0, 0};
          for (int k = 0; k < op->dim; k++) {
            t[k] = tp.verts[(static_cast<size_t>(e) * nv + c1) * op->dim + k] - tp.verts[(static_cast<size_t>(e) * nv + c0) * op->dim + k];
            mod += t[k] * t[k];
          }
          for (int k = 0; k < 3; k++) t1[k] = t[k] / std::sqrt(mod);
        }
      }
      if (tp.num_shared > 0 && !op->reduce)
        throw std::runtime_error("halo: a partitioned mesh with non-reflecting patches needs runtime.reduce");
      op->n_nr_faces = static_cast<int>(nrf.size());
      op->d_bc_sums = dev_alloc<double>(MAXBC * (TPSRHS_MAXEQUATIONS + 1));
      HIP_CHECK(hipMemset(op->d_bc_sums, 0, MAXBC * (TPSRHS_MAXEQUATIONS + 1) * sizeof(double)));
      if (!nrf.empty()) {
        op->d_nr_faces = dev_upload(nrf);
        op->d_nr_ordinal = dev_upload(ordinal);
        const size_t n = nrf.size() * static_cast<size_t>(op->nq) * op->neq;
        for (int k = 0; k < 2; k++) {
          op->d_bstate[k] = dev_alloc<double>(n);
          HIP_CHECK(hipMemset(op->d_bstate[k], 0, n * sizeof(double)));
        }
      }
    }
    op->has_nr = any_nr;
  }
  if (op->ndofs >= (int64_t(1) << 31)) throw Unsupported("more than 2^31 nodes per rank");
  const int64_t nslots = static_cast<int64_t>(op->ne) * op->nfaces + tp.num_shared;
  op->d_Up = dev_alloc<double>(op->neq * op->ndofs);
  op->d_gradUp = dev_alloc<double>(op->dim * op->neq * op->ndofs);
  op->d_TA = dev_alloc<double>(nslots * 2 * op->neq * op->nf);
  op->d_TB = dev_alloc<double>(nslots * (op->neq - 1) * op->nq);
  HIP_CHECK(hipMemset(op->d_TA, 0, nslots * 2 * op->neq * op->nf * sizeof(double)));
  HIP_CHECK(hipMemset(op->d_TB, 0, nslots * (op->neq - 1) * op->nq * sizeof(double)));
  op->d_speed = dev_alloc<double>(1);
  HIP_CHECK(hipMemset(op->d_speed, 0, sizeof(double)));
  if (tp.num_shared > 0) {
    op->d_shared_slot = dev_upload(tp.shared_slot);
    op->d_shared_orient = dev_upload(tp.shared_orient);
    const int64_t per0 = 2 * op->neq * op->nf, per1 = static_cast<int64_t>(op->neq - 1) * op->nq;
    op->d_send = dev_alloc<double>(tp.num_shared * std::max(per0, per1));
    for (size_t i = 0; i < tp.nbr_offsets.size(); i++) {
      op->send_off[0].push_back(tp.nbr_offsets[i] * per0);
      op->send_off[1].push_back(tp.nbr_offsets[i] * per1);
    }
    op->recv_off[0] = op->send_off[0];
    op->recv_off[1] = op->send_off[1];
  }
  for (auto &set : op->evs)
    for (auto &e : set) HIP_CHECK(hipEventCreate(&e));
}

int fail(int code, const std::string &msg) {
  g_last_error = msg;
  return code;
}

template <class F>
int guarded(F &&f) {
  try {
    f();
    return TPSRHS_OK;
  } catch (const Unsupported &e) {
    return fail(TPSRHS_ERR_UNSUPPORTED, e.what());
  } catch (const DeviceError &e) {
    return fail(TPSRHS_ERR_DEVICE, e.what());
  } catch (const std::invalid_argument &e) {
    return fail(TPSRHS_ERR_INVALID_ARGUMENT, e.what());
  } catch (const std::runtime_error &e) {
    const std::string w = e.what();
    if (w.find("no HIP device") != std::string::npos) return fail(TPSRHS_ERR_NO_DEVICE, w);
    if (w.find("halo") != std::string::npos) return fail(TPSRHS_ERR_HALO, w);
    return fail(TPSRHS_ERR_MESH, w);
  } catch (const std::exception &e) {
    return fail(TPSRHS_ERR_INVALID_ARGUMENT, e.what());
  }
}

}  // namespace

#if TPSRHS_STAMP
// diagnostic builds of this unit (-DTPSRHS_STAMP=1 / 2, tools/stamp_phases.py with a dry-air workload): the phase cycles of
// the first `nblocks` blocks of the last stamped kernel
extern "C" int tpsrhs_debug_stamps(unsigned int *out, int nblocks) {
  const size_t bytes = static_cast<size_t>(nblocks) * tpsrhs::NSTAMP * sizeof(unsigned int);
  return hipMemcpyFromSymbol(out, HIP_SYMBOL(tpsrhs::g_stamp), bytes) == hipSuccess ? 0 : 1;
}
#endif

extern "C" {

int tpsrhs_create(const tpsrhs_mesh *mesh, const tpsrhs_disc *disc, const tpsrhs_physics *physics, int num_bcs,
                  const tpsrhs_bc *bcs, const tpsrhs_runtime *runtime, tpsrhs_handle *out) {
  if (!mesh || !disc || !physics || !out || (num_bcs > 0 && !bcs))
    return fail(TPSRHS_ERR_INVALID_ARGUMENT, "tpsrhs_create: NULL argument");
  *out = nullptr;
  std::unique_ptr<tpsrhs_operator> op(new tpsrhs_operator());
  const int st = guarded([&] { setup(op.get(), mesh, disc, physics, num_bcs, bcs, runtime); });
  if (st == TPSRHS_OK) *out = op.release();
  return st;
}

int tpsrhs_destroy(tpsrhs_handle h) {
  delete h;
  return TPSRHS_OK;
}

int tpsrhs_mult(tpsrhs_handle h, const double *x, double *y, double /*time*/, double *max_char_speed) {
  if (!h || !x || !y) return fail(TPSRHS_ERR_INVALID_ARGUMENT, "tpsrhs_mult: NULL argument");
  return guarded([&] {
    HIP_CHECK(hipSetDevice(h->device));
    h->launch(h, x, y, false);
    if (max_char_speed) {
      hipLaunchKernelGGL(k_reduce_max<1024>, dim3(1), dim3(1024), 0, h->stream, h->flux_grid, h->d_block_speed, h->d_speed);
      HIP_CHECK(hipGetLastError());
      HIP_CHECK(hipMemcpyAsync(max_char_speed, h->d_speed, sizeof(double), hipMemcpyDeviceToHost, h->stream));
      HIP_CHECK(hipStreamSynchronize(h->stream));
    }
  });
}

int tpsrhs_mult_host(tpsrhs_handle h, const double *x, double *y, double time, double *max_char_speed) {
  if (!h || !x || !y) return fail(TPSRHS_ERR_INVALID_ARGUMENT, "tpsrhs_mult_host: NULL argument");
  const size_t bytes = static_cast<size_t>(h->neq) * h->ndofs * sizeof(double);
  int st = guarded([&] {
    HIP_CHECK(hipSetDevice(h->device));
    if (!h->d_xh) {
      h->d_xh = dev_alloc<double>(h->neq * h->ndofs);
      h->d_yh = dev_alloc<double>(h->neq * h->ndofs);
    }
    HIP_CHECK(hipMemcpyAsync(h->d_xh, x, bytes, hipMemcpyHostToDevice, h->stream));
  });
  if (st != TPSRHS_OK) return st;
  st = tpsrhs_mult(h, h->d_xh, h->d_yh, time, max_char_speed);
  if (st != TPSRHS_OK) return st;
  return guarded([&] {
    HIP_CHECK(hipMemcpyAsync(y, h->d_yh, bytes, hipMemcpyDeviceToHost, h->stream));
    HIP_CHECK(hipStreamSynchronize(h->stream));
  });
}

namespace {
// the four stages of MFEM's RK4Solver::Step around Mult; dt by value or from device memory (dt_dev)
void rk4_stages(tpsrhs_operator *h, double *x, double dt, const double *dt_dev) {
  const int64_t n = static_cast<int64_t>(h->neq) * h->ndofs;
  if (!h->d_rk) {
    h->d_rk = dev_alloc<double>(3 * n);
    h->d_nan = dev_alloc<unsigned long long>(1);
    HIP_CHECK(hipMemsetAsync(h->d_nan, 0, sizeof(unsigned long long), h->stream));
  }
  double *k = h->d_rk, *y = k + n, *z = y + n;
  if (!h->ta_chain) h->ta_valid = false;  // a step starts with its own k_traces sweep: x may have been touched since the
                                          // last call (tpsrhs_advance chains its steps: nothing touches x between them)
  const bool mixture = h->phys.working_fluid == TPSRHS_USER_DEFINED;
  const int sp_first = h->nvel + 2;
  const int sp_last = mixture ? sp_first + (h->phys.mixture.ambipolar ? h->phys.mixture.num_species - 2
                                                                       : h->phys.mixture.num_species - 1)
                              : sp_first;
  const int grid = static_cast<int>(std::min<int64_t>((n + 255) / 256, 8192));
  if (!h->forcing_active) {
    // The stage combinations in k_flux's epilogue (RkDev, kernels.hpp): k never goes to memory, the accumulator of
    // RK4Solver::Step is recovered from the stage states at the end.  (The optional forcing terms add to the residual in
    // a pass of their own after k_flux: with them the separate stage kernel below stays.)
    double *y2 = k, *y3 = y, *y4 = z;  // the three stage states
    const double *ins[4] = {x, y2, y3, y4};
    double *outs[4] = {y2, y3, y4, x};
    for (int stage = 1; stage <= 4; stage++) {
      RkDev r = {};
      r.mode = stage;
      r.sp_first = sp_first;
      r.sp_last = sp_last;
      r.dt_host = dt;
      r.dt_dev = dt_dev;
      r.x0 = x;
      r.y2 = y2;
      r.y3 = y3;
      r.y4 = y4;
      r.out = outs[stage - 1];
      r.nan_count = h->d_nan;
      h->rk = r;
      try {
        h->launch(h, ins[stage - 1], outs[stage - 1], false);
      } catch (...) {
        h->rk = RkDev{};
        h->ta_valid = false;
        throw;
      }
      h->rk = RkDev{};
    }
    if (!h->ta_chain) h->ta_valid = false;
    return;
  }
  const double *in = x;
  for (int stage = 1; stage <= 4; stage++) {
    h->launch(h, in, k, false);  // k_s = f(stage input); SetTime is a no-op for this operator
    hipLaunchKernelGGL(k_rk4_stage<256>, dim3(grid), dim3(256), 0, h->stream, stage, n, h->ndofs, sp_first, sp_last, dt,
                       dt_dev, x, k, y, z, h->d_nan);
    HIP_CHECK(hipGetLastError());
    in = y;
  }
}
}  // namespace

int tpsrhs_rk4_step(tpsrhs_handle h, double *x, double *time, double dt, double *max_char_speed, int64_t *nan_count) {
  if (!h || !x || !time) return fail(TPSRHS_ERR_INVALID_ARGUMENT, "tpsrhs_rk4_step: NULL argument");
  return guarded([&] {
    HIP_CHECK(hipSetDevice(h->device));
    h->nr_dt = dt;  // the boundary conditions see M2ulPhyS::dt (src/BoundaryCondition.hpp:54)
    if (h->d_nan) HIP_CHECK(hipMemsetAsync(h->d_nan, 0, sizeof(unsigned long long), h->stream));
    rk4_stages(h, x, dt, nullptr);
    *time += dt;
    if (max_char_speed || nan_count) {
      hipLaunchKernelGGL(k_reduce_max<1024>, dim3(1), dim3(1024), 0, h->stream, h->flux_grid, h->d_block_speed, h->d_speed);
      HIP_CHECK(hipGetLastError());
      double speed = 0.0;
      unsigned long long bad = 0;
      HIP_CHECK(hipMemcpyAsync(&speed, h->d_speed, sizeof(double), hipMemcpyDeviceToHost, h->stream));
      HIP_CHECK(hipMemcpyAsync(&bad, h->d_nan, sizeof(bad), hipMemcpyDeviceToHost, h->stream));
      HIP_CHECK(hipStreamSynchronize(h->stream));
      if (max_char_speed) *max_char_speed = speed;
      if (nan_count) *nan_count = static_cast<int64_t>(bad);
    }
  });
}

int tpsrhs_advance(tpsrhs_handle h, double *x, double *time, double *dt, int num_steps, int constant_dt, double cfl,
                   double hmin, int64_t *nan_count) {
  if (!h || !x || !time || !dt) return fail(TPSRHS_ERR_INVALID_ARGUMENT, "tpsrhs_advance: NULL argument");
  if (num_steps < 0 || !(*dt > 0.0) || (!constant_dt && !(cfl > 0.0 && hmin > 0.0)))
    return fail(TPSRHS_ERR_INVALID_ARGUMENT, "tpsrhs_advance: needs num_steps >= 0, dt > 0 and (constant dt or cfl, hmin > 0)");
  return guarded([&] {
    HIP_CHECK(hipSetDevice(h->device));
    if (!constant_dt && h->topo.num_shared > 0 && !h->reduce)
      throw std::runtime_error("halo: a variable time step on a partitioned mesh needs runtime.reduce");
    if (!h->d_ctl) h->d_ctl = dev_alloc<double>(3);
    const double ctl0[3] = {*dt, *time, 0.0};
    HIP_CHECK(hipMemcpyAsync(h->d_ctl, ctl0, sizeof(ctl0), hipMemcpyHostToDevice, h->stream));
    HIP_CHECK(hipStreamSynchronize(h->stream));  // ctl0 lives on this stack frame
    if (h->d_nan) HIP_CHECK(hipMemsetAsync(h->d_nan, 0, sizeof(unsigned long long), h->stream));
    h->nr_dt_dev = h->d_ctl;
    auto one_step = [&] {
      rk4_stages(h, x, 0.0, h->d_ctl);
      hipLaunchKernelGGL(k_step_end<1024>, dim3(1), dim3(1024), 0, h->stream, h->flux_grid, h->d_block_speed, h->d_ctl,
                         constant_dt ? 1 : 0, cfl * hmin / static_cast<double>(h->dim));
      HIP_CHECK(hipGetLastError());
      if (!constant_dt && h->reduce && h->topo.num_shared > 0) {  // MPI_Allreduce(MIN) of src/M2ulPhyS.cpp:2015
        if (h->reduce(h->reduce_ctx, h->d_ctl, 1, TPSRHS_REDUCE_MIN, h->stream) != 0)
          throw std::runtime_error("halo: reduce callback failed");
      }
    };
    // On one rank a step is a fixed sequence of ~17 launches whose arguments do not change from step to step (dt
    // and the time are in device memory; the two boundary-state buffers swap four times per step): captured once
    // into a hipGraph and replayed -- the launch overhead matters on small meshes.  Needs a capturable stream (not
    // the NULL stream), no host callbacks in the step (partitioned meshes keep the plain loop), no timing events.
    const char *genv = std::getenv("TPSRHS_GRAPH");
    const bool use_graph = h->stream != nullptr && h->topo.num_shared == 0 && !h->timing && num_steps >= 3 &&
                           !(genv && genv[0] == '0');
    h->ta_valid = false;
    h->ta_chain = !h->forcing_active;  // (with forcing terms the stage kernel runs and nothing is fused)
    try {
      int step = 0;
      if (use_graph) {
        one_step();  // first step outside the graph: allocations, initial boundary state, its own k_traces sweep
        step = 1;
        tpsrhs_operator::StepKey key;
        key.x = x;
        key.constant_dt = constant_dt ? 1 : 0;
        key.bstate_cur = h->bstate_cur;
        key.epoch = h->config_epoch;
        key.coef = cfl * hmin / static_cast<double>(h->dim);
        if (!h->step_graph || !(h->step_key == key)) {
          if (h->step_graph) {
            HIP_CHECK(hipGraphExecDestroy(h->step_graph));
            h->step_graph = nullptr;
          }
          hipGraph_t g = nullptr;
          HIP_CHECK(hipStreamBeginCapture(h->stream, hipStreamCaptureModeThreadLocal));
          try {
            one_step();
          } catch (...) {
            (void)hipStreamEndCapture(h->stream, &g);
            if (g) (void)hipGraphDestroy(g);
            throw;
          }
          HIP_CHECK(hipStreamEndCapture(h->stream, &g));
          const hipError_t ie = hipGraphInstantiate(&h->step_graph, g, nullptr, nullptr, 0);
          (void)hipGraphDestroy(g);
          HIP_CHECK(ie);
          h->step_key = key;
        }
        for (; step < num_steps; step++) HIP_CHECK(hipGraphLaunch(h->step_graph, h->stream));
      }
      for (; step < num_steps; step++) one_step();
    } catch (...) {
      h->nr_dt_dev = nullptr;
      h->ta_chain = h->ta_valid = false;
      throw;
    }
    h->nr_dt_dev = nullptr;
    h->ta_chain = h->ta_valid = false;  // the caller owns x again
    double ctl[3] = {0, 0, 0};
    unsigned long long bad = 0;
    HIP_CHECK(hipMemcpyAsync(ctl, h->d_ctl, sizeof(ctl), hipMemcpyDeviceToHost, h->stream));
    if (h->d_nan) HIP_CHECK(hipMemcpyAsync(&bad, h->d_nan, sizeof(bad), hipMemcpyDeviceToHost, h->stream));
    HIP_CHECK(hipStreamSynchronize(h->stream));
    *dt = ctl[0];
    *time = ctl[1];
    h->nr_dt = ctl[0];
    if (nan_count) *nan_count = static_cast<int64_t>(bad);
  });
}

int tpsrhs_set_dt(tpsrhs_handle h, double dt) {
  if (!h) return fail(TPSRHS_ERR_INVALID_ARGUMENT, "tpsrhs_set_dt: NULL handle");
  h->nr_dt = dt;
  return TPSRHS_OK;
}

namespace {
void upload_forcing(tpsrhs_operator *h) {
  const ForcingDev &f = h->forcing;
  h->forcing_active = f.has_pg || f.nheat > 0 || f.nsponge > 0 || f.nps > 0 || f.joule != nullptr;
  if (!h->forcing_active) return;
  HIP_CHECK(hipSetDevice(h->device));
  if (!h->d_forcing) h->d_forcing = dev_alloc<ForcingDev>(1);
  HIP_CHECK(hipStreamSynchronize(h->stream));  // an earlier Mult may still read the block
  HIP_CHECK(hipMemcpy(h->d_forcing, &h->forcing, sizeof(ForcingDev), hipMemcpyHostToDevice));
}
}  // namespace

int tpsrhs_set_forcing(tpsrhs_handle h, const tpsrhs_forcing *in) {
  if (!h) return fail(TPSRHS_ERR_INVALID_ARGUMENT, "tpsrhs_set_forcing: NULL handle");
  return guarded([&] {
    ForcingDev f = {};
    f.joule = h->forcing.joule;
    // plane-node lists and sum buffers of the mixed-out zones of THIS call; they replace the previous call's, which
    // are freed once the new block is in place (or these, if the call fails: the old forcing then stays as it was)
    std::vector<void *> fresh;
    struct Guard {
      std::vector<void *> &v;
      bool keep = false;
      ~Guard() {
        if (!keep)
          for (void *q : v) (void)hipFree(q);
      }
    } guard{fresh};
    if (in) {
      if (in->num_heat_sources < 0 || in->num_heat_sources > TPSRHS_MAXHEATSOURCES || in->num_sponge_zones < 0 ||
          in->num_sponge_zones > TPSRHS_MAXSPONGEZONES)
        throw std::invalid_argument("tpsrhs_set_forcing: heat source / sponge zone count out of range");
      if (in->num_passive_scalars < 0 || in->num_passive_scalars > TPSRHS_MAXPASSIVESCALARS)
        throw std::invalid_argument("tpsrhs_set_forcing: passive scalar count out of range");
      f.nps = in->num_passive_scalars;
      for (int i = 0; i < f.nps; i++) {  // PassiveScalar constructor, src/forcing_terms.cpp:778-790
        const tpsrhs_passive_scalar &s = in->passive_scalars[i];
        if (!(s.radius > 0.0)) throw std::invalid_argument("tpsrhs_set_forcing: passive scalar radius must be positive");
        for (int d = 0; d < 3; d++) f.ps[i].x0[d] = s.coords[d];
        f.ps[i].radius = s.radius;
        f.ps[i].value = s.value;
      }
      const int dim = h->dim;
      f.has_pg = in->has_pressure_gradient ? 1 : 0;
      for (int d = 0; d < 3; d++) f.pg[d] = in->pressure_gradient[d];
      f.nheat = in->num_heat_sources;
      for (int i = 0; i < f.nheat; i++) {  // HeatSource constructor, src/forcing_terms.cpp:890-899
        const tpsrhs_heat_source &s = in->heat_sources[i];
        ForcingDev::Heat &d = f.heat[i];
        d.value = s.value;
        d.radius = s.radius;
        double mod = 0.0;
        for (int k = 0; k < dim; k++) mod += (s.point2[k] - s.point1[k]) * (s.point2[k] - s.point1[k]);
        mod = std::sqrt(mod);
        if (!(mod > 0.0)) throw std::invalid_argument("tpsrhs_set_forcing: heat source with point1 == point2");
        d.len = mod;
        for (int k = 0; k < 3; k++) {
          d.p1[k] = s.point1[k];
          d.axis[k] = k < dim ? (s.point2[k] - s.point1[k]) / mod : 0.0;
        }
      }
      f.nsponge = in->num_sponge_zones;
      for (int i = 0; i < f.nsponge; i++) {
        const tpsrhs_sponge_zone &s = in->sponge_zones[i];
        ForcingDev::Sponge &d = f.sponge[i];
        if (s.type != TPSRHS_SPONGE_PLANAR && s.type != TPSRHS_SPONGE_ANNULUS)
          throw std::invalid_argument("tpsrhs_set_forcing: unknown sponge zone type");
        if (s.type == TPSRHS_SPONGE_ANNULUS && dim != 3)
          throw Unsupported("annular sponge zone needs dim == 3 (the reference's transform indexes 3 components)");
        d.type = s.type;
        double mod = 0.0;  // "make sure normal is unitary", src/forcing_terms.cpp:528-532
        for (int k = 0; k < dim; k++) mod += s.normal[k] * s.normal[k];
        mod = std::sqrt(mod);
        if (!(mod > 0.0)) throw std::invalid_argument("tpsrhs_set_forcing: sponge zone with a zero normal");
        for (int k = 0; k < 3; k++) {
          d.normal[k] = k < dim ? s.normal[k] / mod : 0.0;
          d.p0[k] = s.point0[k];
          d.pinit[k] = s.point_init[k];
        }
        d.r1 = s.r1;
        d.r2 = s.r2;
        d.mult = s.mult_factor;
        if (s.solution_type == TPSRHS_SPONGE_MIXEDOUT) {
          // nodesInMixedOutPlane of the constructor, src/forcing_terms.cpp:553-606
          if (h->phys.working_fluid != TPSRHS_DRY_AIR || h->nvel != dim)
            throw Unsupported("mixed-out sponge zone: built for dry air, planar 2-D and 3-D");
          if (!(s.tol > 0.0)) throw std::invalid_argument("tpsrhs_set_forcing: mixed-out sponge zone needs tol > 0");
          const Tables1D t = make_tables(h->order, dim, h->nc, h->nc);
          const int n1 = h->order + 1, npe = (dim == 3) ? n1 * n1 * n1 : n1 * n1, nv = 1 << dim;
          std::vector<int> nodes;
          for (int e = 0; e < h->ne; e++) {
            const double *V = &h->topo.verts[static_cast<size_t>(e) * nv * dim];
            for (int k = 0; k < npe; k++) {
              const int idx[3] = {k % n1, (k / n1) % n1, k / (n1 * n1)};
              double X[3] = {0.0, 0.0, 0.0};
              for (int v = 0; v < nv; v++) {
                double shp = 1.0;
                for (int a = 0; a < dim; a++) shp *= ((v >> a) & 1) ? t.x[idx[a]] : 1.0 - t.x[idx[a]];
                for (int i = 0; i < dim; i++) X[i] += shp * V[v * dim + i];
              }
              double dist_init = 0.0;
              for (int a = 0; a < dim; a++) dist_init -= d.normal[a] * (X[a] - d.pinit[a]);
              bool in_plane;
              if (d.type == TPSRHS_SPONGE_PLANAR) {
                in_plane = std::fabs(dist_init) < s.tol;
              } else {
                double R = 0.0;
                for (int a = 0; a < dim; a++) {
                  const double tt = X[a] - d.pinit[a] + dist_init * d.normal[a];
                  R += tt * tt;
                }
                in_plane = std::fabs(std::sqrt(R) - d.r1) < s.tol;
              }
              if (in_plane) nodes.push_back(e * npe + k);
            }
          }
          if (nodes.empty() && h->topo.num_shared == 0)
            throw std::invalid_argument("tpsrhs_set_forcing: no node within tol of the mix-out plane");
          // the plane sums are added over the ranks (MPI_Allreduce of src/forcing_terms.cpp:732-735): without the hook a
          // partitioned run would use rank-local means, and a rank without plane nodes 0 / 0
          if (h->topo.num_shared > 0 && !h->reduce)
            throw std::invalid_argument("tpsrhs_set_forcing: a mixed-out sponge zone on a partitioned mesh needs runtime.reduce");
          d.mixed_out = 1;
          d.n_plane = static_cast<int>(nodes.size());
          int *dn = dev_upload(nodes);
          double *ds = dev_alloc<double>(TPSRHS_MAXEQUATIONS + 1);
          fresh.push_back(dn);
          fresh.push_back(ds);
          d.plane_nodes = dn;
          d.msum = ds;
          for (int eq = 0; eq < TPSRHS_MAXEQUATIONS; eq++) d.target[eq] = 0.0;
        } else {
          if (s.solution_type != TPSRHS_SPONGE_USERDEF) throw std::invalid_argument("tpsrhs_set_forcing: unknown sponge solution type");
          for (int eq = 0; eq < TPSRHS_MAXEQUATIONS; eq++) d.target[eq] = eq < h->neq ? s.target_U[eq] : 0.0;
          if (!(d.target[0] > 0.0)) throw std::invalid_argument("tpsrhs_set_forcing: sponge target density must be positive");
        }
      }
    }
    // the device block first; the host copy, the buffers of the mixed-out zones and the configuration epoch are committed
    // only once it is in place -- if the upload fails the operator keeps its previous forcing and the new buffers are
    // released (the guard), the previous ones are not
    const ForcingDev prev = h->forcing;
    const bool prev_active = h->forcing_active;
    h->forcing = f;
    try {
      upload_forcing(h);  // synchronises the stream before it overwrites the device block
      HIP_CHECK(hipStreamSynchronize(h->stream));
    } catch (...) {
      h->forcing = prev;
      h->forcing_active = prev_active;
      throw;
    }
    std::vector<void *> old;
    old.swap(h->d_mixed_out);
    h->d_mixed_out = fresh;
    guard.keep = true;
    h->config_epoch++;
    for (void *q : old) (void)hipFree(q);
  });
}

int tpsrhs_set_mixing_length(tpsrhs_handle h, const double *distance, const tpsrhs_mixing_length *in) {
  if (!h) return fail(TPSRHS_ERR_INVALID_ARGUMENT, "tpsrhs_set_mixing_length: NULL handle");
  return guarded([&] {
    if (distance) {
      if (!in) throw std::invalid_argument("tpsrhs_set_mixing_length: parameters missing");
      // the 2-D kernels with the `heavy` closure interface: mixtures (planar, axisymmetric), axisymmetric dry air
      const bool plasma = h->phys.working_fluid == TPSRHS_USER_DEFINED;
      if (h->dim != 2 || !(plasma || h->nvel == 3))
        throw Unsupported("mixing-length model: built for the 2-D kernels (mixtures planar / axisymmetric, dry air axisymmetric)");
      if (h->nc) throw Unsupported("mixing-length model: Gauss-Legendre pair");
      if (!(in->max_mixing_length >= 0.0)) throw std::invalid_argument("tpsrhs_set_mixing_length: negative max_mixing_length");
      h->mixlen.distance = distance;
      h->mixlen.lmax = in->max_mixing_length;
      h->mixlen.prt = in->pr_ratio;
      h->mixlen.bulk = in->bulk_multiplier;
    } else {
      h->mixlen.distance = nullptr;
    }
    h->config_epoch++;  // a captured time-loop graph holds the old pointer
  });
}

int tpsrhs_set_joule_heating(tpsrhs_handle h, const double *joule_heating) {
  if (!h) return fail(TPSRHS_ERR_INVALID_ARGUMENT, "tpsrhs_set_joule_heating: NULL handle");
  return guarded([&] {
    // JouleHeating::updateTerms asserts nvel == 3 (src/forcing_terms.cpp:444)
    if (joule_heating && h->nvel != 3) throw Unsupported("JouleHeating needs three velocity components (3-D or axisymmetric)");
    h->forcing.joule = joule_heating;
    h->config_epoch++;
    upload_forcing(h);
  });
}

int tpsrhs_update_gradients(tpsrhs_handle h, const double *x) {
  if (!h || !x) return fail(TPSRHS_ERR_INVALID_ARGUMENT, "tpsrhs_update_gradients: NULL argument");
  return guarded([&] {
    HIP_CHECK(hipSetDevice(h->device));
    h->launch(h, x, nullptr, true);
  });
}

int tpsrhs_get_primitives(tpsrhs_handle h, double *up_out) {
  if (!h || !up_out) return fail(TPSRHS_ERR_INVALID_ARGUMENT, "tpsrhs_get_primitives: NULL argument");
  return guarded([&] {
    HIP_CHECK(hipMemcpyAsync(up_out, h->d_Up, sizeof(double) * h->neq * h->ndofs, hipMemcpyDeviceToDevice, h->stream));
  });
}

int tpsrhs_get_gradients(tpsrhs_handle h, double *gradup_out) {
  if (!h || !gradup_out) return fail(TPSRHS_ERR_INVALID_ARGUMENT, "tpsrhs_get_gradients: NULL argument");
  return guarded([&] {
    HIP_CHECK(hipMemcpyAsync(gradup_out, h->d_gradUp, sizeof(double) * h->dim * h->neq * h->ndofs,
                             hipMemcpyDeviceToDevice, h->stream));
  });
}

int64_t tpsrhs_height(tpsrhs_handle h) { return h ? h->neq * h->ndofs : -1; }
int64_t tpsrhs_num_dofs(tpsrhs_handle h) { return h ? h->ndofs : -1; }
int tpsrhs_num_equation(tpsrhs_handle h) { return h ? h->neq : -1; }

int tpsrhs_enable_kernel_timing(tpsrhs_handle h, int enable) {
  if (!h) return TPSRHS_ERR_INVALID_ARGUMENT;
  h->timing = enable != 0;
  h->sets_recorded = 0;
  return TPSRHS_OK;
}

int tpsrhs_kernel_times(tpsrhs_handle h, int capacity, const char **names, double *milliseconds) {
  if (!h || h->sets_recorded == 0) return 0;
  if (hipStreamSynchronize(h->stream) != hipSuccess) return 0;
  const int nsets = static_cast<int>(std::min<int64_t>(h->sets_recorded, tpsrhs_operator::MAXSETS));
  int n = 0;
  for (int k = 0; k < NKERN && n < capacity; k++, n++) {
    double sum = 0.0;
    for (int i = 0; i < nsets; i++) {
      float ms = 0.f;
      if (hipEventElapsedTime(&ms, h->evs[i][k], h->evs[i][k + 1]) != hipSuccess) return n;
      sum += ms;
    }
    if (names) names[n] = kKernelNames[k];
    if (milliseconds) milliseconds[n] = sum / nsets;  // average over the recorded Mults
  }
  return n;
}

int tpsrhs_eval_pointwise(tpsrhs_handle h, int quantity, int64_t n, const double *U, double *out) {
  if (!h || n < 0 || !U || !out || quantity < 0 || quantity > 3) return fail(TPSRHS_ERR_INVALID_ARGUMENT, "eval_pointwise");
  return guarded([&] {
    HIP_CHECK(hipSetDevice(h->device));
    h->point_eval(h, quantity, n, U, out);
    HIP_CHECK(hipStreamSynchronize(h->stream));
  });
}

namespace {
__global__ void k_table_eval(TableDev t, int64_t n, const double *__restrict__ x, double *__restrict__ f) {
  const int64_t i = static_cast<int64_t>(blockIdx.x) * 256 + threadIdx.x;
  if (i < n) f[i] = table_eval(t, x[i]);
}
}  // namespace

namespace {
__global__ void k_math_eval(int fn, int64_t n, const double *__restrict__ x, double *__restrict__ y) {
  const int64_t i = static_cast<int64_t>(blockIdx.x) * 256 + threadIdx.x;
  if (i >= n) return;
  const double v = x[i];
  double r;
  switch (fn) {
    case 0: r = fexp(v); break;
    case 1: r = fexp<false>(v); break;
    case 2: r = flog(v); break;
    case 3: r = flog_pos(v); break;
    case 4: r = fast_rcp(v); break;
    case 5: r = fast_sqrt(v); break;
    default: r = fast_rsqrt(v); break;
  }
  y[i] = r;
}
}  // namespace

int tpsrhs_math_eval(int function, int64_t n, const double *x, double *y) {
  if (function < 0 || function > 6 || n < 0 || !x || !y) return fail(TPSRHS_ERR_INVALID_ARGUMENT, "math_eval");
  return guarded([&] {
    int ndev = 0;
    if (hipGetDeviceCount(&ndev) != hipSuccess || ndev == 0) throw std::runtime_error("no HIP device visible");
    if (n > 0) {
      hipLaunchKernelGGL(k_math_eval, dim3(static_cast<unsigned>((n + 255) / 256)), dim3(256), 0, nullptr, function, n, x, y);
      HIP_CHECK(hipGetLastError());
    }
    HIP_CHECK(hipDeviceSynchronize());
  });
}

int tpsrhs_table_eval(const tpsrhs_table *table, int64_t n, const double *x, double *f) {
  if (!table || n < 0 || !x || !f) return fail(TPSRHS_ERR_INVALID_ARGUMENT, "table_eval");
  return guarded([&] {
    int ndev = 0;
    if (hipGetDeviceCount(&ndev) != hipSuccess || ndev == 0) throw std::runtime_error("no HIP device visible");
    tpsrhs_operator tmp;  // owns the device copy of the table for the duration of the call
    (void)hipGetDevice(&tmp.device);
    const TableDev td = upload_table(&tmp, *table);
    if (n > 0) {
      hipLaunchKernelGGL(k_table_eval, dim3(static_cast<unsigned>((n + 255) / 256)), dim3(256), 0, nullptr, td, n, x, f);
      HIP_CHECK(hipGetLastError());
    }
    HIP_CHECK(hipDeviceSynchronize());
  });
}

int tpsrhs_mult_times(tpsrhs_handle h, int capacity, double *milliseconds) {
  if (!h || h->sets_recorded == 0 || !milliseconds) return 0;
  if (hipStreamSynchronize(h->stream) != hipSuccess) return 0;
  const int nsets = static_cast<int>(std::min<int64_t>(h->sets_recorded, tpsrhs_operator::MAXSETS));
  int n = 0;
  for (; n < nsets && n < capacity; n++) {
    float ms = 0.f;
    if (hipEventElapsedTime(&ms, h->evs[n][0], h->evs[n][NKERN]) != hipSuccess) return n;
    milliseconds[n] = ms;
  }
  return n;
}

int tpsrhs_kernel_bytes(tpsrhs_handle h, int capacity, const char **names, double *bytes) {
  if (!h) return 0;
  // algorithmic HBM traffic of the three sweeps ("bytes per unit"), in bytes per Mult
  const double N = static_cast<double>(h->ndofs), neq = h->neq, dim = h->dim;
  const double slots = static_cast<double>(h->ne) * h->nfaces;
  const double ta = slots * 2 * neq * h->nf, tb = slots * (neq - 1) * h->nq;
  const double geo = static_cast<double>(h->ne) * (1 << h->dim) * dim;
  const double b[NKERN] = {
      8.0 * (neq * N /*U*/ + ta),
      8.0 * (neq * N + 0.5 * ta /*neighbour Up traces*/ + neq * N /*Up*/ + dim * neq * N /*gradUp*/ + tb + geo),
      8.0 * (neq * N + dim * neq * N + 0.5 * ta /*neighbour U traces*/ + 2.0 * tb + neq * N /*y*/ + geo)};
  int n = 0;
  for (int k = 0; k < NKERN && n < capacity; k++, n++) {
    if (names) names[n] = kKernelNames[k];
    if (bytes) bytes[n] = b[k];
  }
  return n;
}

int tpsrhs_face_tables(const tpsrhs_mesh *mesh, int num_bcs, const tpsrhs_bc *bcs, int32_t *face_nbr,
                       uint8_t *face_orient, int32_t *shared_slot, uint8_t *shared_orient) {
  if (!mesh || !face_nbr || !face_orient) return fail(TPSRHS_ERR_INVALID_ARGUMENT, "tpsrhs_face_tables: NULL argument");
  return guarded([&] {
    const Topology T = build_topology(*mesh, num_bcs, bcs);
    std::memcpy(face_nbr, T.face_nbr.data(), T.face_nbr.size() * sizeof(int32_t));
    std::memcpy(face_orient, T.face_orient.data(), T.face_orient.size());
    if (shared_slot && T.num_shared) std::memcpy(shared_slot, T.shared_slot.data(), T.num_shared * sizeof(int32_t));
    if (shared_orient && T.num_shared) std::memcpy(shared_orient, T.shared_orient.data(), T.num_shared);
  });
}

const char *tpsrhs_status_string(int status) {
  switch (status) {
    case TPSRHS_OK: return "TPSRHS_OK";
    case TPSRHS_ERR_INVALID_ARGUMENT: return "TPSRHS_ERR_INVALID_ARGUMENT";
    case TPSRHS_ERR_UNSUPPORTED: return "TPSRHS_ERR_UNSUPPORTED";
    case TPSRHS_ERR_MESH: return "TPSRHS_ERR_MESH";
    case TPSRHS_ERR_DEVICE: return "TPSRHS_ERR_DEVICE";
    case TPSRHS_ERR_NO_DEVICE: return "TPSRHS_ERR_NO_DEVICE";
    case TPSRHS_ERR_HALO: return "TPSRHS_ERR_HALO";
    default: return "TPSRHS_ERR_UNKNOWN";
  }
}
const char *tpsrhs_last_error(void) { return g_last_error.c_str(); }
int tpsrhs_get_plasma_conductivity(tpsrhs_handle h, const double *x, double *sigma_out) {
  if (!h || !x || !sigma_out) return fail(TPSRHS_ERR_INVALID_ARGUMENT, "tpsrhs_get_plasma_conductivity: NULL argument");
  return guarded([&] {
    if (h->phys.working_fluid == TPSRHS_DRY_AIR) throw Unsupported("plasma conductivity output: dry air has no SourceTerm");
    if (h->phys.working_fluid == TPSRHS_USER_DEFINED && !h->phys.mixture.ambipolar && h->phys.chemistry.num_reactions > 0)
      throw Unsupported("plasma conductivity output: the reference's SourceTerm does not store it for a reacting mixture that "
                        "is not ambipolar (src/source_term.cpp:178-197)");
    HIP_CHECK(hipSetDevice(h->device));
    h->point_eval(h, 4, h->ndofs, x, sigma_out);
  });
}

const char *tpsrhs_version(void) { return "tpsrhs 0.1.0 (gfx950)"; }

}  // extern "C"
