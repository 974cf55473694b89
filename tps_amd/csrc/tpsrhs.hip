// libtpsrhs.so -- implementation of include/tpsrhs.h for gfx950 (MI355X).
//
// Host side: builds the face topology and the 1-D operator tables, keeps every field resident in
// HBM, and enqueues the three sweeps of kernels.hpp on one HIP stream per operator.  There is no
// CPU fallback: without a HIP device tpsrhs_create returns TPSRHS_ERR_NO_DEVICE.
#include <hip/hip_runtime.h>

#include <cstdio>
#include <cstring>
#include <memory>
#include <stdexcept>
#include <string>
#include <vector>

#include "../../include/tpsrhs.h"
#include "basis.hpp"
#include "kernels.hpp"
#include "physics_dryair.hpp"
#include "topology.hpp"

using namespace tpsrhs;

static thread_local std::string g_last_error;

namespace {

struct DeviceError : std::runtime_error {
  explicit DeviceError(const std::string &s) : std::runtime_error(s) {}
};
struct Unsupported : std::runtime_error {
  explicit Unsupported(const std::string &s) : std::runtime_error(s) {}
};

#define HIP_CHECK(expr)                                                                                   \
  do {                                                                                                    \
    hipError_t _e = (expr);                                                                               \
    if (_e != hipSuccess)                                                                                 \
      throw DeviceError(std::string(#expr) + ": " + hipGetErrorString(_e) + " (" __FILE__ ":" + std::to_string(__LINE__) + ")"); \
  } while (0)

template <class T>
T *dev_alloc(size_t n) {
  T *p = nullptr;
  HIP_CHECK(hipMalloc(reinterpret_cast<void **>(&p), std::max<size_t>(n, 1) * sizeof(T)));
  return p;
}
template <class T>
T *dev_upload(const std::vector<T> &v) {
  T *p = dev_alloc<T>(v.size());
  if (!v.empty()) HIP_CHECK(hipMemcpy(p, v.data(), v.size() * sizeof(T), hipMemcpyHostToDevice));
  return p;
}

constexpr int NKERN = 3;
const char *kKernelNames[NKERN] = {"k_traces", "k_gradient", "k_flux"};

}  // namespace

struct tpsrhs_operator {
  int dim = 0, order = 0, neq = 0, nvel = 0;
  int ne = 0, nfaces = 0, nf = 0, nq = 0;
  int64_t ndofs = 0;
  int device = 0;
  hipStream_t stream = nullptr;
  Topology topo;
  tpsrhs_physics phys;
  DryAirParams dry;
  // device data
  double *d_verts = nullptr;
  int2 *d_face_info = nullptr;
  double *d_Up = nullptr, *d_gradUp = nullptr, *d_TA = nullptr, *d_TB = nullptr;
  double *d_speed = nullptr, *d_block_speed = nullptr;
  int flux_grid = 0;
  double *d_xh = nullptr, *d_yh = nullptr;  // staging for tpsrhs_mult_host
  // halo
  tpsrhs_halo_fn halo = nullptr;
  void *halo_ctx = nullptr;
  int32_t *d_shared_slot = nullptr;
  uint8_t *d_shared_orient = nullptr;
  double *d_send = nullptr;
  std::vector<int64_t> send_off[2], recv_off[2];
  // timing
  // per-kernel timing: a ring of event sets so that a timed loop never synchronises
  static constexpr int MAXSETS = 128;
  bool timing = false;
  hipEvent_t evs[MAXSETS][NKERN + 1] = {};
  int64_t sets_recorded = 0;
  hipEvent_t *ev = evs[0];

  void (*launch)(tpsrhs_operator *, const double *, double *, bool) = nullptr;

  MeshDev mesh_dev() const {
    MeshDev m;
    m.ne = ne;
    m.ndofs = ndofs;
    m.verts = d_verts;
    m.face_info = d_face_info;
    return m;
  }
  ~tpsrhs_operator() {
    (void)hipSetDevice(device);
    for (void *p : {static_cast<void *>(d_verts), static_cast<void *>(d_face_info),
                    static_cast<void *>(d_Up), static_cast<void *>(d_gradUp),
                    static_cast<void *>(d_TA), static_cast<void *>(d_TB), static_cast<void *>(d_speed), static_cast<void *>(d_block_speed),
                    static_cast<void *>(d_xh), static_cast<void *>(d_yh), static_cast<void *>(d_shared_slot),
                    static_cast<void *>(d_shared_orient), static_cast<void *>(d_send)})
      if (p) (void)hipFree(p);
    for (auto &set : evs)
      for (auto &e : set)
        if (e) (void)hipEventDestroy(e);
  }
};

namespace {

void exchange(tpsrhs_operator *op, int phase, double *T, int nfld, int per) {
  const Topology &tp = op->topo;
  if (tp.num_shared == 0) return;
  const int n1 = (phase == 0) ? op->order + 1 : ((op->dim - 1) + 2 * op->order) / 2 + 1;
  const int64_t total = static_cast<int64_t>(tp.num_shared) * nfld * per;
  const int grid = static_cast<int>(std::min<int64_t>((total + 255) / 256, 2048));
  if (op->dim == 3)
    hipLaunchKernelGGL(k_pack<3>, dim3(grid), dim3(256), 0, op->stream, tp.num_shared, nfld, n1, op->d_shared_slot,
                       op->d_shared_orient, T, op->d_send);
  else
    hipLaunchKernelGGL(k_pack<2>, dim3(grid), dim3(256), 0, op->stream, tp.num_shared, nfld, n1, op->d_shared_slot,
                       op->d_shared_orient, T, op->d_send);
  HIP_CHECK(hipGetLastError());
  double *recv = T + static_cast<int64_t>(op->ne) * op->nfaces * nfld * per;
  const int st = op->halo(op->halo_ctx, phase, op->d_send, recv, static_cast<int>(tp.nbr_ranks.size()),
                          tp.nbr_ranks.data(), op->send_off[phase].data(), op->recv_off[phase].data(), op->stream);
  if (st != 0) throw std::runtime_error("halo callback failed in phase " + std::to_string(phase));
}

template <int DIM, int P, class PH>
void launch_all(tpsrhs_operator *op, const double *x, double *y, bool gradients_only) {
  typedef Cfg<DIM, P> C;
  const MeshDev m = op->mesh_dev();
  const int grid = (op->ne + C::EPB - 1) / C::EPB;
  const typename PH::Params &prm = op->dry;
  hipStream_t s = op->stream;
  if (op->timing) {
    op->ev = op->evs[op->sets_recorded % tpsrhs_operator::MAXSETS];
    HIP_CHECK(hipEventRecord(op->ev[0], s));
  }
  hipLaunchKernelGGL((k_traces<C, PH>), dim3(grid), dim3(C::BLOCK), 0, s, m, prm, x, op->d_Up, op->d_TA);
  HIP_CHECK(hipGetLastError());
  if (op->timing) HIP_CHECK(hipEventRecord(op->ev[1], s));
  exchange(op, 0, op->d_TA, 2 * PH::NEQ, C::NF);
  hipLaunchKernelGGL((k_gradient<C, PH>), dim3(grid), dim3(C::BLOCK), 0, s, m, prm, x, op->d_TA, op->d_gradUp, op->d_TB);
  HIP_CHECK(hipGetLastError());
  if (op->timing) HIP_CHECK(hipEventRecord(op->ev[2], s));
  if (gradients_only) return;
  exchange(op, 1, op->d_TB, PH::NEQ, C::NQ);
  if (!op->d_block_speed) {
    op->d_block_speed = dev_alloc<double>(grid);
    op->flux_grid = grid;
  }
  hipLaunchKernelGGL((k_flux<C, PH>), dim3(grid), dim3(C::BLOCK), 0, s, m, prm, x, op->d_gradUp, op->d_TA, op->d_TB, y,
                     op->d_block_speed);
  HIP_CHECK(hipGetLastError());
  if (op->timing) {
    HIP_CHECK(hipEventRecord(op->ev[3], s));
    op->sets_recorded++;
  }
}

template <int DIM, class PH>
void pick_order(tpsrhs_operator *op) {
  switch (op->order) {
    case 1: op->launch = &launch_all<DIM, 1, PH>; break;
    case 2: op->launch = &launch_all<DIM, 2, PH>; break;
    case 3: op->launch = &launch_all<DIM, 3, PH>; break;
    case 4: op->launch = &launch_all<DIM, 4, PH>; break;
    default: throw Unsupported("polynomial order " + std::to_string(op->order) + " is not built (1..4)");
  }
}

void setup(tpsrhs_operator *op, const tpsrhs_mesh *mesh, const tpsrhs_disc *disc, const tpsrhs_physics *phys,
           int num_bcs, const tpsrhs_bc *bcs, const tpsrhs_runtime *rt) {
  if (disc->basis_type != TPSRHS_BASIS_GAUSS_LEGENDRE || disc->int_rule_type != 0)
    throw Unsupported(
        "only the collocated Gauss-Legendre basis + Gauss-Legendre rule (basisType 0, integrationRule 0) is built");
  if (disc->axisymmetric) throw Unsupported("axisymmetric formulation is not built yet");
  if (phys->working_fluid != TPSRHS_DRY_AIR) throw Unsupported("only WorkingFluid::DRY_AIR is built yet");
  if (phys->eq_system != TPSRHS_EULER && phys->eq_system != TPSRHS_NS) throw Unsupported("NS_PASSIVE is out of scope");
  if (num_bcs > MAXBC) throw Unsupported("too many boundary conditions");
  for (int i = 0; i < num_bcs; i++) {
    const tpsrhs_bc &b = bcs[i];
    const bool ok = (b.category == TPSRHS_INLET && b.type == TPSRHS_SUB_DENS_VEL) ||
                    (b.category == TPSRHS_OUTLET && b.type == TPSRHS_SUB_P) ||
                    (b.category == TPSRHS_WALL &&
                     (b.type == TPSRHS_INV || b.type == TPSRHS_VISC_ADIAB || b.type == TPSRHS_VISC_ISOTH));
    if (!ok) throw Unsupported("boundary condition type outside the hot-path scope (attribute " + std::to_string(b.attribute) + ")");
  }
  op->dim = mesh->dim;
  op->order = disc->order;
  op->nvel = op->dim;
  op->neq = op->dim + 2;
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
  op->halo = rt ? rt->halo : nullptr;
  op->halo_ctx = rt ? rt->halo_ctx : nullptr;

  op->topo = build_topology(*mesh, num_bcs, bcs);
  const Topology &tp = op->topo;
  if (tp.num_shared > 0 && !op->halo) throw std::invalid_argument("mesh has shared faces but runtime.halo is NULL");
  op->ne = tp.ne;
  op->nfaces = tp.nfaces;
  const int n1 = op->order + 1, q1 = ((op->dim - 1) + 2 * op->order) / 2 + 1;
  op->nf = (op->dim == 3) ? n1 * n1 : n1;
  op->nq = (op->dim == 3) ? q1 * q1 : q1;
  const int npe = (op->dim == 3) ? n1 * n1 * n1 : n1 * n1;
  op->ndofs = static_cast<int64_t>(op->ne) * npe;

  DryAirParams &d = op->dry;
  std::memset(&d, 0, sizeof(d));
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
  d.num_bcs = num_bcs;
  for (int i = 0; i < num_bcs; i++) {
    d.bc[i].category = bcs[i].category;
    d.bc[i].type = bcs[i].type;
    for (int k = 0; k < 4 + TPSRHS_MAXSPECIES; k++) d.bc[i].data[k] = bcs[i].data[k];
  }

  if (op->dim == 3)
    pick_order<3, DryAirPhys<3>>(op);
  else
    pick_order<2, DryAirPhys<2>>(op);

  op->d_verts = dev_upload(tp.verts);
  {
    std::vector<int2> fi(tp.face_nbr.size());
    for (size_t i = 0; i < fi.size(); i++) fi[i] = make_int2(tp.face_nbr[i], tp.face_orient[i]);
    op->d_face_info = dev_upload(fi);
  }
  {
    // 1-D operator tables -> __constant__ memory; a function of (dim, order) only
    const Tables1D tabs = make_tables(op->order, op->dim);
    const size_t off = (static_cast<size_t>(op->dim - 2) * (TPSRHS_MAXORDER + 1) + op->order) * sizeof(Tables1D);
    HIP_CHECK(hipMemcpyToSymbol(HIP_SYMBOL(c_tab), &tabs, sizeof(Tables1D), off, hipMemcpyHostToDevice));
  }
  if (op->ndofs >= (int64_t(1) << 31)) throw Unsupported("more than 2^31 nodes per rank");
  const int64_t nslots = static_cast<int64_t>(op->ne) * op->nfaces + tp.num_shared;
  op->d_Up = dev_alloc<double>(op->neq * op->ndofs);
  op->d_gradUp = dev_alloc<double>(op->dim * op->neq * op->ndofs);
  op->d_TA = dev_alloc<double>(nslots * 2 * op->neq * op->nf);
  op->d_TB = dev_alloc<double>(nslots * op->neq * op->nq);
  HIP_CHECK(hipMemset(op->d_TA, 0, nslots * 2 * op->neq * op->nf * sizeof(double)));
  HIP_CHECK(hipMemset(op->d_TB, 0, nslots * op->neq * op->nq * sizeof(double)));
  op->d_speed = dev_alloc<double>(1);
  HIP_CHECK(hipMemset(op->d_speed, 0, sizeof(double)));
  if (tp.num_shared > 0) {
    op->d_shared_slot = dev_upload(tp.shared_slot);
    op->d_shared_orient = dev_upload(tp.shared_orient);
    const int64_t per0 = 2 * op->neq * op->nf, per1 = static_cast<int64_t>(op->neq) * op->nq;
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
      hipLaunchKernelGGL(k_reduce_max, dim3(1), dim3(256), 0, h->stream, h->flux_grid, h->d_block_speed, h->d_speed);
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

int tpsrhs_kernel_bytes(tpsrhs_handle h, int capacity, const char **names, double *bytes) {
  if (!h) return 0;
  // algorithmic HBM traffic of the three sweeps ("bytes per unit"), in bytes per Mult
  const double N = static_cast<double>(h->ndofs), neq = h->neq, dim = h->dim;
  const double slots = static_cast<double>(h->ne) * h->nfaces;
  const double ta = slots * 2 * neq * h->nf, tb = slots * neq * h->nq;
  const double geo = static_cast<double>(h->ne) * (1 << h->dim) * dim;
  const double b[NKERN] = {
      8.0 * (neq * N /*U*/ + neq * N /*Up*/ + ta),
      8.0 * (neq * N + 0.5 * ta /*neighbour Up traces*/ + dim * neq * N /*gradUp*/ + tb + geo),
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
const char *tpsrhs_version(void) { return "tpsrhs 0.1.0 (gfx950)"; }

}  // extern "C"
