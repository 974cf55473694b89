// Shared by the translation units of libtpsrhs.so: the operator object behind tpsrhs_handle and the
// templated launch sequence of one RHS evaluation.  Kernels are instantiated per physics family in
// separate .hip files (compiled in parallel); each has its own copy of the __constant__ tables.
#ifndef TPSRHS_OPERATOR_HPP_
#define TPSRHS_OPERATOR_HPP_

#include <hip/hip_runtime.h>

#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <memory>
#include <stdexcept>
#include <string>
#include <vector>

#include "../../include/tpsrhs.h"
#include "basis.hpp"
#include "kernels.hpp"
#include "topology.hpp"

using namespace tpsrhs;

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

// TPSRHS_POISON=1 (debug / tests/test_gpu_poison.py): every device allocation of the library starts as all-ones bytes
// (a NaN as a double, -1 as an index), so that a read of memory no kernel has written shows up as a NaN in the
// residual instead of depending on what the allocator happened to return.  Read at every call: cheap next to hipMalloc.
inline bool poison_allocations() {
  const char *e = std::getenv("TPSRHS_POISON");
  return e && e[0] == '1';
}
template <class T>
T *dev_alloc(size_t n) {
  T *p = nullptr;
  const size_t bytes = std::max<size_t>(n, 1) * sizeof(T);
  HIP_CHECK(hipMalloc(reinterpret_cast<void **>(&p), bytes));
  if (poison_allocations()) HIP_CHECK(hipMemset(p, 0xFF, bytes));
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
  int nc = 0;  // 1: the non-collocated Gauss-Lobatto pair (basisType 1, integrationRule 1), 0: the collocated Gauss-Legendre pair
  double *d_minv = nullptr;  // nc: [ne][npe][npe] inverse element mass matrices
  int ne = 0, nfaces = 0, nf = 0, nq = 0;
  int64_t ndofs = 0;
  int device = 0;
  hipStream_t stream = nullptr;
  Topology topo;
  tpsrhs_physics phys;
  alignas(16) unsigned char params[4096];  // PH::Params of the selected physics, passed by value
  void *d_chem = nullptr;                   // ChemDev block + table storage (plasma)
  void *d_params = nullptr;                 // device image of `params` for the physics that read it through a pointer (plasma)
  std::vector<void *> d_extra;
  std::vector<void *> d_mixed_out;  // plane-node lists / sum buffers of the current forcing's mixed-out sponge zones
  // device data
  double *d_verts = nullptr;
  int2 *d_face_info = nullptr;
  double *d_Up = nullptr, *d_gradUp = nullptr, *d_TA = nullptr, *d_TB = nullptr;
  double *d_speed = nullptr, *d_block_speed = nullptr;
  int flux_grid = 0;
  double *d_xh = nullptr, *d_yh = nullptr;  // staging for tpsrhs_mult_host
  double *d_rk = nullptr;                   // k | y | z of tpsrhs_rk4_step
  unsigned long long *d_nan = nullptr;
  RkDev rk = {};  // mode 0 outside tpsrhs_rk4_step / tpsrhs_advance: k_flux writes the residual
  MixLenDev mixlen = {nullptr, 0.0, 1.0, 0.0};  // MixingLengthTransport (tpsrhs_set_mixing_length); distance NULL: off
  ForcingDev forcing = {};                  // host copy of the optional forcing terms (tpsrhs_set_forcing / _joule_heating)
  ForcingDev *d_forcing = nullptr;
  bool forcing_active = false;
  // non-reflecting inlet / outlet patches (dry air): their faces {slot, bc index}, slot -> ordinal, the
  // double-buffered boundary state, the patch sums of the primitives, the time step they integrate with
  bool has_nr = false;  // some boundary condition is of a non-reflecting type (possibly without local faces)
  int n_nr_faces = 0;
  int2 *d_nr_faces = nullptr;
  int *d_nr_ordinal = nullptr;
  double *d_bstate[2] = {nullptr, nullptr};
  int bstate_cur = 0;
  bool bstate_init = false;
  double *d_bc_sums = nullptr;
  double nr_dt = 0.0;
  double *d_ctl = nullptr;          // {dt, time, max speed} of tpsrhs_advance
  const double *nr_dt_dev = nullptr;  // non-NULL while tpsrhs_advance runs: the boundary conditions read dt there
  tpsrhs_reduce_fn reduce = nullptr;
  void *reduce_ctx = nullptr;
  // one RK4 step of tpsrhs_advance as an executable graph (dt lives in device memory, so the step replays as is)
  hipGraphExec_t step_graph = nullptr;
  struct StepKey {
    const void *x = nullptr;
    int constant_dt = 0, bstate_cur = 0, epoch = 0;
    double coef = 0.0;
    bool operator==(const StepKey &o) const {
      return x == o.x && constant_dt == o.constant_dt && bstate_cur == o.bstate_cur && epoch == o.epoch && coef == o.coef;
    }
  } step_key;
  int config_epoch = 0;  // bumped by everything that changes what a step launches (forcing terms, ...)
  // halo
  tpsrhs_halo_fn halo = nullptr;
  void *halo_ctx = nullptr;
  hipStream_t comm_stream = nullptr;  // second stream: pack + exchange overlap the interior blocks
  hipEvent_t ev_halo[4] = {};         // traces of the halo blocks ready / TA received / TB ready / TB received
  int *d_blocks_halo = nullptr, *d_blocks_interior = nullptr;
  int n_blocks_halo = 0, n_blocks_interior = 0;
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
  void (*point_eval)(tpsrhs_operator *, int, int64_t, const double *, double *) = nullptr;
  // Time loop: k_flux of stages 1..3 forms the next stage's face-node traces in its epilogue (RkDev::ta_out, kernels.hpp),
  // alternating between d_TA and d_TA2; the next stage then starts at k_gradient.  Single-rank 3-D collocated kernels only
  // (flux_fuses_traces); TPSRHS_FUSE_TRACES=0 keeps the separate sweep.
  bool fuse_traces = true;
  bool ta_valid = false;     // d_ta_next holds the traces of the input of the stage that runs next
  bool ta_chain = false;     // inside tpsrhs_advance: stage 4 leaves the traces of the new solution for the next step
  double *d_TA2 = nullptr, *d_ta_next = nullptr;
  bool sweep_alt = true;  // alternate the direction of consecutive sweeps (launch_all); TPSRHS_SWEEP_ALT
  int sweep_parity = 0;
  VsDev vs2d = {};  // viscous sponge of the 2-D heavy kernels (MeshDev::vs); enabled = 0: none

  MeshDev mesh_dev() const {
    MeshDev m;
    m.blocks = nullptr;
    m.ne = ne;
    m.reverse = 0;
    m.ndofs = ndofs;
    m.verts = d_verts;
    m.face_info = d_face_info;
    m.minv = d_minv;
    m.ml = mixlen;
    m.vs = vs2d;
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
    if (d_chem) (void)hipFree(d_chem);
    if (d_params) (void)hipFree(d_params);
    if (d_minv) (void)hipFree(d_minv);
    if (d_TA2) (void)hipFree(d_TA2);
    if (d_rk) (void)hipFree(d_rk);
    if (d_nan) (void)hipFree(d_nan);
    if (d_forcing) (void)hipFree(d_forcing);
    if (step_graph) (void)hipGraphExecDestroy(step_graph);
    for (void *p : {static_cast<void *>(d_nr_faces), static_cast<void *>(d_nr_ordinal), static_cast<void *>(d_bstate[0]),
                    static_cast<void *>(d_bstate[1]), static_cast<void *>(d_bc_sums), static_cast<void *>(d_ctl)})
      if (p) (void)hipFree(p);
    if (d_blocks_halo) (void)hipFree(d_blocks_halo);
    if (d_blocks_interior) (void)hipFree(d_blocks_interior);
    for (auto &e : ev_halo)
      if (e) (void)hipEventDestroy(e);
    if (comm_stream) (void)hipStreamDestroy(comm_stream);
    for (void *p : d_extra) (void)hipFree(p);
    for (void *p : d_mixed_out) (void)hipFree(p);
    for (auto &set : evs)
      for (auto &e : set)
        if (e) (void)hipEventDestroy(e);
  }
};

namespace {

// The parameter block as the kernels of physics PH take it: by value (dry air) or as a pointer to its device image
// (plasma: PlasmaPhys::KArg), uploaded once -- the block is complete when the first kernel is launched and nothing
// changes it afterwards (the dry-air block, which the non-reflecting conditions update per Mult, travels by value).
template <class PH>
typename PH::KArg kernel_params(tpsrhs_operator *op) {
  if constexpr (std::is_pointer<typename PH::KArg>::value) {
    if (!op->d_params) {
      op->d_params = dev_alloc<unsigned char>(sizeof(op->params));
      HIP_CHECK(hipMemcpy(op->d_params, op->params, sizeof(op->params), hipMemcpyHostToDevice));
    }
    return reinterpret_cast<typename PH::KArg>(op->d_params);
  } else {
    return *reinterpret_cast<const typename PH::Params *>(op->params);
  }
}

inline void exchange(tpsrhs_operator *op, int phase, double *T, int nfld, int per, hipStream_t stream) {
  const Topology &tp = op->topo;
  if (tp.num_shared == 0) return;
  const int n1 = (phase == 0) ? op->order + 1 : rule_points(op->nc, (op->dim - 1) + 2 * op->order);
  const int64_t total = static_cast<int64_t>(tp.num_shared) * nfld * per;
  const int grid = static_cast<int>(std::min<int64_t>((total + 255) / 256, 2048));
  if (op->dim == 3)
    hipLaunchKernelGGL(k_pack<3>, dim3(grid), dim3(256), 0, stream, tp.num_shared, nfld, n1, op->d_shared_slot,
                       op->d_shared_orient, T, op->d_send);
  else
    hipLaunchKernelGGL(k_pack<2>, dim3(grid), dim3(256), 0, stream, tp.num_shared, nfld, n1, op->d_shared_slot,
                       op->d_shared_orient, T, op->d_send);
  HIP_CHECK(hipGetLastError());
  double *recv = T + static_cast<int64_t>(op->ne) * op->nfaces * nfld * per;
  const int st = op->halo(op->halo_ctx, phase, op->d_send, recv, static_cast<int>(tp.nbr_ranks.size()),
                          tp.nbr_ranks.data(), op->send_off[phase].data(), op->recv_off[phase].data(), stream);
  if (st != 0) throw std::runtime_error("halo callback failed in phase " + std::to_string(phase));
}

template <int DIM, int P, class PH, int NC = 0>
void launch_all(tpsrhs_operator *op, const double *x, double *y, bool gradients_only) {
  typedef Cfg<DIM, P, NC> C;
  static_assert(sizeof(typename PH::Params) <= sizeof(op->params), "parameter block too large");
  const typename PH::Params &prm = *reinterpret_cast<const typename PH::Params *>(op->params);  // host copy
  const int nblocks = (op->ne + C::EPB - 1) / C::EPB;
  hipStream_t s = op->stream;
  if constexpr (PH::HAS_NR_BC) {  // this Mult's view of the non-reflecting boundary state
    typename PH::Params &w = *reinterpret_cast<typename PH::Params *>(op->params);
    w.bstate = op->d_bstate[op->bstate_cur];
    w.nr_ordinal = op->d_nr_ordinal;
    w.nr_dt = op->nr_dt;
  }
  const typename PH::KArg prm_k = kernel_params<PH>(op);  // after the update above: what the kernels of this Mult get
  if (!op->d_block_speed && !gradients_only) {
    op->d_block_speed = dev_alloc<double>(nblocks);
    op->flux_grid = nblocks;
  }
  // Alternating sweep direction (TPSRHS_SWEEP_ALT): k_traces and k_flux of a Mult walk the block list one way, k_gradient
  // the other, and the next Mult starts the other way round -- every sweep begins with what its predecessor wrote last
  // (xcd_block).  The direction is per sweep, the same for its halo and interior launches.
  const int dir0 = op->sweep_alt ? op->sweep_parity : 0;
  // which trace buffer this Mult reads, whether its k_traces sweep is already done, and where k_flux puts the next one's
  double *ta = op->d_TA;
  bool have_traces = false;
  if constexpr (flux_fuses_traces<C, PH>()) {
    const int stage = op->rk.mode;
    // the previous stage of this step left them (rk4_stages: its output is this input) -- or, inside tpsrhs_advance
    // (ta_chain), the last stage of the previous step, whose output x is this step's input
    if ((stage >= 2 || (stage == 1 && op->ta_chain)) && op->ta_valid) {
      ta = op->d_ta_next;
      have_traces = true;
    }
    op->ta_valid = false;
    if (op->fuse_traces && !gradients_only && op->topo.num_shared == 0 && stage >= 1 && (stage <= 3 || op->ta_chain)) {
      if (!op->d_TA2) op->d_TA2 = dev_alloc<double>(static_cast<int64_t>(op->ne) * op->nfaces * 2 * PH::NEQ * C::NF);
      op->rk.ta_out = (ta == op->d_TA) ? op->d_TA2 : op->d_TA;
      op->d_ta_next = op->rk.ta_out;
      op->ta_valid = true;  // (cleared again below if a launch throws)
    }
  } else {
    op->ta_valid = false;
  }
  auto traces = [&](MeshDev m, int grid) {
    if (have_traces) return;
    m.reverse = dir0;
    hipLaunchKernelGGL((k_traces<C, PH>), dim3(grid), dim3(C::BLOCK), 0, s, m, prm_k, x, ta);
    HIP_CHECK(hipGetLastError());
  };
  auto gradient = [&](MeshDev m, int grid) {
    m.reverse = op->sweep_alt ? 1 - dir0 : 0;
    hipLaunchKernelGGL((k_gradient<C, PH>), dim3(grid), dim3(C::BLOCK), 0, s, m, prm_k, x, ta, op->d_Up, op->d_gradUp,
                       op->d_TB);
    HIP_CHECK(hipGetLastError());
  };
  auto flux = [&](MeshDev m, int grid) {
    m.reverse = dir0;
    hipLaunchKernelGGL((k_flux<C, PH>), dim3(grid), dim3(C::BLOCK), 0, s, m, prm_k, x, op->d_gradUp, ta, op->d_TB, y,
                       op->d_block_speed, op->rk);
    HIP_CHECK(hipGetLastError());
  };
  // non-reflecting patches: mean of the primitives (+ sum over the ranks), then the boundary-state update;
  // after the last k_gradient launch, before the first k_flux launch
  auto nr_update = [&](const MeshDev &m) {
    if constexpr (PH::HAS_NR_BC) {
      if (!op->has_nr) return;
      const int nbc = prm.num_bcs;
      hipLaunchKernelGGL((k_bc_mean<C, PH>), dim3(nbc), dim3(256), 0, s, op->n_nr_faces, op->d_nr_faces, ta,
                         op->d_bc_sums);
      HIP_CHECK(hipGetLastError());
      if (op->reduce && op->topo.num_shared > 0) {
        const int st = op->reduce(op->reduce_ctx, op->d_bc_sums, nbc * (TPSRHS_MAXEQUATIONS + 1), TPSRHS_REDUCE_SUM, s);
        if (st != 0) throw std::runtime_error("halo: reduce callback failed");
      }
      if (op->n_nr_faces > 0) {
        hipLaunchKernelGGL((k_bc_nr<C, PH>), dim3(op->n_nr_faces), dim3(C::BLOCK), 0, s, m, prm_k, op->d_nr_faces, op->d_bc_sums, x,
                           op->d_Up, op->d_gradUp, op->d_bstate[op->bstate_cur], op->d_bstate[1 - op->bstate_cur],
                           op->bstate_init ? 0 : 1, op->nr_dt_dev);
        HIP_CHECK(hipGetLastError());
      }
    }
  };
  // ConstantPressureGradient / SpongeZone / HeatSource / JouleHeating, after the last k_flux launch
  auto forcing = [&](const MeshDev &m) {
    if (!op->forcing_active) return;
    if constexpr (PH::HAS_MIXED_OUT) {  // SpongeZone::updateTerms: computeMixedOutValues first (src/forcing_terms.cpp:631-635)
      for (int z = 0; z < op->forcing.nsponge; z++) {
        if (!op->forcing.sponge[z].mixed_out) continue;
        hipLaunchKernelGGL((k_mixed_out_sum<C, PH>), dim3(1), dim3(256), 0, s, op->ndofs, prm_k, op->d_forcing, z, x);
        HIP_CHECK(hipGetLastError());
        if (op->reduce && op->topo.num_shared > 0) {  // MPI_Allreduce of meanNormalFluxes, :732-735
          const int st = op->reduce(op->reduce_ctx, op->forcing.sponge[z].msum, PH::NEQ + 1, TPSRHS_REDUCE_SUM, s);
          if (st != 0) throw std::runtime_error("mixed-out sponge: reduce callback failed");
        }
        hipLaunchKernelGGL((k_mixed_out_finish<C, PH>), dim3(1), dim3(1), 0, s, prm_k, op->d_forcing, z);
        HIP_CHECK(hipGetLastError());
      }
    }
    const int grid = static_cast<int>((op->ndofs + 255) / 256);
    hipLaunchKernelGGL((k_forcing<C, PH>), dim3(grid), dim3(256), 0, s, m, prm_k, op->d_forcing, x, op->d_gradUp, y);
    HIP_CHECK(hipGetLastError());
  };
  if (op->timing) {
    op->ev = op->evs[op->sets_recorded % tpsrhs_operator::MAXSETS];
    HIP_CHECK(hipEventRecord(op->ev[0], s));
  }
  const MeshDev all = op->mesh_dev();
  if (op->sweep_alt) op->sweep_parity ^= 1;  // (read into dir0 above: the next Mult starts from the other end)
  if (op->topo.num_shared == 0) {
    traces(all, nblocks);
    if (op->timing) HIP_CHECK(hipEventRecord(op->ev[1], s));
    gradient(all, nblocks);
    if (op->timing) HIP_CHECK(hipEventRecord(op->ev[2], s));
    if (gradients_only) return;
    nr_update(all);
    flux(all, nblocks);
  } else {
    // Partitioned mesh.  Blocks that touch a shared face ("halo blocks") run first in the producing
    // sweeps and last in the consuming ones; packing and the neighbour exchange run on the second
    // stream while the interior blocks compute (the reference overlaps its MPI exchange with the
    // volume work the same way, src/rhs_operator.cpp:361-372,775-831).
    if (!op->d_blocks_halo) {
      std::vector<int> halo, interior;
      std::vector<char> is_halo(nblocks, 0);
      for (int i = 0; i < op->topo.num_shared; i++) is_halo[(op->topo.shared_slot[i] / op->nfaces) / C::EPB] = 1;
      for (int b = 0; b < nblocks; b++) (is_halo[b] ? halo : interior).push_back(b);
      op->n_blocks_halo = static_cast<int>(halo.size());
      op->n_blocks_interior = static_cast<int>(interior.size());
      op->d_blocks_halo = dev_upload(halo);
      op->d_blocks_interior = dev_upload(interior);
      HIP_CHECK(hipStreamCreateWithFlags(&op->comm_stream, hipStreamNonBlocking));
      for (auto &e : op->ev_halo) HIP_CHECK(hipEventCreateWithFlags(&e, hipEventDisableTiming));
    }
    MeshDev mh = all, mi = all;
    mh.blocks = op->d_blocks_halo;
    mi.blocks = op->d_blocks_interior;
    const int nh = op->n_blocks_halo, ni = op->n_blocks_interior;
    hipStream_t c = op->comm_stream;
    // host order: the interior launches are enqueued BEFORE the (host-side, slow) exchange callback, so
    // the compute stream never waits for the host
    traces(mh, nh);
    HIP_CHECK(hipEventRecord(op->ev_halo[0], s));
    if (ni) traces(mi, ni);
    if (op->timing) HIP_CHECK(hipEventRecord(op->ev[1], s));
    if (ni) gradient(mi, ni);
    HIP_CHECK(hipStreamWaitEvent(c, op->ev_halo[0], 0));
    exchange(op, 0, op->d_TA, 2 * PH::NEQ, C::NF, c);  // overlaps the two interior launches above
    HIP_CHECK(hipEventRecord(op->ev_halo[1], c));
    HIP_CHECK(hipStreamWaitEvent(s, op->ev_halo[1], 0));
    gradient(mh, nh);
    if (op->timing) HIP_CHECK(hipEventRecord(op->ev[2], s));
    if (gradients_only) return;
    HIP_CHECK(hipEventRecord(op->ev_halo[2], s));
    nr_update(all);
    if (ni) flux(mi, ni);
    HIP_CHECK(hipStreamWaitEvent(c, op->ev_halo[2], 0));
    exchange(op, 1, op->d_TB, PH::NEQ - 1, C::NQ, c);  // overlaps the interior flux launch
    HIP_CHECK(hipEventRecord(op->ev_halo[3], c));
    HIP_CHECK(hipStreamWaitEvent(s, op->ev_halo[3], 0));
    flux(mh, nh);
  }
  forcing(all);
  if constexpr (PH::HAS_NR_BC) {
    if (op->n_nr_faces > 0) {  // the updated states are what the next Mult's Riemann solver sees
      op->bstate_cur = 1 - op->bstate_cur;
      op->bstate_init = true;
    }
  }
  if (op->timing) {
    HIP_CHECK(hipEventRecord(op->ev[3], s));
    op->sets_recorded++;
  }
}

template <class PH>
void launch_point_eval(tpsrhs_operator *op, int quantity, int64_t n, const double *U, double *out) {
  const int grid = static_cast<int>((n + 255) / 256);
  if (grid == 0) return;
  hipLaunchKernelGGL((k_point_eval<PH>), dim3(grid), dim3(256), 0, op->stream, kernel_params<PH>(op), quantity, n, U, out);
  HIP_CHECK(hipGetLastError());
}

// 1-D operator tables -> this translation unit's __constant__ copy; a function of (dim, order) only
inline void upload_tables(int dim, int order, int nc = 0) {
  const Tables1D tabs = make_tables(order, dim, nc, nc);
  const size_t off = ((static_cast<size_t>(nc) * 2 + (dim - 2)) * (TPSRHS_MAXORDER + 1) + order) * sizeof(Tables1D);
  HIP_CHECK(hipMemcpyToSymbol(HIP_SYMBOL(c_tab), &tabs, sizeof(Tables1D), off, hipMemcpyHostToDevice));
}

// non-collocated (Gauss-Lobatto / Gauss-Lobatto) kernels: orders 1..3
template <int DIM, class PH>
void pick_order_nc(tpsrhs_operator *op) {
  upload_tables(DIM, op->order, 1);
  op->point_eval = &launch_point_eval<PH>;
  if constexpr (PH::AXISYM) {
    throw Unsupported("the Gauss-Lobatto pair is built for the planar 2-D and the 3-D formulation");
  } else {
    switch (op->order) {
      case 1: op->launch = &launch_all<DIM, 1, PH, 1>; break;
      case 2: op->launch = &launch_all<DIM, 2, PH, 1>; break;
      case 3: op->launch = &launch_all<DIM, 3, PH, 1>; break;
      default: throw Unsupported("Gauss-Lobatto basis + rule: polynomial orders 1..3 are built");
    }
  }
}

template <int DIM, class PH>
void pick_order(tpsrhs_operator *op) {
  if (op->nc) {
    pick_order_nc<DIM, PH>(op);
    return;
  }
  upload_tables(DIM, op->order);
  op->point_eval = &launch_point_eval<PH>;
  switch (op->order) {
    case 1: op->launch = &launch_all<DIM, 1, PH>; break;
    case 2: op->launch = &launch_all<DIM, 2, PH>; break;
    case 3: op->launch = &launch_all<DIM, 3, PH>; break;
    case 4: op->launch = &launch_all<DIM, 4, PH>; break;
    case 5:  // MAXDOFS = 216 of the reference (src/dataStructures.hpp:41-65): light physics only
      if constexpr (PH::MAX_ORDER >= 5) {
        op->launch = &launch_all<DIM, 5, PH>;
        break;
      }
      [[fallthrough]];
    default:
      throw Unsupported("polynomial order " + std::to_string(op->order) + " is not built for this physics (1.." +
                        std::to_string(PH::MAX_ORDER) + ")");
  }
}

}  // namespace

#endif
