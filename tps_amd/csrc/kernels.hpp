// HIP kernels (gfx950) of the collocated DG right-hand side.
//
// One RHS evaluation = three element-centric streaming sweeps, none of which scatters:
//   k_traces   U -> face-node traces of U and Up                    (replaces the neighbour gathers of
//                                                                     Gradients::interpFaceData_gpu, src/gradients.cpp:292-384)
//   k_gradient U + neighbour Up traces -> gradUp, viscous normal-flux traces at the face quadrature
//              points                                               (src/gradients.cpp:144-232, src/faceGradientIntegration.cpp:40-140)
//   k_flux     U + gradUp + neighbour traces -> y                   (src/face_integrator.cpp:194-352, src/BCintegrator.cpp:295-441,
//                                                                     src/rhs_operator.cpp:493-559, src/domain_integrator.cpp:45-99,
//                                                                     src/rhs_operator.cpp:432-461)
//
// Mapping to the hardware.  A workgroup owns EPB whole elements, one lane per node: one 64-lane wave
// per hex at p=3.  Nodal fields live in LDS.  The element-independent 1-D operators (differentiation
// matrix, end-point values, node -> face-quadrature interpolation) are applied by sum factorisation.
// Face work is done one reference direction at a time (the two opposite faces xi_d = 0, 1), in
// "line" stages: a lane takes one line of N1 (or Q1) values out of LDS, produces all Q1 (or N1)
// outputs of that line in registers and writes them back -- every table coefficient is a compile
// time index into __constant__ memory, i.e. a scalar operand, and each LDS value is read once per
// line instead of once per output.  Geometry (Jacobians, area-weighted normals) is recomputed from
// the 2^dim vertex coordinates of the element: 192 B per hex instead of 10 doubles per node.
// LDS is one hand-allocated pool whose regions are re-used as fields die (10-18 KB per hex at p=3),
// so that 8-14 single-wave workgroups are resident per CU and hide each other's latencies.
#ifndef TPSRHS_KERNELS_HPP_
#define TPSRHS_KERNELS_HPP_

#include <hip/hip_runtime.h>

#include <type_traits>

#include "basis.hpp"
#include "physics_dryair.hpp"

namespace tpsrhs {

#ifndef TPSRHS_MINW_GRAD
#define TPSRHS_MINW_GRAD 1
#endif
#ifndef TPSRHS_MINW_FLUX
#define TPSRHS_MINW_FLUX 3  // <= 168 VGPRs: 3 waves per SIMD (the allocator otherwise lands on 170)
#endif
#ifndef TPSRHS_NT_STORES
#define TPSRHS_NT_STORES 0  // experiment: non-temporal stores of the streams k_gradient never reads back (Up, TB)
#endif
#ifndef TPSRHS_NO_MFMA
#define TPSRHS_NO_MFMA 0  // A/B switch: 1 = the dense inverse mass of the p = 3 Gauss-Lobatto hex on the vector ALU (rounds 2-3)
#endif
#ifndef TPSRHS_FLUX_LATE
#define TPSRHS_FLUX_LATE 0  // experiment: k_flux issues the neighbour records of its first direction pair after the nodal physics
#endif
#ifndef TPSRHS_ABLATE
#define TPSRHS_ABLATE 0  // timing experiments only (wrong results)
#endif
constexpr int cmax(int a, int b) { return a > b ? a : b; }

// Diagnostic builds only (-DTPSRHS_STAMP=1, tools/stamp_phases.py): s_memtime at the phase boundaries of
// k_gradient / k_flux of every block, summed per phase into a __device__ array that no kernel reads.  The
// production library carries none of this.
#ifndef TPSRHS_STAMP
#define TPSRHS_STAMP 0
#endif
#if TPSRHS_STAMP
constexpr int NSTAMP = 16, STAMP_BLOCKS = 1 << 16;
static __device__ unsigned int g_stamp[STAMP_BLOCKS][NSTAMP];  // cycles per phase of every block (one kernel at a time)
struct Stamper {
  unsigned long long t0;
  unsigned int acc[NSTAMP];
  __device__ inline unsigned long long now() {
    unsigned long long t;
    __builtin_amdgcn_sched_barrier(0);
    asm volatile("s_memtime %0\n\ts_waitcnt lgkmcnt(0)" : "=s"(t)::"memory");
    __builtin_amdgcn_sched_barrier(0);
    return t;
  }
  __device__ inline void start() {
    for (int i = 0; i < NSTAMP; i++) acc[i] = 0;
    t0 = now();
  }
  __device__ inline void mark(int phase) {
    const unsigned long long t = now();
    acc[phase] += static_cast<unsigned int>(t - t0);
    t0 = t;
  }
  __device__ inline void flush() {
    if (threadIdx.x == 0 && blockIdx.x < STAMP_BLOCKS)
      for (int i = 0; i < NSTAMP; i++) g_stamp[blockIdx.x][i] = acc[i];
  }
};
#endif
// Diagnostic builds only (-DTPSRHS_DUMPF=1, tools/probe_dumpf.py): the nodal flux and the point sources of every block of the
// last k_flux launch of this translation unit, copied from LDS / registers into a __device__ array that no kernel reads.
#ifndef TPSRHS_DUMPF
#define TPSRHS_DUMPF 0
#endif
#if TPSRHS_DUMPF
constexpr int DUMPF_MAX = 1 << 22;
static __device__ double g_dumpf[DUMPF_MAX];
#endif
#if TPSRHS_STAMP == 1  // k_gradient
#define STAMP_DECL Stamper stamper
#define STAMP_ARG , stamper
#define STAMP_PARAM , Stamper &stamper
#define STAMP_START() stamper.start()
#define STAMP(phase) stamper.mark(phase)
#define STAMP_FLUSH() stamper.flush()
#else
#define STAMP_DECL
#define STAMP_ARG
#define STAMP_PARAM
#define STAMP_START()
#define STAMP(phase)
#define STAMP_FLUSH()
#endif
#if TPSRHS_STAMP == 2  // k_flux
#define FSTAMP_DECL \
  Stamper stamper;  \
  stamper.start()
#define FSTAMP(phase) stamper.mark(phase)
#define FSTAMP_FLUSH() stamper.flush()
#else
#define FSTAMP_DECL
#define FSTAMP(phase)
#define FSTAMP_FLUSH()
#endif

// 1-D operator tables of every (dim, order), filled by tpsrhs_create.  Identical for all operators of
// a process (they depend on (dim, p) only).  Indexed with compile-time constants they are scalar loads.
// (first index: 0 the collocated Gauss-Legendre pair, 1 the non-collocated Gauss-Lobatto pair)
static __constant__ Tables1D c_tab[2][2][TPSRHS_MAXORDER + 1];

// A fresh view of a __constant__ table: loads through the result cannot be moved above this point (nor merged with
// earlier ones), so a stage placed after it fetches its coefficients with scalar loads where it uses them.  Without
// it the loop-invariant code motion keeps all 1-D operators (~70 SGPRs at p = 3) live across the point physics of
// the heavy kernels, where they are spilled to VGPR lanes and come back through v_readlane (VALU issue).
__device__ inline const Tables1D &fresh_table(const Tables1D &ct) {
  typedef const Tables1D __attribute__((address_space(4))) *CP;
  const unsigned long long a = reinterpret_cast<unsigned long long>(&ct);
  unsigned lo = __builtin_amdgcn_readfirstlane(static_cast<unsigned>(a));
  unsigned hi = __builtin_amdgcn_readfirstlane(static_cast<unsigned>(a >> 32));
  asm volatile("" : "+s"(lo), "+s"(hi));
  return *(const Tables1D *)reinterpret_cast<CP>((static_cast<unsigned long long>(hi) << 32) | lo);
}

// LDS read of one double (a hook: a volatile-typed variant that stops hipcc from fusing neighbouring
// 8-byte reads into ds_read2_b64 was measured and dropped -- volatile LDS accesses are followed by a
// full s_waitcnt, which serialises the line stages; see DESIGN.md "What was tried").
__device__ inline double ldsr(const double *p) { return *p; }

// Nodal field access fld[k*stride + n]: the field base is wave-uniform (SGPR pair) and the lane offset
// 32-bit, so that every load/store uses the scalar-base addressing form and no 64-bit per-lane
// address is kept in VGPRs (the host guarantees NDofs < 2^31).
__device__ inline const double *field_ptr(const double *base, int k, int64_t stride) { return base + k * stride; }
__device__ inline double *field_ptr(double *base, int k, int64_t stride) { return base + k * stride; }

// Synchronisation between the LDS stages of a block.  A single-wave workgroup needs no hardware
// barrier and no wait: the LDS executes the DS instructions of one wave in issue order, so a ds_read
// that follows a ds_write in program order sees it.  __syncthreads() would also drain the VMEM
// counter (vmcnt(0)) at every stage and expose the latency of every prefetched global load; the
// wave barrier only stops the compiler from reordering across it.
template <int BLOCK>
__device__ inline void block_sync() {
  if (BLOCK == 64) {
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
    __builtin_amdgcn_wave_barrier();
    __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
  } else {
    __syncthreads();
  }
}

// NC_ = 0: Gauss-Legendre basis + Gauss-Legendre rules (basisType 0, integrationRule 0): nodes and volume
//          quadrature points coincide, diagonal mass matrix -- the pair of the reference's cylinder / wedge / torch inputs;
// NC_ = 1: Gauss-Lobatto basis + Gauss-Lobatto rules (1, 1; the reference's defaults, src/M2ulPhyS.cpp:2671-2672):
//          p+2 volume points per direction, dense element mass matrix, every integral through quadrature points.
template <int DIM_, int P_, int NC_ = 0>
struct Cfg {
  static constexpr int DIM = DIM_, P = P_, N1 = P_ + 1, NC = NC_;
  static constexpr int NPE = (DIM_ == 3) ? N1 * N1 * N1 : N1 * N1;
  static constexpr int NF = (DIM_ == 3) ? N1 * N1 : N1;  // nodes of a face = lines through the element
  static constexpr int Q1 = ((DIM_ - 1) + 2 * P_) / 2 + 1 + NC_;  // face rule, order OrderW + 2p
  static constexpr int QV = P_ + 1 + NC_;                          // volume rule, order 2p
  static constexpr int NQV = (DIM_ == 3) ? QV * QV * QV : QV * QV;
  static constexpr int NQ = (DIM_ == 3) ? Q1 * Q1 : Q1;
  static constexpr int NW = (DIM_ == 3) ? Q1 * N1 : 1;  // half-interpolated face values (3-D only)
  static constexpr int NFACES = 2 * DIM_;
  static constexpr int NV = 1 << DIM_;
  // (the Gauss-Lobatto hexes of p = 2, 3 get two waves: their 100 / 72 face quadrature points per direction pair then
  // are ONE round of the block's lanes -- see EPB below)
  static constexpr int BLOCK = (NC_ && DIM_ == 3 && P_ >= 2) ? 128 : ((NPE <= 64) ? 64 : ((NPE <= 128) ? 128 : 256));
  // elements per block.  3-D p = 1: 3 elements = 54 face quadrature points per direction pair, one round of the
  // 64 lanes (4 elements needed a second round for 8 points, and with 11 equations that instantiation kept
  // 2 x 3 sets of prefetched traces live: 259 spilled VGPRs); the non-collocated pair has 16 points per face: 2
  // elements = 64 points, one round.  Round 2 ran that pair with 4 elements -- two rounds of face points in one wave,
  // the shape of round 2's failing 11-equation kernel -- and in round 3 the mixtures of 9+ equations with the argon
  // mixture transport returned a wrong species residual in the lanes of the block's third and fourth element, one of
  // them a memory fault (tools/probe_gll_p1.py): that shape is gone.
  // Every failing instantiation of rounds 2 and 3 had the same shape -- a single-wave block that takes TWO rounds of face
  // quadrature points (and then more than 256 registers with SGPR spills) --, and no other shape has failed in the
  // exhaustive sweep (tools/sweep_instantiations.py): in 3-D no block takes a second round any more.  Gauss-Lobatto
  // pair: p = 1: 2 hexes x 2 faces x 16 points = 64; p = 2: 2 hexes, 100 points, 128 lanes; p = 3: 1 hex, 72 points,
  // 128 lanes.  2-D, Gauss-Lobatto p = 1: 10 quads (60 points; 16 quads = 96 before).
  static constexpr int EPB = (DIM_ == 3 && NC_) ? (P_ == 3 ? 1 : 2)
                             : (DIM_ == 3 && P_ == 1) ? 3
                             : (DIM_ == 2 && NC_ && P_ == 1) ? 10
                                                             : BLOCK / NPE;
  static constexpr int NODES = EPB * NPE;
  // one direction pair (faces 2d, 2d+1) of a block
  static constexpr int PF = 2 * EPB;
  static constexpr int TN = PF * NF;
  static constexpr int TW = PF * NW;
  static constexpr int TQ = PF * NQ;
  static constexpr int LN = EPB * NF;
  static constexpr int Q_ROUNDS = (TQ + BLOCK - 1) / BLOCK;
  static_assert(DIM_ == 2 || Q_ROUNDS == 1, "3-D: one round of face quadrature points per direction pair");
  static_assert(!(DIM_ == 2 && NC_) || Q_ROUNDS == 1, "2-D Gauss-Lobatto pair: one round per direction pair");
  // 2-D: the quadrature points of BOTH direction pairs of a block are worked on together (at p = 3 they
  // are 2 x 32 = one full 64-lane round; one pair at a time leaves half of the lanes idle in the physics)
  static constexpr int TQ2 = 2 * TQ;
  static constexpr int Q2_ROUNDS = (TQ2 + BLOCK - 1) / BLOCK;
};

// LDS copy of the tables that are indexed per lane
template <class C>
struct Tab {
  double x[C::N1], w[C::N1], iw[C::N1], D[C::N1 * C::N1], b0[C::N1], b1[C::N1];
  double xq[C::Q1], wq[C::Q1], B[C::Q1 * C::N1];
  double xv[C::QV], wv[C::QV];  // volume rule (read by the non-collocated variant)
};

// ConstantPressureGradient / HeatSource / SpongeZone / JouleHeating of RHSoperator's forcing array
// (src/rhs_operator.cpp:101-166), evaluated per node in k_flux next to the point sources.  One block in
// device memory; the address is wave-uniform, so its fields arrive through scalar loads.
struct ForcingDev {
  int has_pg, nheat, nsponge, pad;
  double pg[3];
  struct Heat {
    double value, radius, len;  // len = |point2 - point1|
    double p1[3], axis[3];      // axis = unit vector point1 -> point2
  } heat[TPSRHS_MAXHEATSOURCES];
  struct Sponge {
    int type, pad;
    double normal[3], p0[3], pinit[3];
    double r1, r2, mult;
    double target[TPSRHS_MAXEQUATIONS];
    // mixed-out target (SpongeZoneSolution::MIXEDOUT): nodes of the mix-out plane, and the sums over them
    int mixed_out, n_plane;
    const int *plane_nodes;  // [n_plane] device
    double *msum;            // [NEQ + 1] device: sum of F_c . n per equation, number of nodes
  } sponge[TPSRHS_MAXSPONGEZONES];
  const double *joule;  // [ndofs] or NULL
  int nps, pad2;
  struct Scalar {
    double x0[3], radius, value;
  } ps[TPSRHS_MAXPASSIVESCALARS];
};

// RK4 stage combination fused into the epilogue of k_flux inside tpsrhs_rk4_step / tpsrhs_advance (src/M2ulPhyS.cpp:
// 2004-2008 around MFEM's RK4Solver::Step): the residual k of a node never goes to memory, the lane that owns the node
// writes the next stage's state (or the new solution) instead.
//   mode 0  plain Mult: Y = k
//   mode 1  out = U + dt/2 k           (stage 1: the stage input is x itself, taken from LDS)
//   mode 2  out = x0 + dt/2 k          mode 3  out = x0 + dt k
//   mode 4  out = x0 + [(y2 - x0) + 2 (y3 - x0) + (y4 - x0)] / 3 + dt/6 k,  then the NaN census and the species clamp
//           (Check_NAN, Check_Undershoot).  With y2 = x0 + dt/2 k1, y3 = x0 + dt/2 k2, y4 = x0 + dt k3 this is
//           x0 + dt/6 (k1 + 2 k2 + 2 k3 + k4): the accumulator z of RK4Solver::Step is recovered from the stage states
//           instead of being streamed through memory at every stage (6 vector passes per step instead of 14); the
//           differences are exact, the result differs from the reference's order of operations by rounding only.
struct RkDev {
  int mode, sp_first, sp_last, pad;
  double dt_host;
  const double *dt_dev;  // tpsrhs_advance keeps dt in device memory
  const double *x0, *y2, *y3, *y4;
  double *out;
  unsigned long long *nan_count;
  double *ta_out;  // non-NULL (stages 1..3 of a step, launch_all): the face-node traces of `out` go here -- the NEXT stage's
                   // k_traces sweep done where the new state sits in registers (3-D collocated kernels, flux_fuses_traces)
};

struct MeshDev {
  const int *blocks;           // workgroup -> block of EPB consecutive elements (NULL: identity); lets one
                               // launch cover the interior and another the blocks that touch shared faces
  int ne;
  int reverse;                 // 1: the sweep walks each XCD's chunk of the block list backwards (xcd_block)
  int64_t ndofs;
  const double *verts;         // [ne][NV][DIM] lexicographic corners
  const int2 *face_info;       // [ne*NFACES] {neighbour slot | -(bc+1), orientation code}
  const double *minv;          // non-collocated variant: [ne][NPE][NPE] inverse element mass matrices (symmetric)
  MixLenDev ml;                // MixingLengthTransport (2-D kernels): wall-distance grid function, or distance = NULL
  VsDev vs;                    // viscous sponge of the 2-D kernels with the heavy interface (enabled = 0: none)
};
// what the closures of the 2-D heavy kernels take of the mixing-length model and the viscous sponge at one point
// (`park`: the block's LDS array of sponge weights, one word per lane -- see EddyCtx::vsw)
__device__ inline EddyCtx closure_ctx(const MeshDev &m, bool dist_on, double dist, const double *X, double *park) {
  EddyCtx ec = dist_on ? eddy_at(m.ml, dist) : eddy_off();
  if (m.vs.enabled) {  // uniform over the grid
    park[threadIdx.x] = visc_sponge_weight_2d(m.vs, X);
    ec.vsw = park;
  }
  return ec;
}

// XCD-aware block order.  The hardware hands workgroup b of a launch to XCD b % 8, each XCD with an L2 of its own: with
// the identity order an element and its face neighbours (e +- 1, e +- nr, ...) sit on eight different L2s, and the trace
// records that BOTH sides of a face read are never found in the L2 by the second reader.  Here the workgroups of one XCD
// take one contiguous eighth of the block list, so that neighbours (all but those across the seven chunk borders) share
// an L2.  A bijection of [0, n): XCD x runs the workgroups x, x + 8, ... -- q + (x < r) of them -- and owns the chunk
// [x q + min(x, r), ...) of that length.
#ifndef TPSRHS_GRAD_LATE
#define TPSRHS_GRAD_LATE 1  // (0: A/B)
#endif
#ifndef TPSRHS_GRAD_LATE_NC
#define TPSRHS_GRAD_LATE_NC 0  // (measured: gll_dry unchanged, 264 -> 244 registers notwithstanding)
#endif
// k_flux of the two-step (plasma) kernels: the nodal gradient (NEQ * DIM values per lane) is parked in the lane's own column
// of the -- still unused -- flux region of the LDS pool while the state-only closure runs, instead of being held in
// registers across it (A/B switch)
#ifndef TPSRHS_FLUX_PARK_GRAD
#define TPSRHS_FLUX_PARK_GRAD 0
#endif
#ifndef TPSRHS_GRAD_LATE_ALL
#define TPSRHS_GRAD_LATE_ALL 0
#endif
#ifndef TPSRHS_XCD_ORDER
#define TPSRHS_XCD_ORDER 1
#endif
__device__ inline int xcd_block(int b, int n, int reverse = 0) {
#if TPSRHS_XCD_ORDER
  constexpr int X = 8;
  const int q = n / X, r = n - q * X;
  const int x = b % X, k = b / X;
  // reverse: the same chunk, last block first.  Consecutive sweeps of a Mult run in opposite directions (operator.hpp):
  // what the previous sweep wrote LAST -- still in this XCD's L2 and in the memory-side cache -- is read FIRST, instead of
  // every sweep streaming 0.6 - 1 GB through a least-recently-used cache in the one order that never hits
  const int kk = reverse ? (q + (x < r ? 1 : 0)) - 1 - k : k;
  return x * q + (x < r ? x : r) + kk;
#else
  return reverse ? n - 1 - b : b;
#endif
}

// Face records of the block's elements -> LDS, once, so that no later stage has a global load on the
// path to a neighbour address (nb = INT32_MIN marks the faces of elements past the end of the mesh).
template <class C>
__device__ inline void load_face_info(int2 *sFI, const MeshDev &m, int e0) {
  for (int i = threadIdx.x; i < C::EPB * C::NFACES; i += C::BLOCK) {
    const int e = e0 + i / C::NFACES;
    sFI[i] = (e < m.ne) ? m.face_info[static_cast<int64_t>(e0) * C::NFACES + i] : make_int2(INT32_MIN, 0);
  }
}

// ---------------------------------------------------------------------------------------------
template <class C>
__device__ constexpr int stride_of(int d) {
  return d == 0 ? 1 : (d == 1 ? C::N1 : C::N1 * C::N1);
}
template <class C>
__device__ constexpr int tan_a(int d) {
  return (C::DIM == 2) ? 1 - d : ((d == 0) ? 1 : 0);
}
template <class C>
__device__ constexpr int tan_b(int d) {
  return (d == 2) ? 1 : 2;
}
// permutation of a tangential index pair under an orientation code (n symmetric points per
// direction): my (ia, ib) -> neighbour's flat index
template <int DIM>
__device__ inline int permute(int o, int n, int ia, int ib) {
  const int fa = (o >> 1) & 1, fb = (o >> 2) & 1;
  if (DIM == 2) return fa ? n - 1 - ia : ia;
  int ja, jb;
  if (!(o & 1)) {
    ja = fa ? n - 1 - ia : ia;
    jb = fb ? n - 1 - ib : ib;
  } else {
    ja = fa ? n - 1 - ib : ib;
    jb = fb ? n - 1 - ia : ia;
  }
  return ja + n * jb;
}

// Jacobian J[i + m*DIM] = dx_i/dxi_m of the multilinear element at reference point xi
template <int DIM>
__device__ inline void jacobian(const double *V, const double *xi, double *J) {
  if (DIM == 2) {
    const double x = xi[0], y = xi[1];
#pragma unroll
    for (int i = 0; i < 2; i++) {
      const double v00 = V[0 * 2 + i], v10 = V[1 * 2 + i], v01 = V[2 * 2 + i], v11 = V[3 * 2 + i];
      J[i + 0 * 2] = (v10 - v00) * (1.0 - y) + (v11 - v01) * y;
      J[i + 1 * 2] = (v01 - v00) * (1.0 - x) + (v11 - v10) * x;
    }
  } else {
    const double x = xi[0], y = xi[1], z = xi[2];
    const double x0 = 1.0 - x, y0 = 1.0 - y, z0 = 1.0 - z;
#pragma unroll
    for (int i = 0; i < 3; i++) {
      const double v000 = V[0 * 3 + i], v100 = V[1 * 3 + i], v010 = V[2 * 3 + i], v110 = V[3 * 3 + i];
      const double v001 = V[4 * 3 + i], v101 = V[5 * 3 + i], v011 = V[6 * 3 + i], v111 = V[7 * 3 + i];
      J[i + 0 * 3] = (v100 - v000) * y0 * z0 + (v110 - v010) * y * z0 + (v101 - v001) * y0 * z + (v111 - v011) * y * z;
      J[i + 1 * 3] = (v010 - v000) * x0 * z0 + (v110 - v100) * x * z0 + (v011 - v001) * x0 * z + (v111 - v101) * x * z;
      J[i + 2 * 3] = (v001 - v000) * x0 * y0 + (v101 - v100) * x * y0 + (v011 - v010) * x0 * y + (v111 - v110) * x * y;
    }
  }
}
// physical coordinates of reference point xi (radius of the axisymmetric formulation; node positions of
// the optional forcing terms)
template <int DIM>
__device__ inline void position(const double *V, const double *xi, double *X) {
  const double x = xi[0], y = xi[1];
  if constexpr (DIM == 2) {
#pragma unroll
    for (int i = 0; i < 2; i++) {
      const double v00 = V[0 * 2 + i], v10 = V[1 * 2 + i], v01 = V[2 * 2 + i], v11 = V[3 * 2 + i];
      X[i] = (v00 * (1.0 - x) + v10 * x) * (1.0 - y) + (v01 * (1.0 - x) + v11 * x) * y;
    }
  } else {
    const double z = xi[2];
#pragma unroll
    for (int i = 0; i < 3; i++) {
      const double v000 = V[0 * 3 + i], v100 = V[1 * 3 + i], v010 = V[2 * 3 + i], v110 = V[3 * 3 + i];
      const double v001 = V[4 * 3 + i], v101 = V[5 * 3 + i], v011 = V[6 * 3 + i], v111 = V[7 * 3 + i];
      const double lo = (v000 * (1.0 - x) + v100 * x) * (1.0 - y) + (v010 * (1.0 - x) + v110 * x) * y;
      const double hi = (v001 * (1.0 - x) + v101 * x) * (1.0 - y) + (v011 * (1.0 - x) + v111 * x) * y;
      X[i] = lo * (1.0 - z) + hi * z;
    }
  }
}
// adjugate A[m + i*DIM] = det(J) * dxi_m/dx_i and determinant
template <int DIM>
__device__ inline double adjugate(const double *J, double *A) {
  if (DIM == 2) {
    A[0 + 0 * 2] = J[1 + 1 * 2];
    A[0 + 1 * 2] = -J[0 + 1 * 2];
    A[1 + 0 * 2] = -J[1 + 0 * 2];
    A[1 + 1 * 2] = J[0 + 0 * 2];
    return J[0] * J[3] - J[2] * J[1];
  } else {
    A[0 + 0 * 3] = J[1 + 1 * 3] * J[2 + 2 * 3] - J[1 + 2 * 3] * J[2 + 1 * 3];
    A[0 + 1 * 3] = J[0 + 2 * 3] * J[2 + 1 * 3] - J[0 + 1 * 3] * J[2 + 2 * 3];
    A[0 + 2 * 3] = J[0 + 1 * 3] * J[1 + 2 * 3] - J[0 + 2 * 3] * J[1 + 1 * 3];
    A[1 + 0 * 3] = J[1 + 2 * 3] * J[2 + 0 * 3] - J[1 + 0 * 3] * J[2 + 2 * 3];
    A[1 + 1 * 3] = J[0 + 0 * 3] * J[2 + 2 * 3] - J[0 + 2 * 3] * J[2 + 0 * 3];
    A[1 + 2 * 3] = J[0 + 2 * 3] * J[1 + 0 * 3] - J[0 + 0 * 3] * J[1 + 2 * 3];
    A[2 + 0 * 3] = J[1 + 0 * 3] * J[2 + 1 * 3] - J[1 + 1 * 3] * J[2 + 0 * 3];
    A[2 + 1 * 3] = J[0 + 1 * 3] * J[2 + 0 * 3] - J[0 + 0 * 3] * J[2 + 1 * 3];
    A[2 + 2 * 3] = J[0 + 0 * 3] * J[1 + 1 * 3] - J[0 + 1 * 3] * J[1 + 0 * 3];
    return J[0 + 0 * 3] * A[0 + 0 * 3] + J[0 + 1 * 3] * A[1 + 0 * 3] + J[0 + 2 * 3] * A[2 + 0 * 3];
  }
}

// Area-weighted outward normal at face quadrature point q of face (D, s), from the face's own corner
// coordinates: on a multilinear element the two tangents of a face are linear in the other
// tangential coordinate, so n = +-(dx/dta x dx/dtb) costs two lerps and a cross product
// (CalcOrtho of the face Jacobian, src/face_integrator.cpp:323).  X = position (2-D only).
template <class C, int D>
__device__ inline void face_geometry(const double *V, const Tab<C> &tab, int s, int q, double *n, double &wq,
                                     double *X) {
  if (C::DIM == 2) {
    constexpr int a = 1 - D;
    const int c0 = (s << D), c1 = (s << D) | (1 << a);
    const double tx = V[c1 * 2 + 0] - V[c0 * 2 + 0], ty = V[c1 * 2 + 1] - V[c0 * 2 + 1];
    const double sg = (s ? 1.0 : -1.0) * (D == 0 ? 1.0 : -1.0);
    n[0] = sg * ty;
    n[1] = -sg * tx;
    wq = tab.wq[q];
    const double t = tab.xq[q];
    X[0] = V[c0 * 2 + 0] + t * tx;
    X[1] = V[c0 * 2 + 1] + t * ty;
  } else {
    constexpr int a = tan_a<C>(D), b = tan_b<C>(D);
    const int qa = q % C::Q1, qb = q / C::Q1;
    const double ta = tab.xq[qa], tb = tab.xq[qb];
    wq = tab.wq[qa] * tab.wq[qb];
    const int c00 = (s << D), c10 = c00 | (1 << a), c01 = c00 | (1 << b), c11 = c10 | (1 << b);
    double va[3], vb[3];
#pragma unroll
    for (int i = 0; i < 3; i++) {
      const double p00 = V[c00 * 3 + i], p10 = V[c10 * 3 + i], p01 = V[c01 * 3 + i], p11 = V[c11 * 3 + i];
      const double ea0 = p10 - p00, ea1 = p11 - p01, eb0 = p01 - p00, eb1 = p11 - p10;
      va[i] = ea0 + tb * (ea1 - ea0);
      vb[i] = eb0 + ta * (eb1 - eb0);
    }
    const double sg = (s ? 1.0 : -1.0) * (D == 1 ? -1.0 : 1.0);
    n[0] = sg * (va[1] * vb[2] - va[2] * vb[1]);
    n[1] = sg * (va[2] * vb[0] - va[0] * vb[2]);
    n[2] = sg * (va[0] * vb[1] - va[1] * vb[0]);
#pragma unroll
    for (int i = 0; i < 3; i++)  // position (read by the viscous sponge only; dropped by the compiler elsewhere)
      X[i] = V[c00 * 3 + i] + ta * (V[c10 * 3 + i] - V[c00 * 3 + i]) + tb * vb[i];
  }
}

// ---------------------------------------------------------------------------------------------
// Line stages.  Field-major LDS arrays: T[fld][TN] face nodes, W[fld][TW] half-interpolated,
// R[fld][TQ] quadrature values, L[fld][TN] projected values; pf = 2*le + s indexes the faces of the
// direction pair.  `ct` is the __constant__ table: all its indices below are compile-time.

// traces on both faces of direction D of NFLD nodal fields F[fld][NODES]: one lane per line
template <class C, int D, int NFLD>
__device__ inline void trace_lines(const double *F, double *T, const Tables1D &ct, int tid) {
  constexpr int DD = ((TPSRHS_ABLATE & 64) && C::DIM == 3 && D == 1) ? 2 : D;  // timing experiment: conflict-free pattern
  constexpr int sd = stride_of<C>(DD);
  constexpr int sa = stride_of<C>(tan_a<C>(DD));
  constexpr int sb = (C::DIM == 3) ? stride_of<C>(tan_b<C>(DD)) : 0;
  for (int item = tid; item < NFLD * C::LN; item += C::BLOCK) {
    const int fld = item / C::LN, r = item - fld * C::LN;
    const int le = r / C::NF, ln = r - le * C::NF;
    const int base = le * C::NPE + ((C::DIM == 2) ? ln * sa : (ln % C::N1) * sa + (ln / C::N1) * sb);
    const double *f = F + fld * C::NODES + base;
    double t0 = 0.0, t1 = 0.0;
#pragma unroll
    for (int i = 0; i < C::N1; i++) {
      const double v = ldsr(&f[i * sd]);
      t0 += ct.b0[i] * v;
      t1 += ct.b1[i] * v;
    }
    T[fld * C::TN + (2 * le) * C::NF + ln] = t0;
    T[fld * C::TN + (2 * le + 1) * C::NF + ln] = t1;
  }
}

// 3-D: W[fld][pf][qa*N1 + jb] = sum_ja B[qa][ja] T[fld][pf][ja + N1*jb]; one lane per (fld, pf, jb)
template <class C, int NFLD>
__device__ inline void interp1_lines(const double *T, double *W, const Tables1D &ct, int tid) {
  if (C::DIM == 2) return;
  for (int item = tid; item < NFLD * C::PF * C::N1; item += C::BLOCK) {
    const int fld = item / (C::PF * C::N1), r = item - fld * (C::PF * C::N1);
    const int pf = r / C::N1, jb = r - pf * C::N1;
    const double *t = T + fld * C::TN + pf * C::NF + C::N1 * jb;
    double v[C::N1];
#pragma unroll
    for (int ja = 0; ja < C::N1; ja++) v[ja] = ldsr(&t[ja]);
    double *w = W + fld * C::TW + pf * C::NW + jb;
#pragma unroll
    for (int qa = 0; qa < C::Q1; qa++) {
      double acc = 0.0;
#pragma unroll
      for (int ja = 0; ja < C::N1; ja++) acc += ct.B[qa * C::N1 + ja] * v[ja];
      w[qa * C::N1] = acc;
    }
  }
}
// value of one field at quadrature point (pf, q); bq = B[qb][.] (3-D) or B[q][.] (2-D) of this lane
template <class C>
__device__ inline double interp2_point(const double *T, const double *W, const double *bq, int pf, int q) {
  double acc = 0.0;
  if (C::DIM == 2) {
#pragma unroll
    for (int a = 0; a < C::N1; a++) acc += bq[a] * ldsr(&T[pf * C::NF + a]);
  } else {
    const int qa = q % C::Q1;
    const double *w = W + pf * C::NW + qa * C::N1;
#pragma unroll
    for (int jb = 0; jb < C::N1; jb++) acc += bq[jb] * ldsr(&w[jb]);
  }
  return acc;
}
// 3-D: W2[fld][pf][ja*Q1 + qb] = sum_qa B[qa][ja] R[fld][pf][qa + Q1*qb]; one lane per (fld, pf, qb)
template <class C, int NFLD>
__device__ inline void project1_lines(const double *R, double *W2, const Tables1D &ct, int tid) {
  if (C::DIM == 2) return;
  for (int item = tid; item < NFLD * C::PF * C::Q1; item += C::BLOCK) {
    const int fld = item / (C::PF * C::Q1), r = item - fld * (C::PF * C::Q1);
    const int pf = r / C::Q1, qb = r - pf * C::Q1;
    const double *rr = R + fld * C::TQ + pf * C::NQ + C::Q1 * qb;
    double v[C::Q1];
#pragma unroll
    for (int qa = 0; qa < C::Q1; qa++) v[qa] = ldsr(&rr[qa]);
    double *w = W2 + fld * C::TW + pf * C::NW + qb;
#pragma unroll
    for (int ja = 0; ja < C::N1; ja++) {
      double acc = 0.0;
#pragma unroll
      for (int qa = 0; qa < C::Q1; qa++) acc += ct.B[qa * C::N1 + ja] * v[qa];
      w[ja * C::Q1] = acc;
    }
  }
}
// L[fld][pf][ja + N1*jb] = sum_qb B[qb][jb] W2[fld][pf][ja*Q1 + qb] (3-D, one lane per (fld, pf, ja));
// 2-D: L[fld][pf][ja] = sum_q B[q][ja] R[fld][pf][q] (one lane per (fld, pf))
template <class C, int NFLD>
__device__ inline void project2_lines(const double *RW, double *L, const Tables1D &ct, int tid) {
  if (C::DIM == 2) {
    for (int item = tid; item < NFLD * C::PF; item += C::BLOCK) {
      const int fld = item / C::PF, pf = item - fld * C::PF;
      const double *rr = RW + fld * C::TQ + pf * C::NQ;
      double v[C::Q1];
#pragma unroll
      for (int q = 0; q < C::Q1; q++) v[q] = ldsr(&rr[q]);
#pragma unroll
      for (int ja = 0; ja < C::N1; ja++) {
        double acc = 0.0;
#pragma unroll
        for (int q = 0; q < C::Q1; q++) acc += ct.B[q * C::N1 + ja] * v[q];
        L[fld * C::TN + pf * C::NF + ja] = acc;
      }
    }
  } else {
    for (int item = tid; item < NFLD * C::PF * C::N1; item += C::BLOCK) {
      const int fld = item / (C::PF * C::N1), r = item - fld * (C::PF * C::N1);
      const int pf = r / C::N1, ja = r - pf * C::N1;
      const double *w = RW + fld * C::TW + pf * C::NW + ja * C::Q1;
      double v[C::Q1];
#pragma unroll
      for (int qb = 0; qb < C::Q1; qb++) v[qb] = ldsr(&w[qb]);
      double *l = L + fld * C::TN + pf * C::NF + ja;
#pragma unroll
      for (int jb = 0; jb < C::N1; jb++) {
        double acc = 0.0;
#pragma unroll
        for (int qb = 0; qb < C::Q1; qb++) acc += ct.B[qb * C::N1 + jb] * v[qb];
        l[jb * C::N1] = acc;
      }
    }
  }
}
// lifting of the pair's two faces to a volume node: b0(idx_D) L[pf=2le] + b1(idx_D) L[pf=2le+1]
// area-weighted normal of face (D, s) at tangential reference coordinates (ta, tb) (tb unused in 2-D)
template <class C, int D>
__device__ inline void face_normal_at(const double *V, int s, double ta, double tb, double *n) {
  if (C::DIM == 2) {
    constexpr int a = 1 - D;
    const int c0 = (s << D), c1 = (s << D) | (1 << a);
    const double tx = V[c1 * 2 + 0] - V[c0 * 2 + 0], ty = V[c1 * 2 + 1] - V[c0 * 2 + 1];
    const double sg = (s ? 1.0 : -1.0) * (D == 0 ? 1.0 : -1.0);
    n[0] = sg * ty;
    n[1] = -sg * tx;
  } else {
    constexpr int a = tan_a<C>(D), b = tan_b<C>(D);
    const int c00 = (s << D), c10 = c00 | (1 << a), c01 = c00 | (1 << b), c11 = c10 | (1 << b);
    double va[3], vb[3];
#pragma unroll
    for (int i = 0; i < 3; i++) {
      const double p00 = V[c00 * 3 + i], p10 = V[c10 * 3 + i], p01 = V[c01 * 3 + i], p11 = V[c11 * 3 + i];
      const double ea0 = p10 - p00, ea1 = p11 - p01, eb0 = p01 - p00, eb1 = p11 - p10;
      va[i] = ea0 + tb * (ea1 - ea0);
      vb[i] = eb0 + ta * (eb1 - eb0);
    }
    const double sg = (s ? 1.0 : -1.0) * (D == 1 ? -1.0 : 1.0);
    n[0] = sg * (va[1] * vb[2] - va[2] * vb[1]);
    n[1] = sg * (va[2] * vb[0] - va[0] * vb[2]);
    n[2] = sg * (va[0] * vb[1] - va[1] * vb[0]);
  }
}
template <class C, int D>
__device__ inline double lift_pair(const double *Lf, const Tab<C> &tab, int le, const int *idx) {
  const int fn = (C::DIM == 2) ? idx[tan_a<C>(D)] : idx[tan_a<C>(D)] + C::N1 * idx[tan_b<C>(D)];
  return tab.b0[idx[D]] * ldsr(&Lf[(2 * le) * C::NF + fn]) + tab.b1[idx[D]] * ldsr(&Lf[(2 * le + 1) * C::NF + fn]);
}

template <class C>
__device__ inline void load_tables(Tab<C> &t, const Tables1D &src) {
  const int tid = threadIdx.x;
  if (tid < C::N1) {
    t.x[tid] = src.x[tid];
    t.w[tid] = src.w[tid];
    t.iw[tid] = 1.0 / src.w[tid];
    t.b0[tid] = src.b0[tid];
    t.b1[tid] = src.b1[tid];
  }
  if (tid < C::Q1) {
    t.xq[tid] = src.xq[tid];
    t.wq[tid] = src.wq[tid];
  }
  if (tid < C::QV) {
    t.xv[tid] = src.xv[tid];
    t.wv[tid] = src.wv[tid];
  }
  for (int i = tid; i < C::N1 * C::N1; i += C::BLOCK) t.D[i] = src.D[i];
  for (int i = tid; i < C::Q1 * C::N1; i += C::BLOCK) t.B[i] = src.B[i];
}
template <class C>
__device__ inline void load_vertices(double *sV, const MeshDev &m, int e0) {
  constexpr int PER = C::NV * C::DIM;
  for (int i = threadIdx.x; i < C::EPB * PER; i += C::BLOCK) {
    const int le = i / PER;
    if (e0 + le < m.ne) sV[i] = m.verts[static_cast<int64_t>(e0) * PER + i];
  }
}
// Neighbour face-node traces of fields [f0, f0+NFLD) of the TA records of one direction pair, held in
// registers: issued early (the global latency overlaps other work), stored to LDS when the pass runs.
template <class C, int NFLD>
struct NbTraces {
  static constexpr int ROUNDS = (C::TN + C::BLOCK - 1) / C::BLOCK;
  double v[ROUNDS][NFLD];
  int nb[ROUNDS];  // INT32_MIN: no item; < 0: boundary face; >= 0: neighbour slot (values loaded)
};
template <class C, int D, int NFLD>
__device__ inline void issue_neighbour_traces(const int2 *sFI, const double *__restrict__ TA, int rec, int f0,
                                              NbTraces<C, NFLD> &t, int tid) {
#pragma unroll
  for (int r = 0; r < NbTraces<C, NFLD>::ROUNDS; r++) {
    const int item = tid + r * C::BLOCK;
    t.nb[r] = INT32_MIN;
#pragma unroll
    for (int k = 0; k < NFLD; k++) t.v[r][k] = 0.0;
    if (item < C::TN) {
      const int pf = item / C::NF, fn = item - pf * C::NF;
      const int2 fi = sFI[(pf >> 1) * C::NFACES + 2 * D + (pf & 1)];
      t.nb[r] = fi.x;
      if (fi.x >= 0 && !(TPSRHS_ABLATE & 1)) {
        const int pn = permute<C::DIM>(fi.y, C::N1, fn % C::N1, fn / C::N1);
        const double *src = TA + static_cast<int64_t>(fi.x) * rec + f0 * C::NF + pn;
#pragma unroll
        for (int k = 0; k < NFLD; k++) t.v[r][k] = src[k * C::NF];
      }
    }
  }
}
// -> Tnb[fld][TN]; boundary faces copy my own trace (already in Town[fld][TN])
template <class C, int NFLD>
__device__ inline void store_neighbour_traces(const NbTraces<C, NFLD> &t, const double *Town, double *Tnb, int tid) {
#pragma unroll
  for (int r = 0; r < NbTraces<C, NFLD>::ROUNDS; r++) {
    const int item = tid + r * C::BLOCK;
    if (t.nb[r] == INT32_MIN) continue;
    if (t.nb[r] >= 0) {
#pragma unroll
      for (int k = 0; k < NFLD; k++) Tnb[k * C::TN + item] = t.v[r][k];
    } else {
#pragma unroll
      for (int k = 0; k < NFLD; k++) Tnb[k * C::TN + item] = ldsr(&Town[k * C::TN + item]);
    }
  }
}
// Viscous normal-flux traces (own and neighbour) at this lane's quadrature points of one direction pair
template <class C, int NEQ>
struct NbFlux {
  double own[C::Q_ROUNDS][NEQ], nbv[C::Q_ROUNDS][NEQ];
  int nb[C::Q_ROUNDS];
};
template <class C, int D, int NEQ>
__device__ inline void issue_visc_traces(const int2 *sFI, int e0, const double *__restrict__ TB, NbFlux<C, NEQ> &t,
                                         int tid) {
#pragma unroll
  for (int r = 0; r < C::Q_ROUNDS; r++) {
    const int item = tid + r * C::BLOCK;
    t.nb[r] = INT32_MIN;
#pragma unroll
    for (int k = 0; k < NEQ; k++) t.own[r][k] = t.nbv[r][k] = 0.0;
    if (item < C::TQ) {
      const int pf = item / C::NQ, q = item - pf * C::NQ;
      const int lslot = (pf >> 1) * C::NFACES + 2 * D + (pf & 1);
      const int2 fi = sFI[lslot];
      t.nb[r] = fi.x;
      if (fi.x != INT32_MIN && !(TPSRHS_ABLATE & 1)) {
        // the records hold equations 1 .. NEQ-1: the viscous flux of the continuity equation is zero
        const double *o = TB + (static_cast<int64_t>(e0) * C::NFACES + lslot) * ((NEQ - 1) * C::NQ) + q;
#pragma unroll
        for (int k = 1; k < NEQ; k++) t.own[r][k] = o[(k - 1) * C::NQ];
        if (fi.x >= 0) {
          const int pq = permute<C::DIM>(fi.y, C::Q1, q % C::Q1, q / C::Q1);
          const double *b2 = TB + static_cast<int64_t>(fi.x) * ((NEQ - 1) * C::NQ) + pq;
#pragma unroll
          for (int k = 1; k < NEQ; k++) t.nbv[r][k] = b2[(k - 1) * C::NQ];
        }
      }
    }
  }
}

// =============================================================================================
// sweep 0: traces of U and Up at the face nodes.  TA[slot][2*NEQ][NF], slot = e*NFACES + f;
// fields 0..NEQ-1 = U, NEQ..2NEQ-1 = Up
// =============================================================================================
template <class C, class PH, int D>
__device__ inline void traces_dir(const MeshDev &m, int e0, const double *sF, double *sT, double *__restrict__ TA,
                                  const Tables1D &ct, int tid) {
  constexpr int NEQ = PH::NEQ;
  trace_lines<C, D, 2 * NEQ>(sF, sT, ct, tid);
  block_sync<C::BLOCK>();
  for (int item = tid; item < C::TN; item += C::BLOCK) {
    const int pf = item / C::NF, fn = item - pf * C::NF;
    const int e = e0 + (pf >> 1);
    if (e >= m.ne) continue;
    double *out = TA + (static_cast<int64_t>(e) * C::NFACES + 2 * D + (pf & 1)) * (2 * NEQ * C::NF) + fn;
#pragma unroll
    for (int fld = 0; fld < 2 * NEQ; fld++) out[fld * C::NF] = ldsr(&sT[fld * C::TN + item]);
  }
  block_sync<C::BLOCK>();
}

template <class C, class PH>
__global__ __launch_bounds__(C::BLOCK) void k_traces(MeshDev m, typename PH::KArg prm_k, const double *__restrict__ U,
                                                     double *__restrict__ TA) {
  typename PH::PRef prm = PH::pref(prm_k);
  constexpr int NEQ = PH::NEQ;
  const Tables1D &ct = c_tab[C::NC][C::DIM - 2][C::P];
  __shared__ double sF[2 * NEQ * C::NODES];
  __shared__ double sT[(C::DIM == 2 ? 2 : 1) * 2 * NEQ * C::TN];
  const int tid = threadIdx.x;
  const int lin = xcd_block(static_cast<int>(blockIdx.x), static_cast<int>(gridDim.x), m.reverse);
  const int bid = m.blocks ? m.blocks[lin] : lin;
  const int e0 = bid * C::EPB;
  if (tid < C::NODES) {
    const int le = tid / C::NPE, nd = tid - le * C::NPE;
    const int e = e0 + le;
    if (e < m.ne) {
      const unsigned n = static_cast<unsigned>(e) * C::NPE + nd;
      double u[NEQ], up[NEQ];
#pragma unroll
      for (int eq = 0; eq < NEQ; eq++) u[eq] = field_ptr(U, eq, m.ndofs)[n];
      PH::prim(prm, u, up);
#pragma unroll
      for (int eq = 0; eq < NEQ; eq++) {
        sF[eq * C::NODES + tid] = u[eq];
        sF[(NEQ + eq) * C::NODES + tid] = up[eq];
      }
    }
  }
  block_sync<C::BLOCK>();
  if constexpr (C::DIM == 2) {
    // quads: a face record field is p+1 doubles (32 B at p = 3), so the per-direction store of traces_dir writes
    // 32-byte pieces.  Both direction pairs go to LDS first; the records of a block's elements are contiguous in TA
    // (slot = e * NFACES + f), so the block then streams them out as one run of full 512-byte rows.
    trace_lines<C, 0, 2 * NEQ>(sF, sT, ct, tid);
    trace_lines<C, 1, 2 * NEQ>(sF, sT + 2 * NEQ * C::TN, ct, tid);
    block_sync<C::BLOCK>();
    constexpr int REC = 2 * NEQ * C::NF, PER_E = C::NFACES * REC;
    double *out = TA + static_cast<int64_t>(e0) * PER_E;
    for (int o = tid; o < C::EPB * PER_E; o += C::BLOCK) {
      const int le = o / PER_E, r = o - le * PER_E;
      const int f = r / REC, r2 = r - f * REC;
      const int fld = r2 / C::NF, fn = r2 - fld * C::NF;
      if (e0 + le < m.ne) out[o] = ldsr(&sT[(f >> 1) * (2 * NEQ * C::TN) + fld * C::TN + (2 * le + (f & 1)) * C::NF + fn]);
    }
  } else {
    traces_dir<C, PH, 0>(m, e0, sF, sT, TA, ct, tid);
    traces_dir<C, PH, 1>(m, e0, sF, sT, TA, ct, tid);
    traces_dir<C, PH, (C::DIM == 3 ? 2 : 0)>(m, e0, sF, sT, TA, ct, tid);
  }
}

// =============================================================================================
// sweep 1: gradient of the primitives (BR1-type: volume derivative + face jump lifting, diagonal
// inverse mass) and the viscous normal-flux traces TB[slot][NEQ-1][NQ] (equations 1 .. NEQ-1):
//   interior / shared face: F_v(U_q, gradUp_q) . n_out   (the consumer forms -1/2 (own - neighbour))
//   boundary face:          the complete additive viscous term of the boundary flux
// =============================================================================================

// =============================================================================================
// Non-collocated variant (Cfg::NC: Gauss-Lobatto basis + Gauss-Lobatto rules, the reference's default pair).
// Nodes and quadrature points differ, so the three element operators of the reference -- Ke (src/gradients.cpp:
// 84-133), the (v, grad w) blocks of Aflux (src/domain_integrator.cpp:45-99) and Me_inv (src/rhs_operator.cpp:
// 173-224) -- no longer collapse to point-wise products.  Ke and Aflux are applied by sum factorisation through
// the QV^dim volume quadrature points (never formed: at p = 3 they are 98 KB per element each); the inverse
// mass matrix of a non-affine element has no tensor structure and is applied as the dense NPE x NPE block the
// reference stores too (32 KB per hex at p = 3, streamed once per sweep).
// =============================================================================================
#ifndef TPSRHS_NC_NFC
#define TPSRHS_NC_NFC 3  // (round 4: 2 -> 3, gll_dry 3.23 -> 3.07 ms; 5 -- one chunk for dry air -- halves the occupancy: 4.58 ms)
#endif
constexpr int NC_NFC = TPSRHS_NC_NFC;  // fields per chunk of the volume operators (LDS scratch per chunk ~ 9 KB per field at p = 3)

// One axis of a tensor-product operator on `nbatch` arrays [d2][d1][d0] in LDS:
//   out[b][..r..] = sum_a c(r, a) in[b][..a..] along axis AX (0 = fastest), a < NIN = d_AX, r < NOUT;
//   c(r, a) = M[r * NIN + a], or M[a * NOUT + r] with TR (the transposed operator: quadrature -> nodes).
// M is a __constant__ table indexed at compile time (scalar operands); one lane per line.
template <int BLOCK, int AX, int NIN, int NOUT, bool TR>
__device__ inline void tensor_axis(const double *in, double *out, const double *M, int nbatch, int d0, int d1, int d2,
                                   int tid) {
  const int size_in = d0 * d1 * d2, lines = size_in / NIN, size_out = lines * NOUT;
  for (int item = tid; item < nbatch * lines; item += BLOCK) {
    const int b = item / lines, l = item - b * lines;
    int off_in, off_out, s_in, s_out;
    if (AX == 0) {  // l = j + d1 k
      off_in = l * d0;
      off_out = l * NOUT;
      s_in = s_out = 1;
    } else if (AX == 1) {  // l = i + d0 k
      const int k = l / d0, i = l - k * d0;
      off_in = k * d1 * d0 + i;
      off_out = k * NOUT * d0 + i;
      s_in = s_out = d0;
    } else {  // l = i + d0 j
      off_in = off_out = l;
      s_in = s_out = d0 * d1;
    }
    const double *src = in + b * size_in + off_in;
    double v[NIN];
#pragma unroll
    for (int a = 0; a < NIN; a++) v[a] = ldsr(&src[a * s_in]);
    double *dst = out + b * size_out + off_out;
#pragma unroll
    for (int r = 0; r < NOUT; r++) {
      double acc = 0.0;
#pragma unroll
      for (int a = 0; a < NIN; a++) acc += (TR ? M[a * NOUT + r] : M[r * NIN + a]) * v[a];
      dst[r * s_out] = acc;
    }
  }
}

// LDS scratch of the volume operators for one chunk of NC_NFC fields of a block:
//   A: 3 arrays [nb][N1]^(dim-1)[QV]   B (3-D): 3 arrays [nb][N1][QV][QV]   Cq: dim arrays [nb][QV^dim]   OUT: dim arrays [nb][NPE]
template <class C>
struct NcScratch {
  static constexpr int NB = NC_NFC * C::EPB;
  static constexpr int SA = (C::DIM == 3) ? C::N1 * C::N1 * C::QV : C::N1 * C::QV;
  static constexpr int SB = (C::DIM == 3) ? C::N1 * C::QV * C::QV : 0;
  static constexpr int O_A = 0, O_B = O_A + 3 * NB * SA, O_C = O_B + 3 * NB * SB, O_OUT = O_C + C::DIM * NB * C::NQV;
  static constexpr int TOTAL = O_OUT + C::DIM * NB * C::NPE;
};

// adjugate (det J  dxi_m/dx_d) and rule weight at volume quadrature point q of local element le
template <class C>
__device__ inline double nc_point_geometry(const double *sV, const Tab<C> &tab, int le, int q, double *A) {
  constexpr int DIM = C::DIM;
  double xi[DIM], J[DIM * DIM], w = 1.0;
  int r = q;
#pragma unroll
  for (int d = 0; d < DIM; d++) {
    const int qd = r % C::QV;
    r /= C::QV;
    xi[d] = tab.xv[qd];
    w *= tab.wv[qd];
  }
  jacobian<DIM>(&sV[le * C::NV * DIM], xi, J);
  adjugate<DIM>(J, A);
  return w;
}

// Volume part of the gradient: g[eq + d*NEQ] += sum_q phi_j(q) w_q (adj(J)^T grad_xi Up_eq)_d (q)   (Ke Up, not yet
// divided by the mass matrix).  sF: nodal fields [NEQ][NODES].
template <class C, int NEQ>
__device__ inline void nc_volume_gradient(const double *sF, double *scr, const double *sV, const Tab<C> &tab,
                                          const Tables1D &ct, bool node_on, int le_n, int nd, int tid, double *g) {
  constexpr int DIM = C::DIM, N1 = C::N1, QV = C::QV, B = C::BLOCK;
  typedef NcScratch<C> S;
  double *A1 = scr + S::O_A, *A2 = A1 + S::NB * S::SA, *A3 = A2 + S::NB * S::SA;
  double *B1 = scr + S::O_B, *B2 = B1 + S::NB * S::SB, *B3 = B2 + S::NB * S::SB;
  double *Cq = scr + S::O_C, *OUT = scr + S::O_OUT;
#pragma clang loop unroll(disable)
  for (int f0 = 0; f0 < NEQ; f0 += NC_NFC) {
    const int nf = (NEQ - f0 < NC_NFC) ? NEQ - f0 : NC_NFC, nb = nf * C::EPB;
    const double *in = sF + f0 * C::NODES;
    // nodes -> quadrature points: the three reference derivatives
    tensor_axis<B, 0, N1, QV, false>(in, A1, ct.Bv, nb, N1, N1, (DIM == 3) ? N1 : 1, tid);  // B(x)
    tensor_axis<B, 0, N1, QV, false>(in, A2, ct.Dv, nb, N1, N1, (DIM == 3) ? N1 : 1, tid);  // D(x)
    block_sync<B>();
    if constexpr (DIM == 3) {
      tensor_axis<B, 1, N1, QV, false>(A2, B1, ct.Bv, nb, QV, N1, N1, tid);  // D B .
      tensor_axis<B, 1, N1, QV, false>(A1, B2, ct.Dv, nb, QV, N1, N1, tid);  // B D .
      tensor_axis<B, 1, N1, QV, false>(A1, B3, ct.Bv, nb, QV, N1, N1, tid);  // B B .
      block_sync<B>();
      tensor_axis<B, 2, N1, QV, false>(B1, Cq, ct.Bv, nb, QV, QV, N1, tid);                       // d/dxi0
      tensor_axis<B, 2, N1, QV, false>(B2, Cq + S::NB * C::NQV, ct.Bv, nb, QV, QV, N1, tid);      // d/dxi1
      tensor_axis<B, 2, N1, QV, false>(B3, Cq + 2 * S::NB * C::NQV, ct.Dv, nb, QV, QV, N1, tid);  // d/dxi2
    } else {
      tensor_axis<B, 1, N1, QV, false>(A2, Cq, ct.Bv, nb, QV, N1, 1, tid);                    // d/dxi0
      tensor_axis<B, 1, N1, QV, false>(A1, Cq + S::NB * C::NQV, ct.Dv, nb, QV, N1, 1, tid);   // d/dxi1
    }
    block_sync<B>();
    // physical gradient times det J and the weight, at the quadrature points (in place)
    for (int item = tid; item < nb * C::NQV; item += B) {
      const int b = item / C::NQV, q = item - b * C::NQV;
      double A[DIM * DIM];
      const double w = nc_point_geometry<C>(sV, tab, b % C::EPB, q, A);
      double dr[DIM], G[DIM];
#pragma unroll
      for (int mm = 0; mm < DIM; mm++) dr[mm] = ldsr(&Cq[mm * S::NB * C::NQV + item]);
#pragma unroll
      for (int d = 0; d < DIM; d++) {
        double sgm = 0.0;
#pragma unroll
        for (int mm = 0; mm < DIM; mm++) sgm += A[mm + d * DIM] * dr[mm];
        G[d] = w * sgm;
      }
#pragma unroll
      for (int d = 0; d < DIM; d++) Cq[d * S::NB * C::NQV + item] = G[d];
    }
    block_sync<B>();
    // test with the basis: quadrature points -> nodes, per direction d (batched as [d][b])
    // NOTE the [d][NB] layout of Cq keeps the nb <= NB used arrays of each d apart: one call per d
#pragma unroll
    for (int d = 0; d < DIM; d++) {
      if constexpr (DIM == 3)
        tensor_axis<B, 2, QV, N1, true>(Cq + d * S::NB * C::NQV, B1 + d * S::NB * S::SB, ct.Bv, nb, QV, QV, QV, tid);
      else
        tensor_axis<B, 1, QV, N1, true>(Cq + d * S::NB * C::NQV, A1 + d * S::NB * S::SA, ct.Bv, nb, QV, QV, 1, tid);
    }
    block_sync<B>();
    if constexpr (DIM == 3) {
#pragma unroll
      for (int d = 0; d < DIM; d++)
        tensor_axis<B, 1, QV, N1, true>(B1 + d * S::NB * S::SB, A1 + d * S::NB * S::SA, ct.Bv, nb, QV, QV, N1, tid);
      block_sync<B>();
    }
#pragma unroll
    for (int d = 0; d < DIM; d++)
      tensor_axis<B, 0, QV, N1, true>(A1 + d * S::NB * S::SA, OUT + d * S::NB * C::NPE, ct.Bv, nb, QV, N1,
                                      (DIM == 3) ? N1 : 1, tid);
    block_sync<B>();
    if (node_on) {
      for (int f = 0; f < nf; f++)
#pragma unroll
        for (int d = 0; d < DIM; d++) {
          const double val = ldsr(&OUT[d * S::NB * C::NPE + (f * C::EPB + le_n) * C::NPE + nd]);
          // (f0 + f) is a run-time index: select, do not index the register array dynamically
#pragma unroll
          for (int eq = 0; eq < NEQ; eq++)
            if (eq == f0 + f) g[eq + d * NEQ] += val;
        }
    }
    block_sync<B>();
  }
}

// Volume term of the residual: z[eq] += sum_q w_q sum_m dphi_j/dxi_m(q) sum_d adj(m,d)(q) F_{eq,d}(q), with the nodal
// flux interpolated to the quadrature points -- Aflux F of the reference (src/domain_integrator.cpp:45-99,
// src/rhs_operator.cpp:379-391).  sF: nodal physical flux [(eq + d*NEQ)][NODES].
template <class C, int NEQ>
__device__ inline void nc_volume_divergence(const double *sF, double *scr, const double *sV, const Tab<C> &tab,
                                            const Tables1D &ct, bool node_on, int le_n, int nd, int tid, double *z) {
  constexpr int DIM = C::DIM, N1 = C::N1, QV = C::QV, B = C::BLOCK;
  typedef NcScratch<C> S;
  double *A1 = scr + S::O_A, *B1 = scr + S::O_B, *Cq = scr + S::O_C, *OUT = scr + S::O_OUT;
#pragma clang loop unroll(disable)
  for (int f0 = 0; f0 < NEQ; f0 += NC_NFC) {
    const int nf = (NEQ - f0 < NC_NFC) ? NEQ - f0 : NC_NFC, nb = nf * C::EPB;
    // nodes -> quadrature points of the DIM flux components (B in every direction)
#pragma unroll
    for (int d = 0; d < DIM; d++)
      tensor_axis<B, 0, N1, QV, false>(sF + (f0 + d * NEQ) * C::NODES, A1 + d * S::NB * S::SA, ct.Bv, nb, N1, N1,
                                       (DIM == 3) ? N1 : 1, tid);
    block_sync<B>();
    if constexpr (DIM == 3) {
#pragma unroll
      for (int d = 0; d < DIM; d++)
        tensor_axis<B, 1, N1, QV, false>(A1 + d * S::NB * S::SA, B1 + d * S::NB * S::SB, ct.Bv, nb, QV, N1, N1, tid);
      block_sync<B>();
#pragma unroll
      for (int d = 0; d < DIM; d++)
        tensor_axis<B, 2, N1, QV, false>(B1 + d * S::NB * S::SB, Cq + d * S::NB * C::NQV, ct.Bv, nb, QV, QV, N1, tid);
    } else {
#pragma unroll
      for (int d = 0; d < DIM; d++)
        tensor_axis<B, 1, N1, QV, false>(A1 + d * S::NB * S::SA, Cq + d * S::NB * C::NQV, ct.Bv, nb, QV, N1, 1, tid);
    }
    block_sync<B>();
    // contravariant components times the weight (in place)
    for (int item = tid; item < nb * C::NQV; item += B) {
      const int b = item / C::NQV, q = item - b * C::NQV;
      double A[DIM * DIM];
      const double w = nc_point_geometry<C>(sV, tab, b % C::EPB, q, A);
      double F[DIM], G[DIM];
#pragma unroll
      for (int d = 0; d < DIM; d++) F[d] = ldsr(&Cq[d * S::NB * C::NQV + item]);
#pragma unroll
      for (int mm = 0; mm < DIM; mm++) {
        double sgm = 0.0;
#pragma unroll
        for (int d = 0; d < DIM; d++) sgm += A[mm + d * DIM] * F[d];
        G[mm] = w * sgm;
      }
#pragma unroll
      for (int mm = 0; mm < DIM; mm++) Cq[mm * S::NB * C::NQV + item] = G[mm];
    }
    block_sync<B>();
    // quadrature points -> nodes: component m with the transposed derivative in direction m
    if constexpr (DIM == 3) {
      tensor_axis<B, 2, QV, N1, true>(Cq, B1, ct.Bv, nb, QV, QV, QV, tid);
      tensor_axis<B, 2, QV, N1, true>(Cq + S::NB * C::NQV, B1 + S::NB * S::SB, ct.Bv, nb, QV, QV, QV, tid);
      tensor_axis<B, 2, QV, N1, true>(Cq + 2 * S::NB * C::NQV, B1 + 2 * S::NB * S::SB, ct.Dv, nb, QV, QV, QV, tid);
      block_sync<B>();
      tensor_axis<B, 1, QV, N1, true>(B1, A1, ct.Bv, nb, QV, QV, N1, tid);
      tensor_axis<B, 1, QV, N1, true>(B1 + S::NB * S::SB, A1 + S::NB * S::SA, ct.Dv, nb, QV, QV, N1, tid);
      tensor_axis<B, 1, QV, N1, true>(B1 + 2 * S::NB * S::SB, A1 + 2 * S::NB * S::SA, ct.Bv, nb, QV, QV, N1, tid);
      block_sync<B>();
      tensor_axis<B, 0, QV, N1, true>(A1, OUT, ct.Dv, nb, QV, N1, N1, tid);
      tensor_axis<B, 0, QV, N1, true>(A1 + S::NB * S::SA, OUT + S::NB * C::NPE, ct.Bv, nb, QV, N1, N1, tid);
      tensor_axis<B, 0, QV, N1, true>(A1 + 2 * S::NB * S::SA, OUT + 2 * S::NB * C::NPE, ct.Bv, nb, QV, N1, N1, tid);
    } else {
      tensor_axis<B, 1, QV, N1, true>(Cq, A1, ct.Bv, nb, QV, QV, 1, tid);
      tensor_axis<B, 1, QV, N1, true>(Cq + S::NB * C::NQV, A1 + S::NB * S::SA, ct.Dv, nb, QV, QV, 1, tid);
      block_sync<B>();
      tensor_axis<B, 0, QV, N1, true>(A1, OUT, ct.Dv, nb, QV, N1, 1, tid);
      tensor_axis<B, 0, QV, N1, true>(A1 + S::NB * S::SA, OUT + S::NB * C::NPE, ct.Bv, nb, QV, N1, 1, tid);
    }
    block_sync<B>();
    if (node_on) {
      for (int f = 0; f < nf; f++) {
        double val = 0.0;
#pragma unroll
        for (int mm = 0; mm < DIM; mm++) val += ldsr(&OUT[mm * S::NB * C::NPE + (f * C::EPB + le_n) * C::NPE + nd]);
#pragma unroll
        for (int eq = 0; eq < NEQ; eq++)
          if (eq == f0 + f) z[eq] += val;
      }
    }
    block_sync<B>();
  }
}

// v <- Me^-1 v for NV nodal vectors held one entry per node lane (src/gradients.cpp:198-229, src/rhs_operator.cpp:
// 432-448).  The inverse is symmetric: lane j walks column j, i.e. every step of the loop reads one contiguous row.
template <class C, int NV>
__device__ inline void nc_apply_minv(const double *__restrict__ minv, double *sR, bool node_on, int e, int le_n, int nd,
                                     int tid, double *v) {
  if (node_on) {
#pragma unroll
    for (int k = 0; k < NV; k++) sR[k * C::NODES + tid] = v[k];
  }
  block_sync<C::BLOCK>();
  if (node_on) {
    const double *Mi = minv + static_cast<int64_t>(e) * (C::NPE * C::NPE) + nd;
    const double *r = sR + le_n * C::NPE;
    double acc[NV];
#pragma unroll
    for (int k = 0; k < NV; k++) acc[k] = 0.0;
    for (int a = 0; a < C::NPE; a++) {
      const double mm = Mi[a * C::NPE];
#pragma unroll
      for (int k = 0; k < NV; k++) acc[k] += mm * ldsr(&r[k * C::NODES + a]);
    }
#pragma unroll
    for (int k = 0; k < NV; k++) v[k] = acc[k];
  }
  block_sync<C::BLOCK>();
}

// The same product on the matrix cores, for the p = 3 hex (64 nodes, two waves per block): Y[64 x NV] = Minv[64 x 64] .
// R[64 x NV] as v_mfma_f64_16x16x4_f64 tiles -- the one dense element-local GEMM of the path (round 4; measured first in
// tools/microbench/mfma_minv.hip: 8 500 cycles per hex and wave against 22 700 for the loop above, whose 64 dependent
// row loads it is bound by).  Each wave owns two 16-row tiles of the result; per k-step (4 columns of Minv) a lane
// supplies A[i = lane % 16][k = lane / 16] straight from global memory -- the block is symmetric, so the tile is read by
// rows: 4 segments of 128 bytes per load -- and B[k][j = lane % 16] from a transposed, padded LDS copy of the vectors
// (stride 17: conflict-free); D comes back as D[tile * 16 + lane / 16 + 4 r][lane % 16] and returns to the node-per-lane
// layout through the same LDS buffer.  NV > 16: passes of 16 vectors.  FP64 matrix and vector peaks are equal on this
// part: what the matrix cores buy here is the operand delivery (no 960 broadcast LDS reads, 16 independent loads in
// flight), not arithmetic rate.  Same sums in the same order as the loop above (k ascending): results agree bit for bit
// in the micro-benchmark.   sB: >= 64 * 17 doubles.
typedef double mfma_v4d __attribute__((ext_vector_type(4)));
template <class C, int NV>
__device__ inline void nc_apply_minv_mfma(const double *__restrict__ minv, double *sB, bool node_on, int e, int nd, int tid,
                                          double *v) {
  static_assert(C::NPE == 64 && C::BLOCK == 128 && C::EPB == 1, "the p = 3 hex: one element, two waves");
  constexpr int LD = 17;
  const double *Mi = minv + static_cast<int64_t>(e) * (64 * 64);
  const int wave = tid >> 6, lane = tid & 63, col = lane & 15, kk = lane >> 4;
#pragma unroll
  for (int k0 = 0; k0 < NV; k0 += 16) {
    if (tid < 64) {  // the node lanes lay their vectors out as B[k = node][j]
#pragma unroll
      for (int j = 0; j < 16; j++) sB[tid * LD + j] = (k0 + j < NV && node_on) ? v[(k0 + j < NV) ? k0 + j : 0] : 0.0;
    }
    block_sync<C::BLOCK>();
    mfma_v4d d0 = {0.0, 0.0, 0.0, 0.0}, d1 = {0.0, 0.0, 0.0, 0.0};
    const double *row = Mi + kk * 64 + (2 * wave) * 16 + col;  // Minv[k][i], i in this wave's two tiles
#pragma unroll 4
    for (int ks = 0; ks < 16; ks++) {
      const double b = ldsr(&sB[(ks * 4 + kk) * LD + col]);
      const double a0 = row[ks * 256], a1 = row[ks * 256 + 16];
      d0 = __builtin_amdgcn_mfma_f64_16x16x4f64(a0, b, d0, 0, 0, 0);
      d1 = __builtin_amdgcn_mfma_f64_16x16x4f64(a1, b, d1, 0, 0, 0);
    }
    block_sync<C::BLOCK>();  // every read of B is done: the buffer takes D
#pragma unroll
    for (int r = 0; r < 4; r++) {
      sB[((2 * wave) * 16 + kk + 4 * r) * LD + col] = d0[r];
      sB[((2 * wave + 1) * 16 + kk + 4 * r) * LD + col] = d1[r];
    }
    block_sync<C::BLOCK>();
    if (tid < 64 && node_on) {
#pragma unroll
      for (int j = 0; j < 16; j++)
        if (k0 + j < NV) v[k0 + j] = ldsr(&sB[tid * LD + j]);
    }
    block_sync<C::BLOCK>();
  }
  (void)nd;
}

// Face term of the gradient through the face quadrature points (GradFaceIntegrator::AssembleFaceVector,
// src/faceGradientIntegration.cpp:40-140): g[eq + d*NEQ] += sum_q w_q phi_j(q) 1/2 (Up2 - Up1)(q) n_d(q) over the two
// faces of direction D.  sT: own | neighbour traces of Up at the face nodes, [2*NEQ][TN]; scr: >= 2*NEQ*TW + NEQ*TQ.
template <class C, class PH, int D>
__device__ inline void nc_grad_jump(const int2 *sFI, typename PH::PRef prm, double *sT, double *scr, const double *sV,
                                    const Tab<C> &tab, const Tables1D &ct, bool node_on, int le_n, const int *idx, int tid,
                                    double *g) {
  constexpr int NEQ = PH::NEQ, DIM = C::DIM;
  double *Wb = scr, *R = scr + 2 * NEQ * C::TW;
  interp1_lines<C, 2 * NEQ>(sT, Wb, ct, tid);
  block_sync<C::BLOCK>();
  double jump[C::Q_ROUNDS][NEQ], nw[C::Q_ROUNDS][DIM];
#pragma unroll
  for (int rd = 0; rd < C::Q_ROUNDS; rd++) {
    const int item = tid + rd * C::BLOCK;
#pragma unroll
    for (int eq = 0; eq < NEQ; eq++) jump[rd][eq] = 0.0;
#pragma unroll
    for (int d = 0; d < DIM; d++) nw[rd][d] = 0.0;
    if (item < C::TQ) {
      const int pf = item / C::NQ, q = item - pf * C::NQ;
      const int le = pf >> 1, s = pf & 1;
      const int nb = sFI[le * C::NFACES + 2 * D + s].x;
      if (nb != INT32_MIN) {
        double bq[C::N1];
        const int qrow = (DIM == 2) ? q : q / C::Q1;
#pragma unroll
        for (int a = 0; a < C::N1; a++) bq[a] = tab.B[qrow * C::N1 + a];
        double u1[NEQ], u2[NEQ];
#pragma unroll
        for (int eq = 0; eq < NEQ; eq++) {
          u1[eq] = interp2_point<C>(sT + eq * C::TN, Wb + eq * C::TW, bq, pf, q);
          u2[eq] = interp2_point<C>(sT + (NEQ + eq) * C::TN, Wb + (NEQ + eq) * C::TW, bq, pf, q);
        }
        if (nb < 0) {  // boundary: u2 = u1, or the wall ghost of useBCinGrad (:96-113)
#pragma unroll
          for (int eq = 0; eq < NEQ; eq++) u2[eq] = u1[eq];
          if (prm.use_bc_in_grad) PH::bc_grad_prim(prm, prm.bc[-nb - 1], u1, u2);
        }
        double n[DIM], wq, Xq[DIM];
        face_geometry<C, D>(&sV[le * C::NV * DIM], tab, s, q, n, wq, Xq);
#pragma unroll
        for (int eq = 0; eq < NEQ; eq++) jump[rd][eq] = 0.5 * (u2[eq] - u1[eq]);
#pragma unroll
        for (int d = 0; d < DIM; d++) nw[rd][d] = wq * n[d];
      }
    }
  }
  block_sync<C::BLOCK>();  // Wb is dead
#pragma unroll
  for (int d = 0; d < DIM; d++) {
#pragma unroll
    for (int rd = 0; rd < C::Q_ROUNDS; rd++) {
      const int item = tid + rd * C::BLOCK;
      if (item < C::TQ) {
#pragma unroll
        for (int eq = 0; eq < NEQ; eq++) R[eq * C::TQ + item] = nw[rd][d] * jump[rd][eq];
      }
    }
    block_sync<C::BLOCK>();
    double *W2 = (DIM == 3) ? Wb : R;  // 2-D: project2_lines reads R directly
    project1_lines<C, NEQ>(R, Wb, ct, tid);
    if (DIM == 3) block_sync<C::BLOCK>();
    project2_lines<C, NEQ>(W2, sT, ct, tid);  // L over the traces (consumed)
    block_sync<C::BLOCK>();
    if (node_on) {
#pragma unroll
      for (int eq = 0; eq < NEQ; eq++) g[eq + d * NEQ] += lift_pair<C, D>(sT + eq * C::TN, tab, le_n, idx);
    }
    block_sync<C::BLOCK>();
  }
}

template <class C, class PH>
struct GradLds {
  static constexpr int NEQ = PH::NEQ, DIM = C::DIM;
  static constexpr int NVF = NEQ * (1 + DIM);  // fields interpolated for the viscous traces: U, gradUp
  static constexpr int CH = NEQ;               // ... NEQ at a time: U, then one gradient direction per chunk
  // The pool is kept small on purpose: the resident workgroups per CU (LDS bytes) set how well the
  // latencies of the line stages overlap.  [ sU | sUp (viscous phase: T chunk) | J | W ]
  //   J: jump phase  -- own | neighbour Up traces of ONE direction pair
  //      viscous phase -- the nodal values of one gradient direction (re-written from registers per chunk)
  static constexpr int SUP = cmax(NEQ * C::NODES, (DIM == 2 ? 2 : 1) * CH * C::TN);  // 2-D: T chunks of both directions
  // (heavy point physics is register-bound, not LDS-bound: there the whole nodal gradient stays in J and the
  //  node lanes' registers are released before the physics)
  // (the lean 3-D face kernel of the ternary mixtures -- visc_phase_lean3d -- re-reads the nodal gradient it has just written
  //  from the L2, one Cartesian direction at a time: 12 KB per block, three waves per SIMD)
  static constexpr bool LEAN3D = PH::LEAN_TRACE && DIM == 3 && !C::NC && C::BLOCK == 64 && C::Q_ROUNDS == 1;
  static constexpr bool G_IN_LDS = PH::HEAVY && !LEAN3D;
  static constexpr int J = cmax(2 * NEQ * C::TN, (G_IN_LDS ? DIM : 1) * NEQ * C::NODES);
  static constexpr int W = CH * C::TW;
  // non-collocated variant: scratch of the volume operator / the quadrature-point gradient jump / the vectors of
  // the dense inverse mass (one after the other)
  static constexpr int V = C::NC ? cmax(cmax(NcScratch<C>::TOTAL, 2 * NEQ * C::TW + NEQ * C::TQ), NEQ * DIM * C::NODES) : 0;
  static constexpr int O_U = 0, O_UP = NEQ * C::NODES, O_J = O_UP + SUP, O_W = O_J + J, O_V = O_W + W;
  static constexpr int TOTAL = O_V + V;
};

// Gradient jump of one direction pair, collocated: g += M^-1 sum_faces <phi, (u^ - u) n>.
//
// The reference integrates this face term with the (p+2)-point face rule (src/faceGradientIntegration.cpp:
// 40-140): sum_q w_q phi_j(q) 1/2 (Up2 - Up1)(q) N(q).  On a bilinear face the integrand is a polynomial
// of degree <= p (phi_j) + p (the jump of two degree-p traces) + 1 (the area-weighted normal N of a
// bilinear patch) = 2p+1 per direction, which the (p+1)-point Gauss-Legendre rule AT THE FACE NODES
// integrates exactly as well.  There phi_j(x_k) = delta_jk, so the whole interpolate-to-quadrature /
// project-back pipeline collapses to a product at the face nodes: L_j = w_j 1/2 (Up2_j - Up1_j) N(x_j).
// Identical in exact arithmetic for every element this library accepts (straight-sided, order-1
// geometry); rounding differs at the 1e-15 level.  The wall ghost of useBCinGrad is affine per component
// (src/wallBC.cpp:241-266), so it commutes with the interpolation too.
template <class C>
__device__ inline void face_normal_rt(int d, const double *V, int s, double ta, double tb, double *n) {
  if (d == 0)
    face_normal_at<C, 0>(V, s, ta, tb, n);
  else if (d == 1 || C::DIM == 2)
    face_normal_at<C, 1>(V, s, ta, tb, n);
  else
    face_normal_at<C, (C::DIM == 3 ? 2 : 0)>(V, s, ta, tb, n);
}
// `D` is a run-time value and the face loop is not unrolled: one face's operands are live at a time
// (with the six faces unrolled the kernel needs > 256 VGPRs)
template <class C, class PH>
__device__ inline void grad_jump_nodal(int D, const int2 *sFI, typename PH::PRef prm, const double *Town,
                                       const double *Tnb, const double *sV, const Tab<C> &tab, int le_n, const int *idx,
                                       double inv_mass, double *g) {
  constexpr int NEQ = PH::NEQ, DIM = C::DIM;
  const int da = (DIM == 2) ? 1 - D : (D == 0 ? 1 : 0), db = (D == 2) ? 1 : 2;  // tan_a, tan_b
  const int ia = idx[da], ib = (DIM == 3) ? idx[db] : 0;
  const int fn = (DIM == 2) ? ia : ia + C::N1 * ib;
  const double wf = (DIM == 2) ? tab.w[ia] : tab.w[ia] * tab.w[ib];
#pragma clang loop unroll(disable)
  for (int s = 0; s < 2; s++) {
    const int pf = 2 * le_n + s;
    const int nb = sFI[le_n * C::NFACES + 2 * D + s].x;
    double u1[NEQ], u2[NEQ];
#pragma unroll
    for (int eq = 0; eq < NEQ; eq++) {
      u1[eq] = ldsr(&Town[eq * C::TN + pf * C::NF + fn]);
      u2[eq] = ldsr(&Tnb[eq * C::TN + pf * C::NF + fn]);
    }
    if (nb < 0 && prm.use_bc_in_grad) PH::bc_grad_prim(prm, prm.bc[-nb - 1], u1, u2);
    double n[DIM];
    face_normal_rt<C>(D, &sV[le_n * C::NV * DIM], s, tab.x[ia], (DIM == 3) ? tab.x[ib] : 0.0, n);
    const double c = 0.5 * wf * inv_mass * (s ? tab.b1[idx[D]] : tab.b0[idx[D]]);
#pragma unroll
    for (int dd = 0; dd < DIM; dd++) {
      const double cn = c * n[dd];
#pragma unroll
      for (int eq = 0; eq < NEQ; eq++) g[eq + dd * NEQ] += cn * (u2[eq] - u1[eq]);
    }
  }
}

// viscous normal-flux traces of one direction pair: [U | gradUp] at the face quadrature points ...
// (chunks of NEQ fields: U from sU, then one gradient direction at a time, written from the registers of
// the node lanes into the small nodal buffer sJ)
template <class C, class PH, int D>
__device__ inline void visc_interp_dir(const double *sU, const double *g, bool node_on, double *sJ, double *Tb, double *Wb,
                                       const Tab<C> &tab, const Tables1D &ct,
                                       double (&v)[C::Q_ROUNDS][GradLds<C, PH>::NVF], int tid) {
  constexpr int NEQ = PH::NEQ, DIM = C::DIM;
  // gradient rows the viscous flux reads: a perfect gas never looks at grad(rho), and 4 fields x 16
  // lines fill one 64-lane round of the line stages where 5 need two
  constexpr int G0 = PH::VISC_USES_GRAD_RHO ? 0 : 1, NG = NEQ - G0;
  auto point_stage = [&](auto nfld_tag, int v0, const double *T, const double *W) {
    constexpr int NF_ = decltype(nfld_tag)::value;
#pragma unroll
    for (int rd = 0; rd < C::Q_ROUNDS; rd++) {
      const int item = tid + rd * C::BLOCK;
      if (item < C::TQ) {
        const int pf = item / C::NQ, q = item - pf * C::NQ;
        double bq[C::N1];
        const int qrow = (DIM == 2) ? q : q / C::Q1;
#pragma unroll
        for (int a = 0; a < C::N1; a++) bq[a] = tab.B[qrow * C::N1 + a];
#pragma unroll
        for (int k = 0; k < NF_; k++) v[rd][v0 + k] = interp2_point<C>(T + k * C::TN, W + k * C::TW, bq, pf, q);
      }
    }
  };
  // chunk 0: the conserved state
  trace_lines<C, D, NEQ>(sU, Tb, ct, tid);
  block_sync<C::BLOCK>();
  interp1_lines<C, NEQ>(Tb, Wb, ct, tid);
  if (DIM == 3) block_sync<C::BLOCK>();
  point_stage(std::integral_constant<int, NEQ>(), 0, Tb, Wb);
  block_sync<C::BLOCK>();
  // chunks 1..DIM: one gradient direction each
#pragma unroll
  for (int c = 1; c <= DIM; c++) {
    const double *src;
    if constexpr (GradLds<C, PH>::G_IN_LDS) {
      src = sJ + ((c - 1) * NEQ + G0) * C::NODES;
    } else {
      if (node_on) {
#pragma unroll
        for (int eq = G0; eq < NEQ; eq++) sJ[eq * C::NODES + tid] = g[eq + (c - 1) * NEQ];
      }
      block_sync<C::BLOCK>();
      src = sJ + G0 * C::NODES;
    }
    trace_lines<C, D, NG>(src, Tb, ct, tid);
    block_sync<C::BLOCK>();
    interp1_lines<C, NG>(Tb, Wb, ct, tid);
    if (DIM == 3) block_sync<C::BLOCK>();
    if (G0) {
#pragma unroll
      for (int rd = 0; rd < C::Q_ROUNDS; rd++) v[rd][c * NEQ] = 0.0;
    }
    point_stage(std::integral_constant<int, NG>(), c * NEQ + G0, Tb, Wb);
    block_sync<C::BLOCK>();
  }
}
template <class C>
__device__ inline void face_geometry_rt(int d, const double *verts, const Tab<C> &tab, int s, int q, double *n, double &wq,
                                        double *Xq) {
  if (d == 0)
    face_geometry<C, 0>(verts, tab, s, q, n, wq, Xq);
  else if (d == 1 || C::DIM == 2)
    face_geometry<C, 1>(verts, tab, s, q, n, wq, Xq);
  else
    face_geometry<C, (C::DIM == 3 ? 2 : 0)>(verts, tab, s, q, n, wq, Xq);
}
// ... and the point physics on them.  `d` is a run-time value here so that physics with a large body
// (PH::HEAVY: plasma transport) is instantiated once per kernel, not once per direction pair.
// the wall distance at a lane's face quadrature points (mixing-length model), by value
template <int ROUNDS>
struct FaceDist {
  bool on = false;
  double d[ROUNDS] = {};
};
template <class C, class PH, bool BOTH = false>
__device__ inline void visc_points(const MeshDev &m, const int2 *sFI, typename PH::PRef prm0, int e0, int d,
                                   double (&v)[BOTH ? C::Q2_ROUNDS : C::Q_ROUNDS][GradLds<C, PH>::NVF], const double *sV,
                                   const Tab<C> &tab, double *__restrict__ TB, int tid,
                                   const FaceDist<BOTH ? C::Q2_ROUNDS : C::Q_ROUNDS> &dq = FaceDist<BOTH ? C::Q2_ROUNDS : C::Q_ROUNDS>()) {
  constexpr int NEQ = PH::NEQ, DIM = C::DIM;
  constexpr int ROUNDS = BOTH ? C::Q2_ROUNDS : C::Q_ROUNDS, ITEMS = BOTH ? C::TQ2 : C::TQ;
#pragma unroll
  for (int rd = 0; rd < ROUNDS; rd++) {
    int item = tid + rd * C::BLOCK;
    if (item >= ITEMS) continue;
    if (BOTH) {  // items of direction pair 0, then of direction pair 1
      d = item / C::TQ;
      item -= d * C::TQ;
    }
    const int pf = item / C::NQ, q = item - pf * C::NQ;
    const int le = pf >> 1, s = pf & 1;
    const int e = e0 + le;
    if (e >= m.ne) continue;
    const int slot = e * C::NFACES + 2 * d + s;
    const int nb = sFI[le * C::NFACES + 2 * d + s].x;
    typename PH::PRef prm = PH::relaunder(prm0);
    double fn[NEQ];
    PH::clamp_species(v[rd]);
    double n[DIM], wq, Xq[DIM];
    face_geometry_rt<C>(d, &sV[le * C::NV * DIM], tab, s, q, n, wq, Xq);
    if constexpr (PH::HEAVY && DIM == 2) {  // dq: the wall distance at the points (mixing-length model) or NULL
      __shared__ double sVsw[C::BLOCK];
      PH::visc_trace(prm, nb, v[rd], v[rd] + NEQ, n, PH::AXISYM ? Xq[0] : -1.0, fn, closure_ctx(m, dq.on, dq.d[rd], Xq, sVsw));
    } else if constexpr (PH::HEAVY) {
      PH::visc_trace(prm, nb, v[rd], v[rd] + NEQ, n, PH::AXISYM ? Xq[0] : -1.0, fn);
    } else if constexpr (PH::LES) {  // delta1 of src/face_integrator.cpp:253, transip of :333
      PointCtx pc;
      pc.delta = prm.elem_delta[e];
#pragma unroll
      for (int k = 0; k < DIM; k++) pc.X[k] = Xq[k];
      if (nb >= 0) {
        PH::visc_flux_n(prm, v[rd], v[rd] + NEQ, n, fn, &pc);
      } else {
        PH::bc_visc_term(prm, prm.bc[-nb - 1], v[rd], v[rd] + NEQ, n, fn, &pc);
      }
    } else {
      if (nb >= 0) {
        PH::visc_flux_n(prm, v[rd], v[rd] + NEQ, n, fn);
      } else {
        PH::bc_visc_term(prm, prm.bc[-nb - 1], v[rd], v[rd] + NEQ, n, fn);
      }
    }
    double *out = TB + static_cast<int64_t>(slot) * ((NEQ - 1) * C::NQ) + q;
#pragma unroll
    for (int eq = 1; eq < NEQ; eq++) out[(eq - 1) * C::NQ] = fn[eq];  // fn[0] == 0 (src/fluxes.cpp:284)
  }
}
// ---- heavy point physics, 3-D, one round of quadrature points per direction pair: the viscous traces in
// two steps.  (1) The conserved state at the face quadrature points -> the state-only closure (collision
// integrals: the transcendental-heavy part) while only NEQ interpolated values are live.  (2) The gradient,
// one Cartesian direction per chunk: the velocity rows are kept (9 values), of the scalar rows only the
// normal derivative sum_d n_d d(.)/dx_d is accumulated -- heat and diffusion fluxes are linear in it.  12-15
// interpolated values per point instead of 24, and none of them live across the closure.
template <class C, class PH, int D>
__device__ inline void visc_state_dir(const double *sU, double *Tb, double *Wb, const Tab<C> &tab, const Tables1D &ct0,
                                      double *u, int tid) {
  constexpr int NEQ = PH::NEQ;
  const Tables1D &ct = fresh_table(ct0);
  trace_lines<C, D, NEQ>(sU, Tb, ct, tid);
  block_sync<C::BLOCK>();
  interp1_lines<C, NEQ>(Tb, Wb, ct, tid);
  block_sync<C::BLOCK>();
  if (tid < C::TQ) {
    const int pf = tid / C::NQ, q = tid - pf * C::NQ;
    double bq[C::N1];
#pragma unroll
    for (int a = 0; a < C::N1; a++) bq[a] = tab.B[(q / C::Q1) * C::N1 + a];
#pragma unroll
    for (int k = 0; k < NEQ; k++) u[k] = interp2_point<C>(Tb + k * C::TN, Wb + k * C::TW, bq, pf, q);
  }
  block_sync<C::BLOCK>();
}
// sG: the nodal gradient [d][eq][NODES]; gv[i + j*DIM] = d u_i / d x_j; gn[eq] = normal derivative (scalar rows)
template <class C, class PH, int D>
__device__ inline void visc_grad_dir(const double *sG, double *Tb, double *Wb, const Tab<C> &tab, const Tables1D &ct0,
                                     const double *n, double *gv, double *gn, int tid) {
  constexpr int NEQ = PH::NEQ, DIM = C::DIM, NVEL = PH::NVEL;
#pragma unroll
  for (int eq = 0; eq < NEQ; eq++) gn[eq] = 0.0;
#pragma unroll
  for (int c = 0; c < DIM; c++) {
    const Tables1D &ct = fresh_table(ct0);
    trace_lines<C, D, NEQ>(sG + c * NEQ * C::NODES, Tb, ct, tid);
    block_sync<C::BLOCK>();
    interp1_lines<C, NEQ>(Tb, Wb, ct, tid);
    block_sync<C::BLOCK>();
    if (tid < C::TQ) {
      const int pf = tid / C::NQ, q = tid - pf * C::NQ;
      double bq[C::N1];
#pragma unroll
      for (int a = 0; a < C::N1; a++) bq[a] = tab.B[(q / C::Q1) * C::N1 + a];
#pragma unroll
      for (int k = 0; k < NEQ; k++) {
        const double val = interp2_point<C>(Tb + k * C::TN, Wb + k * C::TW, bq, pf, q);
        if (k >= 1 && k <= NVEL)
          gv[(k - 1) + c * DIM] = val;
        else
          gn[k] += n[c] * val;
      }
    }
    block_sync<C::BLOCK>();
  }
}
template <class C, class PH>
__device__ inline void visc_phase_heavy3d(const MeshDev &m, const int2 *sFI, typename PH::PRef prm0, int e0,
                                          const double *sU, const double *sG, double *Tb, double *Wb, const double *sV,
                                          const Tab<C> &tab, const Tables1D &ct, double *__restrict__ TB, int tid0 STAMP_PARAM) {
  constexpr int NEQ = PH::NEQ, DIM = C::DIM;
  static_assert(DIM == 3 && C::Q_ROUNDS == 1 && PH::NVEL == 3, "3-D, one round of face points per direction pair");
#pragma clang loop unroll(disable)
  for (int d = 0; d < DIM; d++) {
    typename PH::PRef prm = PH::relaunder(prm0);  // the parameter loads of this direction stay inside it
    const int tid = tid0;
    const int pf = tid / C::NQ, q = tid - pf * C::NQ;
    const int le = pf >> 1, s = pf & 1;
    const bool on = tid < C::TQ && (e0 + le) < m.ne;
    double u[NEQ];
#pragma unroll
    for (int k = 0; k < NEQ; k++) u[k] = 1.0;
    if (d == 0)
      visc_state_dir<C, PH, 0>(sU, Tb, Wb, tab, ct, u, tid);
    else if (d == 1)
      visc_state_dir<C, PH, 1>(sU, Tb, Wb, tab, ct, u, tid);
    else
      visc_state_dir<C, PH, 2>(sU, Tb, Wb, tab, ct, u, tid);
    STAMP(5);
    PH::clamp_species(u);
    int nb = 0;
    double n[DIM] = {1.0, 0.0, 0.0}, wq, Xq[DIM];
    if (on) {
      nb = sFI[le * C::NFACES + 2 * d + s].x;
      face_geometry_rt<C>(d, &sV[le * C::NV * DIM], tab, s, q, n, wq, Xq);
    }
    // 0: no viscous term on this face, 1: interior face, 2: wall face (interior state, then wall-side state)
    const int np_lane = on ? PH::visc_passes(prm, nb) : 0;
    // the pass loop holds block-wide stages: its trip count is uniform over the block (one wave: a ballot;
    // p = 4, 5: two / four waves share the element)
    int npass;
    if constexpr (C::BLOCK == 64)
      npass = (__ballot(np_lane == 2) != 0) ? 2 : ((__ballot(np_lane >= 1) != 0) ? 1 : 0);
    else
      npass = __syncthreads_or(np_lane == 2) ? 2 : (__syncthreads_or(np_lane >= 1) ? 1 : 0);
    double fn[NEQ];
#pragma unroll
    for (int eq = 0; eq < NEQ; eq++) fn[eq] = 0.0;
    // the second pass exists in the few waves that touch a wall; it repeats the gradient interpolation
    // rather than keeping 12 more values alive across the closure of every wave
#pragma clang loop unroll(disable)
    for (int pass = 0; pass < npass; pass++) {
      double Us[NEQ];
      typename PH::WallFlux w;
      typename PH::ViscCoef cf;
      if (pass < np_lane) {
        typename PH::PRef pq = PH::relaunder(prm);
        PH::visc_pass_state(pq, nb, pass, u, n, Us, w);
        PH::visc_point_coeffs(pq, Us, !w.species, cf);
      }
      STAMP(6);
      double gv[DIM * DIM], gn[NEQ];
      if (d == 0)
        visc_grad_dir<C, PH, 0>(sG, Tb, Wb, tab, ct, n, gv, gn, tid);
      else if (d == 1)
        visc_grad_dir<C, PH, 1>(sG, Tb, Wb, tab, ct, n, gv, gn, tid);
      else
        visc_grad_dir<C, PH, 2>(sG, Tb, Wb, tab, ct, n, gv, gn, tid);
      STAMP(7);
      if (pass < np_lane) {
        double f[NEQ];
        PH::visc_normal_flux_n(PH::relaunder(prm), Us, cf, gv, gn, n, w, f);
        if (nb >= 0) {
#pragma unroll
          for (int eq = 0; eq < NEQ; eq++) fn[eq] = f[eq];
        } else {
#pragma unroll
          for (int eq = 1; eq < NEQ; eq++) fn[eq] -= 0.5 * f[eq];
        }
      }
    }
    if (on) {
      const int slot = (e0 + le) * C::NFACES + 2 * d + s;
      double *out = TB + static_cast<int64_t>(slot) * ((NEQ - 1) * C::NQ) + q;
#pragma unroll
      for (int eq = 1; eq < NEQ; eq++) out[(eq - 1) * C::NQ] = fn[eq];  // fn[0] == 0 (src/fluxes.cpp:284)
    }
    STAMP(8);
  }
}
// ---- the same phase in its LEAN form (round 4; ternary mixtures, collocated p <= 3 hexes, one wave per block): sized for
// THREE waves per SIMD.  Differences from visc_phase_heavy3d:
//   * the closure of a face point leaves PH::ViscLean (12-18 values) instead of ViscCoef + state + wall prescriptions (38);
//   * nothing is carried across the closure for a second pass: a wall face's second pass (the few waves that touch a wall)
//     interpolates the state again and adds its half to the record it wrote in the first;
//   * the nodal gradient is not held in LDS (18 fields x 64 nodes = 9 KB): each Cartesian direction of it is re-read from
//     the gradUp vector this block has just written -- an L2 hit -- into the 3 KB nodal buffer, the next one in flight
//     while the current one is interpolated.
template <class C, class PH, int D, int CD>
__device__ inline void lean_grad_chunk(const double *sJ, double *Tb, double *Wb, const Tab<C> &tab, const Tables1D &ct0,
                                       const double *n, double *gv, double *gn, int tid) {
  constexpr int NEQ = PH::NEQ, DIM = C::DIM, NVEL = PH::NVEL;
  const Tables1D &ct = fresh_table(ct0);
  trace_lines<C, D, NEQ>(sJ, Tb, ct, tid);
  block_sync<C::BLOCK>();
  interp1_lines<C, NEQ>(Tb, Wb, ct, tid);
  block_sync<C::BLOCK>();
  if (tid < C::TQ) {
    const int pf = tid / C::NQ, q = tid - pf * C::NQ;
    double bq[C::N1];
#pragma unroll
    for (int a = 0; a < C::N1; a++) bq[a] = tab.B[(q / C::Q1) * C::N1 + a];
#pragma unroll
    for (int k = 0; k < NEQ; k++) {
      const double val = interp2_point<C>(Tb + k * C::TN, Wb + k * C::TW, bq, pf, q);
      if (k >= 1 && k <= NVEL)
        gv[(k - 1) + CD * DIM] = val;
      else
        gn[k] += n[CD] * val;
    }
  }
  block_sync<C::BLOCK>();
}
// the three Cartesian directions of the nodal gradient through the nodal buffer sJ, the next one in flight (registers of
// the node lanes) while the current one is interpolated to the face points of direction pair D
template <class C, class PH, int D>
__device__ inline void lean_grad_dir(const MeshDev &m, const double *gradUp_q, bool node_on, unsigned node, double *gnext, double *sJ,
                                     double *Tb, double *Wb, const Tab<C> &tab, const Tables1D &ct, const double *n, double *gv,
                                     double *gn, int tid) {
  constexpr int NEQ = PH::NEQ;
  auto stage = [&](auto ctag) {
    constexpr int CD = decltype(ctag)::value;
    if (node_on) {
#pragma unroll
      for (int eq = 0; eq < NEQ; eq++) sJ[eq * C::NODES + tid] = gnext[eq];
      if (CD + 1 < C::DIM) {
        // (node index and field stride through empty asm statements: the 18 field addresses of a node are formed where
        //  they are used -- scalar arithmetic -- instead of being hoisted out of the loops and held in 36 VGPRs)
        unsigned nq = node;
        int64_t stride = m.ndofs;
        asm volatile("" : "+v"(nq), "+s"(stride));
#pragma unroll
        for (int eq = 0; eq < NEQ; eq++) gnext[eq] = field_ptr(gradUp_q, (CD + 1) * NEQ + eq, stride)[nq];
      }
    }
    block_sync<C::BLOCK>();
    lean_grad_chunk<C, PH, D, CD>(sJ, Tb, Wb, tab, ct, n, gv, gn, tid);
  };
  stage(std::integral_constant<int, 0>());
  stage(std::integral_constant<int, 1>());
  stage(std::integral_constant<int, 2>());
}
template <class C, class PH>
__device__ inline void visc_phase_lean3d(const MeshDev &m, const int2 *sFI, typename PH::PRef prm0, int e0, const double *sU,
                                         const double *gradUp_q, bool node_on, unsigned node, double *sJ, double *Tb, double *Wb,
                                         const double *sV, const Tab<C> &tab, const Tables1D &ct, double *__restrict__ TB,
                                         int tid0 STAMP_PARAM) {
  constexpr int NEQ = PH::NEQ, DIM = C::DIM;
  static_assert(DIM == 3 && C::Q_ROUNDS == 1 && PH::NVEL == 3 && C::BLOCK == 64, "3-D, one wave, one round of face points per pair");
#pragma clang loop unroll(disable)
  for (int d = 0; d < DIM; d++) {
    typename PH::PRef prm = PH::relaunder(prm0);  // the parameter loads of this direction stay inside it
    // (the lane index through an empty asm: everything derived from it -- a dozen LDS addresses per line stage -- is
    //  recomputed in the direction that uses it instead of being hoisted out of this loop and held, or spilled, across it)
    int tid = tid0;
    asm volatile("" : "+v"(tid));
    const int pf = tid / C::NQ, q = tid - pf * C::NQ;
    const int le = pf >> 1, s = pf & 1;
    const bool on = tid < C::TQ && (e0 + le) < m.ne;
    int nb = 0;
    double n[DIM] = {1.0, 0.0, 0.0};
    if (on) {
      double wq, Xq[DIM];
      nb = sFI[le * C::NFACES + 2 * d + s].x;
      face_geometry_rt<C>(d, &sV[le * C::NV * DIM], tab, s, q, n, wq, Xq);
    }
    // 0: no viscous term on this face, 1: interior face, 2: wall face (interior state, then wall-side state)
    const int np_lane = on ? PH::visc_passes(prm, nb) : 0;
    const int npass = (__ballot(np_lane == 2) != 0) ? 2 : ((__ballot(np_lane >= 1) != 0) ? 1 : 0);
    double *out = TB + static_cast<int64_t>((e0 + le) * C::NFACES + 2 * d + s) * ((NEQ - 1) * C::NQ) + q;
    if (on && np_lane == 0) {
#pragma unroll
      for (int eq = 1; eq < NEQ; eq++) out[(eq - 1) * C::NQ] = 0.0;
    }
#pragma clang loop unroll(disable)
    for (int pass = 0; pass < npass; pass++) {
      double u[NEQ];
#pragma unroll
      for (int k = 0; k < NEQ; k++) u[k] = 1.0;
      if (d == 0)
        visc_state_dir<C, PH, 0>(sU, Tb, Wb, tab, ct, u, tid);
      else if (d == 1)
        visc_state_dir<C, PH, 1>(sU, Tb, Wb, tab, ct, u, tid);
      else
        visc_state_dir<C, PH, 2>(sU, Tb, Wb, tab, ct, u, tid);
      STAMP(5);
      PH::clamp_species(u);
      // the first Cartesian direction of the nodal gradient goes out before the closure (an L2 round trip behind ~1 500
      // FP64 instructions), the others while their predecessor is interpolated
      double gnext[NEQ];
#pragma unroll
      for (int eq = 0; eq < NEQ; eq++) gnext[eq] = 0.0;
      if (node_on) {
        unsigned nq = node;
        int64_t stride = m.ndofs;
        asm volatile("" : "+v"(nq), "+s"(stride));
#pragma unroll
        for (int eq = 0; eq < NEQ; eq++) gnext[eq] = field_ptr(gradUp_q, eq, stride)[nq];
      }
      typename PH::ViscLean cf;
      if (pass < np_lane) {
        typename PH::PRef pq = PH::relaunder(prm);
        double Us[NEQ];
        typename PH::WallFlux w;
        PH::visc_pass_state(pq, nb, pass, u, n, Us, w);
        PH::visc_point_lean(pq, Us, w, cf);
      }
      STAMP(6);
      double gv[DIM * DIM], gn[NEQ];
#pragma unroll
      for (int k = 0; k < DIM * DIM; k++) gv[k] = 0.0;
#pragma unroll
      for (int eq = 0; eq < NEQ; eq++) gn[eq] = 0.0;
      if (d == 0)
        lean_grad_dir<C, PH, 0>(m, gradUp_q, node_on, node, gnext, sJ, Tb, Wb, tab, ct, n, gv, gn, tid);
      else if (d == 1)
        lean_grad_dir<C, PH, 1>(m, gradUp_q, node_on, node, gnext, sJ, Tb, Wb, tab, ct, n, gv, gn, tid);
      else
        lean_grad_dir<C, PH, 2>(m, gradUp_q, node_on, node, gnext, sJ, Tb, Wb, tab, ct, n, gv, gn, tid);
      STAMP(7);
      if (pass < np_lane) {
        double f[NEQ];
        PH::visc_normal_flux_lean(cf, gv, gn, n, f);
        if (nb >= 0) {
#pragma unroll
          for (int eq = 1; eq < NEQ; eq++) {  // f[0] == 0 (src/fluxes.cpp:284)
            if (TPSRHS_NT_STORES)
              __builtin_nontemporal_store(f[eq], &out[(eq - 1) * C::NQ]);
            else
              out[(eq - 1) * C::NQ] = f[eq];
          }
        } else if (pass == 0) {  // wall face: -1/2 (Fv_in + Fv_wall) . n, the interior half first
#pragma unroll
          for (int eq = 1; eq < NEQ; eq++) out[(eq - 1) * C::NQ] = -0.5 * f[eq];
        } else {
#pragma unroll
          for (int eq = 1; eq < NEQ; eq++) out[(eq - 1) * C::NQ] -= 0.5 * f[eq];
        }
      }
      STAMP(8);
    }
    if (on && npass == 0) {
      // (np_lane == 0 for every lane: the zeros were written above)
    }
  }
}
// 2-D viscous phase: both direction pairs at once (see Cfg::TQ2)
template <class C, class PH>
__device__ inline void visc_phase_2d(const MeshDev &m, const int2 *sFI, typename PH::PRef prm, int e0,
                                     const double *sU, const double *g, bool node_on, double *sJ, double *Tb,
                                     const double *sV, const Tab<C> &tab, const Tables1D &ct, double *__restrict__ TB,
                                     int tid, const double *sDist = nullptr) {
  static_assert(C::DIM == 2, "2-D only");
  constexpr int NEQ = PH::NEQ;
  typedef GradLds<C, PH> L;
  constexpr int G0 = PH::VISC_USES_GRAD_RHO ? 0 : 1, NG = NEQ - G0;
  double v[C::Q2_ROUNDS][L::NVF];
  auto chunk = [&](auto nfld_tag, const double *src, int v0) {
    constexpr int NF_ = decltype(nfld_tag)::value;
    trace_lines<C, 0, NF_>(src, Tb, ct, tid);
    trace_lines<C, 1, NF_>(src, Tb + L::CH * C::TN, ct, tid);
    block_sync<C::BLOCK>();
#pragma unroll
    for (int rd = 0; rd < C::Q2_ROUNDS; rd++) {
      int item = tid + rd * C::BLOCK;
      if (item < C::TQ2) {
        const int d = item / C::TQ;
        item -= d * C::TQ;
        const int pf = item / C::NQ, q = item - pf * C::NQ;
        const double *T = Tb + d * (L::CH * C::TN);
        double bq[C::N1];
#pragma unroll
        for (int a = 0; a < C::N1; a++) bq[a] = tab.B[q * C::N1 + a];
#pragma unroll
        for (int k = 0; k < NF_; k++) v[rd][v0 + k] = interp2_point<C>(T + k * C::TN, nullptr, bq, pf, q);
      }
    }
    block_sync<C::BLOCK>();
  };
  chunk(std::integral_constant<int, NEQ>(), sU, 0);
  // mixing-length model: the nodal wall distance interpolated to the face points like a field
  // (src/face_integrator.cpp:303-308, src/BCintegrator.cpp:409-412); block-uniform branch
  FaceDist<C::Q2_ROUNDS> dq;
  dq.on = sDist != nullptr;
  if (sDist) {
    trace_lines<C, 0, 1>(sDist, Tb, ct, tid);
    trace_lines<C, 1, 1>(sDist, Tb + L::CH * C::TN, ct, tid);
    block_sync<C::BLOCK>();
#pragma unroll
    for (int rd = 0; rd < C::Q2_ROUNDS; rd++) {
      int item = tid + rd * C::BLOCK;
      dq.d[rd] = 0.0;
      if (item < C::TQ2) {
        const int d = item / C::TQ;
        item -= d * C::TQ;
        const int pf = item / C::NQ, q = item - pf * C::NQ;
        double bq[C::N1];
#pragma unroll
        for (int a = 0; a < C::N1; a++) bq[a] = tab.B[q * C::N1 + a];
        dq.d[rd] = interp2_point<C>(Tb + d * (L::CH * C::TN), nullptr, bq, pf, q);
      }
    }
    block_sync<C::BLOCK>();
  }
#pragma unroll
  for (int c = 1; c <= 2; c++) {
    const double *src;
    if constexpr (L::G_IN_LDS) {
      src = sJ + ((c - 1) * NEQ + G0) * C::NODES;
    } else {
      if (node_on) {
#pragma unroll
        for (int eq = G0; eq < NEQ; eq++) sJ[eq * C::NODES + tid] = g[eq + (c - 1) * NEQ];
      }
      block_sync<C::BLOCK>();
      src = sJ + G0 * C::NODES;
    }
    if (G0) {
#pragma unroll
      for (int rd = 0; rd < C::Q2_ROUNDS; rd++) v[rd][c * NEQ] = 0.0;
    }
    chunk(std::integral_constant<int, NG>(), src, c * NEQ + G0);
  }
  visc_points<C, PH, true>(m, sFI, prm, e0, 0, v, sV, tab, TB, tid, dq);
}
template <class C, class PH, int D>
__device__ inline void visc_traces_dir(const MeshDev &m, const int2 *sFI, typename PH::PRef prm, int e0,
                                       const double *sU, const double *g, bool node_on, double *sJ, double *Tb, double *Wb,
                                       const double *sV, const Tab<C> &tab, const Tables1D &ct, double *__restrict__ TB,
                                       int tid) {
  double v[C::Q_ROUNDS][GradLds<C, PH>::NVF];
  visc_interp_dir<C, PH, D>(sU, g, node_on, sJ, Tb, Wb, tab, ct, v, tid);
  visc_points<C, PH>(m, sFI, prm, e0, D, v, sV, tab, TB, tid);
}

template <class C, class PH>
__global__ __launch_bounds__(C::BLOCK, (C::NC && PH::HEAVY) ? 1 : PH::minw_grad(C::DIM, C::P, C::NC)) void k_gradient(MeshDev m, typename PH::KArg prm_k,
                                                       const double *__restrict__ U, const double *__restrict__ TA,
                                                       double *__restrict__ Upout, double *__restrict__ gradUp,
                                                       double *__restrict__ TB) {
  typename PH::PRef prm = PH::pref(prm_k);
  constexpr int NEQ = PH::NEQ, DIM = C::DIM;
  typedef GradLds<C, PH> L;
  const Tables1D &ct = c_tab[C::NC][DIM - 2][C::P];
  __shared__ Tab<C> tab;
  __shared__ double sV[C::EPB * C::NV * DIM];
  __shared__ double pool[L::TOTAL];
  double *sU = pool + L::O_U;    // [NEQ][NODES]
  double *sUp = pool + L::O_UP;  // [NEQ][NODES]; viscous phase: T chunk
  double *sJ = pool + L::O_J;    // jump phase: traces of one direction pair; viscous phase: nodal gradient chunk
  double *sW = pool + L::O_W;    // W chunk of the viscous phase

  const int tid = threadIdx.x;
  const int lin = xcd_block(static_cast<int>(blockIdx.x), static_cast<int>(gridDim.x), m.reverse);
  const int bid = m.blocks ? m.blocks[lin] : lin;
  // (TPSRHS_ABLATE & 512, timing experiment: every block works on one of 8 elements -- all loads become cache hits)
  const int e0 = (TPSRHS_ABLATE & 512) ? (bid % 8) * C::EPB : bid * C::EPB;
  __shared__ int2 sFI[C::EPB * C::NFACES];
  STAMP_DECL;
  STAMP_START();
  const bool node_on = tid < C::NODES && (e0 + tid / C::NPE) < m.ne;
  const int le_n = tid / C::NPE, nd = tid - le_n * C::NPE;
  int idx[3] = {nd % C::N1, (nd / C::N1) % C::N1, nd / (C::N1 * C::N1)};
  // the state goes out first: its latency overlaps that of the tables, vertices and face records
  double u[NEQ];
  if (node_on) {
    const unsigned n = static_cast<unsigned>(e0 + le_n) * C::NPE + nd;
#pragma unroll
    for (int eq = 0; eq < NEQ; eq++) u[eq] = field_ptr(U, eq, m.ndofs)[n];
  }
  load_face_info<C>(sFI, m, e0);
  load_tables<C>(tab, ct);
  load_vertices<C>(sV, m, e0);
  block_sync<C::BLOCK>();
  STAMP(0);
  // neighbour Up traces of all direction pairs: issued first, consumed by the jump passes
  NbTraces<C, NEQ> ta0, ta1, ta2;
  // The light (dry-air) 3-D kernels issue the records of the second and third pair one jump pass ahead of their use instead
  // of all at the start: 2 x NEQ fewer values live across the volume gradient, 134 -> 128 registers at p = 3 (143 / 145 ->
  // 127 / 128 at p = 4, 5) = FOUR waves per SIMD without a spill (round 2's 128-register cap cost 4 spilled registers).
  // Measured, alternating core libraries on one box (profiles/r04_ab_grad_late.txt): cfg2 k_gradient 0.400 -> 0.375 ms, the
  // Mult 0.823 -> 0.804.  The plasma kernels peak in their viscous phase (152 registers either way) and keep the early issue.
  constexpr bool LATE = TPSRHS_GRAD_LATE && ((DIM == 3 && (!C::NC || TPSRHS_GRAD_LATE_NC) && !PH::HEAVY) || TPSRHS_GRAD_LATE_ALL);
  issue_neighbour_traces<C, 0, NEQ>(sFI, TA, 2 * NEQ * C::NF, NEQ, ta0, tid);
  if (!LATE) {
    issue_neighbour_traces<C, 1, NEQ>(sFI, TA, 2 * NEQ * C::NF, NEQ, ta1, tid);
    if (DIM == 3) issue_neighbour_traces<C, (DIM == 3 ? 2 : 0), NEQ>(sFI, TA, 2 * NEQ * C::NF, NEQ, ta2, tid);
  }
  if (node_on) {
    const unsigned n = static_cast<unsigned>(e0 + le_n) * C::NPE + nd;
    double up[NEQ];
    PH::prim(prm, u, up);
#pragma unroll
    for (int eq = 0; eq < NEQ; eq++) {
      sU[eq * C::NODES + tid] = u[eq];
      sUp[eq * C::NODES + tid] = up[eq];
      // the Up grid function (a side effect Mult owns, src/rhs_operator.cpp:623-651) is written here: this
      // sweep is not bandwidth-bound, k_traces is
      if (TPSRHS_NT_STORES)
        __builtin_nontemporal_store(up[eq], &field_ptr(Upout, eq, m.ndofs)[n]);
      else
        field_ptr(Upout, eq, m.ndofs)[n] = up[eq];
    }
  }
  block_sync<C::BLOCK>();
  STAMP(1);

  // ---- volume part: collocation derivative (Ke then M^-1 of the reference collapse to it)
  double g[NEQ * DIM];  // g[eq + d*NEQ]
#pragma unroll
  for (int k = 0; k < NEQ * DIM; k++) g[k] = 0.0;
  double inv_mass = 0.0;
  if constexpr (C::NC) {
    nc_volume_gradient<C, NEQ>(sUp, pool + L::O_V, sV, tab, ct, node_on, le_n, nd, tid, g);
  } else if (node_on) {
    double xi[DIM], J[DIM * DIM], A[DIM * DIM];
    double iwn = 1.0;
#pragma unroll
    for (int d = 0; d < DIM; d++) {
      xi[d] = tab.x[idx[d]];
      iwn *= tab.iw[idx[d]];
    }
    jacobian<DIM>(&sV[le_n * C::NV * DIM], xi, J);
    const double det = adjugate<DIM>(J, A);
    const double idet = fast_rcp(det);
    inv_mass = iwn * idet;
    double Dr[DIM][C::N1];
#pragma unroll
    for (int mm = 0; mm < DIM; mm++)
#pragma unroll
      for (int a = 0; a < C::N1; a++) Dr[mm][a] = tab.D[idx[mm] * C::N1 + a];
#pragma unroll
    for (int eq = 0; eq < NEQ; eq++) {
      double dr[DIM];
#pragma unroll
      for (int mm = 0; mm < DIM; mm++) {
        const int m2 = ((TPSRHS_ABLATE & 128) && DIM == 3) ? 2 : mm;  // timing experiment
        const int sd = stride_of<C>(m2);
        const double *F = &sUp[eq * C::NODES + le_n * C::NPE + nd - idx[m2] * sd];
        double acc = 0.0;
#pragma unroll
        for (int a = 0; a < C::N1; a++) acc += Dr[mm][a] * ldsr(&F[a * sd]);
        dr[mm] = acc;
      }
#pragma unroll
      for (int d = 0; d < DIM; d++) {
        double s = 0.0;
#pragma unroll
        for (int mm = 0; mm < DIM; mm++) s += A[mm + d * DIM] * dr[mm];
        g[eq + d * NEQ] = s * idet;
      }
    }
  }

  STAMP(2);
  // ---- face part of the gradient, collocated (grad_jump_nodal), one direction pair at a time
  if (!(TPSRHS_ABLATE & 8)) {
    // the jump of one direction pair: collocated (a product at the face nodes), or through the face quadrature
    // points when nodes and points differ
    auto jump = [&](auto dtag) {
      constexpr int D = decltype(dtag)::value;
      if constexpr (C::NC) {
        nc_grad_jump<C, PH, D>(sFI, prm, sJ, pool + L::O_V, sV, tab, ct, node_on, le_n, idx, tid, g);
      } else {
        if (node_on) grad_jump_nodal<C, PH>(D, sFI, prm, sJ, sJ + NEQ * C::TN, sV, tab, le_n, idx, inv_mass, g);
        block_sync<C::BLOCK>();
      }
    };
    if (LATE) issue_neighbour_traces<C, 1, NEQ>(sFI, TA, 2 * NEQ * C::NF, NEQ, ta1, tid);
    trace_lines<C, 0, NEQ>(sUp, sJ, ct, tid);
    block_sync<C::BLOCK>();  // own traces complete (boundary faces copy them)
    store_neighbour_traces<C, NEQ>(ta0, sJ, sJ + NEQ * C::TN, tid);
    block_sync<C::BLOCK>();
    jump(std::integral_constant<int, 0>());
    if (LATE && DIM == 3) issue_neighbour_traces<C, (DIM == 3 ? 2 : 0), NEQ>(sFI, TA, 2 * NEQ * C::NF, NEQ, ta2, tid);
    trace_lines<C, 1, NEQ>(sUp, sJ, ct, tid);
    block_sync<C::BLOCK>();
    store_neighbour_traces<C, NEQ>(ta1, sJ, sJ + NEQ * C::TN, tid);
    block_sync<C::BLOCK>();
    jump(std::integral_constant<int, 1>());
    if (DIM == 3) {
      trace_lines<C, (DIM == 3 ? 2 : 0), NEQ>(sUp, sJ, ct, tid);
      block_sync<C::BLOCK>();
      store_neighbour_traces<C, NEQ>(ta2, sJ, sJ + NEQ * C::TN, tid);
      block_sync<C::BLOCK>();
      jump(std::integral_constant<int, (DIM == 3 ? 2 : 0)>());
    }
  }
  // non-collocated: gradUp = Me^-1 (Ke Up + face terms), dense (src/gradients.cpp:198-229)
  if constexpr (C::NC) {
    if constexpr (C::NPE == 64 && C::BLOCK == 128 && !TPSRHS_NO_MFMA) {
      static_assert(L::V >= 64 * 17, "LDS buffer of the MFMA inverse mass");
      nc_apply_minv_mfma<C, NEQ * DIM>(m.minv, pool + L::O_V, node_on, e0, nd, tid, g);
    }
    else
      nc_apply_minv<C, NEQ * DIM>(m.minv, pool + L::O_V, node_on, e0 + le_n, le_n, nd, tid, g);
  }

  STAMP(3);
  if (node_on) {
    const unsigned n = static_cast<unsigned>(e0 + le_n) * C::NPE + nd;
#pragma unroll
    for (int k = 0; k < NEQ * DIM; k++) field_ptr(gradUp, k, m.ndofs)[n] = g[k];
    if constexpr (L::G_IN_LDS) {
#pragma unroll
      for (int k = 0; k < NEQ * DIM; k++) sJ[k * C::NODES + tid] = g[k];
    }
  }
  if (L::G_IN_LDS) block_sync<C::BLOCK>();
  STAMP(4);
  if constexpr (L::LEAN3D) {
    if (!(TPSRHS_ABLATE & 16)) {
      // the lean face kernel reads the gradient back from the vector written above (same lanes, same addresses), after
      // the stores have been acknowledged.  The memory clobber keeps the compiler from satisfying those reads from the
      // registers that held g (which would keep all 36 of them alive across the closure).
      asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
      const double *gq = gradUp;
      const unsigned node = static_cast<unsigned>(e0 + le_n) * C::NPE + nd;
      visc_phase_lean3d<C, PH>(m, sFI, prm, e0, sU, gq, node_on, node, sJ, sUp, sW, sV, tab, ct, TB, tid STAMP_ARG);
    }
    STAMP_FLUSH();
    return;
  }

  // ---- viscous normal-flux traces (T chunk in the sUp region: the nodal Up values are dead)
  if (!(TPSRHS_ABLATE & 16)) {
    if constexpr (DIM == 2) {
      if constexpr (PH::HEAVY) {
        // mixing-length model (tpsrhs_set_mixing_length): the nodal wall distance of the block's elements
        __shared__ double sDist[C::NODES];
        const bool ml_on = m.ml.distance != nullptr;  // uniform over the grid
        if (ml_on) {
          if (tid < C::NODES)
            sDist[tid] = node_on ? m.ml.distance[static_cast<int64_t>(e0 + le_n) * C::NPE + nd] : 0.0;
          block_sync<C::BLOCK>();
        }
        visc_phase_2d<C, PH>(m, sFI, prm, e0, sU, g, node_on, sJ, sUp, sV, tab, ct, TB, tid, ml_on ? sDist : nullptr);
      } else {
        visc_phase_2d<C, PH>(m, sFI, prm, e0, sU, g, node_on, sJ, sUp, sV, tab, ct, TB, tid);
      }
    } else if constexpr (PH::TWO_STEP && C::Q_ROUNDS == 1 && !(TPSRHS_ABLATE & 256)) {
      visc_phase_heavy3d<C, PH>(m, sFI, prm, e0, sU, sJ, sUp, sW, sV, tab, ct, TB, tid STAMP_ARG);
    } else if constexpr (PH::HEAVY) {
#pragma clang loop unroll(disable)
      for (int d = 0; d < DIM; d++) {
        double v[C::Q_ROUNDS][L::NVF];
        if (d == 0)
          visc_interp_dir<C, PH, 0>(sU, g, node_on, sJ, sUp, sW, tab, ct, v, tid);
        else if (d == 1)
          visc_interp_dir<C, PH, 1>(sU, g, node_on, sJ, sUp, sW, tab, ct, v, tid);
        else
          visc_interp_dir<C, PH, (DIM == 3 ? 2 : 0)>(sU, g, node_on, sJ, sUp, sW, tab, ct, v, tid);
        visc_points<C, PH>(m, sFI, prm, e0, d, v, sV, tab, TB, tid);
      }
    } else {
      visc_traces_dir<C, PH, 0>(m, sFI, prm, e0, sU, g, node_on, sJ, sUp, sW, sV, tab, ct, TB, tid);
      visc_traces_dir<C, PH, 1>(m, sFI, prm, e0, sU, g, node_on, sJ, sUp, sW, sV, tab, ct, TB, tid);
      if (DIM == 3)
        visc_traces_dir<C, PH, (DIM == 3 ? 2 : 0)>(m, sFI, prm, e0, sU, g, node_on, sJ, sUp, sW, sV, tab, ct, TB, tid);
    }
  }
  STAMP_FLUSH();
}

// =============================================================================================
// sweep 2: y = M^-1 [ (grad phi, F_c - F_v)  -  <phi, F^ . n> ]  (+ point sources)
// =============================================================================================
// The optional forcing terms at one node (added to y after the inverse mass, src/rhs_operator.cpp:451-461).
//   X: node position; u, st: conserved state and its closure; gr: gradUp[eq + d*NEQ]
template <class C, class PH>
__device__ inline void apply_forcing(const ForcingDev &f, typename PH::PRef prm, int64_t node, const double *X,
                                     const double *u, const typename PH::State &st, const double *gr, double *src) {
  constexpr int NEQ = PH::NEQ, DIM = C::DIM, NVEL = PH::NVEL;
  // ConstantPressureGradient::updateTerms, src/forcing_terms.cpp:150-170 (dim, not nvel: no pressure
  // gradient in the theta direction)
  if (f.has_pg) {
    double gpv = 0.0;
#pragma unroll
    for (int d = 0; d < DIM; d++) {
      src[1 + d] -= f.pg[d];
      gpv -= st.vel[d] * f.pg[d];
      gpv -= st.p * gr[(1 + d) + d * NEQ];
    }
    src[1 + NVEL] += gpv;
  }
  // PassiveScalar::updateTerms, src/forcing_terms.cpp:826-848 (node list of the constructor, :795-818): the last equation
  if (f.nps > 0) {
    double up[NEQ];
    PH::prim(prm, u, up);
    for (int i = 0; i < f.nps; i++) {
      const ForcingDev::Scalar &ps = f.ps[i];
      double dist = 0.0;
#pragma unroll
      for (int d = 0; d < DIM; d++) dist += (X[d] - ps.x0[d]) * (X[d] - ps.x0[d]);
      if (sqrt(dist) < ps.radius) {
        double vel = 0.0;
#pragma unroll
        for (int d = 0; d < DIM; d++) vel += up[1 + d] * up[1 + d];
        vel = sqrt(vel);
        src[NEQ - 1] -= vel * (up[NEQ - 1] - up[0] * ps.value) / ps.radius;
      }
    }
  }
  // SpongeZone::addSpongeZoneForcing, src/forcing_terms.cpp:637-711; sigma of the constructor, :553-606
  for (int z = 0; z < f.nsponge; z++) {
    const ForcingDev::Sponge &sz = f.sponge[z];
    double dist_init = 0.0, dist_f = 0.0;
#pragma unroll
    for (int d = 0; d < DIM; d++) {
      dist_init -= sz.normal[d] * (X[d] - sz.pinit[d]);
      dist_f += sz.normal[d] * (X[d] - sz.p0[d]);
    }
    double sigma = 0.0;
    double tgt[NEQ];
#pragma unroll
    for (int eq = 0; eq < NEQ; eq++) tgt[eq] = sz.target[eq];
    if (sz.type == TPSRHS_SPONGE_PLANAR) {
      if (dist_init > 0.0 && dist_f > 0.0) {
        const double len = dist_f + dist_init;
        sigma = dist_init / len / len;
      }
    } else if constexpr (DIM == 3) {  // annulus: radial distance to the axis through pinit along normal
      double t[3], R = 0.0;
#pragma unroll
      for (int d = 0; d < 3; d++) {
        t[d] = X[d] - sz.pinit[d] + dist_init * sz.normal[d];
        R += t[d] * t[d];
      }
      R = sqrt(R);
      if (dist_init > 0.0 && dist_f > 0.0 && R - sz.r1 > 0.0) {
        const double len = sz.r2 - sz.r1;
        sigma = (R - sz.r1) / len / len;
        // target momentum given as (radial, azimuthal, axial): rows of MM are ur, uth = uz x ur, uz;
        // targetCyl(1..3) = MM^-1 targetU(1..3)  (:686-705)
        double ur[3], uz[3], ut[3];
#pragma unroll
        for (int d = 0; d < 3; d++) {
          ur[d] = t[d] / R;
          uz[d] = sz.normal[d];
        }
        ut[0] = uz[1] * ur[2] - ur[1] * uz[2];
        ut[1] = uz[2] * ur[0] - uz[0] * ur[2];
        ut[2] = uz[0] * ur[1] - ur[0] * uz[1];
        const double M[9] = {ur[0], ur[1], ur[2], ut[0], ut[1], ut[2], uz[0], uz[1], uz[2]};  // row-major
        const double c00 = M[4] * M[8] - M[5] * M[7], c01 = M[5] * M[6] - M[3] * M[8], c02 = M[3] * M[7] - M[4] * M[6];
        const double det = M[0] * c00 + M[1] * c01 + M[2] * c02;
        const double inv[9] = {c00, M[2] * M[7] - M[1] * M[8], M[1] * M[5] - M[2] * M[4],
                               c01, M[0] * M[8] - M[2] * M[6], M[2] * M[3] - M[0] * M[5],
                               c02, M[1] * M[6] - M[0] * M[7], M[0] * M[4] - M[1] * M[3]};
#pragma unroll
        for (int i = 0; i < 3; i++)
          tgt[1 + i] = (inv[3 * i] * sz.target[1] + inv[3 * i + 1] * sz.target[2] + inv[3 * i + 2] * sz.target[3]) / det;
      }
    }
    if (sigma > 0.0) {
      const double cs = PH::sound_speed(prm, sz.target);
      const double s = sigma * sz.mult;
      // Un = GetConservativesFromPrimitives(Up) of the reference is the node's conserved state again
#pragma unroll
      for (int eq = 0; eq < NEQ; eq++) src[eq] -= cs * s * (u[eq] - tgt[eq]);
    }
  }
  // HeatSource (type "cylinder"): node list of the constructor :890-917, updateTerms :923-936.
  // The reference indexes equation dim+1 (not nvel+1).
  for (int h = 0; h < f.nheat; h++) {
    const ForcingDev::Heat &hs = f.heat[h];
    double proj = 0.0, Xr[DIM];
#pragma unroll
    for (int d = 0; d < DIM; d++) {
      Xr[d] = X[d] - hs.p1[d];
      proj += Xr[d] * hs.axis[d];
    }
    double r2 = 0.0;
#pragma unroll
    for (int d = 0; d < DIM; d++) {
      const double r = Xr[d] - proj * hs.axis[d];
      r2 += r * r;
    }
    if (sqrt(r2) < hs.radius && proj > 0.0 && proj < hs.len) src[DIM + 1] += hs.value;
  }
  // JouleHeating::updateTerms, src/forcing_terms.cpp:443-471
  if (f.joule) {
    const double heating = f.joule[node];
    if (heating > 0.0) {
      src[NVEL + 1] += heating;
      if (PH::TWO_TEMPERATURE) src[NEQ - 1] += heating;
    }
  }
}

// =============================================================================================
// Non-reflecting inlet / outlet conditions (src/inletBC.cpp:576-727,
// src/outletBC.cpp:573-1027).  They carry a boundary state per face quadrature point that every Mult
// advances by dt, and they need the patch mean of the primitives (BCintegrator::updateBCMean,
// src/rhs_operator.cpp:364).  Two small kernels over the faces of such patches only:
//   k_bc_mean  after k_traces:   sums of the interpolated Up over the boundary quadrature points
//   k_bc_nr    after k_gradient: the characteristic update old state -> new state of every point
// k_flux reads the OLD state as the ghost of the Riemann solver (as the reference does: the update and the
// flux of one point use the state before the update); the host swaps the two buffers after the Mult.
// =============================================================================================
template <class C, class PH>
__device__ inline const double *nr_state(typename PH::PRef prm, int nb, int slot, int q) {
  if constexpr (PH::HAS_NR_BC) {
    const auto &bc = prm.bc[-nb - 1];
    if (is_non_reflecting(bc.category, bc.type))
      return prm.bstate + (static_cast<int64_t>(prm.nr_ordinal[slot]) * C::NQ + q) * PH::NEQ;
  }
  return nullptr;
}

// sums[b][0..NEQ-1] = sum over the faces of patch b and their quadrature points of the interpolated Up,
// sums[b][NEQ] = number of points (src/outletBC.cpp:485-531).  The interpolation is linear, so the sum over the
// points of a face is a weighted sum of its face-node traces (already in TA): weight = product of the column
// sums of the 1-D node -> quadrature matrix.  One block per boundary condition, fixed summation order.
template <class C, class PH>
__global__ __launch_bounds__(256) void k_bc_mean(int nfaces, const int2 *__restrict__ faces,
                                                 const double *__restrict__ TA, double *__restrict__ sums) {
  constexpr int NEQ = PH::NEQ;
  const Tables1D &ct = c_tab[C::NC][C::DIM - 2][C::P];
  __shared__ double red[256];
  const int b = blockIdx.x, tid = threadIdx.x;
  double cw[C::N1];
#pragma unroll
  for (int a = 0; a < C::N1; a++) {
    double c = 0.0;
#pragma unroll
    for (int q = 0; q < C::Q1; q++) c += ct.B[q * C::N1 + a];
    cw[a] = c;
  }
  double acc[NEQ + 1];
#pragma unroll
  for (int k = 0; k <= NEQ; k++) acc[k] = 0.0;
  for (int i = tid; i < nfaces; i += 256) {
    if (faces[i].y != b) continue;
    const double *rec = TA + static_cast<int64_t>(faces[i].x) * (2 * NEQ * C::NF) + NEQ * C::NF;  // Up traces
    for (int fn = 0; fn < C::NF; fn++) {
      double w = 0.0;
#pragma unroll
      for (int a = 0; a < C::N1; a++)
#pragma unroll
        for (int bb = 0; bb < (C::DIM == 3 ? C::N1 : 1); bb++)
          if (fn == a + C::N1 * bb) w = cw[a] * (C::DIM == 3 ? cw[bb] : 1.0);
#pragma unroll
      for (int eq = 0; eq < NEQ; eq++) acc[eq] += w * rec[eq * C::NF + fn];
    }
    acc[NEQ] += C::NQ;
  }
  for (int k = 0; k <= NEQ; k++) {
    red[tid] = acc[k];
    __syncthreads();
    for (int off = 128; off > 0; off >>= 1) {
      if (tid < off) red[tid] += red[tid + off];
      __syncthreads();
    }
    if (tid == 0) sums[b * (TPSRHS_MAXEQUATIONS + 1) + k] = red[0];
    __syncthreads();
  }
}

// One block per face of a non-reflecting patch, one lane per face quadrature point.
template <class C, class PH>
__global__ __launch_bounds__(C::BLOCK) void k_bc_nr(MeshDev m, typename PH::KArg prm_k, const int2 *__restrict__ faces,
                                              const double *__restrict__ sums, const double *__restrict__ U,
                                              const double *__restrict__ Up, const double *__restrict__ gradUp,
                                              double *__restrict__ state_old, double *__restrict__ state_new, int first,
                                              const double *__restrict__ dt_dev) {
  typename PH::PRef prm = PH::pref(prm_k);
  constexpr int NEQ = PH::NEQ, DIM = C::DIM;
  // tpsrhs_advance keeps dt in device memory.  (Not written into `prm`: a kernel that modifies its by-value
  // parameter block gets a private copy of all of it -- 1.9 KB of scratch per lane in round 2.)
  const double nr_dt = dt_dev ? *dt_dev : prm.nr_dt;
  static_assert(C::NQ <= C::BLOCK, "one lane per face quadrature point");
  const Tables1D &ct = c_tab[C::NC][DIM - 2][C::P];
  __shared__ Tab<C> tab;
  load_tables<C>(tab, ct);
  __syncthreads();
  const int q = threadIdx.x;
  if (q >= C::NQ) return;
  const int slot = faces[blockIdx.x].x, b = faces[blockIdx.x].y;
  const int e = slot / C::NFACES, lf = slot - e * C::NFACES, D = lf >> 1, s = lf & 1;
  const int da = (DIM == 2) ? 1 - D : (D == 0 ? 1 : 0), db = (D == 2) ? 1 : 2;
  int str[3] = {1, C::N1, C::N1 * C::N1};
  const int sd = str[D], sa = str[da], sb = (DIM == 3) ? str[db] : 0;
  const int qa = (DIM == 3) ? q % C::Q1 : q, qb = (DIM == 3) ? q / C::Q1 : 0;
  double Uq[NEQ], Upq[NEQ], g[NEQ * DIM];
#pragma unroll
  for (int eq = 0; eq < NEQ; eq++) Uq[eq] = Upq[eq] = 0.0;
#pragma unroll
  for (int k = 0; k < NEQ * DIM; k++) g[k] = 0.0;
  const int64_t n0 = static_cast<int64_t>(e) * C::NPE;
  for (int jd = 0; jd < C::N1; jd++) {
    const double wd = s ? tab.b1[jd] : tab.b0[jd];
    for (int jb = 0; jb < (DIM == 3 ? C::N1 : 1); jb++) {
      const double wb = (DIM == 3) ? tab.B[qb * C::N1 + jb] : 1.0;
      for (int ja = 0; ja < C::N1; ja++) {
        const double w = wd * wb * tab.B[qa * C::N1 + ja];
        const int64_t n = n0 + jd * sd + ja * sa + jb * sb;
#pragma unroll
        for (int eq = 0; eq < NEQ; eq++) {
          Uq[eq] += w * field_ptr(U, eq, m.ndofs)[n];
          if (first) Upq[eq] += w * field_ptr(Up, eq, m.ndofs)[n];
        }
#pragma unroll
        for (int k = 0; k < NEQ * DIM; k++) g[k] += w * field_ptr(gradUp, k, m.ndofs)[n];
      }
    }
  }
  double nrm[DIM], wq, Xq[DIM];
  face_geometry_rt<C>(D, m.verts + static_cast<int64_t>(e) * C::NV * DIM, tab, s, q, nrm, wq, Xq);
  const double *sm = sums + b * (TPSRHS_MAXEQUATIONS + 1);
  double meanUp[NEQ];
#pragma unroll
  for (int eq = 0; eq < NEQ; eq++) meanUp[eq] = sm[eq] / sm[NEQ];
  const int64_t rec = (static_cast<int64_t>(blockIdx.x) * C::NQ + q) * NEQ;
  double s2[NEQ], newU[NEQ];
  if (first) {  // initBoundaryU / first updateMean: cons(interpolated Up), src/outletBC.cpp:420-468,541-560
    PH::cons(prm, Upq, s2);
#pragma unroll
    for (int eq = 0; eq < NEQ; eq++) state_old[rec + eq] = s2[eq];
  } else {
#pragma unroll
    for (int eq = 0; eq < NEQ; eq++) s2[eq] = state_old[rec + eq];
  }
  PH::nr_update(prm, prm.bc[b], nr_dt, meanUp, nrm, Uq, g, s2, newU);
#pragma unroll
  for (int eq = 0; eq < NEQ; eq++) state_new[rec + eq] = newU[eq];
}

// SpongeZone::computeMixedOutValues (src/forcing_terms.cpp:713-743): the convective normal flux summed over the
// nodes of the mix-out plane -- one block, fixed summation order -- then (after the sum over the ranks) the
// mixed-out state of the mean flux becomes the zone's target.
template <class C, class PH>
__global__ __launch_bounds__(256) void k_mixed_out_sum(int64_t ndofs, typename PH::KArg prm_k, const ForcingDev *fd, int z,
                                                       const double *__restrict__ U) {
  typename PH::PRef prm = PH::pref(prm_k);
  constexpr int NEQ = PH::NEQ;
  const ForcingDev::Sponge &sz = fd->sponge[z];
  __shared__ double red[256];
  const int tid = threadIdx.x;
  double acc[NEQ];
#pragma unroll
  for (int eq = 0; eq < NEQ; eq++) acc[eq] = 0.0;
  for (int i = tid; i < sz.n_plane; i += 256) {
    const int64_t n = sz.plane_nodes[i];
    double u[NEQ], fn[NEQ];
#pragma unroll
    for (int eq = 0; eq < NEQ; eq++) u[eq] = field_ptr(U, eq, ndofs)[n];
    PH::conv_flux_n(prm, u, PH::make_state(prm, u), sz.normal, fn);
#pragma unroll
    for (int eq = 0; eq < NEQ; eq++) acc[eq] += fn[eq];
  }
  for (int eq = 0; eq < NEQ; eq++) {
    red[tid] = acc[eq];
    __syncthreads();
    for (int s = 128; s > 0; s >>= 1) {
      if (tid < s) red[tid] += red[tid + s];
      __syncthreads();
    }
    if (tid == 0) sz.msum[eq] = red[0];
    __syncthreads();
  }
  if (tid == 0) sz.msum[NEQ] = static_cast<double>(sz.n_plane);
}
template <class C, class PH>
__global__ void k_mixed_out_finish(typename PH::KArg prm_k, ForcingDev *fd, int z) {
  typename PH::PRef prm = PH::pref(prm_k);
  constexpr int NEQ = PH::NEQ;
  ForcingDev::Sponge &sz = fd->sponge[z];
  double mean[NEQ], tgt[NEQ];
#pragma unroll
  for (int eq = 0; eq < NEQ; eq++) mean[eq] = sz.msum[eq] / sz.msum[NEQ];
  // no node of ANY rank lies within tol of the plane: 0 / 0 as in the reference (src/forcing_terms.cpp:737-739); the
  // NaN target is left to Check_NAN (k_rk4_stage's census) rather than silently replaced
  PH::state_from_mean_flux(prm, mean, sz.normal, tgt);
#pragma unroll
  for (int eq = 0; eq < NEQ; eq++) sz.target[eq] = tgt[eq];
}

// y += optional forcing terms: a streaming pass of its own (one lane per node), launched after k_flux
// only when such a term is configured -- the hot sweeps carry neither its registers nor a branch.
template <class C, class PH>
__global__ __launch_bounds__(256) void k_forcing(MeshDev m, typename PH::KArg prm_k, const ForcingDev *__restrict__ fd,
                                                 const double *__restrict__ U, const double *__restrict__ gradUp,
                                                 double *__restrict__ Y) {
  typename PH::PRef prm = PH::pref(prm_k);
  constexpr int NEQ = PH::NEQ, DIM = C::DIM;
  const Tables1D &ct = c_tab[C::NC][DIM - 2][C::P];
  const int64_t n = static_cast<int64_t>(blockIdx.x) * 256 + threadIdx.x;
  if (n >= m.ndofs) return;
  const int e = static_cast<int>(n / C::NPE), nd = static_cast<int>(n - static_cast<int64_t>(e) * C::NPE);
  const int idx[3] = {nd % C::N1, (nd / C::N1) % C::N1, nd / (C::N1 * C::N1)};
  double xi[DIM], X[DIM];
#pragma unroll
  for (int d = 0; d < DIM; d++) xi[d] = ct.x[idx[d]];
  position<DIM>(m.verts + static_cast<int64_t>(e) * C::NV * DIM, xi, X);
  double u[NEQ], gr[NEQ * DIM], src[NEQ];
#pragma unroll
  for (int eq = 0; eq < NEQ; eq++) {
    u[eq] = field_ptr(U, eq, m.ndofs)[n];
    src[eq] = 0.0;
  }
  if (fd->has_pg) {  // the only term that reads the gradient: div u
#pragma unroll
    for (int d = 0; d < DIM; d++) gr[(1 + d) + d * NEQ] = field_ptr(gradUp, (1 + d) + d * NEQ, m.ndofs)[n];
  }
  const typename PH::State st = PH::make_state(prm, u);  // the reference reads Up of the unclamped state
  apply_forcing<C, PH>(*fd, prm, n, X, u, st, gr, src);
#pragma unroll
  for (int eq = 0; eq < NEQ; eq++)
    if (src[eq] != 0.0) field_ptr(Y, eq, m.ndofs)[n] += src[eq];
}

template <class C, class PH>
struct FluxLds {
  static constexpr int NEQ = PH::NEQ, DIM = C::DIM;
  // One direction pair at a time: X = T | R (3-D) | L, Y = W | W2 (3-D) or R (2-D).  2-D with few equations:
  // both direction pairs at once (Cfg::TQ2): X = (own | neighbour traces, later L) x 2, Y = R x 2.  (With the
  // 11 equations of the six-species mixture the second set of prefetched traces costs more registers than
  // the fuller rounds give back: measured 0.84 -> 1.04 ms.)
  static constexpr bool BOTH_2D = (DIM == 2) && (NEQ <= 8);
  static constexpr int X = BOTH_2D ? 4 * NEQ * C::TN : cmax(cmax(2 * NEQ * C::TN, NEQ * C::TQ), NEQ * C::TN);
  static constexpr int Y = BOTH_2D ? 2 * NEQ * C::TQ
                                   : cmax(cmax(2 * NEQ * C::TW, NEQ * C::TW), (DIM == 2) ? NEQ * C::TQ : 0);
  // non-collocated variant: the nodal flux and the scratch of the volume operator live together; the vectors of
  // the dense inverse mass are staged in X at the end
  static constexpr int NCV = C::NC ? NcScratch<C>::TOTAL : 0;
  static constexpr int GF = cmax(NEQ * DIM * C::NODES + NCV, X + Y);  // sGf (+ scratch), then X and Y
  static constexpr int TOTAL = NEQ * C::NODES + GF;
  static_assert(!C::NC || X >= NEQ * C::NODES, "staging of the inverse-mass vectors");
};

// k_flux of the time loop can form the next stage's face-node traces in its epilogue (RkDev::ta_out) when the two nodal field
// sets and the traces of one direction pair fit into its LDS pool: the 3-D collocated kernels.
template <class C, class PH>
constexpr bool flux_fuses_traces() {
  return C::DIM == 3 && !C::NC && FluxLds<C, PH>::TOTAL >= 2 * PH::NEQ * C::NODES + 2 * PH::NEQ * C::TN;
}

template <class PH>
__device__ inline typename PH::PRef flux_params(typename PH::PRef p) {
  if constexpr (PH::LAUNDER_FLUX)
    return PH::relaunder(p);
  else
    return p;
}
template <class C, class PH, int D>
__device__ inline void face_flux_dir(const MeshDev &m, typename PH::PRef prm, int e0, const double *sU,
                                     double *X, double *Yb, const double *sV, const Tab<C> &tab, const Tables1D &ct,
                                     const NbTraces<C, PH::NEQ> &ta, const NbFlux<C, PH::NEQ> &tb, bool node_on,
                                     int le_n, const int *idx, double *z, int tid) {
  constexpr int NEQ = PH::NEQ, DIM = C::DIM;
  trace_lines<C, D, NEQ>(sU, X, ct, tid);
  block_sync<C::BLOCK>();
  store_neighbour_traces<C, NEQ>(ta, X, X + NEQ * C::TN, tid);
  block_sync<C::BLOCK>();
  interp1_lines<C, 2 * NEQ>(X, Yb, ct, tid);
  if (DIM == 3) block_sync<C::BLOCK>();
  double fh[C::Q_ROUNDS][NEQ];
#pragma unroll
  for (int rd = 0; rd < C::Q_ROUNDS; rd++) {
    const int item = tid + rd * C::BLOCK;
#pragma unroll
    for (int eq = 0; eq < NEQ; eq++) fh[rd][eq] = 0.0;
    const int nb = tb.nb[rd];
    if (nb == INT32_MIN) continue;
    const int pf = item / C::NQ, q = item - pf * C::NQ;
    const int le = pf >> 1, s = pf & 1;
    double bq[C::N1];
    const int qrow = (DIM == 2) ? q : q / C::Q1;
#pragma unroll
    for (int a = 0; a < C::N1; a++) bq[a] = tab.B[qrow * C::N1 + a];
    double u1[NEQ], u2[NEQ];
#pragma unroll
    for (int eq = 0; eq < NEQ; eq++) {
      u1[eq] = interp2_point<C>(X + eq * C::TN, Yb + eq * C::TW, bq, pf, q);
      u2[eq] = interp2_point<C>(X + (NEQ + eq) * C::TN, Yb + (NEQ + eq) * C::TW, bq, pf, q);
    }
    PH::clamp_species(u1);
    double n[DIM], wq, Xq[DIM];
    face_geometry<C, D>(&sV[le * C::NV * DIM], tab, s, q, n, wq, Xq);
    if (TPSRHS_ABLATE & 4) {
#pragma unroll
      for (int eq = 0; eq < NEQ; eq++) fh[rd][eq] = u1[eq] + u2[eq] * n[0];
    } else if (nb >= 0) {
      PH::clamp_species(u2);
      PH::riemann(prm, u1, u2, n, fh[rd]);
#pragma unroll
      for (int eq = 0; eq < NEQ; eq++) fh[rd][eq] -= 0.5 * (tb.own[rd][eq] - tb.nbv[rd][eq]);
    } else {
      double ug[NEQ];
      PH::bc_ghost(prm, prm.bc[-nb - 1], u1, n, ug, nr_state<C, PH>(prm, nb, (e0 + le) * C::NFACES + 2 * D + s, q));
      PH::riemann_bc(prm, prm.bc[-nb - 1], u1, ug, n, fh[rd]);
#pragma unroll
      for (int eq = 0; eq < NEQ; eq++) fh[rd][eq] += tb.own[rd][eq];
    }
    if (PH::AXISYM) wq *= Xq[0];  // r-weighted face integrals (src/face_integrator.cpp:338, BCintegrator.cpp:427)
#pragma unroll
    for (int eq = 0; eq < NEQ; eq++) fh[rd][eq] *= wq;
  }
  block_sync<C::BLOCK>();  // X (T) and Y (W) are dead
  double *R = (DIM == 2) ? Yb : X;
#pragma unroll
  for (int rd = 0; rd < C::Q_ROUNDS; rd++) {
    const int item = tid + rd * C::BLOCK;
    if (item < C::TQ) {
#pragma unroll
      for (int eq = 0; eq < NEQ; eq++) R[eq * C::TQ + item] = fh[rd][eq];
    }
  }
  block_sync<C::BLOCK>();
  project1_lines<C, NEQ>(X, Yb, ct, tid);
  if (DIM == 3) block_sync<C::BLOCK>();
  project2_lines<C, NEQ>(Yb, X, ct, tid);
  block_sync<C::BLOCK>();
  if (node_on) {
#pragma unroll
    for (int eq = 0; eq < NEQ; eq++) z[eq] -= lift_pair<C, D>(X + eq * C::TN, tab, le_n, idx);
  }
  block_sync<C::BLOCK>();
}

// ---- 2-D: the face term of both direction pairs at once (full rounds of quadrature points, see Cfg::TQ2)
template <class C, int NEQ>
struct NbFlux2 {
  double own[C::Q2_ROUNDS][NEQ], nbv[C::Q2_ROUNDS][NEQ];
  int nb[C::Q2_ROUNDS];
};
template <class C, int NEQ>
__device__ inline void issue_visc_traces_2d(const int2 *sFI, int e0, const double *__restrict__ TB, NbFlux2<C, NEQ> &t,
                                            int tid) {
#pragma unroll
  for (int r = 0; r < C::Q2_ROUNDS; r++) {
    int item = tid + r * C::BLOCK;
    t.nb[r] = INT32_MIN;
#pragma unroll
    for (int k = 0; k < NEQ; k++) t.own[r][k] = t.nbv[r][k] = 0.0;
    if (item < C::TQ2) {
      const int d = item / C::TQ;
      item -= d * C::TQ;
      const int pf = item / C::NQ, q = item - pf * C::NQ;
      const int lslot = (pf >> 1) * C::NFACES + 2 * d + (pf & 1);
      const int2 fi = sFI[lslot];
      t.nb[r] = fi.x;
      if (fi.x != INT32_MIN && !(TPSRHS_ABLATE & 1)) {
        const double *o = TB + (static_cast<int64_t>(e0) * C::NFACES + lslot) * ((NEQ - 1) * C::NQ) + q;
#pragma unroll
        for (int k = 1; k < NEQ; k++) t.own[r][k] = o[(k - 1) * C::NQ];
        if (fi.x >= 0) {
          const int pq = permute<2>(fi.y, C::Q1, q, 0);
          const double *b2 = TB + static_cast<int64_t>(fi.x) * ((NEQ - 1) * C::NQ) + pq;
#pragma unroll
          for (int k = 1; k < NEQ; k++) t.nbv[r][k] = b2[(k - 1) * C::NQ];
        }
      }
    }
  }
}
template <class C, class PH>
__device__ inline void face_flux_2d(const MeshDev &m, typename PH::PRef prm0, int e0, const double *sU, double *X,
                                    double *Yb, const double *sV, const Tab<C> &tab, const Tables1D &ct,
                                    const NbTraces<C, PH::NEQ> &ta0, const NbTraces<C, PH::NEQ> &ta1,
                                    const NbFlux2<C, PH::NEQ> &tb, bool node_on, int le_n, const int *idx, double *z,
                                    int tid) {
  static_assert(C::DIM == 2, "2-D only");
  // (the table gas: its table records are fetched from the parameter image HERE, not at the top of the kernel, where they
  //  would sit in -- or be spilled from -- 100 SGPRs across the nodal physics)
  typename PH::PRef prm = flux_params<PH>(prm0);
  constexpr int NEQ = PH::NEQ, DIM = 2;
  constexpr int XS = 2 * NEQ * C::TN;  // own | neighbour traces of one direction pair
  trace_lines<C, 0, NEQ>(sU, X, ct, tid);
  trace_lines<C, 1, NEQ>(sU, X + XS, ct, tid);
  block_sync<C::BLOCK>();
  store_neighbour_traces<C, NEQ>(ta0, X, X + NEQ * C::TN, tid);
  store_neighbour_traces<C, NEQ>(ta1, X + XS, X + XS + NEQ * C::TN, tid);
  block_sync<C::BLOCK>();
  double fh[C::Q2_ROUNDS][NEQ];
#pragma unroll
  for (int rd = 0; rd < C::Q2_ROUNDS; rd++) {
    int item = tid + rd * C::BLOCK;
#pragma unroll
    for (int eq = 0; eq < NEQ; eq++) fh[rd][eq] = 0.0;
    const int nb = tb.nb[rd];
    if (nb == INT32_MIN) continue;
    const int d = item / C::TQ;
    item -= d * C::TQ;
    const int pf = item / C::NQ, q = item - pf * C::NQ;
    const int le = pf >> 1, s = pf & 1;
    const double *T = X + d * XS;
    double bq[C::N1];
#pragma unroll
    for (int a = 0; a < C::N1; a++) bq[a] = tab.B[q * C::N1 + a];
    double u1[NEQ], u2[NEQ];
#pragma unroll
    for (int eq = 0; eq < NEQ; eq++) {
      u1[eq] = interp2_point<C>(T + eq * C::TN, nullptr, bq, pf, q);
      u2[eq] = interp2_point<C>(T + (NEQ + eq) * C::TN, nullptr, bq, pf, q);
    }
    PH::clamp_species(u1);
    double n[DIM], wq, Xq[DIM];
    face_geometry_rt<C>(d, &sV[le * C::NV * DIM], tab, s, q, n, wq, Xq);
    if (nb >= 0) {
      PH::clamp_species(u2);
      PH::riemann(prm, u1, u2, n, fh[rd]);
#pragma unroll
      for (int eq = 0; eq < NEQ; eq++) fh[rd][eq] -= 0.5 * (tb.own[rd][eq] - tb.nbv[rd][eq]);
    } else {
      double ug[NEQ];
      PH::bc_ghost(prm, prm.bc[-nb - 1], u1, n, ug, nr_state<C, PH>(prm, nb, (e0 + le) * C::NFACES + 2 * d + s, q));
      PH::riemann_bc(prm, prm.bc[-nb - 1], u1, ug, n, fh[rd]);
#pragma unroll
      for (int eq = 0; eq < NEQ; eq++) fh[rd][eq] += tb.own[rd][eq];
    }
    if (PH::AXISYM) wq *= Xq[0];
#pragma unroll
    for (int eq = 0; eq < NEQ; eq++) fh[rd][eq] *= wq;
  }
  // R of direction pair d in Yb + d*NEQ*TQ (the traces are read, the region beyond them is free)
#pragma unroll
  for (int rd = 0; rd < C::Q2_ROUNDS; rd++) {
    int item = tid + rd * C::BLOCK;
    if (item < C::TQ2) {
      const int d = item / C::TQ;
      item -= d * C::TQ;
#pragma unroll
      for (int eq = 0; eq < NEQ; eq++) Yb[(d * NEQ + eq) * C::TQ + item] = fh[rd][eq];
    }
  }
  block_sync<C::BLOCK>();  // R complete; every read of the traces in X is done
  project2_lines<C, NEQ>(Yb, X, ct, tid);
  project2_lines<C, NEQ>(Yb + NEQ * C::TQ, X + XS, ct, tid);
  block_sync<C::BLOCK>();
  if (node_on) {
#pragma unroll
    for (int eq = 0; eq < NEQ; eq++)
      z[eq] -= lift_pair<C, 0>(X + eq * C::TN, tab, le_n, idx) + lift_pair<C, 1>(X + XS + eq * C::TN, tab, le_n, idx);
  }
  block_sync<C::BLOCK>();
}

template <class C, class PH>
__global__ __launch_bounds__(C::BLOCK, (C::NC && (PH::HEAVY || PH::MINW_FLUX > 2)) ? (PH::HEAVY ? 1 : 2) : PH::MINW_FLUX) void k_flux(MeshDev m, typename PH::KArg prm_k, const double *__restrict__ U,
                                                   const double *__restrict__ gradUp, const double *__restrict__ TA,
                                                   const double *__restrict__ TB, double *__restrict__ Y,
                                                   double *__restrict__ block_speed, RkDev rk) {
  typename PH::PRef prm = PH::pref(prm_k);
  constexpr int NEQ = PH::NEQ, DIM = C::DIM;
  typedef FluxLds<C, PH> L;
  const Tables1D &ct = c_tab[C::NC][DIM - 2][C::P];
  __shared__ Tab<C> tab;
  __shared__ double sV[C::EPB * C::NV * DIM];
  __shared__ double pool[L::TOTAL];
  double *sU = pool;                    // [NEQ][NODES]
  double *sGf = pool + NEQ * C::NODES;  // [NEQ*DIM][NODES] contravariant nodal flux; later X | Y
  double *sX = sGf, *sY = sGf + L::X;

  const int tid = threadIdx.x;
  const int lin = xcd_block(static_cast<int>(blockIdx.x), static_cast<int>(gridDim.x), m.reverse);
  const int bid = m.blocks ? m.blocks[lin] : lin;
  const int e0 = (TPSRHS_ABLATE & 512) ? (bid % 8) * C::EPB : bid * C::EPB;  // (512: timing experiment, as in k_gradient)
  __shared__ int2 sFI[C::EPB * C::NFACES];
  FSTAMP_DECL;
  const bool node_on = tid < C::NODES && (e0 + tid / C::NPE) < m.ne;
  const int le_n = tid / C::NPE, nd = tid - le_n * C::NPE;
  int idx[3] = {nd % C::N1, (nd / C::N1) % C::N1, nd / (C::N1 * C::N1)};
  // the nodal loads go out first: their latency (a full trip to HBM under load) then overlaps that of the
  // tables, vertices and face records instead of following it
  double u[NEQ], gr[NEQ * DIM];
  if (node_on) {
    const unsigned n = static_cast<unsigned>(e0 + le_n) * C::NPE + nd;
#pragma unroll
    for (int eq = 0; eq < NEQ; eq++) u[eq] = field_ptr(U, eq, m.ndofs)[n];
#pragma unroll
    for (int k = 0; k < NEQ * DIM; k++) gr[k] = field_ptr(gradUp, k, m.ndofs)[n];
  }
  load_face_info<C>(sFI, m, e0);
  load_tables<C>(tab, ct);
  load_vertices<C>(sV, m, e0);
  block_sync<C::BLOCK>();
  FSTAMP(0);
  NbTraces<C, NEQ> ta0;
  NbFlux<C, NEQ> tb0;
  // the neighbour records of the first direction pair go out before the nodal physics (their latency hides behind it)
  // -- except in the wide 2-D kernels (more than 8 equations): 3 x NEQ values held across the closure there are 66
  // VGPRs of a kernel that otherwise spills 90 (torch6, round 2); those issue them where the face term starts
  constexpr bool EARLY = !(DIM == 2 && NEQ > 8) && !TPSRHS_FLUX_LATE;
  if (EARLY) {
    issue_neighbour_traces<C, 0, NEQ>(sFI, TA, 2 * NEQ * C::NF, 0, ta0, tid);
    if (!L::BOTH_2D) issue_visc_traces<C, 0, NEQ>(sFI, e0, TB, tb0, tid);
  }
  constexpr bool PARK = TPSRHS_FLUX_PARK_GRAD && PH::TWO_STEP && !C::NC && DIM == 3;
  if (node_on) {
#pragma unroll
    for (int eq = 0; eq < NEQ; eq++) sU[eq * C::NODES + tid] = u[eq];
    if constexpr (PARK) {  // (the lane's own column of sGf: overwritten by the lane's own flux at the end of the nodal physics)
#pragma unroll
      for (int k = 0; k < NEQ * DIM; k++) sGf[k * C::NODES + tid] = gr[k];
    }
  }
  block_sync<C::BLOCK>();  // tables + vertices + sU
  FSTAMP(1);

  // ---- nodal flux F_c - F_v (src/rhs_operator.cpp:493-559), contravariant components.
  // Ordered to keep the register peak low: physics first (U, gradUp -> 15 flux entries), then the
  // geometry, then the contraction with the metric rows.
  double inv_mass = 0.0, speed = 0.0;
  double src[NEQ];
#pragma unroll
  for (int eq = 0; eq < NEQ; eq++) src[eq] = 0.0;
  if (node_on) {
    double F[NEQ * DIM];
    double radius = 1.0;  // axisymmetric: r of the node (x-coordinate), weights the mass and the volume term
    double Xn[DIM] = {};  // position of the node: the radius, the viscous sponge of the 2-D heavy kernels
    if constexpr (PH::AXISYM || (PH::HEAVY && DIM == 2)) {
      if (PH::AXISYM || m.vs.enabled) {
        double xi[DIM];
#pragma unroll
        for (int d = 0; d < DIM; d++) xi[d] = tab.x[idx[d]];
        position<DIM>(&sV[le_n * C::NV * DIM], xi, Xn);
      }
      if (PH::AXISYM) radius = Xn[0];
    }
    {
      double uc[NEQ];
#pragma unroll
      for (int eq = 0; eq < NEQ; eq++) uc[eq] = u[eq];
      PH::clamp_species(uc);
      const typename PH::State st = PH::make_state(prm, uc);
      speed = PH::max_char_speed(prm, uc, st);
      typename PH::FluxCoef fc;
      if constexpr (PH::TWO_STEP) PH::flux_coeffs(prm, uc, st, fc);
      if constexpr (PARK) {  // back from the LDS, through a pointer the compiler cannot see through (no store-to-load forwarding,
                             // which would keep the 2 * NEQ * DIM registers alive across the closure)
        const double *pg = sGf + tid;
        asm volatile("" : "+v"(pg) : : "memory");
#pragma unroll
        for (int k = 0; k < NEQ * DIM; k++) gr[k] = pg[k * C::NODES];
      }
      if (PH::HAS_SOURCE) {
        double up[NEQ];
        PH::prim(prm, u, up);
        PH::source(prm, u, up, gr, src);
        if constexpr (PH::AXISYM) PH::axisym_source(prm, u, up, gr, radius, src);
      }

      if (TPSRHS_ABLATE & 2) {
#pragma unroll
        for (int k = 0; k < NEQ * DIM; k++) F[k] = uc[k % NEQ] + gr[k];
      } else {
        if constexpr (PH::HEAVY && DIM == 2) {  // (+ the mixing-length eddy viscosity when a distance function is set)
          const bool ml_on = m.ml.distance != nullptr;
          __shared__ double sVsw[C::BLOCK];
          const EddyCtx ec = closure_ctx(m, true, ml_on ? m.ml.distance[static_cast<int64_t>(e0 + le_n) * C::NPE + nd] : 0.0, Xn, sVsw);
          if constexpr (PH::TWO_STEP)
            PH::total_flux(prm, uc, st, fc, gr, PH::AXISYM ? radius : -1.0, F, ec);
          else
            PH::total_flux(prm, uc, st, gr, radius, F, ec);
        } else if constexpr (PH::TWO_STEP)
          PH::total_flux(prm, uc, st, fc, gr, PH::AXISYM ? radius : -1.0, F);
        else if constexpr (PH::AXISYM)
          PH::total_flux(prm, uc, st, gr, radius, F);
        else if constexpr (PH::LES) {  // elSize and xyz of the node, src/rhs_operator.cpp:526-539
          PointCtx pc;
          double xn[DIM];
#pragma unroll
          for (int d = 0; d < DIM; d++) xn[d] = tab.x[idx[d]];
          position<DIM>(&sV[le_n * C::NV * DIM], xn, pc.X);
          pc.delta = prm.elem_delta[e0 + le_n];
          PH::total_flux(prm, uc, st, gr, F, &pc);
        } else
          PH::total_flux(prm, uc, st, gr, F);
      }
    }
    __builtin_amdgcn_sched_barrier(0);
    double xi[DIM], J[DIM * DIM], A[DIM * DIM];
    double wn = 1.0, iwn = 1.0;
#pragma unroll
    for (int d = 0; d < DIM; d++) {
      xi[d] = tab.x[idx[d]];
      wn *= tab.w[idx[d]];
      iwn *= tab.iw[idx[d]];
    }
    jacobian<DIM>(&sV[le_n * C::NV * DIM], xi, J);
    const double det = adjugate<DIM>(J, A);
    inv_mass = iwn * fast_rcp(det);
    if (PH::AXISYM) {  // Me_inv_rad and the r-weighted DomainIntegrator (src/rhs_operator.cpp:198-201,
      inv_mass *= fast_rcp(radius);  // src/domain_integrator.cpp:78-81)
      wn *= radius;
    }
    __builtin_amdgcn_sched_barrier(0);
    if constexpr (C::NC) {  // the physical nodal flux: the metric is applied at the volume quadrature points
#pragma unroll
      for (int k = 0; k < NEQ * DIM; k++) sGf[k * C::NODES + tid] = F[k];
    } else {
#pragma unroll
      for (int eq = 0; eq < NEQ; eq++)
#pragma unroll
        for (int mm = 0; mm < DIM; mm++) {
          double s = 0.0;
#pragma unroll
          for (int d = 0; d < DIM; d++) s += A[mm + d * DIM] * F[eq + d * NEQ];
          sGf[(eq + mm * NEQ) * C::NODES + tid] = wn * s;
        }
    }
  }
  // max |u|+c of the block -> one slot per block.  (A single global atomicMax per wave serialises
  // at the L2: ~12 ns each, 0.6 ms for 50k waves -- measured; per-block stores cost nothing and a
  // tiny reduction kernel runs only when the caller asks for the value.)
  {
    double v = speed;
#pragma unroll
    for (int off = 32; off > 0; off >>= 1) v = fmax(v, __shfl_xor(v, off));
    if (C::BLOCK == 64) {
      if (tid == 0) block_speed[bid] = v;
    } else {
      __shared__ double swave[C::BLOCK / 64];
      if ((tid & 63) == 0) swave[tid >> 6] = v;
      block_sync<C::BLOCK>();
      if (tid == 0) {
        double b = swave[0];
#pragma unroll
        for (int w = 1; w < C::BLOCK / 64; w++) b = fmax(b, swave[w]);
        block_speed[bid] = b;
      }
    }
  }
  block_sync<C::BLOCK>();  // sGf complete
  FSTAMP(2);
#if TPSRHS_DUMPF
  if (tid < C::NODES) {  // [block][F (NEQ*DIM) | src (NEQ)][NODES]
    constexpr int ROWS = NEQ * DIM + NEQ;
    if ((static_cast<int64_t>(bid) + 1) * ROWS * C::NODES <= DUMPF_MAX) {
      double *out = g_dumpf + static_cast<int64_t>(bid) * ROWS * C::NODES + tid;
      for (int k = 0; k < NEQ * DIM; k++) out[k * C::NODES] = sGf[k * C::NODES + tid];
#pragma unroll
      for (int eq = 0; eq < NEQ; eq++) out[(NEQ * DIM + eq) * C::NODES] = src[eq];
    }
  }
#endif

  // ---- volume term: z_j = sum_m sum_a D[a][j_m] Ghat_m(a)   (src/domain_integrator.cpp:45-99)
  double z[NEQ];
#pragma unroll
  for (int eq = 0; eq < NEQ; eq++) z[eq] = 0.0;
  if constexpr (C::NC) {
    static_assert(!PH::AXISYM, "the non-collocated variant is planar / 3-D");
    nc_volume_divergence<C, NEQ>(sGf, sGf + NEQ * DIM * C::NODES, sV, tab, ct, node_on, le_n, nd, tid, z);
  } else if (node_on) {
    double Dc[DIM][C::N1];
#pragma unroll
    for (int mm = 0; mm < DIM; mm++)
#pragma unroll
      for (int a = 0; a < C::N1; a++) Dc[mm][a] = tab.D[a * C::N1 + idx[mm]];
#pragma unroll
    for (int eq = 0; eq < NEQ; eq++) {
      double acc = 0.0;
#pragma unroll
      for (int mm = 0; mm < DIM; mm++) {
        const int m2 = ((TPSRHS_ABLATE & 128) && DIM == 3) ? 2 : mm;  // timing experiment
        const int sd = stride_of<C>(m2);
        const double *F = &sGf[(eq + mm * NEQ) * C::NODES + le_n * C::NPE + nd - idx[m2] * sd];
#pragma unroll
        for (int a = 0; a < C::N1; a++) acc += Dc[mm][a] * ldsr(&F[a * sd]);
      }
      z[eq] = acc;
    }
  }
  block_sync<C::BLOCK>();  // sGf dead: its region becomes X | Y
  FSTAMP(3);

  // ---- face term
  if constexpr (L::BOTH_2D) {
    if (!(TPSRHS_ABLATE & 8)) {
      NbTraces<C, NEQ> ta1;
      NbFlux2<C, NEQ> tb;
      issue_neighbour_traces<C, 1, NEQ>(sFI, TA, 2 * NEQ * C::NF, 0, ta1, tid);
      issue_visc_traces_2d<C, NEQ>(sFI, e0, TB, tb, tid);
      face_flux_2d<C, PH>(m, prm, e0, sU, sX, sY, sV, tab, ct, ta0, ta1, tb, node_on, le_n, idx, z, tid);
    }
  } else if (!(TPSRHS_ABLATE & 8)) {  // one direction pair at a time
    // software pipeline over the direction pairs: the traces of pair d+1 are in flight while pair d runs
    NbTraces<C, NEQ> ta1, ta2;
    NbFlux<C, NEQ> tb1, tb2;
    // (not in 2-D with more than 8 equations: the prefetched pair costs 60+ registers of a kernel that already
    // spills -- torch6 k_flux 1.41 -> 1.28 ms without it)
    constexpr bool PIPE = !(DIM == 2 && NEQ > 8);
    if (!EARLY) {
      issue_neighbour_traces<C, 0, NEQ>(sFI, TA, 2 * NEQ * C::NF, 0, ta0, tid);
      issue_visc_traces<C, 0, NEQ>(sFI, e0, TB, tb0, tid);
    }
    if (PIPE) {
      issue_neighbour_traces<C, 1, NEQ>(sFI, TA, 2 * NEQ * C::NF, 0, ta1, tid);
      issue_visc_traces<C, 1, NEQ>(sFI, e0, TB, tb1, tid);
    }
    face_flux_dir<C, PH, 0>(m, prm, e0, sU, sX, sY, sV, tab, ct, ta0, tb0, node_on, le_n, idx, z, tid);
    if (!PIPE) {
      issue_neighbour_traces<C, 1, NEQ>(sFI, TA, 2 * NEQ * C::NF, 0, ta1, tid);
      issue_visc_traces<C, 1, NEQ>(sFI, e0, TB, tb1, tid);
    }
    FSTAMP(4);
    if (DIM == 3) {
      issue_neighbour_traces<C, (DIM == 3 ? 2 : 0), NEQ>(sFI, TA, 2 * NEQ * C::NF, 0, ta2, tid);
      issue_visc_traces<C, (DIM == 3 ? 2 : 0), NEQ>(sFI, e0, TB, tb2, tid);
    }
    face_flux_dir<C, PH, 1>(m, prm, e0, sU, sX, sY, sV, tab, ct, ta1, tb1, node_on, le_n, idx, z, tid);
    FSTAMP(5);
    if (DIM == 3)
      face_flux_dir<C, PH, (DIM == 3 ? 2 : 0)>(m, prm, e0, sU, sX, sY, sV, tab, ct, ta2, tb2, node_on, le_n, idx, z,
                                                tid);
    FSTAMP(6);
  }

  if constexpr (C::NC) {  // y = Me^-1 z (src/rhs_operator.cpp:432-448), then the point sources
    if constexpr (C::NPE == 64 && C::BLOCK == 128 && !TPSRHS_NO_MFMA) {
      static_assert(L::GF >= 64 * 17, "LDS buffer of the MFMA inverse mass");
      nc_apply_minv_mfma<C, NEQ>(m.minv, sGf, node_on, e0, nd, tid, z);  // (the flux / scratch region is free by now; sU is not: stage 1 of the time loop reads it)
    }
    else
      nc_apply_minv<C, NEQ>(m.minv, sX, node_on, e0 + le_n, le_n, nd, tid, z);
    inv_mass = 1.0;
  }
  double vout[NEQ];  // the state this lane writes in the time loop (the traces below take it from here)
#pragma unroll
  for (int eq = 0; eq < NEQ; eq++) vout[eq] = 0.0;
  if (node_on) {
    const unsigned n = static_cast<unsigned>(e0 + le_n) * C::NPE + nd;
    if (rk.mode == 0) {
#pragma unroll
      for (int eq = 0; eq < NEQ; eq++) field_ptr(Y, eq, m.ndofs)[n] = inv_mass * z[eq] + src[eq];
    } else {  // RK4 stage combination (see RkDev); uniform branch
      const double dt = rk.dt_dev ? *rk.dt_dev : rk.dt_host;
      unsigned long long bad = 0;
#pragma unroll
      for (int eq = 0; eq < NEQ; eq++) {
        const double k = inv_mass * z[eq] + src[eq];
        double v;
        if (rk.mode == 1) {
          v = ldsr(&sU[eq * C::NODES + tid]) + (dt / 2) * k;
        } else if (rk.mode == 2) {
          v = field_ptr(rk.x0, eq, m.ndofs)[n] + (dt / 2) * k;
        } else if (rk.mode == 3) {
          v = field_ptr(rk.x0, eq, m.ndofs)[n] + dt * k;
        } else {
          const double x0 = field_ptr(rk.x0, eq, m.ndofs)[n];
          const double d2 = field_ptr(rk.y2, eq, m.ndofs)[n] - x0, d3 = field_ptr(rk.y3, eq, m.ndofs)[n] - x0,
                       d4 = field_ptr(rk.y4, eq, m.ndofs)[n] - x0;
          v = x0 + ((d2 + 2.0 * d3 + d4) * (1.0 / 3.0) + (dt / 6) * k);
          if (v != v) bad++;                                        // Check_NAN
          if (eq >= rk.sp_first && eq < rk.sp_last) v = fmax(v, 0.0);  // Check_Undershoot
        }
        field_ptr(rk.out, eq, m.ndofs)[n] = v;
        vout[eq] = v;
      }
      if (bad) atomicAdd(rk.nan_count, bad);
    }
  }
  if constexpr (flux_fuses_traces<C, PH>()) {
    if (rk.mode != 0 && rk.ta_out) {  // uniform: the k_traces sweep of the next stage, from registers (src/rhs_operator.cpp:361-372)
      block_sync<C::BLOCK>();  // stage 1 has read its input from sU: the pool is free
      double *sF = pool, *sT = pool + 2 * NEQ * C::NODES;
      if (node_on) {
        double up[NEQ];
        PH::prim(prm, vout, up);
#pragma unroll
        for (int eq = 0; eq < NEQ; eq++) {
          sF[eq * C::NODES + tid] = vout[eq];
          sF[(NEQ + eq) * C::NODES + tid] = up[eq];
        }
      }
      block_sync<C::BLOCK>();
      traces_dir<C, PH, 0>(m, e0, sF, sT, rk.ta_out, ct, tid);
      traces_dir<C, PH, 1>(m, e0, sF, sT, rk.ta_out, ct, tid);
      traces_dir<C, PH, (C::DIM == 3 ? 2 : 0)>(m, e0, sF, sT, rk.ta_out, ct, tid);
    }
  }
  FSTAMP(7);
  FSTAMP_FLUSH();
}

// Point-wise closures of the gas model for n states U[eq + ...]: U is [NEQ][n] (byNODES), one lane per state
// (tpsrhs_eval_pointwise).  quantity: 0 primitives -> out[NEQ][n], 1 pressure, 2 speed of sound, 3 |u| + c, 4 electric
// conductivity (tpsrhs_get_plasma_conductivity) -> out[n]
template <class PH>
__global__ __launch_bounds__(256) void k_point_eval(typename PH::KArg prm_k, int quantity, int64_t n,
                                                    const double *__restrict__ U, double *__restrict__ out) {
  typename PH::PRef prm = PH::pref(prm_k);
  constexpr int NEQ = PH::NEQ;
  const int64_t i = static_cast<int64_t>(blockIdx.x) * 256 + threadIdx.x;
  if (i >= n) return;
  double u[NEQ];
#pragma unroll
  for (int eq = 0; eq < NEQ; eq++) u[eq] = U[eq * n + i];
  if (quantity == 0) {
    double up[NEQ];
    PH::prim(prm, u, up);
#pragma unroll
    for (int eq = 0; eq < NEQ; eq++) out[eq * n + i] = up[eq];
  } else if (quantity == 1) {
    out[i] = PH::pressure(prm, u);
  } else if (quantity == 2) {
    out[i] = PH::sound_speed(prm, u);
  } else if (quantity == 4) {  // SrcTrns::ELECTRIC_CONDUCTIVITY as SourceTerm stores it (src/source_term.cpp:125-199): clamped species
    PH::clamp_species(u);
    out[i] = PH::electric_conductivity(prm, u);
  } else {
    out[i] = PH::max_char_speed(prm, u);
  }
}

// One block reduces the 50 176 per-block maxima of cfg2: a lane's loads must not wait for each other (round 2: one
// dependent load per iteration, 196 iterations of a full memory latency = 75 us per call of the time loop's k_step_end)
template <int BLOCK>
__device__ inline double strided_max(int n, const double *__restrict__ v) {
  constexpr int UN = 8;
  double m[UN];
#pragma unroll
  for (int u = 0; u < UN; u++) m[u] = 0.0;
  int i = threadIdx.x;
  for (; i + (UN - 1) * BLOCK < n; i += UN * BLOCK) {
    double t[UN];
#pragma unroll
    for (int u = 0; u < UN; u++) t[u] = v[i + u * BLOCK];  // UN independent loads in flight
#pragma unroll
    for (int u = 0; u < UN; u++) m[u] = fmax(m[u], t[u]);
  }
  for (; i < n; i += BLOCK) m[0] = fmax(m[0], v[i]);
#pragma unroll
  for (int u = 1; u < UN; u++) m[0] = fmax(m[0], m[u]);
  return m[0];
}

// max over the per-block maxima written by k_flux (one block)
template <int BLOCK>
__global__ void k_reduce_max(int n, const double *__restrict__ v, double *__restrict__ out) {
  __shared__ double s[BLOCK];
  const double m = strided_max<BLOCK>(n, v);
  s[threadIdx.x] = m;
  __syncthreads();
  for (int off = BLOCK / 2; off > 0; off >>= 1) {
    if (threadIdx.x < off) s[threadIdx.x] = fmax(s[threadIdx.x], s[threadIdx.x + off]);
    __syncthreads();
  }
  if (threadIdx.x == 0) *out = s[0];
}

// =============================================================================================
// RK4 stage combinations (MFEM RK4Solver::Step) as one streaming pass per stage:
//   stage 1: y = x + dt/2 k, z = x + dt/6 k     stage 2: y = x + dt/2 k, z += dt/3 k
//   stage 3: y = x + dt k,   z += dt/3 k        stage 4: x = z + dt/6 k (+ NaN census, species clamp)
// =============================================================================================
template <int BLOCK>
__global__ void k_rk4_stage(int stage, int64_t n, int64_t ndofs, int sp_first, int sp_last, double dt_host,
                            const double *__restrict__ dt_dev, double *__restrict__ x, const double *__restrict__ k,
                            double *__restrict__ y, double *__restrict__ z, unsigned long long *__restrict__ nan_count) {
  const double dt = dt_dev ? *dt_dev : dt_host;  // tpsrhs_advance keeps dt in device memory
  unsigned long long bad = 0;
  for (int64_t i = blockIdx.x * static_cast<int64_t>(BLOCK) + threadIdx.x; i < n; i += static_cast<int64_t>(gridDim.x) * BLOCK) {
    const double ki = k[i];
    if (stage == 1) {
      const double xi = x[i];
      y[i] = xi + (dt / 2) * ki;
      z[i] = xi + (dt / 6) * ki;
    } else if (stage == 2) {
      y[i] = x[i] + (dt / 2) * ki;
      z[i] += (dt / 3) * ki;
    } else if (stage == 3) {
      y[i] = x[i] + dt * ki;
      z[i] += (dt / 3) * ki;
    } else {
      double v = z[i] + (dt / 6) * ki;
      if (v != v) bad++;
      const int64_t eq = i / ndofs;
      if (eq >= sp_first && eq < sp_last) v = fmax(v, 0.0);  // Check_Undershoot
      x[i] = v;
    }
  }
  if (stage == 4 && bad) atomicAdd(nan_count, bad);
}

// End of a step of tpsrhs_advance (src/M2ulPhyS.cpp:2004-2016): time += dt; with a variable time step
// dt = CFL hmin / max_char_speed / dim from the per-block maxima the last k_flux left.  ctl = {dt, time, speed}.
template <int BLOCK>
__global__ void k_step_end(int n, const double *__restrict__ block_speed, double *__restrict__ ctl, int constant_dt,
                           double cfl_hmin_over_dim) {
  __shared__ double s[BLOCK];
  const double m = strided_max<BLOCK>(n, block_speed);
  s[threadIdx.x] = m;
  __syncthreads();
  for (int off = BLOCK / 2; off > 0; off >>= 1) {
    if (threadIdx.x < off) s[threadIdx.x] = fmax(s[threadIdx.x], s[threadIdx.x + off]);
    __syncthreads();
  }
  if (threadIdx.x == 0) {
    ctl[1] += ctl[0];
    ctl[2] = s[0];
    if (!constant_dt) ctl[0] = cfl_hmin_over_dim / s[0];
  }
}

// =============================================================================================
// halo packing: permute the traces of the shared faces into the canonical frame both ranks agree on
// (role of the pack kernel of initNBlockDataTransfer, src/rhs_operator.cpp:798-803)
// =============================================================================================
template <int DIM>
__global__ void k_pack(int nshared, int nfld, int n1, const int32_t *__restrict__ shared_slot,
                       const uint8_t *__restrict__ shared_orient, const double *__restrict__ T,
                       double *__restrict__ out) {
  const int per = (DIM == 3) ? n1 * n1 : n1;
  const int64_t total = static_cast<int64_t>(nshared) * nfld * per;
  for (int64_t i = blockIdx.x * static_cast<int64_t>(blockDim.x) + threadIdx.x; i < total;
       i += static_cast<int64_t>(gridDim.x) * blockDim.x) {
    const int s = static_cast<int>(i / (nfld * per));
    const int rem = static_cast<int>(i - static_cast<int64_t>(s) * nfld * per);
    const int fld = rem / per, k = rem - fld * per;
    const int pk = permute<DIM>(shared_orient[s], n1, k % n1, k / n1);
    out[(static_cast<int64_t>(s) * nfld + fld) * per + pk] = T[(static_cast<int64_t>(shared_slot[s]) * nfld + fld) * per + k];
  }
}

}  // namespace tpsrhs
#endif
