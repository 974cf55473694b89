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
// A workgroup owns EPB whole elements (one lane per node; one 64-lane wave per hex at p=3), keeps
// their nodal fields in LDS, and applies the element-independent 1-D operators (differentiation
// matrix, end-point values, node -> face-quadrature interpolation) by sum factorisation out of a
// 0.5 KB LDS table; two-dimensional face operators are factorised in two lane-parallel stages
// through LDS.  Geometry (Jacobians, area-weighted normals) is recomputed from the 2^dim vertex
// coordinates of the element: 192 B per hex instead of 10 doubles per node.
// LDS is one hand-allocated pool whose regions are re-used as fields die, so that 7-8 single-wave
// workgroups fit a CU at p=3.
#ifndef TPSRHS_KERNELS_HPP_
#define TPSRHS_KERNELS_HPP_

#include <hip/hip_runtime.h>

#include "basis.hpp"

namespace tpsrhs {

#ifndef TPSRHS_MINW
#define TPSRHS_MINW 1
#endif
#ifndef TPSRHS_ABLATE
#define TPSRHS_ABLATE 0  // timing experiments only (wrong results): 1 no trace reads, 2 no nodal physics, 4 no face physics
#endif
constexpr int cmax(int a, int b) { return a > b ? a : b; }

// LDS read of one double.  hipcc fuses neighbouring 8-byte LDS loads into ds_read2_b64, which the
// LDS serves at half to a quarter of the ds_read_b64 rate on gfx950 (MI355X_MICROARCH.md, LDS
// table: 8-16 cycles per wave-instruction against 2).  A volatile access is never fused.
typedef const volatile __attribute__((address_space(3))) double *lds_cvptr;
__device__ inline double ldsr(const double *p) { return *(lds_cvptr)(p); }

// Nodal field access fld[k*stride + n]: the field base is wave-uniform (SGPR pair) and the lane offset
// 32-bit, so that every load/store uses the scalar-base addressing form and no 64-bit per-lane
// address is kept in VGPRs (the host guarantees NDofs < 2^31).
__device__ inline const double *field_ptr(const double *base, int k, int64_t stride) { return base + k * stride; }
__device__ inline double *field_ptr(double *base, int k, int64_t stride) { return base + k * stride; }

template <int DIM_, int P_>
struct Cfg {
  static constexpr int DIM = DIM_, P = P_, N1 = P_ + 1;
  static constexpr int NPE = (DIM_ == 3) ? N1 * N1 * N1 : N1 * N1;
  static constexpr int NF = (DIM_ == 3) ? N1 * N1 : N1;
  static constexpr int Q1 = ((DIM_ - 1) + 2 * P_) / 2 + 1;
  static constexpr int NQ = (DIM_ == 3) ? Q1 * Q1 : Q1;
  static constexpr int NW = (DIM_ == 3) ? Q1 * N1 : 1;  // half-interpolated face values (3-D only)
  static constexpr int NFACES = 2 * DIM_;
  static constexpr int NV = 1 << DIM_;
  // Elements per block and block size: enough lanes that the face-quadrature loop of a block is a
  // single round (no per-round register arrays) while several waves share one LDS pool -> more
  // resident waves per CU to hide the HBM latency of the trace reads.
  static constexpr int EPB = (DIM_ == 3) ? (P_ == 1 ? 4 : (P_ == 2 ? 2 : 1)) : cmax(1, 64 / NPE);
  static constexpr int FQ_RAW = EPB * 2 * DIM_ * NQ;
  static constexpr int BLOCK = cmax(64, ((cmax(FQ_RAW, EPB * NPE) + 63) / 64) * 64) > 256
                                   ? 256
                                   : cmax(64, ((cmax(FQ_RAW, EPB * NPE) + 63) / 64) * 64);
  static constexpr int MINW = TPSRHS_MINW;  // launch-bounds waves per SIMD (register cap)
  static constexpr int NODES = EPB * NPE;  // active lanes in node loops
  static constexpr int LF = EPB * NFACES;  // local faces per block
  static constexpr int FN_ITEMS = LF * NF;
  static constexpr int FQ_ITEMS = LF * NQ;
  static constexpr int FW_ITEMS = LF * NW;
  static constexpr int FQ_ROUNDS = (FQ_ITEMS + BLOCK - 1) / BLOCK;
};

// compact LDS copy of the 1-D tables of one order
template <class C>
struct Tab {
  double x[C::N1], w[C::N1], iw[C::N1], D[C::N1 * C::N1], b0[C::N1], b1[C::N1];
  double xq[C::Q1], wq[C::Q1], B[C::Q1 * C::N1];
};

struct MeshDev {
  int ne;
  int64_t ndofs;
  const double *verts;         // [ne][NV][DIM] lexicographic corners
  const int32_t *face_nbr;     // [ne*NFACES]
  const uint8_t *face_orient;  // [ne*NFACES]
  const Tables1D *tables;      // device copy
};

// ---------------------------------------------------------------------------------------------
// index helpers
template <class C>
__device__ inline int stride_of(int d) {
  return d == 0 ? 1 : (d == 1 ? C::N1 : C::N1 * C::N1);
}
template <class C>
__device__ inline void tangential(int d, int &a, int &b) {
  if (C::DIM == 2) {
    a = 1 - d;
    b = -1;
  } else {
    a = (d == 0) ? 1 : 0;
    b = (d == 2) ? 1 : 2;
  }
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

// a face item decoded from a flat id: local face lf = le*NFACES + f and index inside the face
struct FaceItem {
  int le, f, lf, idx;
};
template <class C>
__device__ inline FaceItem face_item(int item, int per) {
  FaceItem r;
  r.lf = item / per;
  r.idx = item - r.lf * per;
  r.le = r.lf / C::NFACES;
  r.f = r.lf - r.le * C::NFACES;
  return r;
}

// Area-weighted outward normal at face quadrature point q of local face f, from the face's own
// corner coordinates: on a multilinear element the two tangents of a face are linear in the other
// tangential coordinate, so n = +-(dx/dta x dx/dtb) costs two lerps and a cross product
// (CalcOrtho of the face Jacobian, src/face_integrator.cpp:323).  X = position (2-D only).
template <class C>
__device__ inline void face_geometry(const double *V, const Tab<C> &tab, int f, int q, double *n, double &wq,
                                     double *X) {
  const int d = f >> 1, s = f & 1;
  if (C::DIM == 2) {
    const int a = 1 - d;
    const int c0 = (s << d), c1 = (s << d) | (1 << a);
    const double tx = V[c1 * 2 + 0] - V[c0 * 2 + 0], ty = V[c1 * 2 + 1] - V[c0 * 2 + 1];
    const double sg = (s ? 1.0 : -1.0) * (d == 0 ? 1.0 : -1.0);
    n[0] = sg * ty;
    n[1] = -sg * tx;
    wq = tab.wq[q];
    const double t = tab.xq[q];
    X[0] = V[c0 * 2 + 0] + t * tx;
    X[1] = V[c0 * 2 + 1] + t * ty;
  } else {
    const int a = (d == 0) ? 1 : 0, b = (d == 2) ? 1 : 2;
    const int qa = q % C::Q1, qb = q / C::Q1;
    const double ta = tab.xq[qa], tb = tab.xq[qb];
    wq = tab.wq[qa] * tab.wq[qb];
    const int c00 = (s << d), c10 = c00 | (1 << a), c01 = c00 | (1 << b), c11 = c10 | (1 << b);
    double va[3], vb[3];
#pragma unroll
    for (int i = 0; i < 3; i++) {
      const double p00 = V[c00 * 3 + i], p10 = V[c10 * 3 + i], p01 = V[c01 * 3 + i], p11 = V[c11 * 3 + i];
      const double ea0 = p10 - p00, ea1 = p11 - p01, eb0 = p01 - p00, eb1 = p11 - p10;
      va[i] = ea0 + tb * (ea1 - ea0);
      vb[i] = eb0 + ta * (eb1 - eb0);
    }
    const double sg = (s ? 1.0 : -1.0) * (d == 1 ? -1.0 : 1.0);
    n[0] = sg * (va[1] * vb[2] - va[2] * vb[1]);
    n[1] = sg * (va[2] * vb[0] - va[0] * vb[2]);
    n[2] = sg * (va[0] * vb[1] - va[1] * vb[0]);
  }
}

// trace at face node (f, fn) of nodal field F (LDS, NPE values of one element): sum_i b_s(i) F[...]
template <class C>
__device__ inline double face_trace(const double *F, const Tab<C> &tab, int f, int fn) {
  const int d = f >> 1, s = f & 1;
  int a, b;
  tangential<C>(d, a, b);
  int base;
  if (C::DIM == 2) {
    base = fn * stride_of<C>(a);
  } else {
    base = (fn % C::N1) * stride_of<C>(a) + (fn / C::N1) * stride_of<C>(b);
  }
  const int sd = stride_of<C>(d);
  const double *bs = s ? tab.b1 : tab.b0;
  double acc = 0.0;
#pragma unroll
  for (int i = 0; i < C::N1; i++) acc += bs[i] * ldsr(&F[base + i * sd]);
  return acc;
}

// ---- two-stage face interpolation: T[fld][FN_ITEMS] -> W[fld][FW_ITEMS] -> value at a quadrature point
// stage 1 (3-D): W[lf][qa + Q1*jb] = sum_ja B[qa][ja] T[lf][ja + N1*jb]
template <class C, int NFLD>
__device__ inline void interp_stage1(const double *T, double *W, const Tab<C> &tab, int tid) {
  if (C::DIM == 2) return;
  for (int item = tid; item < C::FW_ITEMS; item += C::BLOCK) {
    const int lf = item / C::NW, r = item - lf * C::NW;
    const int jb = r / C::Q1, qa = r - jb * C::Q1;
    const double *Bq = &tab.B[qa * C::N1];
    const double *t = T + lf * C::NF + C::N1 * jb;
#pragma unroll
    for (int fld = 0; fld < NFLD; fld++) {
      double acc = 0.0;
#pragma unroll
      for (int ja = 0; ja < C::N1; ja++) acc += Bq[ja] * ldsr(&t[fld * C::FN_ITEMS + ja]);
      W[fld * C::FW_ITEMS + item] = acc;
    }
  }
}
// stage 2: value of field `fld` at quadrature point (lf, q).  3-D reads W, 2-D reads T directly.
template <class C>
__device__ inline double interp_stage2(const double *T, const double *W, const Tab<C> &tab, int lf, int q) {
  double acc = 0.0;
  if (C::DIM == 2) {
#pragma unroll
    for (int a = 0; a < C::N1; a++) acc += tab.B[q * C::N1 + a] * ldsr(&T[lf * C::NF + a]);
  } else {
    const int qa = q % C::Q1, qb = q / C::Q1;
#pragma unroll
    for (int jb = 0; jb < C::N1; jb++) acc += tab.B[qb * C::N1 + jb] * ldsr(&W[lf * C::NW + qa + C::Q1 * jb]);
  }
  return acc;
}
// ---- two-stage projection (transpose): R[fld][FQ_ITEMS] -> W2[fld][FW_ITEMS] -> L[fld][FN_ITEMS]
// stage 1 (3-D): W2[lf][ja + N1*qb] = sum_qa B[qa][ja] R[lf][qa + Q1*qb]
template <class C, int NFLD>
__device__ inline void project_stage1(const double *R, double *W2, const Tab<C> &tab, int tid) {
  if (C::DIM == 2) return;
  for (int item = tid; item < C::FW_ITEMS; item += C::BLOCK) {
    const int lf = item / C::NW, r = item - lf * C::NW;
    const int qb = r / C::N1, ja = r - qb * C::N1;
    const double *rr = R + lf * C::NQ + C::Q1 * qb;
#pragma unroll
    for (int fld = 0; fld < NFLD; fld++) {
      double acc = 0.0;
#pragma unroll
      for (int qa = 0; qa < C::Q1; qa++) acc += tab.B[qa * C::N1 + ja] * ldsr(&rr[fld * C::FQ_ITEMS + qa]);
      W2[fld * C::FW_ITEMS + item] = acc;
    }
  }
}
template <class C, int NFLD>
__device__ inline void project_stage2(const double *R, const double *W2, double *L, const Tab<C> &tab, int tid) {
  for (int item = tid; item < C::FN_ITEMS; item += C::BLOCK) {
    const int lf = item / C::NF, fn = item - lf * C::NF;
    if (C::DIM == 2) {
#pragma unroll
      for (int fld = 0; fld < NFLD; fld++) {
        double acc = 0.0;
#pragma unroll
        for (int q = 0; q < C::Q1; q++) acc += tab.B[q * C::N1 + fn] * ldsr(&R[fld * C::FQ_ITEMS + lf * C::NQ + q]);
        L[fld * C::FN_ITEMS + item] = acc;
      }
    } else {
      const int ja = fn % C::N1, jb = fn / C::N1;
#pragma unroll
      for (int fld = 0; fld < NFLD; fld++) {
        double acc = 0.0;
#pragma unroll
        for (int qb = 0; qb < C::Q1; qb++)
          acc += tab.B[qb * C::N1 + jb] * ldsr(&W2[fld * C::FW_ITEMS + lf * C::NW + ja + C::N1 * qb]);
        L[fld * C::FN_ITEMS + item] = acc;
      }
    }
  }
}
// lifting to a volume node: sum over the element's faces of b_s(idx_d) L[f][fn(node)]
template <class C>
__device__ inline double face_lift(const double *L, const Tab<C> &tab, const int *idx) {
  double acc = 0.0;
#pragma unroll
  for (int d = 0; d < C::DIM; d++) {
    int a, b;
    tangential<C>(d, a, b);
    const int fn = (C::DIM == 2) ? idx[a] : idx[a] + C::N1 * idx[b];
    acc += tab.b0[idx[d]] * ldsr(&L[(2 * d) * C::NF + fn]) + tab.b1[idx[d]] * ldsr(&L[(2 * d + 1) * C::NF + fn]);
  }
  return acc;
}

template <class C>
__device__ inline void load_tables(Tab<C> &t, const Tables1D *src) {
  const int tid = threadIdx.x;
  if (tid < C::N1) {
    t.x[tid] = src->x[tid];
    t.w[tid] = src->w[tid];
    t.iw[tid] = 1.0 / src->w[tid];
    t.b0[tid] = src->b0[tid];
    t.b1[tid] = src->b1[tid];
  }
  if (tid < C::Q1) {
    t.xq[tid] = src->xq[tid];
    t.wq[tid] = src->wq[tid];
  }
  for (int i = tid; i < C::N1 * C::N1; i += C::BLOCK) t.D[i] = src->D[i];
  for (int i = tid; i < C::Q1 * C::N1; i += C::BLOCK) t.B[i] = src->B[i];
}
template <class C>
__device__ inline void load_vertices(double *sV, const MeshDev &m, int e0) {
  constexpr int PER = C::NV * C::DIM;
  for (int i = threadIdx.x; i < C::EPB * PER; i += C::BLOCK) {
    const int le = i / PER;
    if (e0 + le < m.ne) sV[i] = m.verts[static_cast<int64_t>(e0) * PER + i];
  }
}

// =============================================================================================
// sweep 0: traces of U and Up at the face nodes.  TA[slot][2*NEQ][NF], slot = e*NFACES + f;
// fields 0..NEQ-1 = U, NEQ..2NEQ-1 = Up
// =============================================================================================
template <class C, class PH>
__global__ __launch_bounds__(C::BLOCK) void k_traces(MeshDev m, typename PH::Params prm, const double *__restrict__ U,
                                                     double *__restrict__ Upout, double *__restrict__ TA) {
  constexpr int NEQ = PH::NEQ;
  __shared__ Tab<C> tab;
  __shared__ double sF[2 * NEQ][C::NODES];
  load_tables<C>(tab, m.tables);
  const int tid = threadIdx.x;
  const int e0 = blockIdx.x * C::EPB;
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
        sF[eq][tid] = u[eq];
        sF[NEQ + eq][tid] = up[eq];
        field_ptr(Upout, eq, m.ndofs)[n] = up[eq];
      }
    }
  }
  __syncthreads();
  for (int item = tid; item < C::FN_ITEMS; item += C::BLOCK) {
    const FaceItem it = face_item<C>(item, C::NF);
    const int e = e0 + it.le;
    if (e >= m.ne) continue;
    double *out = TA + (static_cast<int64_t>(e) * C::NFACES + it.f) * (2 * NEQ * C::NF) + it.idx;
#pragma unroll
    for (int fld = 0; fld < 2 * NEQ; fld++) out[fld * C::NF] = face_trace<C>(&sF[fld][it.le * C::NPE], tab, it.f, it.idx);
  }
}

// =============================================================================================
// sweep 1: gradient of the primitives (BR1-type: volume derivative + face jump lifting, diagonal
// inverse mass) and the viscous normal-flux traces TB[slot][NEQ][NQ]:
//   interior / shared face: F_v(U_q, gradUp_q) . n_out   (the consumer forms -1/2 (own - neighbour))
//   boundary face:          the complete additive viscous term of the boundary flux
// =============================================================================================
template <class C, class PH>
struct GradLds {
  static constexpr int NEQ = PH::NEQ, DIM = C::DIM;
  // chunk of fields interpolated together in the viscous phase
  static constexpr int CH = NEQ;
  static constexpr int R0 = NEQ * C::NODES;                                           // sU
  static constexpr int R1 = cmax(NEQ * DIM * C::NODES, NEQ * C::FQ_ITEMS);            // sG | sQ
  static constexpr int R2 = cmax(cmax(NEQ * C::NODES, NEQ * C::FW_ITEMS), CH * C::FN_ITEMS);  // sUp | W | T chunk
  static constexpr int R3 = cmax(NEQ * C::FN_ITEMS, CH * C::FW_ITEMS);                // sFN | sL | W chunk
  static constexpr int O0 = 0, O1 = R0, O2 = R0 + R1, O3 = R0 + R1 + R2;
  static constexpr int TOTAL = R0 + R1 + R2 + R3;
};

template <class C, class PH>
__global__ __launch_bounds__(C::BLOCK, C::MINW) void k_gradient(MeshDev m, typename PH::Params prm,
                                                       const double *__restrict__ U, const double *__restrict__ TA,
                                                       double *__restrict__ gradUp, double *__restrict__ TB) {
  constexpr int NEQ = PH::NEQ, DIM = C::DIM;
  typedef GradLds<C, PH> L;
  __shared__ Tab<C> tab;
  __shared__ double sV[C::EPB * C::NV * DIM];
  __shared__ double pool[L::TOTAL];
  double *sU = pool + L::O0;   // [NEQ][NODES]
  double *sG = pool + L::O1;   // [NEQ*DIM][NODES]      (after the jump phase)
  double *sQ = pool + L::O1;   // [NEQ][FQ_ITEMS]       (jump phase)
  double *sUp = pool + L::O2;  // [NEQ][NODES]
  double *sW = pool + L::O2;   // [NEQ][FW_ITEMS]       (after sUp is dead)
  double *sT = pool + L::O2;   // [CH][FN_ITEMS]        (viscous phase)
  double *sFN = pool + L::O3;  // [NEQ][FN_ITEMS]
  double *sL = pool + L::O3;   // [NEQ][FN_ITEMS]
  double *sWc = pool + L::O3;  // [CH][FW_ITEMS]        (viscous phase)

  load_tables<C>(tab, m.tables);
  const int tid = threadIdx.x;
  const int e0 = blockIdx.x * C::EPB;
  load_vertices<C>(sV, m, e0);
  const bool node_on = tid < C::NODES && (e0 + tid / C::NPE) < m.ne;
  const int le_n = tid / C::NPE, nd = tid - le_n * C::NPE;
  int idx[3] = {nd % C::N1, (nd / C::N1) % C::N1, nd / (C::N1 * C::N1)};
  if (node_on) {
    const unsigned n = static_cast<unsigned>(e0 + le_n) * C::NPE + nd;
    double u[NEQ], up[NEQ];
#pragma unroll
    for (int eq = 0; eq < NEQ; eq++) u[eq] = field_ptr(U, eq, m.ndofs)[n];
    PH::prim(prm, u, up);
#pragma unroll
    for (int eq = 0; eq < NEQ; eq++) {
      sU[eq * C::NODES + tid] = u[eq];
      sUp[eq * C::NODES + tid] = up[eq];
    }
  }
  __syncthreads();

  // ---- volume part: collocation derivative (Ke then M^-1 of the reference collapse to it)
  double g[NEQ * DIM];  // g[eq + d*NEQ]
  double inv_mass = 0.0;
  if (node_on) {
    double xi[DIM], J[DIM * DIM], A[DIM * DIM];
    double iwn = 1.0;
#pragma unroll
    for (int d = 0; d < DIM; d++) {
      xi[d] = tab.x[idx[d]];
      iwn *= tab.iw[idx[d]];
    }
    jacobian<DIM>(&sV[le_n * C::NV * DIM], xi, J);
    const double det = adjugate<DIM>(J, A);
    const double idet = 1.0 / det;
    inv_mass = iwn * idet;
#pragma unroll
    for (int eq = 0; eq < NEQ; eq++) {
      double dr[DIM];
#pragma unroll
      for (int mm = 0; mm < DIM; mm++) {
        const int sd = stride_of<C>(mm);
        const double *F = &sUp[eq * C::NODES + le_n * C::NPE + nd - idx[mm] * sd];
        double acc = 0.0;
#pragma unroll
        for (int a = 0; a < C::N1; a++) acc += tab.D[idx[mm] * C::N1 + a] * ldsr(&F[a * sd]);
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

  // ---- face part: half jump of the primitives at the face nodes (own trace on boundary faces)
  for (int item = tid; item < C::FN_ITEMS; item += C::BLOCK) {
    const FaceItem it = face_item<C>(item, C::NF);
    const int e = e0 + it.le;
    if (e >= m.ne) continue;
    const int slot = e * C::NFACES + it.f;
    const int nb = m.face_nbr[slot];
    double u1[NEQ];
#pragma unroll
    for (int eq = 0; eq < NEQ; eq++) u1[eq] = face_trace<C>(&sUp[eq * C::NODES + it.le * C::NPE], tab, it.f, it.idx);
    if (nb >= 0) {
      const int o = m.face_orient[slot];
      const int pn = permute<DIM>(o, C::N1, it.idx % C::N1, it.idx / C::N1);
      const double *src = TA + static_cast<int64_t>(nb) * (2 * NEQ * C::NF) + NEQ * C::NF + pn;
#pragma unroll
      for (int eq = 0; eq < NEQ; eq++) sFN[eq * C::FN_ITEMS + item] = 0.5 * (src[eq * C::NF] - u1[eq]);
    } else {
      // boundary: u2 = u1 (src/faceGradientIntegration.cpp:113-115); the wall ghost of useBCinGrad
      // is not polynomial in the face nodes and is applied at the quadrature points below
#pragma unroll
      for (int eq = 0; eq < NEQ; eq++) sFN[eq * C::FN_ITEMS + item] = u1[eq];
    }
  }
  __syncthreads();  // sFN complete, sUp dead
  interp_stage1<C, NEQ>(sFN, sW, tab, tid);
  __syncthreads();

  // quadrature-point values of the half jump and the weighted normals, in registers per round
  double jq[C::FQ_ROUNDS][NEQ];
  double nw[C::FQ_ROUNDS][DIM];
#pragma unroll
  for (int r = 0; r < C::FQ_ROUNDS; r++) {
    const int item = tid + r * C::BLOCK;
#pragma unroll
    for (int eq = 0; eq < NEQ; eq++) jq[r][eq] = 0.0;
#pragma unroll
    for (int d = 0; d < DIM; d++) nw[r][d] = 0.0;
    if (item < C::FQ_ITEMS) {
      const FaceItem it = face_item<C>(item, C::NQ);
      const int e = e0 + it.le;
      if (e < m.ne) {
        const int nb = m.face_nbr[e * C::NFACES + it.f];
        double n[DIM], wq, X[DIM];
        face_geometry<C>(&sV[it.le * C::NV * DIM], tab, it.f, it.idx, n, wq, X);
#pragma unroll
        for (int d = 0; d < DIM; d++) nw[r][d] = n[d] * wq;
        if (nb >= 0) {
#pragma unroll
          for (int eq = 0; eq < NEQ; eq++)
            jq[r][eq] = interp_stage2<C>(sFN + eq * C::FN_ITEMS, sW + eq * C::FW_ITEMS, tab, it.lf, it.idx);
        } else if (prm.use_bc_in_grad) {
          double u1[NEQ], u2[NEQ];
#pragma unroll
          for (int eq = 0; eq < NEQ; eq++)
            u1[eq] = interp_stage2<C>(sFN + eq * C::FN_ITEMS, sW + eq * C::FW_ITEMS, tab, it.lf, it.idx);
          PH::bc_grad_prim(prm, prm.bc[-nb - 1], u1, u2);
#pragma unroll
          for (int eq = 0; eq < NEQ; eq++) jq[r][eq] = 0.5 * (u2[eq] - u1[eq]);
        }
      }
    }
  }
  __syncthreads();  // sW, sFN dead
#pragma unroll
  for (int d = 0; d < DIM; d++) {
#pragma unroll
    for (int r = 0; r < C::FQ_ROUNDS; r++) {
      const int item = tid + r * C::BLOCK;
      if (item < C::FQ_ITEMS) {
#pragma unroll
        for (int eq = 0; eq < NEQ; eq++) sQ[eq * C::FQ_ITEMS + item] = jq[r][eq] * nw[r][d];
      }
    }
    __syncthreads();
    project_stage1<C, NEQ>(sQ, sW, tab, tid);
    if (DIM == 3) __syncthreads();
    project_stage2<C, NEQ>(sQ, sW, sL, tab, tid);
    __syncthreads();
    if (node_on) {
#pragma unroll
      for (int eq = 0; eq < NEQ; eq++)
        g[eq + d * NEQ] += inv_mass * face_lift<C>(&sL[eq * C::FN_ITEMS + le_n * C::NFACES * C::NF], tab, idx);
    }
    // the next direction overwrites sQ (read by project_stage1/2 above, complete at the last
    // barrier) and sW/sL (read before the barriers that precede their next writes)
  }
  __syncthreads();  // all reads of sQ done before sG (same region) is written

  if (node_on) {
    const unsigned n = static_cast<unsigned>(e0 + le_n) * C::NPE + nd;
#pragma unroll
    for (int d = 0; d < DIM; d++)
#pragma unroll
      for (int eq = 0; eq < NEQ; eq++) {
        field_ptr(gradUp, eq + d * NEQ, m.ndofs)[n] = g[eq + d * NEQ];
        sG[(eq + d * NEQ) * C::NODES + tid] = g[eq + d * NEQ];
      }
  }
  __syncthreads();

  // ---- viscous normal-flux traces: U and gradUp at the face quadrature points, NEQ fields at a time
  double uq[C::FQ_ROUNDS][NEQ], gq[C::FQ_ROUNDS][NEQ * DIM];
#pragma unroll
  for (int c = 0; c < 1 + DIM; c++) {
    const double *src = (c == 0) ? sU : sG + (c - 1) * NEQ * C::NODES;
    for (int item = tid; item < C::FN_ITEMS; item += C::BLOCK) {
      const FaceItem it = face_item<C>(item, C::NF);
      if (e0 + it.le >= m.ne) continue;
#pragma unroll
      for (int k = 0; k < NEQ; k++)
        sT[k * C::FN_ITEMS + item] = face_trace<C>(&src[k * C::NODES + it.le * C::NPE], tab, it.f, it.idx);
    }
    __syncthreads();
    interp_stage1<C, NEQ>(sT, sWc, tab, tid);
    if (DIM == 3) __syncthreads();
#pragma unroll
    for (int r = 0; r < C::FQ_ROUNDS; r++) {
      const int item = tid + r * C::BLOCK;
      if (item < C::FQ_ITEMS) {
        const int lf = item / C::NQ, q = item - lf * C::NQ;
#pragma unroll
        for (int k = 0; k < NEQ; k++) {
          const double v = interp_stage2<C>(sT + k * C::FN_ITEMS, sWc + k * C::FW_ITEMS, tab, lf, q);
          if (c == 0)
            uq[r][k] = v;
          else
            gq[r][k + (c - 1) * NEQ] = v;
        }
      }
    }
    __syncthreads();
  }
#pragma unroll
  for (int r = 0; r < C::FQ_ROUNDS; r++) {
    const int item = tid + r * C::BLOCK;
    if (item >= C::FQ_ITEMS) continue;
    const FaceItem it = face_item<C>(item, C::NQ);
    const int e = e0 + it.le;
    if (e >= m.ne) continue;
    const int slot = e * C::NFACES + it.f;
    const int nb = m.face_nbr[slot];
    double fn[NEQ];
    PH::clamp_species(uq[r]);
    double n[DIM], wq, X[DIM];
    face_geometry<C>(&sV[it.le * C::NV * DIM], tab, it.f, it.idx, n, wq, X);
    if (nb >= 0) {
      PH::visc_flux_n(prm, uq[r], gq[r], n, fn);
    } else {
      PH::bc_visc_term(prm, prm.bc[-nb - 1], uq[r], gq[r], n, fn);
    }
    double *out = TB + static_cast<int64_t>(slot) * (NEQ * C::NQ) + it.idx;
#pragma unroll
    for (int eq = 0; eq < NEQ; eq++) out[eq * C::NQ] = fn[eq];
  }
}

// =============================================================================================
// sweep 2: y = M^-1 [ (grad phi, F_c - F_v)  -  <phi, F^ . n> ]  (+ point sources)
// =============================================================================================
template <class C, class PH>
struct FluxLds {
  static constexpr int NEQ = PH::NEQ, DIM = C::DIM;
  // region A: sU + sGf, later W / W2 + sL ; region B: sT1 + sT2, later sQ
  static constexpr int A_W = NEQ * C::FW_ITEMS;
  static constexpr int RA = cmax(NEQ * C::NODES + NEQ * DIM * C::NODES, A_W + NEQ * C::FN_ITEMS);
  static constexpr int RB = cmax(2 * NEQ * C::FN_ITEMS, NEQ * C::FQ_ITEMS);
  static constexpr int TOTAL = RA + RB;
};

template <class C, class PH>
__global__ __launch_bounds__(C::BLOCK, C::MINW) void k_flux(MeshDev m, typename PH::Params prm, const double *__restrict__ U,
                                                   const double *__restrict__ gradUp, const double *__restrict__ TA,
                                                   const double *__restrict__ TB, double *__restrict__ Y,
                                                   unsigned long long *__restrict__ max_speed_bits) {
  constexpr int NEQ = PH::NEQ, DIM = C::DIM;
  typedef FluxLds<C, PH> L;
  __shared__ Tab<C> tab;
  __shared__ double sV[C::EPB * C::NV * DIM];
  __shared__ double pool[L::TOTAL];
  double *sU = pool;                                // [NEQ][NODES]
  double *sGf = pool + NEQ * C::NODES;              // [NEQ*DIM][NODES] contravariant nodal flux
  double *sW = pool;                                // [NEQ][FW_ITEMS]   (after z)
  double *sL = pool + L::A_W;                       // [NEQ][FN_ITEMS]
  double *sT1 = pool + L::RA;                       // [NEQ][FN_ITEMS] own traces of U
  double *sT2 = pool + L::RA + NEQ * C::FN_ITEMS;   // [NEQ][FN_ITEMS] neighbour traces in my frame
  double *sQ = pool + L::RA;                        // [NEQ][FQ_ITEMS]   (after the interpolation)

  load_tables<C>(tab, m.tables);
  const int tid = threadIdx.x;
  const int e0 = blockIdx.x * C::EPB;
  load_vertices<C>(sV, m, e0);
  const bool node_on = tid < C::NODES && (e0 + tid / C::NPE) < m.ne;
  const int le_n = tid / C::NPE, nd = tid - le_n * C::NPE;
  int idx[3] = {nd % C::N1, (nd / C::N1) % C::N1, nd / (C::N1 * C::N1)};
  double u[NEQ], gr[NEQ * DIM];
  if (node_on) {
    const unsigned n = static_cast<unsigned>(e0 + le_n) * C::NPE + nd;
#pragma unroll
    for (int eq = 0; eq < NEQ; eq++) {
      u[eq] = field_ptr(U, eq, m.ndofs)[n];
      sU[eq * C::NODES + tid] = u[eq];
    }
#pragma unroll
    for (int d = 0; d < DIM; d++)
#pragma unroll
      for (int eq = 0; eq < NEQ; eq++) gr[eq + d * NEQ] = field_ptr(gradUp, eq + d * NEQ, m.ndofs)[n];
  }
  __syncthreads();  // tables + vertices + sU

  // ---- nodal flux F_c - F_v (src/rhs_operator.cpp:493-559), contravariant components
  double inv_mass = 0.0, speed = 0.0;
  double src[NEQ];
#pragma unroll
  for (int eq = 0; eq < NEQ; eq++) src[eq] = 0.0;
  if (node_on) {
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
    inv_mass = iwn / det;
    double uc[NEQ];
#pragma unroll
    for (int eq = 0; eq < NEQ; eq++) uc[eq] = u[eq];
    PH::clamp_species(uc);
    const typename PH::State st = PH::make_state(prm, uc);
    speed = PH::max_char_speed(prm, uc, st);
    if (PH::HAS_SOURCE) {
      double up[NEQ];
      PH::prim(prm, u, up);
      PH::source(prm, u, up, gr, src);
    }
    if (PH::HAS_FLUX_DOT) {
      // contravariant flux, one metric row at a time (keeps ~35 doubles live instead of ~70)
      const typename PH::Transport tr = PH::transport(prm, st);
      double divV = 0.0;
#pragma unroll
      for (int i = 0; i < DIM; i++) divV += gr[(1 + i) + i * NEQ];
#pragma unroll
      for (int mm = 0; mm < DIM; mm++) {
        double a[DIM], Fa[NEQ];
#pragma unroll
        for (int d = 0; d < DIM; d++) a[d] = wn * A[mm + d * DIM];
        PH::total_flux_dot(prm, uc, st, tr, divV, gr, a, Fa);
#pragma unroll
        for (int eq = 0; eq < NEQ; eq++) sGf[(eq + mm * NEQ) * C::NODES + tid] = Fa[eq];
        __builtin_amdgcn_sched_barrier(0);  // one metric row at a time: keeps the register peak low
      }
    } else {
      double F[NEQ * DIM], Fv[NEQ * DIM];
      PH::conv_flux(prm, uc, st, F);
      PH::visc_flux(prm, uc, st, gr, Fv);
#pragma unroll
      for (int eq = 0; eq < NEQ; eq++)
#pragma unroll
        for (int mm = 0; mm < DIM; mm++) {
          double s = 0.0;
#pragma unroll
          for (int d = 0; d < DIM; d++) s += A[mm + d * DIM] * (F[eq + d * NEQ] - Fv[eq + d * NEQ]);
          sGf[(eq + mm * NEQ) * C::NODES + tid] = wn * s;
        }
    }
  }
  // max |u|+c over the wave -> global (positive doubles order like their bit patterns)
  {
    double v = speed;
#pragma unroll
    for (int off = 32; off > 0; off >>= 1) v = fmax(v, __shfl_xor(v, off));
    if ((tid & 63) == 0 && v > 0.0) atomicMax(max_speed_bits, static_cast<unsigned long long>(__double_as_longlong(v)));
  }

  // ---- face traces of U: own and neighbour
  for (int item = tid; item < C::FN_ITEMS; item += C::BLOCK) {
    const FaceItem it = face_item<C>(item, C::NF);
    const int e = e0 + it.le;
    if (e >= m.ne) continue;
    const int slot = e * C::NFACES + it.f;
    const int nb = m.face_nbr[slot];
#pragma unroll
    for (int eq = 0; eq < NEQ; eq++)
      sT1[eq * C::FN_ITEMS + item] = face_trace<C>(&sU[eq * C::NODES + it.le * C::NPE], tab, it.f, it.idx);
    if (nb >= 0) {
      const int o = m.face_orient[slot];
      const int pn = permute<DIM>(o, C::N1, it.idx % C::N1, it.idx / C::N1);
      const double *s2 = TA + static_cast<int64_t>(nb) * (2 * NEQ * C::NF) + pn;
#pragma unroll
      for (int eq = 0; eq < NEQ; eq++) sT2[eq * C::FN_ITEMS + item] = (TPSRHS_ABLATE & 1) ? 1.0 : s2[eq * C::NF];
    }
  }
  __syncthreads();  // sGf, sT1, sT2 complete

  // ---- volume term: z_j = sum_m sum_a D[a][j_m] Ghat_m(a)   (src/domain_integrator.cpp:45-99)
  double z[NEQ];
  if (node_on) {
#pragma unroll
    for (int eq = 0; eq < NEQ; eq++) {
      double acc = 0.0;
#pragma unroll
      for (int mm = 0; mm < DIM; mm++) {
        const int sd = stride_of<C>(mm);
        const double *F = &sGf[(eq + mm * NEQ) * C::NODES + le_n * C::NPE + nd - idx[mm] * sd];
#pragma unroll
        for (int a = 0; a < C::N1; a++) acc += tab.D[a * C::N1 + idx[mm]] * ldsr(&F[a * sd]);
      }
      z[eq] = acc;
    }
  }
  if (DIM == 3) __syncthreads();  // sU/sGf dead: region A becomes W

  // ---- U at the face quadrature points, both sides
  double u1[C::FQ_ROUNDS][NEQ], u2[C::FQ_ROUNDS][NEQ];
#pragma unroll
  for (int side = 0; side < 2; side++) {
    const double *T = side ? sT2 : sT1;
    interp_stage1<C, NEQ>(T, sW, tab, tid);
    if (DIM == 3) __syncthreads();
#pragma unroll
    for (int r = 0; r < C::FQ_ROUNDS; r++) {
      const int item = tid + r * C::BLOCK;
      if (item < C::FQ_ITEMS) {
        const int lf = item / C::NQ, q = item - lf * C::NQ;
#pragma unroll
        for (int eq = 0; eq < NEQ; eq++) {
          const double v = interp_stage2<C>(T + eq * C::FN_ITEMS, sW + eq * C::FW_ITEMS, tab, lf, q);
          if (side)
            u2[r][eq] = v;
          else
            u1[r][eq] = v;
        }
      }
    }
    __syncthreads();
  }

  // ---- numerical flux at the face quadrature points (sT1/sT2 dead: region B becomes sQ)
#pragma unroll
  for (int r = 0; r < C::FQ_ROUNDS; r++) {
    const int item = tid + r * C::BLOCK;
    if (item >= C::FQ_ITEMS) continue;
    const FaceItem it = face_item<C>(item, C::NQ);
    const int e = e0 + it.le;
    if (e >= m.ne) continue;
    const int slot = e * C::NFACES + it.f;
    const int nb = m.face_nbr[slot];
    double fh[NEQ];
    PH::clamp_species(u1[r]);
    double n[DIM], wq, X[DIM];
    face_geometry<C>(&sV[it.le * C::NV * DIM], tab, it.f, it.idx, n, wq, X);
    const double *tb_own = TB + static_cast<int64_t>(slot) * (NEQ * C::NQ) + it.idx;
    if (TPSRHS_ABLATE & 4) {
#pragma unroll
      for (int eq = 0; eq < NEQ; eq++) fh[eq] = u1[r][eq] + u2[r][eq] * n[0];
    } else if (nb >= 0) {
      PH::clamp_species(u2[r]);
      PH::lax_friedrichs(prm, u1[r], u2[r], n, fh);
      const int o = m.face_orient[slot];
      const int pq = permute<DIM>(o, C::Q1, it.idx % C::Q1, it.idx / C::Q1);
      const double *tb_nb = TB + static_cast<int64_t>(nb) * (NEQ * C::NQ) + pq;
      if (!(TPSRHS_ABLATE & 1)) {
#pragma unroll
        for (int eq = 0; eq < NEQ; eq++) fh[eq] -= 0.5 * (tb_own[eq * C::NQ] - tb_nb[eq * C::NQ]);
      }
    } else {
      double ug[NEQ];
      PH::bc_ghost(prm, prm.bc[-nb - 1], u1[r], n, ug);
      PH::lax_friedrichs(prm, u1[r], ug, n, fh);
#pragma unroll
      for (int eq = 0; eq < NEQ; eq++) fh[eq] += tb_own[eq * C::NQ];
    }
#pragma unroll
    for (int eq = 0; eq < NEQ; eq++) sQ[eq * C::FQ_ITEMS + item] = fh[eq] * wq;
  }
  __syncthreads();
  project_stage1<C, NEQ>(sQ, sW, tab, tid);
  if (DIM == 3) __syncthreads();
  project_stage2<C, NEQ>(sQ, sW, sL, tab, tid);
  __syncthreads();
  if (node_on) {
    const unsigned n = static_cast<unsigned>(e0 + le_n) * C::NPE + nd;
#pragma unroll
    for (int eq = 0; eq < NEQ; eq++) {
      const double lift = face_lift<C>(&sL[eq * C::FN_ITEMS + le_n * C::NFACES * C::NF], tab, idx);
      field_ptr(Y, eq, m.ndofs)[n] = inv_mass * (z[eq] - lift) + src[eq];
    }
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
