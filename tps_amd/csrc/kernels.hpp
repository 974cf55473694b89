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
// A workgroup owns EPB whole elements (one lane per node), keeps their nodal fields in LDS, and
// applies the element-independent 1-D operators (differentiation matrix, end-point values, node ->
// face-quadrature interpolation) by sum factorisation out of a < 1 KB LDS table.  Geometry
// (Jacobians, area-weighted normals) is recomputed from the 2^dim vertex coordinates of the element.
#ifndef TPSRHS_KERNELS_HPP_
#define TPSRHS_KERNELS_HPP_

#include <hip/hip_runtime.h>

#include "basis.hpp"

namespace tpsrhs {

template <int DIM_, int P_>
struct Cfg {
  static constexpr int DIM = DIM_, P = P_, N1 = P_ + 1;
  static constexpr int NPE = (DIM_ == 3) ? N1 * N1 * N1 : N1 * N1;
  static constexpr int NF = (DIM_ == 3) ? N1 * N1 : N1;
  static constexpr int Q1 = ((DIM_ - 1) + 2 * P_) / 2 + 1;
  static constexpr int NQ = (DIM_ == 3) ? Q1 * Q1 : Q1;
  static constexpr int NFACES = 2 * DIM_;
  static constexpr int NV = 1 << DIM_;
  static constexpr int BLOCK = (NPE <= 64) ? 64 : ((NPE <= 128) ? 128 : 256);
  static constexpr int EPB = BLOCK / NPE;  // elements per block
  static constexpr int NODES = EPB * NPE;  // active lanes in node loops
  static constexpr int FN_ITEMS = EPB * NFACES * NF;
  static constexpr int FQ_ITEMS = EPB * NFACES * NQ;
  static constexpr int FQ_ROUNDS = (FQ_ITEMS + BLOCK - 1) / BLOCK;
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
// tangential axes of face direction d
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
// permutation of a tangential index pair under an orientation code (n points per direction,
// symmetric point sets): my (ia, ib) -> neighbour's flat index
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

// Jacobian J[i + m*DIM] = dx_i/dxi_m of the multilinear element at reference point xi; V = vertex
// coordinates [corner][DIM], lexicographic corners
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
template <int DIM>
__device__ inline void position(const double *V, const double *xi, double *X) {
  if (DIM == 2) {
    const double x = xi[0], y = xi[1];
#pragma unroll
    for (int i = 0; i < 2; i++)
      X[i] = (V[0 + i] * (1.0 - x) + V[2 + i] * x) * (1.0 - y) + (V[4 + i] * (1.0 - x) + V[6 + i] * x) * y;
  } else {
    const double x = xi[0], y = xi[1], z = xi[2];
#pragma unroll
    for (int i = 0; i < 3; i++) {
      const double a = (V[0 + i] * (1.0 - x) + V[3 + i] * x) * (1.0 - y) + (V[6 + i] * (1.0 - x) + V[9 + i] * x) * y;
      const double b = (V[12 + i] * (1.0 - x) + V[15 + i] * x) * (1.0 - y) + (V[18 + i] * (1.0 - x) + V[21 + i] * x) * y;
      X[i] = a * (1.0 - z) + b * z;
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
    // cofactors: A(m,i) = cof(J)(i,m)
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

// a face item (le, f, index-in-face) decoded from a flat item id; `per` = NF or NQ
struct FaceItem {
  int le, f, idx;
};
template <class C>
__device__ inline FaceItem face_item(int item, int per) {
  FaceItem r;
  r.le = item / (C::NFACES * per);
  const int rem = item - r.le * (C::NFACES * per);
  r.f = rem / per;
  r.idx = rem - r.f * per;
  return r;
}

// area-weighted outward normal n[DIM], quadrature weight and position at face quadrature point
// (f, q) of an element with vertices V
template <class C>
__device__ inline void face_geometry(const double *V, const Tables1D &tab, int f, int q, double *n, double &wq,
                                     double *X) {
  constexpr int DIM = C::DIM;
  const int d = f >> 1, s = f & 1;
  int a, b;
  tangential<C>(d, a, b);
  double xi[DIM];
  xi[d] = s;
  if (DIM == 2) {
    xi[a] = tab.xq[q];
    wq = tab.wq[q];
  } else {
    const int qa = q % C::Q1, qb = q / C::Q1;
    xi[a] = tab.xq[qa];
    xi[b] = tab.xq[qb];
    wq = tab.wq[qa] * tab.wq[qb];
  }
  double J[DIM * DIM], A[DIM * DIM];
  jacobian<DIM>(V, xi, J);
  adjugate<DIM>(J, A);
  const double sg = s ? 1.0 : -1.0;
#pragma unroll
  for (int i = 0; i < DIM; i++) n[i] = sg * A[d + i * DIM];
  position<DIM>(V, xi, X);
}

// trace at face node (f, fn) of nodal field F (LDS, NPE values of one element): sum_i b_s(i) F[...]
template <class C>
__device__ inline double face_trace(const double *F, const Tables1D &tab, int f, int fn) {
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
  for (int i = 0; i < C::N1; i++) acc += bs[i] * F[base + i * sd];
  return acc;
}

// value at face quadrature point q of a face-node field T[NF] (LDS): sum-factorised B (x) B
template <class C>
__device__ inline double face_interp(const double *T, const Tables1D &tab, int q) {
  if (C::DIM == 2) {
    double acc = 0.0;
#pragma unroll
    for (int a = 0; a < C::N1; a++) acc += tab.B[q * C::N1 + a] * T[a];
    return acc;
  } else {
    const int qa = q % C::Q1, qb = q / C::Q1;
    double acc = 0.0;
#pragma unroll
    for (int jb = 0; jb < C::N1; jb++) {
      double r = 0.0;
#pragma unroll
      for (int ja = 0; ja < C::N1; ja++) r += tab.B[qa * C::N1 + ja] * T[ja + C::N1 * jb];
      acc += tab.B[qb * C::N1 + jb] * r;
    }
    return acc;
  }
}
// transpose: face node fn <- quadrature values R[NQ] (LDS)
template <class C>
__device__ inline double face_project(const double *R, const Tables1D &tab, int fn) {
  if (C::DIM == 2) {
    double acc = 0.0;
#pragma unroll
    for (int q = 0; q < C::Q1; q++) acc += tab.B[q * C::N1 + fn] * R[q];
    return acc;
  } else {
    const int ja = fn % C::N1, jb = fn / C::N1;
    double acc = 0.0;
#pragma unroll
    for (int qb = 0; qb < C::Q1; qb++) {
      double r = 0.0;
#pragma unroll
      for (int qa = 0; qa < C::Q1; qa++) r += tab.B[qa * C::N1 + ja] * R[qa + C::Q1 * qb];
      acc += tab.B[qb * C::N1 + jb] * r;
    }
    return acc;
  }
}
// lifting to a volume node: sum over the element's faces of b_s(idx_d) L[f][fn(node)]
// L: LDS [NFACES][NF] of one element
template <class C>
__device__ inline double face_lift(const double *L, const Tables1D &tab, const int *idx) {
  double acc = 0.0;
#pragma unroll
  for (int d = 0; d < C::DIM; d++) {
    int a, b;
    tangential<C>(d, a, b);
    const int fn = (C::DIM == 2) ? idx[a] : idx[a] + C::N1 * idx[b];
    acc += tab.b0[idx[d]] * L[(2 * d) * C::NF + fn] + tab.b1[idx[d]] * L[(2 * d + 1) * C::NF + fn];
  }
  return acc;
}

template <class C>
__device__ inline void load_tables(Tables1D &dst, const Tables1D *src) {
  constexpr int NW = sizeof(Tables1D) / sizeof(double);
  double *d = reinterpret_cast<double *>(&dst);
  const double *s = reinterpret_cast<const double *>(src);
  for (int i = threadIdx.x; i < NW; i += C::BLOCK) d[i] = s[i];
}

// =============================================================================================
// sweep 0: traces of U and Up at the face nodes.  TA[slot][2*NEQ][NF], slot = e*NFACES + f;
// fields 0..NEQ-1 = U, NEQ..2NEQ-1 = Up
// =============================================================================================
template <class C, class PH>
__global__ __launch_bounds__(C::BLOCK) void k_traces(MeshDev m, typename PH::Params prm, const double *__restrict__ U,
                                                     double *__restrict__ Upout, double *__restrict__ TA) {
  constexpr int NEQ = PH::NEQ;
  __shared__ Tables1D tab;
  __shared__ double sF[2 * NEQ][C::NODES];
  load_tables<C>(tab, m.tables);
  const int tid = threadIdx.x;
  const int e0 = blockIdx.x * C::EPB;
  if (tid < C::NODES) {
    const int le = tid / C::NPE, nd = tid - le * C::NPE;
    const int e = e0 + le;
    if (e < m.ne) {
      const int64_t n = static_cast<int64_t>(e) * C::NPE + nd;
      double u[NEQ], up[NEQ];
#pragma unroll
      for (int eq = 0; eq < NEQ; eq++) u[eq] = U[n + eq * m.ndofs];
      PH::prim(prm, u, up);
#pragma unroll
      for (int eq = 0; eq < NEQ; eq++) {
        sF[eq][tid] = u[eq];
        sF[NEQ + eq][tid] = up[eq];
        Upout[n + eq * m.ndofs] = up[eq];
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
__global__ __launch_bounds__(C::BLOCK) void k_gradient(MeshDev m, typename PH::Params prm,
                                                       const double *__restrict__ U, const double *__restrict__ TA,
                                                       double *__restrict__ gradUp, double *__restrict__ TB) {
  constexpr int NEQ = PH::NEQ, DIM = C::DIM;
  constexpr int NFLD = NEQ + NEQ * DIM;  // U and gradUp, interpolated to the faces at the end
  __shared__ Tables1D tab;
  __shared__ double sV[C::EPB][C::NV * DIM];
  __shared__ double sU[NEQ][C::NODES];
  __shared__ double sG[NEQ * DIM][C::NODES];
  // scratch: phase 1 {sUp, sFN, sQ, sL}; phase 2 sT (face-node traces of U and gradUp)
  constexpr int SCR1 = NEQ * C::NODES + NEQ * C::FN_ITEMS + NEQ * C::FQ_ITEMS + NEQ * C::FN_ITEMS;
  constexpr int SCR2 = NFLD * C::FN_ITEMS;
  constexpr int SCR = SCR1 > SCR2 ? SCR1 : SCR2;
  __shared__ double scratch[SCR];
  double *sUp = scratch;                    // [NEQ][NODES]
  double *sFN = sUp + NEQ * C::NODES;       // [NEQ][FN_ITEMS]
  double *sQ = sFN + NEQ * C::FN_ITEMS;     // [NEQ][FQ_ITEMS]
  double *sL = sQ + NEQ * C::FQ_ITEMS;      // [NEQ][FN_ITEMS]
  double *sT = scratch;                     // [NFLD][FN_ITEMS]

  load_tables<C>(tab, m.tables);
  const int tid = threadIdx.x;
  const int e0 = blockIdx.x * C::EPB;
  for (int i = tid; i < C::EPB * C::NV * DIM; i += C::BLOCK) {
    const int le = i / (C::NV * DIM);
    if (e0 + le < m.ne) sV[le][i - le * (C::NV * DIM)] = m.verts[static_cast<int64_t>(e0) * C::NV * DIM + i];
  }
  const bool node_on = tid < C::NODES && (e0 + tid / C::NPE) < m.ne;
  const int le_n = tid / C::NPE, nd = tid - le_n * C::NPE;
  int idx[3] = {nd % C::N1, (nd / C::N1) % C::N1, nd / (C::N1 * C::N1)};
  if (node_on) {
    const int64_t n = static_cast<int64_t>(e0 + le_n) * C::NPE + nd;
    double u[NEQ], up[NEQ];
#pragma unroll
    for (int eq = 0; eq < NEQ; eq++) u[eq] = U[n + eq * m.ndofs];
    PH::prim(prm, u, up);
#pragma unroll
    for (int eq = 0; eq < NEQ; eq++) {
      sU[eq][tid] = u[eq];
      sUp[eq * C::NODES + tid] = up[eq];
    }
  }
  __syncthreads();

  // ---- volume part: collocation derivative, Ke then M^-1 of the reference collapse to it
  double g[NEQ * DIM];  // g[eq + d*NEQ]
  double inv_mass = 0.0;
  if (node_on) {
    double xi[DIM], J[DIM * DIM], A[DIM * DIM];
    double wn = 1.0;
#pragma unroll
    for (int d = 0; d < DIM; d++) {
      xi[d] = tab.x[idx[d]];
      wn *= tab.w[idx[d]];
    }
    jacobian<DIM>(sV[le_n], xi, J);
    const double det = adjugate<DIM>(J, A);
    const double idet = 1.0 / det;
    inv_mass = 1.0 / (wn * det);
#pragma unroll
    for (int eq = 0; eq < NEQ; eq++) {
      double dr[DIM];
#pragma unroll
      for (int mm = 0; mm < DIM; mm++) {
        const int sd = stride_of<C>(mm);
        const double *F = &sUp[eq * C::NODES + le_n * C::NPE + nd - idx[mm] * sd];
        double acc = 0.0;
#pragma unroll
        for (int a = 0; a < C::N1; a++) acc += tab.D[idx[mm] * C::N1 + a] * F[a * sd];
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

  // ---- face part: jump of the primitives at the face nodes
  for (int item = tid; item < C::FN_ITEMS; item += C::BLOCK) {
    const FaceItem it = face_item<C>(item, C::NF);
    const int e = e0 + it.le;
    if (e >= m.ne) continue;
    const int slot = e * C::NFACES + it.f;
    const int nb = m.face_nbr[slot];
    double u1[NEQ], u2[NEQ];
#pragma unroll
    for (int eq = 0; eq < NEQ; eq++) u1[eq] = face_trace<C>(&sUp[eq * C::NODES + it.le * C::NPE], tab, it.f, it.idx);
    if (nb >= 0) {
      const int o = m.face_orient[slot];
      const int pn = permute<DIM>(o, C::N1, it.idx % C::N1, it.idx / C::N1);
      const double *src = TA + static_cast<int64_t>(nb) * (2 * NEQ * C::NF) + NEQ * C::NF + pn;
#pragma unroll
      for (int eq = 0; eq < NEQ; eq++) u2[eq] = src[eq * C::NF];
#pragma unroll
      for (int eq = 0; eq < NEQ; eq++) sFN[eq * C::FN_ITEMS + item] = 0.5 * (u2[eq] - u1[eq]);
    } else {
      // boundary: u2 = u1 (src/faceGradientIntegration.cpp:113-115); the wall ghost of useBCinGrad
      // is not polynomial in the face nodes and is applied at the quadrature points below
#pragma unroll
      for (int eq = 0; eq < NEQ; eq++) sFN[eq * C::FN_ITEMS + item] = u1[eq];
    }
  }
  __syncthreads();

  // quadrature-point values of the jump and weighted normals, kept in registers per round
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
        const int slot = e * C::NFACES + it.f;
        const int nb = m.face_nbr[slot];
        double n[DIM], wq, X[DIM];
        face_geometry<C>(sV[it.le], tab, it.f, it.idx, n, wq, X);
#pragma unroll
        for (int d = 0; d < DIM; d++) nw[r][d] = n[d] * wq;
        const double *T = &sFN[(it.le * C::NFACES + it.f) * C::NF];
        if (nb >= 0) {
#pragma unroll
          for (int eq = 0; eq < NEQ; eq++) jq[r][eq] = face_interp<C>(T + eq * C::FN_ITEMS, tab, it.idx);
        } else if (prm.use_bc_in_grad) {
          double u1[NEQ], u2[NEQ];
#pragma unroll
          for (int eq = 0; eq < NEQ; eq++) u1[eq] = face_interp<C>(T + eq * C::FN_ITEMS, tab, it.idx);
          PH::bc_grad_prim(prm, prm.bc[-nb - 1], u1, u2);
#pragma unroll
          for (int eq = 0; eq < NEQ; eq++) jq[r][eq] = 0.5 * (u2[eq] - u1[eq]);
        }
      }
    }
  }
  __syncthreads();
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
    for (int item = tid; item < C::FN_ITEMS; item += C::BLOCK) {
      const FaceItem it = face_item<C>(item, C::NF);
      const double *R = &sQ[(it.le * C::NFACES + it.f) * C::NQ];
#pragma unroll
      for (int eq = 0; eq < NEQ; eq++) sL[eq * C::FN_ITEMS + item] = face_project<C>(R + eq * C::FQ_ITEMS, tab, it.idx);
    }
    __syncthreads();
    if (node_on) {
#pragma unroll
      for (int eq = 0; eq < NEQ; eq++)
        g[eq + d * NEQ] += inv_mass * face_lift<C>(&sL[eq * C::FN_ITEMS + le_n * C::NFACES * C::NF], tab, idx);
    }
    __syncthreads();
  }

  if (node_on) {
    const int64_t n = static_cast<int64_t>(e0 + le_n) * C::NPE + nd;
#pragma unroll
    for (int d = 0; d < DIM; d++)
#pragma unroll
      for (int eq = 0; eq < NEQ; eq++) {
        gradUp[n + eq * m.ndofs + d * NEQ * m.ndofs] = g[eq + d * NEQ];
        sG[eq + d * NEQ][tid] = g[eq + d * NEQ];
      }
  }
  __syncthreads();

  // ---- viscous normal-flux traces: face-node traces of U and gradUp, then quadrature points
  for (int item = tid; item < C::FN_ITEMS; item += C::BLOCK) {
    const FaceItem it = face_item<C>(item, C::NF);
    if (e0 + it.le >= m.ne) continue;
#pragma unroll
    for (int eq = 0; eq < NEQ; eq++) sT[eq * C::FN_ITEMS + item] = face_trace<C>(&sU[eq][it.le * C::NPE], tab, it.f, it.idx);
#pragma unroll
    for (int k = 0; k < NEQ * DIM; k++)
      sT[(NEQ + k) * C::FN_ITEMS + item] = face_trace<C>(&sG[k][it.le * C::NPE], tab, it.f, it.idx);
  }
  __syncthreads();
#pragma unroll 1
  for (int r = 0; r < C::FQ_ROUNDS; r++) {
    const int item = tid + r * C::BLOCK;
    if (item >= C::FQ_ITEMS) continue;
    const FaceItem it = face_item<C>(item, C::NQ);
    const int e = e0 + it.le;
    if (e >= m.ne) continue;
    const int slot = e * C::NFACES + it.f;
    const int nb = m.face_nbr[slot];
    const double *T = &sT[(it.le * C::NFACES + it.f) * C::NF];
    double uq[NEQ], gq[NEQ * DIM], fn[NEQ];
#pragma unroll
    for (int eq = 0; eq < NEQ; eq++) uq[eq] = face_interp<C>(T + eq * C::FN_ITEMS, tab, it.idx);
    PH::clamp_species(uq);
#pragma unroll
    for (int k = 0; k < NEQ * DIM; k++) gq[k] = face_interp<C>(T + (NEQ + k) * C::FN_ITEMS, tab, it.idx);
    double n[DIM], wq, X[DIM];
    face_geometry<C>(sV[it.le], tab, it.f, it.idx, n, wq, X);
    if (nb >= 0) {
      PH::visc_flux_n(prm, uq, gq, n, fn);
    } else {
      PH::bc_visc_term(prm, prm.bc[-nb - 1], uq, gq, n, fn);
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
__global__ __launch_bounds__(C::BLOCK) void k_flux(MeshDev m, typename PH::Params prm, const double *__restrict__ U,
                                                   const double *__restrict__ gradUp, const double *__restrict__ TA,
                                                   const double *__restrict__ TB, double *__restrict__ Y,
                                                   unsigned long long *__restrict__ max_speed_bits) {
  constexpr int NEQ = PH::NEQ, DIM = C::DIM;
  __shared__ Tables1D tab;
  __shared__ double sV[C::EPB][C::NV * DIM];
  __shared__ double sU[NEQ][C::NODES];
  __shared__ double sGf[NEQ * DIM][C::NODES];  // contravariant nodal flux, [eq + m*NEQ]
  __shared__ double sT1[NEQ][C::FN_ITEMS];     // own face-node traces of U
  __shared__ double sT2[NEQ][C::FN_ITEMS];     // neighbour traces, permuted into my frame
  __shared__ double sQ[NEQ][C::FQ_ITEMS];      // weighted numerical flux at the quadrature points
  double(*sL)[C::FN_ITEMS] = sT1;              // projected flux (reuses sT1 after the qpt loop)

  load_tables<C>(tab, m.tables);
  const int tid = threadIdx.x;
  const int e0 = blockIdx.x * C::EPB;
  for (int i = tid; i < C::EPB * C::NV * DIM; i += C::BLOCK) {
    const int le = i / (C::NV * DIM);
    if (e0 + le < m.ne) sV[le][i - le * (C::NV * DIM)] = m.verts[static_cast<int64_t>(e0) * C::NV * DIM + i];
  }
  const bool node_on = tid < C::NODES && (e0 + tid / C::NPE) < m.ne;
  const int le_n = tid / C::NPE, nd = tid - le_n * C::NPE;
  int idx[3] = {nd % C::N1, (nd / C::N1) % C::N1, nd / (C::N1 * C::N1)};
  double u[NEQ], gr[NEQ * DIM];
  if (node_on) {
    const int64_t n = static_cast<int64_t>(e0 + le_n) * C::NPE + nd;
#pragma unroll
    for (int eq = 0; eq < NEQ; eq++) {
      u[eq] = U[n + eq * m.ndofs];
      sU[eq][tid] = u[eq];
    }
#pragma unroll
    for (int d = 0; d < DIM; d++)
#pragma unroll
      for (int eq = 0; eq < NEQ; eq++) gr[eq + d * NEQ] = gradUp[n + eq * m.ndofs + d * NEQ * m.ndofs];
  }
  __syncthreads();  // tables + vertices + sU

  // ---- nodal flux F_c - F_v (src/rhs_operator.cpp:493-559), contravariant components
  double inv_mass = 0.0, speed = 0.0;
  double src[NEQ];
#pragma unroll
  for (int eq = 0; eq < NEQ; eq++) src[eq] = 0.0;
  if (node_on) {
    double xi[DIM], J[DIM * DIM], A[DIM * DIM];
    double wn = 1.0;
#pragma unroll
    for (int d = 0; d < DIM; d++) {
      xi[d] = tab.x[idx[d]];
      wn *= tab.w[idx[d]];
    }
    jacobian<DIM>(sV[le_n], xi, J);
    const double det = adjugate<DIM>(J, A);
    inv_mass = 1.0 / (wn * det);
    double uc[NEQ];
#pragma unroll
    for (int eq = 0; eq < NEQ; eq++) uc[eq] = u[eq];
    PH::clamp_species(uc);
    double F[NEQ * DIM], Fv[NEQ * DIM];
    PH::conv_flux(prm, uc, F);
    PH::visc_flux(prm, uc, gr, Fv);
    speed = PH::max_char_speed(prm, uc);
    if (PH::HAS_SOURCE) {
      double up[NEQ];
      PH::prim(prm, u, up);
      PH::source(prm, u, up, gr, src);
    }
#pragma unroll
    for (int eq = 0; eq < NEQ; eq++)
#pragma unroll
      for (int mm = 0; mm < DIM; mm++) {
        double s = 0.0;
#pragma unroll
        for (int d = 0; d < DIM; d++) s += A[mm + d * DIM] * (F[eq + d * NEQ] - Fv[eq + d * NEQ]);
        sGf[eq + mm * NEQ][tid] = wn * s;
      }
  }
  // max |u|+c over the block -> global (positive doubles order like their bit patterns)
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
    for (int eq = 0; eq < NEQ; eq++) sT1[eq][item] = face_trace<C>(&sU[eq][it.le * C::NPE], tab, it.f, it.idx);
    if (nb >= 0) {
      const int o = m.face_orient[slot];
      const int pn = permute<DIM>(o, C::N1, it.idx % C::N1, it.idx / C::N1);
      const double *s2 = TA + static_cast<int64_t>(nb) * (2 * NEQ * C::NF) + pn;
#pragma unroll
      for (int eq = 0; eq < NEQ; eq++) sT2[eq][item] = s2[eq * C::NF];
    }
  }
  __syncthreads();

  // ---- volume term: z_j = sum_m sum_a D[a][j_m] Ghat_m(a)   (src/domain_integrator.cpp:45-99)
  double z[NEQ];
  if (node_on) {
#pragma unroll
    for (int eq = 0; eq < NEQ; eq++) {
      double acc = 0.0;
#pragma unroll
      for (int mm = 0; mm < DIM; mm++) {
        const int sd = stride_of<C>(mm);
        const double *F = &sGf[eq + mm * NEQ][le_n * C::NPE + nd - idx[mm] * sd];
#pragma unroll
        for (int a = 0; a < C::N1; a++) acc += tab.D[a * C::N1 + idx[mm]] * F[a * sd];
      }
      z[eq] = acc;
    }
  }

  // ---- numerical flux at the face quadrature points
#pragma unroll 1
  for (int r = 0; r < C::FQ_ROUNDS; r++) {
    const int item = tid + r * C::BLOCK;
    if (item >= C::FQ_ITEMS) continue;
    const FaceItem it = face_item<C>(item, C::NQ);
    const int e = e0 + it.le;
    if (e >= m.ne) continue;
    const int slot = e * C::NFACES + it.f;
    const int nb = m.face_nbr[slot];
    const int fbase = (it.le * C::NFACES + it.f) * C::NF;
    double u1[NEQ], u2[NEQ], fh[NEQ];
#pragma unroll
    for (int eq = 0; eq < NEQ; eq++) u1[eq] = face_interp<C>(&sT1[eq][fbase], tab, it.idx);
    PH::clamp_species(u1);
    double n[DIM], wq, X[DIM];
    face_geometry<C>(sV[it.le], tab, it.f, it.idx, n, wq, X);
    const double *tb_own = TB + static_cast<int64_t>(slot) * (NEQ * C::NQ) + it.idx;
    if (nb >= 0) {
#pragma unroll
      for (int eq = 0; eq < NEQ; eq++) u2[eq] = face_interp<C>(&sT2[eq][fbase], tab, it.idx);
      PH::clamp_species(u2);
      PH::lax_friedrichs(prm, u1, u2, n, fh);
      const int o = m.face_orient[slot];
      const int pq = permute<DIM>(o, C::Q1, it.idx % C::Q1, it.idx / C::Q1);
      const double *tb_nb = TB + static_cast<int64_t>(nb) * (NEQ * C::NQ) + pq;
#pragma unroll
      for (int eq = 0; eq < NEQ; eq++) fh[eq] -= 0.5 * (tb_own[eq * C::NQ] - tb_nb[eq * C::NQ]);
    } else {
      PH::bc_ghost(prm, prm.bc[-nb - 1], u1, n, u2);
      PH::lax_friedrichs(prm, u1, u2, n, fh);
#pragma unroll
      for (int eq = 0; eq < NEQ; eq++) fh[eq] += tb_own[eq * C::NQ];
    }
#pragma unroll
    for (int eq = 0; eq < NEQ; eq++) sQ[eq][item] = fh[eq] * wq;
  }
  __syncthreads();
  for (int item = tid; item < C::FN_ITEMS; item += C::BLOCK) {
    const FaceItem it = face_item<C>(item, C::NF);
    if (e0 + it.le >= m.ne) continue;
    const int qbase = (it.le * C::NFACES + it.f) * C::NQ;
#pragma unroll
    for (int eq = 0; eq < NEQ; eq++) sL[eq][item] = face_project<C>(&sQ[eq][qbase], tab, it.idx);
  }
  __syncthreads();
  if (node_on) {
    const int64_t n = static_cast<int64_t>(e0 + le_n) * C::NPE + nd;
#pragma unroll
    for (int eq = 0; eq < NEQ; eq++) {
      const double lift = face_lift<C>(&sL[eq][le_n * C::NFACES * C::NF], tab, idx);
      Y[n + eq * m.ndofs] = inv_mass * (z[eq] - lift) + src[eq];
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
