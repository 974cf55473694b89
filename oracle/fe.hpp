// TEST INFRASTRUCTURE -- CPU oracle, never shipped, never on the product path.
//
// Finite-element substrate of the oracle: everything TPS takes from MFEM (which is absent from
// /root/reference and from this image).  Restated from MFEM's published
// conventions, not from TPS source:
//   * 1-D Gauss-Legendre / Gauss-Lobatto rules on [0,1]; IntegrationRules::Get(geom, order) uses
//     n = order/2+1 (GL) or n = order/2+2 (GLL) points per direction, tensorised for squares/cubes
//   * DG_FECollection nodal bases: tensor Lagrange polynomials on GL (basisType 0) or GLL
//     (basisType 1) nodes, dofs lexicographic (x fastest)     [used at src/M2ulPhyS.cpp:557-579]
//   * order-1 (bi/tri-linear) element geometry, MFEM vertex ordering; OrderW = dim*1-1
//   * face quadrature order OrderW + 2p                        [src/face_integrator.cpp:233-243]
//   * CalcOrtho: area-weighted (non unit) normal, pointing out of element 1
// PARITY UNPINNED for the assembled operator (see oracle/README.md); the gradient operator is
// pinned by the reference's test/gradient.test error windows (tests/test_oracle_pins.py).
#ifndef TPS_ORACLE_FE_HPP_
#define TPS_ORACLE_FE_HPP_

#include <algorithm>
#include <array>
#include <cassert>
#include <cmath>
#include <cstdio>
#include <map>
#include <stdexcept>
#include <string>
#include <vector>

namespace tpsoracle {

// ---------------------------------------------------------------- 1-D rules on [0,1]
struct Rule1D {
  std::vector<double> x, w;
  int n() const { return static_cast<int>(x.size()); }
};

inline void legendre(int n, double x, double &P, double &dP) {
  // P_n and P_n' on [-1,1] by the three-term recurrence
  double p0 = 1.0, p1 = x;
  if (n == 0) {
    P = 1.0;
    dP = 0.0;
    return;
  }
  for (int k = 2; k <= n; k++) {
    double pk = ((2.0 * k - 1.0) * x * p1 - (k - 1.0) * p0) / k;
    p0 = p1;
    p1 = pk;
  }
  P = p1;
  dP = n * (x * p1 - p0) / (x * x - 1.0);
}

inline Rule1D gauss_legendre(int n) {
  Rule1D r;
  r.x.resize(n);
  r.w.resize(n);
  for (int i = 0; i < n; i++) {
    double z = -std::cos(M_PI * (i + 0.75) / (n + 0.5));
    for (int it = 0; it < 100; it++) {
      double P, dP;
      legendre(n, z, P, dP);
      double dz = P / dP;
      z -= dz;
      if (std::fabs(dz) < 1e-16) break;
    }
    double P, dP;
    legendre(n, z, P, dP);
    r.x[i] = 0.5 * (1.0 + z);
    r.w[i] = 1.0 / ((1.0 - z * z) * dP * dP);  // (2/((1-z^2)P'^2)) * 1/2
  }
  // symmetrise exactly
  for (int i = 0; i < n / 2; i++) {
    double xm = 0.5 * (r.x[i] + (1.0 - r.x[n - 1 - i]));
    r.x[i] = xm;
    r.x[n - 1 - i] = 1.0 - xm;
    double wm = 0.5 * (r.w[i] + r.w[n - 1 - i]);
    r.w[i] = r.w[n - 1 - i] = wm;
  }
  if (n % 2 == 1) r.x[n / 2] = 0.5;
  return r;
}

inline Rule1D gauss_lobatto(int n) {
  Rule1D r;
  r.x.resize(n);
  r.w.resize(n);
  if (n == 1) {
    r.x[0] = 0.5;
    r.w[0] = 1.0;
    return r;
  }
  const int N = n - 1;
  r.x[0] = 0.0;
  r.x[N] = 1.0;
  r.w[0] = r.w[N] = 1.0 / (N * (N + 1.0));
  for (int i = 1; i < N; i++) {
    // interior nodes: roots of P_N'(z); Newton on q(z) = P_N'(z) with
    // q' from Legendre's ODE: (1-z^2) P'' = 2 z P' - N(N+1) P
    double z = -std::cos(M_PI * i / N);
    for (int it = 0; it < 100; it++) {
      double P, dP;
      legendre(N, z, P, dP);
      double d2P = (2.0 * z * dP - N * (N + 1.0) * P) / (1.0 - z * z);
      double dz = dP / d2P;
      z -= dz;
      if (std::fabs(dz) < 1e-16) break;
    }
    double P, dP;
    legendre(N, z, P, dP);
    r.x[i] = 0.5 * (1.0 + z);
    r.w[i] = 1.0 / (N * (N + 1.0) * P * P);
  }
  for (int i = 0; i < n / 2; i++) {
    double xm = 0.5 * (r.x[i] + (1.0 - r.x[n - 1 - i]));
    r.x[i] = xm;
    r.x[n - 1 - i] = 1.0 - xm;
    double wm = 0.5 * (r.w[i] + r.w[n - 1 - i]);
    r.w[i] = r.w[n - 1 - i] = wm;
  }
  if (n % 2 == 1) r.x[n / 2] = 0.5;
  return r;
}

// IntegrationRules::Get(Segment, order) for quadrature family `type` (0 GL, 1 GLL)
inline Rule1D segment_rule(int type, int order) {
  if (type == 0) return gauss_legendre(order / 2 + 1);
  return gauss_lobatto(order / 2 + 2);
}

// ---------------------------------------------------------------- 1-D Lagrange basis
struct Basis1D {
  std::vector<double> nodes;
  int n() const { return static_cast<int>(nodes.size()); }
  void eval(double x, double *l) const {
    const int N = n();
    for (int m = 0; m < N; m++) {
      double v = 1.0;
      for (int j = 0; j < N; j++)
        if (j != m) v *= (x - nodes[j]) / (nodes[m] - nodes[j]);
      l[m] = v;
    }
  }
  void eval_d(double x, double *dl) const {
    const int N = n();
    for (int m = 0; m < N; m++) {
      double s = 0.0;
      for (int i = 0; i < N; i++) {
        if (i == m) continue;
        double v = 1.0 / (nodes[m] - nodes[i]);
        for (int j = 0; j < N; j++)
          if (j != m && j != i) v *= (x - nodes[j]) / (nodes[m] - nodes[j]);
        s += v;
      }
      dl[m] = s;
    }
  }
};

inline Basis1D make_basis(int basis_type, int p) {
  Basis1D b;
  b.nodes = (basis_type == 0) ? gauss_legendre(p + 1).x : gauss_lobatto(p + 1).x;
  return b;
}

// ---------------------------------------------------------------- small dense helpers
struct Dense {
  int r = 0, c = 0;
  std::vector<double> a;  // column-major like mfem::DenseMatrix
  Dense() {}
  Dense(int r_, int c_) : r(r_), c(c_), a(static_cast<size_t>(r_) * c_, 0.0) {}
  double &operator()(int i, int j) { return a[i + static_cast<size_t>(j) * r]; }
  double operator()(int i, int j) const { return a[i + static_cast<size_t>(j) * r]; }
};

inline void invert_dense(Dense &M) {
  // Gauss-Jordan with partial pivoting (role of mfem::DenseMatrix::Invert)
  const int n = M.r;
  Dense I(n, n);
  for (int i = 0; i < n; i++) I(i, i) = 1.0;
  for (int col = 0; col < n; col++) {
    int piv = col;
    for (int i = col + 1; i < n; i++)
      if (std::fabs(M(i, col)) > std::fabs(M(piv, col))) piv = i;
    if (M(piv, col) == 0.0) throw std::runtime_error("singular mass matrix");
    if (piv != col) {
      for (int j = 0; j < n; j++) {
        std::swap(M(piv, j), M(col, j));
        std::swap(I(piv, j), I(col, j));
      }
    }
    const double inv = 1.0 / M(col, col);
    for (int j = 0; j < n; j++) {
      M(col, j) *= inv;
      I(col, j) *= inv;
    }
    for (int i = 0; i < n; i++) {
      if (i == col) continue;
      const double f = M(i, col);
      if (f == 0.0) continue;
      for (int j = 0; j < n; j++) {
        M(i, j) -= f * M(col, j);
        I(i, j) -= f * I(col, j);
      }
    }
  }
  M = I;
}

// ---------------------------------------------------------------- mesh + face table
// Local faces are numbered f = 2*d + s: the face xi_d = s of the reference square/cube.
// Tangential axes of a face are the remaining reference axes in increasing order.
struct Face {
  int e1 = -1, f1 = -1;  // element 1 and its local face
  int e2 = -1, f2 = -1;  // element 2 (or -1 on the boundary)
  int attr = -1;         // boundary attribute (boundary faces)
  // map from element-1 tangential coordinates (ta,tb) to element-2 ones:
  //   swap=0: ta' = fa ? 1-ta : ta ; tb' = fb ? 1-tb : tb
  //   swap=1: ta' = fa ? 1-tb : tb ; tb' = fb ? 1-ta : ta
  int swap = 0, fa = 0, fb = 0;
};

struct Mesh {
  int dim = 0, nv = 0, ne = 0;
  int nvpe = 0;                     // vertices per element
  std::vector<int> ev;              // [ne*nvpe] topological ids, LEXICOGRAPHIC corner order
  std::vector<double> ex;           // [ne*nvpe*dim] coordinates, lexicographic corner order
  std::vector<Face> faces;
  std::vector<std::array<int, 6>> elem_faces;  // face index per local face (-1 unused)

  static int lex_of_mfem(int dim, int v) {
    static const int q[4] = {0, 1, 3, 2};
    static const int h[8] = {0, 1, 3, 2, 4, 5, 7, 6};
    return dim == 2 ? q[v] : h[v];
  }
  // lexicographic corner indices of local face f, in (ta,tb) corner order
  void face_corners(int f, int *c) const {
    const int d = f / 2, s = f % 2;
    if (dim == 2) {
      const int a = 1 - d;
      for (int ta = 0; ta < 2; ta++) c[ta] = (s << d) | (ta << a);
    } else {
      int a = (d == 0) ? 1 : 0, b = (d == 2) ? 1 : 2;
      for (int tb = 0; tb < 2; tb++)
        for (int ta = 0; ta < 2; ta++) c[ta + 2 * tb] = (s << d) | (ta << a) | (tb << b);
    }
  }

  void build(int dim_, int nv_, int ne_, const int *elem_vertices, const double *elem_coords,
             int nbf, const int *bdr_vertices, const int *bdr_attr) {
    dim = dim_;
    nv = nv_;
    ne = ne_;
    nvpe = 1 << dim;
    ev.resize(static_cast<size_t>(ne) * nvpe);
    ex.resize(static_cast<size_t>(ne) * nvpe * dim);
    for (int e = 0; e < ne; e++)
      for (int v = 0; v < nvpe; v++) {
        const int l = lex_of_mfem(dim, v);
        ev[e * nvpe + l] = elem_vertices[e * nvpe + v];
        for (int d = 0; d < dim; d++)
          ex[(static_cast<size_t>(e) * nvpe + l) * dim + d] =
              elem_coords[(static_cast<size_t>(e) * nvpe + v) * dim + d];
      }
    const int nfv = 1 << (dim - 1), nlf = 2 * dim;
    typedef std::array<int, 4> Key;
    std::map<Key, int> table;
    elem_faces.assign(ne, {-1, -1, -1, -1, -1, -1});
    faces.clear();
    for (int e = 0; e < ne; e++) {
      for (int f = 0; f < nlf; f++) {
        int c[4], g[4] = {-1, -1, -1, -1};
        face_corners(f, c);
        for (int i = 0; i < nfv; i++) g[i] = ev[e * nvpe + c[i]];
        Key key = {-1, -1, -1, -1};
        for (int i = 0; i < nfv; i++) key[i] = g[i];
        std::sort(key.begin(), key.begin() + nfv);
        for (int i = 1; i < nfv; i++)
          if (key[i] == key[i - 1]) throw std::runtime_error("degenerate face (too few periodic cells)");
        auto it = table.find(key);
        if (it == table.end()) {
          Face F;
          F.e1 = e;
          F.f1 = f;
          table[key] = static_cast<int>(faces.size());
          elem_faces[e][f] = static_cast<int>(faces.size());
          faces.push_back(F);
        } else {
          Face &F = faces[it->second];
          if (F.e2 >= 0) throw std::runtime_error("face shared by more than two elements");
          F.e2 = e;
          F.f2 = f;
          elem_faces[e][f] = it->second;
          // orientation: match element-1 corners to element-2 corners
          int c1[4], g1[4];
          face_corners(F.f1, c1);
          for (int i = 0; i < nfv; i++) g1[i] = ev[F.e1 * nvpe + c1[i]];
          auto find2 = [&](int gv) {
            for (int i = 0; i < nfv; i++)
              if (g[i] == gv) return i;
            throw std::runtime_error("face vertex mismatch");
          };
          if (dim == 2) {
            F.swap = 0;
            F.fa = find2(g1[0]);  // image of ta=0
            F.fb = 0;
          } else {
            const int o = find2(g1[0]);   // image of (0,0)
            const int pa = find2(g1[1]);  // image of (1,0)
            const int oa = o & 1, ob = o >> 1, paa = pa & 1;
            F.swap = (paa != oa) ? 0 : 1;
            F.fa = oa;
            F.fb = ob;
          }
        }
      }
    }
    // boundary attributes
    for (int b = 0; b < nbf; b++) {
      Key key = {-1, -1, -1, -1};
      for (int i = 0; i < nfv; i++) key[i] = bdr_vertices[b * nfv + i];
      std::sort(key.begin(), key.begin() + nfv);
      auto it = table.find(key);
      if (it == table.end()) throw std::runtime_error("boundary face not found in mesh");
      Face &F = faces[it->second];
      if (F.e2 < 0) F.attr = bdr_attr[b];  // interior "boundary" elements are ignored (periodic)
    }
    for (const Face &F : faces)
      if (F.e2 < 0 && F.attr < 0) throw std::runtime_error("boundary face without attribute");
  }

  // element-1 tangential coords -> element-2 tangential coords
  static void map_tangent(const Face &F, int dim, const double *t, double *t2) {
    if (dim == 2) {
      t2[0] = F.fa ? 1.0 - t[0] : t[0];
      return;
    }
    if (!F.swap) {
      t2[0] = F.fa ? 1.0 - t[0] : t[0];
      t2[1] = F.fb ? 1.0 - t[1] : t[1];
    } else {
      t2[0] = F.fa ? 1.0 - t[1] : t[1];
      t2[1] = F.fb ? 1.0 - t[0] : t[0];
    }
  }
  // reference point of an element from local face + tangential coordinates
  void face_ref_point(int f, const double *t, double *xi) const {
    const int d = f / 2, s = f % 2;
    if (dim == 2) {
      xi[d] = s;
      xi[1 - d] = t[0];
    } else {
      int a = (d == 0) ? 1 : 0, b = (d == 2) ? 1 : 2;
      xi[d] = s;
      xi[a] = t[0];
      xi[b] = t[1];
    }
  }
  // x(xi) and J(i,m) = dx_i/dxi_m of the multilinear map of element e
  void transform(int e, const double *xi, double *x, double *J) const {
    double N[3][2], dN[3][2];
    for (int d = 0; d < dim; d++) {
      N[d][0] = 1.0 - xi[d];
      N[d][1] = xi[d];
      dN[d][0] = -1.0;
      dN[d][1] = 1.0;
    }
    for (int i = 0; i < dim; i++) {
      x[i] = 0.0;
      for (int m = 0; m < dim; m++) J[i + m * dim] = 0.0;
    }
    for (int c = 0; c < nvpe; c++) {
      const double *X = &ex[(static_cast<size_t>(e) * nvpe + c) * dim];
      double shp = 1.0;
      for (int d = 0; d < dim; d++) shp *= N[d][(c >> d) & 1];
      for (int i = 0; i < dim; i++) x[i] += shp * X[i];
      for (int m = 0; m < dim; m++) {
        double ds = 1.0;
        for (int d = 0; d < dim; d++) ds *= (d == m) ? dN[d][(c >> d) & 1] : N[d][(c >> d) & 1];
        for (int i = 0; i < dim; i++) J[i + m * dim] += ds * X[i];
      }
    }
  }
  // CalcOrtho of the face Jacobian: area-weighted normal out of the element owning local face f
  void face_normal(int f, const double *J, double *nor) const {
    const int d = f / 2, s = f % 2;
    const double sg = s ? 1.0 : -1.0;
    if (dim == 2) {
      const int a = 1 - d;
      const double tx = J[0 + a * 2], ty = J[1 + a * 2];
      // tangent rotated by -90deg points along +xi_0 for d=0 (a=1) up to handedness
      const double o = (d == 0) ? 1.0 : -1.0;
      nor[0] = sg * o * ty;
      nor[1] = -sg * o * tx;
    } else {
      int a = (d == 0) ? 1 : 0, b = (d == 2) ? 1 : 2;
      const double *ta = &J[a * 3], *tb = &J[b * 3];
      const double o = (d == 1) ? -1.0 : 1.0;
      nor[0] = sg * o * (ta[1] * tb[2] - ta[2] * tb[1]);
      nor[1] = sg * o * (ta[2] * tb[0] - ta[0] * tb[2]);
      nor[2] = sg * o * (ta[0] * tb[1] - ta[1] * tb[0]);
    }
  }
};

// Smallest singular value of the dim x dim matrix J[i + m*dim] (the role of DenseMatrix::CalcSingularvalue(dim-1)
// behind Mesh::GetElementSize(e, 1) [MFEM]): one-sided Jacobi (Hestenes) -- rotate column pairs until they are
// orthogonal, the singular values are the column norms.
inline double min_singular_value(int dim, const double *Jin) {
  double A[9];
  for (int k = 0; k < dim * dim; k++) A[k] = Jin[k];
  for (int sweep = 0; sweep < 60; sweep++) {
    bool rotated = false;
    for (int p = 0; p < dim; p++)
      for (int q = p + 1; q < dim; q++) {
        double app = 0.0, aqq = 0.0, apq = 0.0;
        for (int i = 0; i < dim; i++) {
          app += A[i + p * dim] * A[i + p * dim];
          aqq += A[i + q * dim] * A[i + q * dim];
          apq += A[i + p * dim] * A[i + q * dim];
        }
        if (std::fabs(apq) <= 1e-17 * std::sqrt(app * aqq)) continue;
        rotated = true;
        const double zeta = (aqq - app) / (2.0 * apq);
        const double t = (zeta >= 0.0 ? 1.0 : -1.0) / (std::fabs(zeta) + std::sqrt(1.0 + zeta * zeta));
        const double c = 1.0 / std::sqrt(1.0 + t * t), sn = c * t;
        for (int i = 0; i < dim; i++) {
          const double vp = A[i + p * dim], vq = A[i + q * dim];
          A[i + p * dim] = c * vp - sn * vq;
          A[i + q * dim] = sn * vp + c * vq;
        }
      }
    if (!rotated) break;
  }
  double smin = 1e300;
  for (int p = 0; p < dim; p++) {
    double n2 = 0.0;
    for (int i = 0; i < dim; i++) n2 += A[i + p * dim] * A[i + p * dim];
    smin = std::min(smin, std::sqrt(n2));
  }
  return smin;
}

inline double det_and_inverse(int dim, const double *J, double *Ji) {
  if (dim == 2) {
    const double det = J[0] * J[3] - J[2] * J[1];
    Ji[0] = J[3] / det;
    Ji[2] = -J[2] / det;
    Ji[1] = -J[1] / det;
    Ji[3] = J[0] / det;
    return det;
  }
  const double a = J[0], b = J[3], c = J[6], d = J[1], e = J[4], f = J[7], g = J[2], h = J[5], i = J[8];
  const double A = e * i - f * h, B = -(d * i - f * g), C = d * h - e * g;
  const double det = a * A + b * B + c * C;
  const double id = 1.0 / det;
  Ji[0] = A * id;
  Ji[3] = -(b * i - c * h) * id;
  Ji[6] = (b * f - c * e) * id;
  Ji[1] = B * id;
  Ji[4] = (a * i - c * g) * id;
  Ji[7] = -(a * f - c * d) * id;
  Ji[2] = C * id;
  Ji[5] = -(a * h - b * g) * id;
  Ji[8] = (a * e - b * d) * id;
  return det;
}

// ---------------------------------------------------------------- tensor element
struct Element {
  int dim = 0, p = 0, n1 = 0, dof = 0;
  Basis1D b;
  void init(int dim_, int p_, int basis_type) {
    dim = dim_;
    p = p_;
    n1 = p + 1;
    dof = (dim == 2) ? n1 * n1 : n1 * n1 * n1;
    b = make_basis(basis_type, p);
  }
  void calc_shape(const double *xi, double *shape) const {
    double l[3][8];
    for (int d = 0; d < dim; d++) b.eval(xi[d], l[d]);
    int o = 0;
    if (dim == 2) {
      for (int j = 0; j < n1; j++)
        for (int i = 0; i < n1; i++) shape[o++] = l[0][i] * l[1][j];
    } else {
      for (int k = 0; k < n1; k++)
        for (int j = 0; j < n1; j++)
          for (int i = 0; i < n1; i++) shape[o++] = l[0][i] * l[1][j] * l[2][k];
    }
  }
  // dshape(k, m) = d phi_k / d xi_m, stored [k + m*dof]
  void calc_dshape(const double *xi, double *dshape) const {
    double l[3][8], dl[3][8];
    for (int d = 0; d < dim; d++) {
      b.eval(xi[d], l[d]);
      b.eval_d(xi[d], dl[d]);
    }
    int o = 0;
    if (dim == 2) {
      for (int j = 0; j < n1; j++)
        for (int i = 0; i < n1; i++) {
          dshape[o] = dl[0][i] * l[1][j];
          dshape[o + dof] = l[0][i] * dl[1][j];
          o++;
        }
    } else {
      for (int k = 0; k < n1; k++)
        for (int j = 0; j < n1; j++)
          for (int i = 0; i < n1; i++) {
            dshape[o] = dl[0][i] * l[1][j] * l[2][k];
            dshape[o + dof] = l[0][i] * dl[1][j] * l[2][k];
            dshape[o + 2 * dof] = l[0][i] * l[1][j] * dl[2][k];
            o++;
          }
    }
  }
  void node_ref(int k, double *xi) const {
    xi[0] = b.nodes[k % n1];
    xi[1] = b.nodes[(k / n1) % n1];
    if (dim == 3) xi[2] = b.nodes[k / (n1 * n1)];
  }
};

// tensor-product rule on the reference square/cube (x fastest), or segment/square for faces
struct RuleND {
  int dim = 0, npts = 0;
  std::vector<double> x;  // [npts*dim]
  std::vector<double> w;
  void init(int dim_, const Rule1D &r) {
    dim = dim_;
    const int n = r.n();
    npts = 1;
    for (int d = 0; d < dim; d++) npts *= n;
    x.resize(static_cast<size_t>(npts) * dim);
    w.resize(npts);
    for (int q = 0; q < npts; q++) {
      int rem = q;
      double ww = 1.0;
      for (int d = 0; d < dim; d++) {
        const int i = rem % n;
        rem /= n;
        x[q * dim + d] = r.x[i];
        ww *= r.w[i];
      }
      w[q] = ww;
    }
  }
};

}  // namespace tpsoracle
#endif
