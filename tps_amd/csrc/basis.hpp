// Host-side 1-D tables of the collocated DG discretisation: Gauss-Legendre nodes/weights, the
// nodal differentiation matrix, end-point (face) values of the Lagrange basis and the
// node -> face-quadrature interpolation matrix.  They replace the per-element dense operators of
// the reference (Ke src/gradients.cpp:84-133, Me_inv src/rhs_operator.cpp:173-224, the (v,grad w)
// blocks of src/domain_integrator.cpp:45-99) and its 216x64-padded face shape tables
// (src/M2ulPhyS.cpp:899-902): everything element-independent lives in < 1 KB of LDS.
#ifndef TPSRHS_BASIS_HPP_
#define TPSRHS_BASIS_HPP_

#include <cmath>
#include <vector>

namespace tpsrhs {

constexpr int MAXN1 = 6;  // p <= 5
constexpr int MAXQ1 = 8;  // face rule points per direction for p <= 5

// LDS-resident per-kernel copy; plain doubles, sized for the largest order.
struct Tables1D {
  double x[MAXN1];            // nodes on [0,1]
  double w[MAXN1];            // quadrature weights at the nodes (collocated volume rule, order 2p)
  double D[MAXN1 * MAXN1];    // D[i*N1+a] = l_a'(x_i)
  double b0[MAXN1];           // l_a(0)
  double b1[MAXN1];           // l_a(1)
  double xq[MAXQ1];           // face rule points (order OrderW + 2p)
  double wq[MAXQ1];
  double B[MAXQ1 * MAXN1];    // B[q*N1+a] = l_a(xq_q)
};

inline void gauss_legendre01(int n, double *x, double *w) {
  for (int i = 0; i < (n + 1) / 2; i++) {
    // Newton on P_n, started from the Chebyshev guess
    double z = std::cos(M_PI * (i + 0.75) / (n + 0.5));
    double pp = 0.0;
    for (int it = 0; it < 100; it++) {
      double p1 = 1.0, p2 = 0.0;
      for (int j = 1; j <= n; j++) {
        const double p3 = p2;
        p2 = p1;
        p1 = ((2.0 * j - 1.0) * z * p2 - (j - 1.0) * p3) / j;
      }
      pp = n * (z * p1 - p2) / (z * z - 1.0);
      const double z1 = z;
      z = z1 - p1 / pp;
      if (std::fabs(z - z1) < 1e-16) break;
    }
    x[i] = 0.5 * (1.0 - z);
    x[n - 1 - i] = 0.5 * (1.0 + z);
    w[i] = w[n - 1 - i] = 1.0 / ((1.0 - z * z) * pp * pp);
  }
}

inline double lagrange(const double *nodes, int n, int a, double x) {
  double v = 1.0;
  for (int j = 0; j < n; j++)
    if (j != a) v *= (x - nodes[j]) / (nodes[a] - nodes[j]);
  return v;
}
inline double lagrange_d(const double *nodes, int n, int a, double x) {
  double s = 0.0;
  for (int i = 0; i < n; i++) {
    if (i == a) continue;
    double v = 1.0 / (nodes[a] - nodes[i]);
    for (int j = 0; j < n; j++)
      if (j != a && j != i) v *= (x - nodes[j]) / (nodes[a] - nodes[j]);
    s += v;
  }
  return s;
}

// p: order; dim: 2|3 (face rule order = (dim-1) + 2p: IsoparametricTransformation::OrderW of an
// order-1 Qk element is dim-1, src/face_integrator.cpp:233-243)
inline Tables1D make_tables(int p, int dim) {
  Tables1D t = {};
  const int n1 = p + 1;
  const int q1 = ((dim - 1) + 2 * p) / 2 + 1;
  gauss_legendre01(n1, t.x, t.w);
  gauss_legendre01(q1, t.xq, t.wq);
  for (int i = 0; i < n1; i++)
    for (int a = 0; a < n1; a++) t.D[i * n1 + a] = lagrange_d(t.x, n1, a, t.x[i]);
  for (int a = 0; a < n1; a++) {
    t.b0[a] = lagrange(t.x, n1, a, 0.0);
    t.b1[a] = lagrange(t.x, n1, a, 1.0);
  }
  for (int q = 0; q < q1; q++)
    for (int a = 0; a < n1; a++) t.B[q * n1 + a] = lagrange(t.x, n1, a, t.xq[q]);
  return t;
}

}  // namespace tpsrhs
#endif
