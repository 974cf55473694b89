// Host-side 1-D tables of the DG discretisation: nodes/weights, the nodal differentiation matrix, end-point
// (face) values of the Lagrange basis, the node -> face-quadrature interpolation matrix and -- for the
// non-collocated Gauss-Lobatto pair, the reference's default (src/M2ulPhyS.cpp:2671-2672) -- the values and
// derivatives of the basis at the volume quadrature points.  They replace the per-element dense operators of
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
constexpr int MAXQV = 7;  // volume rule points per direction (order 2p: p + 1 Gauss-Legendre, p + 2 Gauss-Lobatto)

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
  // non-collocated variant: the volume rule (order 2p) and the basis on it
  double xv[MAXQV], wv[MAXQV];
  double Bv[MAXQV * MAXN1];   // Bv[q*N1+a] = l_a(xv_q)
  double Dv[MAXQV * MAXN1];   // Dv[q*N1+a] = l_a'(xv_q)
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

// Gauss-Lobatto points and weights on [0,1]: end points plus the roots of P'_{n-1}
inline void gauss_lobatto01(int n, double *x, double *w) {
  x[0] = 0.0;
  x[n - 1] = 1.0;
  w[0] = w[n - 1] = 1.0 / (n * (n - 1.0));
  for (int i = 1; i <= (n - 1) / 2; i++) {
    // Newton on P'_{n-1}, started from the Chebyshev-Lobatto guess
    double z = std::cos(M_PI * i / (n - 1.0)), pn = 0.0;
    for (int it = 0; it < 100; it++) {
      double p1 = 1.0, p2 = 0.0;  // P_j, P_{j-1}
      for (int j = 1; j <= n - 1; j++) {
        const double p3 = p2;
        p2 = p1;
        p1 = ((2.0 * j - 1.0) * z * p2 - (j - 1.0) * p3) / j;
      }
      pn = p1;
      const int m = n - 1;
      const double dp = m * (z * p1 - p2) / (z * z - 1.0);                   // P'_m
      const double d2p = (2.0 * z * dp - m * (m + 1.0) * p1) / (1.0 - z * z);  // P''_m
      const double z1 = z;
      z = z1 - dp / d2p;
      if (std::fabs(z - z1) < 1e-16) break;
    }
    {  // P_{n-1} at the converged point
      double p1 = 1.0, p2 = 0.0;
      for (int j = 1; j <= n - 1; j++) {
        const double p3 = p2;
        p2 = p1;
        p1 = ((2.0 * j - 1.0) * z * p2 - (j - 1.0) * p3) / j;
      }
      pn = p1;
    }
    x[i] = 0.5 * (1.0 - z);
    x[n - 1 - i] = 0.5 * (1.0 + z);
    w[i] = w[n - 1 - i] = 1.0 / (n * (n - 1.0) * pn * pn);
  }
}
// IntegrationRules::Get(Segment, order) of the quadrature family `rule` (0 Gauss-Legendre, 1 Gauss-Lobatto): points
inline int rule_points(int rule, int order) { return rule == 0 ? order / 2 + 1 : order / 2 + 2; }
inline void segment_rule01(int rule, int n, double *x, double *w) {
  if (rule == 0)
    gauss_legendre01(n, x, w);
  else
    gauss_lobatto01(n, x, w);
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
// basis: 0 Gauss-Legendre nodes, 1 Gauss-Lobatto nodes (DG_FECollection basis type, src/M2ulPhyS.cpp:564-572);
// rule: the quadrature family of every integration rule (src/M2ulPhyS.cpp:557-562)
inline Tables1D make_tables(int p, int dim, int basis = 0, int rule = 0) {
  Tables1D t = {};
  const int n1 = p + 1;
  const int q1 = rule_points(rule, (dim - 1) + 2 * p);
  segment_rule01(basis, n1, t.x, t.w);  // t.w: the collocated weights (meaningful for basis == rule == 0 only)
  segment_rule01(rule, q1, t.xq, t.wq);
  const int qv = rule_points(rule, 2 * p);
  segment_rule01(rule, qv, t.xv, t.wv);
  for (int q = 0; q < qv; q++)
    for (int a = 0; a < n1; a++) {
      t.Bv[q * n1 + a] = lagrange(t.x, n1, a, t.xv[q]);
      t.Dv[q * n1 + a] = lagrange_d(t.x, n1, a, t.xv[q]);
    }
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
