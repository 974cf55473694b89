// TEST INFRASTRUCTURE -- a minimal stand-in for <mfem.hpp>, only so that include/tpsrhs_mfem_adapter.hpp is
// compiled and its Mult(const Vector&, Vector&) is called from C++ in this repository's tests (this image has no
// MFEM).  It mimics the SIGNATURES of the few MFEM classes / methods the adapter uses ([MFEM] >= 4.4: Vector,
// Array, DenseMatrix, ElementTransformation::GetPointMat, Device::IsEnabled, TimeDependentOperator, and the
// ParMesh queries GetElementVertices, GetBdrElementVertices, GetBdrAttribute, GetGlobalVertexIndices,
// GetElementSize, GetNFaceNeighbors, GetFaceNbrGroup, GetFaceNbrRank, GroupNQuadrilaterals / GroupQuadrilateral, GroupNEdges /
// GroupEdge, GetFaceVertices, GetEdgeVertices) over plain arrays.  It computes nothing and is no part of the product.
#ifndef TPSRHS_MOCK_MFEM_HPP_
#define TPSRHS_MOCK_MFEM_HPP_

#include <vector>

typedef long long HYPRE_BigInt;

namespace mfem {

template <class T>
class Array {
 public:
  int Size() const { return static_cast<int>(d_.size()); }
  void SetSize(int n) { d_.resize(n); }
  T &operator[](int i) { return d_[i]; }
  const T &operator[](int i) const { return d_[i]; }
  std::vector<T> d_;
};

class Vector {
 public:
  Vector() {}
  explicit Vector(int n) : d_(n, 0.0) {}
  int Size() const { return static_cast<int>(d_.size()); }
  void SetSize(int n) { d_.assign(n, 0.0); }
  double &operator()(int i) { return d_[i]; }
  const double &operator()(int i) const { return d_[i]; }
  // memory-manager accessors: the mock has host memory only
  const double *Read() const { return d_.data(); }
  double *Write() { return d_.data(); }
  const double *HostRead() const { return d_.data(); }
  double *HostWrite() { return d_.data(); }

 private:
  std::vector<double> d_;
};

class DenseMatrix {
 public:
  DenseMatrix() : h_(0), w_(0) {}
  void SetSize(int h, int w) {
    h_ = h;
    w_ = w;
    d_.assign(static_cast<size_t>(h) * w, 0.0);
  }
  int Height() const { return h_; }
  int Width() const { return w_; }
  double &operator()(int i, int j) { return d_[i + static_cast<size_t>(j) * h_]; }
  const double &operator()(int i, int j) const { return d_[i + static_cast<size_t>(j) * h_]; }

 private:
  int h_, w_;
  std::vector<double> d_;
};

class Device {
 public:
  static bool IsEnabled() { return false; }
};

class Operator {
 public:
  explicit Operator(int s = 0) : height(s), width(s) {}
  virtual ~Operator() {}
  int Height() const { return height; }
  virtual void Mult(const Vector &x, Vector &y) const = 0;

 protected:
  int height, width;
};
class TimeDependentOperator : public Operator {
 public:
  explicit TimeDependentOperator(int n = 0, double t0 = 0.0) : Operator(n), t(t0) {}
  virtual double GetTime() const { return t; }
  virtual void SetTime(double t_) { t = t_; }

 protected:
  double t;
};

class ElementTransformation {
 public:
  const DenseMatrix &GetPointMat() const { return pm; }
  DenseMatrix pm;
};

// one rank's mesh from plain arrays (what a real ParMesh knows)
class ParMesh {
 public:
  int dim = 0, nv = 0;
  std::vector<int> elem_vertices;    // [ne][2^dim], MFEM order
  std::vector<double> elem_coords;   // [ne][2^dim][dim]
  std::vector<int> bdr_vertices;     // [nbe][2^(dim-1)]
  std::vector<int> bdr_attributes;
  std::vector<long long> global_vertex;  // [nv]
  std::vector<double> elem_size;     // [ne] what GetElementSize(e, 1) returns (the test supplies it; nothing is computed)
  struct Nbr {
    int rank;
    std::vector<std::vector<int>> faces;  // local vertices of each shared face
  };
  std::vector<Nbr> nbrs;

  int Dimension() const { return dim; }
  int GetNV() const { return nv; }
  int GetNE() const { return static_cast<int>(elem_vertices.size()) >> dim; }
  int GetNBE() const { return static_cast<int>(bdr_attributes.size()); }
  void GetElementVertices(int e, Array<int> &v) const {
    const int n = 1 << dim;
    v.SetSize(n);
    for (int k = 0; k < n; k++) v[k] = elem_vertices[static_cast<size_t>(e) * n + k];
  }
  ElementTransformation *GetElementTransformation(int e) {
    const int n = 1 << dim;
    tr_.pm.SetSize(dim, n);
    for (int k = 0; k < n; k++)
      for (int d = 0; d < dim; d++) tr_.pm(d, k) = elem_coords[(static_cast<size_t>(e) * n + k) * dim + d];
    return &tr_;
  }
  void GetBdrElementVertices(int b, Array<int> &v) const {
    const int n = 1 << (dim - 1);
    v.SetSize(n);
    for (int k = 0; k < n; k++) v[k] = bdr_vertices[static_cast<size_t>(b) * n + k];
  }
  int GetBdrAttribute(int b) const { return bdr_attributes[b]; }
  double GetElementSize(int e, int /*type*/ = 0) const { return elem_size.at(e); }
  void GetGlobalVertexIndices(Array<HYPRE_BigInt> &g) const {
    g.SetSize(nv);
    for (int i = 0; i < nv; i++) g[i] = global_vertex.empty() ? i : global_vertex[i];
  }
  int GetNFaceNeighbors() const { return static_cast<int>(nbrs.size()); }
  int GetFaceNbrGroup(int fn) const { return fn + 1; }
  int GetFaceNbrRank(int fn) const { return nbrs[fn].rank; }
  int GroupNQuadrilaterals(int g) const { return dim == 3 ? static_cast<int>(nbrs[g - 1].faces.size()) : 0; }
  int GroupNEdges(int g) const { return dim == 2 ? static_cast<int>(nbrs[g - 1].faces.size()) : 0; }
  void GroupQuadrilateral(int g, int i, int &face, int &o) const {
    face = encode(g, i);
    o = 0;
  }
  void GroupEdge(int g, int i, int &edge, int &o) const {
    edge = encode(g, i);
    o = 0;
  }
  void GetFaceVertices(int f, Array<int> &v) const { decode(f, v); }
  void GetEdgeVertices(int f, Array<int> &v) const { decode(f, v); }

 private:
  static int encode(int g, int i) { return (g << 20) | i; }
  void decode(int f, Array<int> &v) const {
    const std::vector<int> &fv = nbrs[(f >> 20) - 1].faces[f & ((1 << 20) - 1)];
    v.SetSize(static_cast<int>(fv.size()));
    for (int k = 0; k < v.Size(); k++) v[k] = fv[k];
  }
  ElementTransformation tr_;
};

}  // namespace mfem
#endif
