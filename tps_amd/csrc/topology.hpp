// Host-side face topology of a conforming quadrilateral / hexahedral mesh: what the reference
// obtains from mfem::ParMesh (faces, FaceElementTransformations orientations, boundary attributes,
// shared faces) and flattens into its indirection arrays (src/M2ulPhyS.cpp:816-1486).
//
// Output is element-centric: for every (element, local face) the slot of the NEIGHBOUR's trace
// record and a 3-bit orientation code, so that kernels never scatter.
//
// Local faces: f = 2*d + s is the face xi_d = s of the reference cube; its tangential axes (a,b)
// are the remaining axes in increasing order.  Orientation code o = swap | fa<<1 | fb<<2 maps MY
// tangential coordinates to the neighbour's:
//    swap=0: ta' = fa ? 1-ta : ta ; tb' = fb ? 1-tb : tb
//    swap=1: ta' = fa ? 1-tb : tb ; tb' = fb ? 1-ta : ta
#ifndef TPSRHS_TOPOLOGY_HPP_
#define TPSRHS_TOPOLOGY_HPP_

#include <algorithm>
#include <array>
#include <cstdint>
#include <stdexcept>
#include <string>
#include <vector>

#include "../../include/tpsrhs.h"

namespace tpsrhs {

struct Topology {
  int dim = 0, ne = 0, nfaces = 0;  // nfaces = 2*dim local faces per element
  std::vector<double> verts;        // [ne][2^dim][dim], LEXICOGRAPHIC corner order
  std::vector<int32_t> face_nbr;    // [ne*nfaces]: >=0 neighbour trace slot; <0: -(bc index + 1)
  std::vector<uint8_t> face_orient; // [ne*nfaces]
  // shared faces, in the caller's order (grouped by neighbour rank)
  int num_shared = 0;
  std::vector<int32_t> shared_slot;    // own slot (e*nfaces+f) of each shared face
  std::vector<uint8_t> shared_orient;  // my frame -> canonical frame
  std::vector<int> nbr_ranks;          // distinct neighbour ranks, in order of appearance
  std::vector<int64_t> nbr_offsets;    // shared-face offsets per neighbour (size nbr_ranks+1)
  int num_interior_pairs = 0, num_boundary = 0;
};

inline int lex_of_mfem(int dim, int v) {
  static const int q[4] = {0, 1, 3, 2};
  static const int h[8] = {0, 1, 3, 2, 4, 5, 7, 6};
  return dim == 2 ? q[v] : h[v];
}

// lexicographic corner indices of local face f listed in (ta,tb) corner order
inline void face_corners(int dim, int f, int *c) {
  const int d = f >> 1, s = f & 1;
  if (dim == 2) {
    const int a = 1 - d;
    for (int ta = 0; ta < 2; ta++) c[ta] = (s << d) | (ta << a);
  } else {
    const int a = (d == 0) ? 1 : 0, b = (d == 2) ? 1 : 2;
    for (int tb = 0; tb < 2; tb++)
      for (int ta = 0; ta < 2; ta++) c[ta + 2 * tb] = (s << d) | (ta << a) | (tb << b);
  }
}

// code mapping the frame whose corners (in its own (ta,tb) order) carry ids `mine` onto the frame
// whose corners carry ids `other`
inline uint8_t orient_code(int dim, const int *mine, const int *other) {
  const int nfv = 1 << (dim - 1);
  auto find = [&](int id) {
    for (int i = 0; i < nfv; i++)
      if (other[i] == id) return i;
    throw std::runtime_error("tpsrhs: face vertex mismatch");
  };
  if (dim == 2) return static_cast<uint8_t>(find(mine[0]) << 1);
  const int o = find(mine[0]), pa = find(mine[1]);
  const int oa = o & 1, ob = o >> 1;
  const int swap = ((pa & 1) != oa) ? 0 : 1;
  return static_cast<uint8_t>(swap | (oa << 1) | (ob << 2));
}

inline Topology build_topology(const tpsrhs_mesh &m, int num_bcs, const tpsrhs_bc *bcs) {
  Topology T;
  const int dim = m.dim;
  if (dim != 2 && dim != 3) throw std::runtime_error("tpsrhs: dim must be 2 or 3");
  const int nvpe = 1 << dim, nfv = 1 << (dim - 1), nlf = 2 * dim;
  T.dim = dim;
  T.ne = m.num_elements;
  T.nfaces = nlf;
  const int ne = T.ne;
  std::vector<int> ev(static_cast<size_t>(ne) * nvpe);
  T.verts.resize(static_cast<size_t>(ne) * nvpe * dim);
  for (int e = 0; e < ne; e++)
    for (int v = 0; v < nvpe; v++) {
      const int l = lex_of_mfem(dim, v);
      const int id = m.elem_vertices[static_cast<size_t>(e) * nvpe + v];
      if (id < 0 || id >= m.num_vertices) throw std::runtime_error("tpsrhs: vertex id out of range");
      ev[static_cast<size_t>(e) * nvpe + l] = id;
      for (int d = 0; d < dim; d++)
        T.verts[(static_cast<size_t>(e) * nvpe + l) * dim + d] = m.elem_coords[(static_cast<size_t>(e) * nvpe + v) * dim + d];
    }

  struct Rec {
    std::array<int, 4> key;
    int kind;  // 0 element face, 1 boundary record, 2 shared record
    int e, f;  // element/local face, or index of the boundary / shared record
  };
  std::vector<Rec> recs;
  recs.reserve(static_cast<size_t>(ne) * nlf + m.num_bdr_faces + m.num_shared_faces);
  auto make_key = [&](const int *ids) {
    std::array<int, 4> k = {-1, -1, -1, -1};
    for (int i = 0; i < nfv; i++) k[i] = ids[i];
    std::sort(k.begin(), k.begin() + nfv);
    return k;
  };
  for (int e = 0; e < ne; e++)
    for (int f = 0; f < nlf; f++) {
      int c[4], g[4];
      face_corners(dim, f, c);
      for (int i = 0; i < nfv; i++) g[i] = ev[static_cast<size_t>(e) * nvpe + c[i]];
      Rec r{make_key(g), 0, e, f};
      for (int i = 1; i < nfv; i++)
        if (r.key[i] == r.key[i - 1])
          throw std::runtime_error("tpsrhs: degenerate face (periodic direction with fewer than 3 cells?)");
      recs.push_back(r);
    }
  for (int b = 0; b < m.num_bdr_faces; b++) recs.push_back(Rec{make_key(&m.bdr_vertices[static_cast<size_t>(b) * nfv]), 1, b, 0});
  for (int s = 0; s < m.num_shared_faces; s++)
    recs.push_back(Rec{make_key(&m.shared_vertices[static_cast<size_t>(s) * nfv]), 2, s, 0});
  std::sort(recs.begin(), recs.end(), [](const Rec &a, const Rec &b) {
    if (a.key != b.key) return a.key < b.key;
    if (a.kind != b.kind) return a.kind < b.kind;
    if (a.e != b.e) return a.e < b.e;
    return a.f < b.f;
  });

  T.face_nbr.assign(static_cast<size_t>(ne) * nlf, INT32_MIN);
  T.face_orient.assign(static_cast<size_t>(ne) * nlf, 0);
  T.num_shared = m.num_shared_faces;
  T.shared_slot.assign(T.num_shared, -1);
  T.shared_orient.assign(T.num_shared, 0);
  for (int s = 0; s < T.num_shared; s++) {
    const int r = m.shared_neighbor_rank[s];
    if (T.nbr_ranks.empty() || T.nbr_ranks.back() != r) {
      for (int q : T.nbr_ranks)
        if (q == r) throw std::runtime_error("tpsrhs: shared faces must be grouped by neighbour rank");
      T.nbr_ranks.push_back(r);
      T.nbr_offsets.push_back(s);
    }
  }
  T.nbr_offsets.push_back(T.num_shared);

  auto corner_ids = [&](int e, int f, int *g) {
    int c[4];
    face_corners(dim, f, c);
    for (int i = 0; i < nfv; i++) g[i] = ev[static_cast<size_t>(e) * nvpe + c[i]];
  };

  size_t i = 0;
  while (i < recs.size()) {
    size_t j = i;
    while (j < recs.size() && recs[j].key == recs[i].key) j++;
    // group [i, j): element faces first (kind 0), then boundary, then shared
    int nel = 0;
    while (i + nel < j && recs[i + nel].kind == 0) nel++;
    if (nel == 0) throw std::runtime_error("tpsrhs: boundary/shared face not found among element faces");
    if (nel > 2) throw std::runtime_error("tpsrhs: face shared by more than two elements");
    const Rec *bd = nullptr, *sh = nullptr;
    for (size_t k = i + nel; k < j; k++) {
      if (recs[k].kind == 1) bd = &recs[k];
      if (recs[k].kind == 2) sh = &recs[k];
    }
    if (nel == 2) {
      const Rec &A = recs[i], &B = recs[i + 1];
      int ga[4], gb[4];
      corner_ids(A.e, A.f, ga);
      corner_ids(B.e, B.f, gb);
      T.face_nbr[static_cast<size_t>(A.e) * nlf + A.f] = B.e * nlf + B.f;
      T.face_orient[static_cast<size_t>(A.e) * nlf + A.f] = orient_code(dim, ga, gb);
      T.face_nbr[static_cast<size_t>(B.e) * nlf + B.f] = A.e * nlf + A.f;
      T.face_orient[static_cast<size_t>(B.e) * nlf + B.f] = orient_code(dim, gb, ga);
      T.num_interior_pairs++;
      // a boundary element on an interior face is an "interior boundary" (periodic meshes): ignored
    } else if (sh) {
      const Rec &A = recs[i];
      int ga[4], gc[4];
      corner_ids(A.e, A.f, ga);
      // canonical frame from the caller's vertex list (ascending global id): origin = first entry;
      // a-axis towards the face-adjacent corner that comes first in the list
      const int *lst = &m.shared_vertices[static_cast<size_t>(sh->e) * nfv];
      if (dim == 2) {
        gc[0] = lst[0];
        gc[1] = lst[1];
      } else {
        auto pos = [&](int id) {
          for (int k = 0; k < nfv; k++)
            if (lst[k] == id) return k;
          return 99;
        };
        int o = -1;
        for (int k = 0; k < 4; k++)
          if (ga[k] == lst[0]) o = k;
        if (o < 0) throw std::runtime_error("tpsrhs: shared face vertices do not match the element");
        const int n1 = ga[o ^ 1], n2 = ga[o ^ 2], dg = ga[o ^ 3];
        const bool first = pos(n1) < pos(n2);
        gc[0] = lst[0];
        gc[1] = first ? n1 : n2;
        gc[2] = first ? n2 : n1;
        gc[3] = dg;
      }
      const int s = sh->e;
      T.shared_slot[s] = A.e * nlf + A.f;
      T.shared_orient[s] = orient_code(dim, ga, gc);
      T.face_nbr[static_cast<size_t>(A.e) * nlf + A.f] = ne * nlf + s;  // halo slot
      T.face_orient[static_cast<size_t>(A.e) * nlf + A.f] = T.shared_orient[s];
    } else {
      const Rec &A = recs[i];
      if (!bd) throw std::runtime_error("tpsrhs: exterior face without a boundary attribute");
      const int attr = m.bdr_attributes[bd->e];
      int idx = -1;
      for (int k = 0; k < num_bcs; k++)
        if (bcs[k].attribute == attr) idx = k;
      if (idx < 0) throw std::runtime_error("tpsrhs: no boundary condition for attribute " + std::to_string(attr));
      T.face_nbr[static_cast<size_t>(A.e) * nlf + A.f] = -(idx + 1);
      T.num_boundary++;
    }
    i = j;
  }
  for (int s = 0; s < T.num_shared; s++)
    if (T.shared_slot[s] < 0) throw std::runtime_error("tpsrhs: shared face is not an exterior face of this rank");
  return T;
}

}  // namespace tpsrhs
#endif
