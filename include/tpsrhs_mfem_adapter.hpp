// tpsrhs_mfem_adapter.hpp -- the class a TPS maintainer puts next to src/rhs_operator.hpp:
//   class RHSoperatorHIP : public mfem::TimeDependentOperator
// with the one method the time integrator calls, Mult(const Vector &x, Vector &y) const
// (src/rhs_operator.hpp:59,157), forwarding to the C ABI of tpsrhs.h.  Header-only; needs <mfem.hpp>
// (MFEM >= 4.4, the reference's requirement, configure.ac:198) or, in this repository's tests,
// tests/mock_mfem/mfem.hpp -- a minimal stand-in of the few MFEM classes used here (TEST INFRASTRUCTURE: this
// image has no MFEM; the mock exists so that this file is compiled and its Mult is run from C++).
//
// What the adapter does at construction (the role of RHSoperator::RHSoperator + M2ulPhyS::initIndirectionArrays,
// src/rhs_operator.cpp:39-322, src/M2ulPhyS.cpp:816-1486):
//   * elements: topological vertices + the coordinates of each element's own corners (periodic meshes keep
//     their geometry), boundary elements + attributes;
//   * shared faces of the ParMesh, grouped by neighbour rank, vertices ordered by ascending GLOBAL vertex id
//     -- the contract of tpsrhs_mesh::shared_vertices;
//   * the PODs of the physics and the boundary conditions are passed through (they are field-for-field images
//     of src/dataStructures.hpp:537-729; pack them from RunConfiguration as INTEGRATION.md shows).
#ifndef TPSRHS_MFEM_ADAPTER_HPP_
#define TPSRHS_MFEM_ADAPTER_HPP_

#include <algorithm>
#include <functional>
#include <stdexcept>
#include <string>
#include <utility>
#include <vector>

#include "tpsrhs.h"

namespace tps_hip {

// the exchange hooks: the integration supplies them (GPU-aware MPI, or the RCCL library of tpsrhs_rccl.h)
struct Exchange {
  tpsrhs_halo_fn halo = nullptr;
  void *halo_ctx = nullptr;
  tpsrhs_reduce_fn reduce = nullptr;
  void *reduce_ctx = nullptr;
  std::function<double(double)> max_over_ranks;  // MPI_Allreduce(MAX) of max_char_speed, src/rhs_operator.cpp:558
};

class RHSoperatorHIP : public mfem::TimeDependentOperator {
 public:
  // vsize = num_equation * NDofs (vfes->GetVSize() of the reference); max_char_speed: the member of M2ulPhyS the
  // reference's operator writes through a reference too (src/rhs_operator.hpp:71)
  RHSoperatorHIP(mfem::ParMesh *mesh, int vsize, const tpsrhs_disc &disc, const tpsrhs_physics &physics,
                 const std::vector<tpsrhs_bc> &bcs, double &max_char_speed, int device = 0, void *stream = nullptr,
                 Exchange exchange = Exchange())
      : mfem::TimeDependentOperator(vsize), max_char_speed_(max_char_speed), exchange_(std::move(exchange)) {
    const int dim = mesh->Dimension(), nv = 1 << dim, nfv = 1 << (dim - 1);
    // ---- elements
    std::vector<int> ev, bv, battr;
    std::vector<double> ex;
    mfem::Array<int> v;
    for (int e = 0; e < mesh->GetNE(); e++) {
      mesh->GetElementVertices(e, v);
      if (v.Size() != nv) throw std::runtime_error("RHSoperatorHIP: quadrilateral / hexahedral meshes only");
      const mfem::DenseMatrix &pm = mesh->GetElementTransformation(e)->GetPointMat();  // dim x nv, own corners
      for (int k = 0; k < nv; k++) {
        ev.push_back(v[k]);
        for (int d = 0; d < dim; d++) ex.push_back(pm(d, k));
      }
    }
    // ---- boundary elements
    for (int b = 0; b < mesh->GetNBE(); b++) {
      mesh->GetBdrElementVertices(b, v);
      for (int k = 0; k < nfv; k++) bv.push_back(v[k]);
      battr.push_back(mesh->GetBdrAttribute(b));
    }
    // ---- shared faces: (neighbour rank, sorted global vertex ids) -> local ids in that order; the list is
    //      sorted by (rank, global key), which both sides of a pair compute identically
    mfem::Array<HYPRE_BigInt> gvid;
    mesh->GetGlobalVertexIndices(gvid);
    struct Shared {
      int rank;
      std::vector<long long> key;
      std::vector<int> loc;
    };
    std::vector<Shared> shared;
    for (int fn = 0; fn < mesh->GetNFaceNeighbors(); fn++) {
      const int group = mesh->GetFaceNbrGroup(fn), rank = mesh->GetFaceNbrRank(fn);
      const int nfaces = (dim == 3) ? mesh->GroupNQuadrilaterals(group) : mesh->GroupNEdges(group);
      for (int i = 0; i < nfaces; i++) {
        int face, orient;
        if (dim == 3)
          mesh->GroupQuadrilateral(group, i, face, orient);
        else
          mesh->GroupEdge(group, i, face, orient);
        if (dim == 3)
          mesh->GetFaceVertices(face, v);
        else
          mesh->GetEdgeVertices(face, v);
        std::vector<std::pair<long long, int>> kv;
        for (int k = 0; k < nfv; k++) kv.push_back({static_cast<long long>(gvid[v[k]]), v[k]});
        std::sort(kv.begin(), kv.end());
        Shared s;
        s.rank = rank;
        for (auto &p : kv) {
          s.key.push_back(p.first);
          s.loc.push_back(p.second);
        }
        shared.push_back(std::move(s));
      }
    }
    std::sort(shared.begin(), shared.end(), [](const Shared &a, const Shared &b) {
      return a.rank != b.rank ? a.rank < b.rank : a.key < b.key;
    });
    std::vector<int> sv, srank;
    for (const Shared &s : shared) {
      for (int l : s.loc) sv.push_back(l);
      srank.push_back(s.rank);
    }
    if (!shared.empty() && !exchange_.halo)
      throw std::runtime_error("RHSoperatorHIP: the mesh is partitioned but no halo exchange was supplied");

    // Mesh::GetElementSize(e, 1) [MFEM]: the grid scale of the sub-grid scale models and of the viscous sponge
    // (src/rhs_operator.cpp:145-156); exact on curved or periodic meshes, where the corner coordinates are not
    std::vector<double> esize(mesh->GetNE());
    for (int e = 0; e < mesh->GetNE(); e++) esize[e] = mesh->GetElementSize(e, 1);

    tpsrhs_mesh m = {};
    m.elem_size = esize.data();
    m.dim = dim;
    m.num_vertices = mesh->GetNV();
    m.num_elements = mesh->GetNE();
    m.elem_vertices = ev.data();
    m.elem_coords = ex.data();
    m.num_bdr_faces = static_cast<int>(battr.size());
    m.bdr_vertices = bv.data();
    m.bdr_attributes = battr.data();
    m.num_shared_faces = static_cast<int>(srank.size());
    m.shared_vertices = sv.data();
    m.shared_neighbor_rank = srank.data();
    tpsrhs_runtime rt = {};
    rt.device = device;
    rt.stream = stream;
    rt.halo = exchange_.halo;
    rt.halo_ctx = exchange_.halo_ctx;
    rt.reduce = exchange_.reduce;
    rt.reduce_ctx = exchange_.reduce_ctx;
    const int st = tpsrhs_create(&m, &disc, &physics, static_cast<int>(bcs.size()), bcs.empty() ? nullptr : bcs.data(),
                                 &rt, &h_);
    if (st != TPSRHS_OK)  // the reference asserts / exits / MPI_Aborts here; the caller decides
      throw std::runtime_error(std::string(tpsrhs_status_string(st)) + ": " + tpsrhs_last_error());
    if (tpsrhs_height(h_) != vsize) {
      tpsrhs_destroy(h_);
      throw std::runtime_error("RHSoperatorHIP: vsize does not match num_equation * NDofs of the operator");
    }
  }
  ~RHSoperatorHIP() override { tpsrhs_destroy(h_); }
  RHSoperatorHIP(const RHSoperatorHIP &) = delete;
  RHSoperatorHIP &operator=(const RHSoperatorHIP &) = delete;

  // src/rhs_operator.hpp:157, src/rhs_operator.cpp:343-464.  With mfem::Device enabled the vectors live in device
  // memory and nothing is copied; otherwise the host arrays are staged (PCIe-inclusive, for bring-up only).
  void Mult(const mfem::Vector &x, mfem::Vector &y) const override {
    double mcs = 0.0;
    const int st = mfem::Device::IsEnabled() ? tpsrhs_mult(h_, x.Read(), y.Write(), this->GetTime(), &mcs)
                                             : tpsrhs_mult_host(h_, x.HostRead(), y.HostWrite(), this->GetTime(), &mcs);
    if (st != TPSRHS_OK) throw std::runtime_error(std::string(tpsrhs_status_string(st)) + ": " + tpsrhs_last_error());
    max_char_speed_ = exchange_.max_over_ranks ? exchange_.max_over_ranks(mcs) : mcs;
  }
  // RHSoperator::updateGradients / the Up and gradUp grid functions (src/rhs_operator.cpp:623-713); device arrays
  void updateGradients(const mfem::Vector &x) const { check(tpsrhs_update_gradients(h_, x.Read())); }
  void getPrimitives(mfem::Vector &up) const { check(tpsrhs_get_primitives(h_, up.Write())); }
  void getGradients(mfem::Vector &gradUp) const { check(tpsrhs_get_gradients(h_, gradUp.Write())); }
  // plasma_conductivity_ of SourceTerm (src/source_term.cpp:184,196) for the EM solver of the coupled runs
  void getPlasmaConductivity(const mfem::Vector &x, mfem::Vector &sigma) const {
    check(tpsrhs_get_plasma_conductivity(h_, x.Read(), sigma.Write()));
  }
  // the optional terms of the reference's constructor: forcing.Append(...) (src/rhs_operator.cpp:101-166), the
  // joule_heating_ and distance_ grid functions (device arrays owned by the caller; NULL-sized vector = off)
  void setForcing(const tpsrhs_forcing *f) { check(tpsrhs_set_forcing(h_, f)); }
  void setJouleHeating(const mfem::Vector *jh) { check(tpsrhs_set_joule_heating(h_, jh ? jh->Read() : nullptr)); }
  void setMixingLength(const mfem::Vector *distance, const tpsrhs_mixing_length &prm) {
    check(tpsrhs_set_mixing_length(h_, distance ? distance->Read() : nullptr, &prm));
  }
  tpsrhs_handle handle() const { return h_; }
  int num_equation() const { return tpsrhs_num_equation(h_); }

 private:
  static void check(int st) {
    if (st != TPSRHS_OK) throw std::runtime_error(std::string(tpsrhs_status_string(st)) + ": " + tpsrhs_last_error());
  }
  tpsrhs_handle h_ = nullptr;
  double &max_char_speed_;
  Exchange exchange_;
};

}  // namespace tps_hip
#endif
