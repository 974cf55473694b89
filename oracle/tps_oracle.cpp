// TEST INFRASTRUCTURE -- CPU oracle, never shipped, never on the product path.
//
// CPU restatement of RHSoperator::Mult (pecos/tps, src/rhs_operator.cpp:343-464) in the reference's
// own dense formulation: per-element Ke (src/gradients.cpp:84-133), inverse mass matrices
// (src/rhs_operator.cpp:173-224), the assembled (v, grad w) block matrix
// (src/domain_integrator.cpp:45-99), per-face quadrature loops with CalcShape at every point
// (src/face_integrator.cpp:194-352, src/faceGradientIntegration.cpp:40-140,
// src/BCintegrator.cpp:295-441).  No sum factorisation, no fusion: it doubles as the timed CPU
// baseline ("port").  Only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg may
// load this library.
//
// PARITY UNPINNED for the assembled operator: the reference's golden solutions are git-LFS
// pointers and MFEM is absent.  Pinned pieces: see oracle/README.md.
#include <omp.h>

#include <cstdint>
#include <cstring>
#include <memory>
#include <string>

#include "fe.hpp"
#include "physics.hpp"
#include "plasma.hpp"
#include "lte.hpp"

namespace tpsoracle {

struct Operator {
  Mesh mesh;
  Element fe;
  tpsrhs_disc disc;
  tpsrhs_physics phys;
  int dim, nvel, neq, dof, ne;
  int64_t ndofs;
  bool axisym;

  std::unique_ptr<GasMixture> mixture;
  std::unique_ptr<TransportProperties> transport;
  std::unique_ptr<Fluxes> fluxes;
  std::unique_ptr<RiemannSolver> rsolver;
  std::unique_ptr<SourceTerm> source;
  LinearTable lteNec;  // net emission table of the table gas' radiation sink
  std::map<int, std::unique_ptr<BoundaryCondition>> bcs;
  // ConstantPressureGradient / SpongeZone / HeatSource / JouleHeating (src/rhs_operator.cpp:101-166)
  bool has_forcing = false;
  tpsrhs_forcing forcing_in;
  std::vector<std::vector<double>> spongeSigma;        // sigma grid function of each zone (:553-606)
  std::vector<std::vector<double>> spongeRadial;       // unit radial vector per node (annulus)
  std::vector<std::vector<int64_t>> spongePlaneNodes;  // nodesInMixedOutPlane (:545-606)
  std::vector<std::vector<int64_t>> heatNodes;         // nodeList_ of each HeatSource (:890-917)
  std::vector<std::vector<int64_t>> scalarNodes;       // psData_[i]->nodes of PassiveScalar (:795-818)
  std::vector<double> joule;                           // joule_heating_ grid function (empty: none)
  std::vector<double> distance;                        // distance_ grid function (empty: none)
  std::unique_ptr<tpsoracle::MixingLengthTransport> mixlen;  // wraps `transport` when useMixingLength (M2ulPhyS.cpp:265-277)

  RuleND volRule;   // order 2p
  RuleND faceRule;  // order OrderW + 2p on the reference segment/square
  std::vector<Dense> Ke, MeInv, Aflux;
  std::vector<double> coords;  // byNODES [n + d*ndofs]
  std::vector<double> elSize;  // per element: GetElementSize(e, 1) / order
  std::vector<double> Up, gradUp;
  double max_char_speed = 0.0;
  // scratch
  std::vector<double> z, flux, faceContrib;

  void setup(const tpsrhs_mesh *m, const tpsrhs_disc *d, const tpsrhs_physics *p, int nbc, const tpsrhs_bc *bc) {
    disc = *d;
    phys = *p;
    dim = m->dim;
    axisym = d->axisymmetric != 0;
    if (axisym && dim != 2) throw std::runtime_error("axisymmetric requires dim == 2");
    nvel = axisym ? 3 : dim;
    mesh.build(m->dim, m->num_vertices, m->num_elements, m->elem_vertices, m->elem_coords, m->num_bdr_faces,
               m->bdr_vertices, m->bdr_attributes);
    ne = mesh.ne;
    fe.init(dim, d->order, d->basis_type);
    dof = fe.dof;
    ndofs = static_cast<int64_t>(ne) * dof;

    if (p->working_fluid == TPSRHS_DRY_AIR) {
      mixture.reset(new DryAir(p->dry_air, dim, nvel));
      transport.reset(new DryAirTransport(mixture.get(), p->dry_air));
    } else if (p->working_fluid == TPSRHS_USER_DEFINED) {
      PerfectMixture *pm = new PerfectMixture(p->mixture, dim, nvel);
      mixture.reset(pm);
      transport.reset(make_transport(pm, *p));
    } else if (p->working_fluid == TPSRHS_LTE_FLUID) {  // src/M2ulPhyS.cpp:164-255, table_dim == 1
      mixture.reset(new LteMixture(p->lte, dim, nvel));
      transport.reset(new LteTransport(mixture.get(), p->lte));
      if (p->radiation.model == TPSRHS_NET_EMISSION) lteNec.init(p->radiation.nec_table);
    } else {
      throw std::runtime_error("unknown working fluid");
    }
    neq = mixture->num_equation;
    fluxes.reset(new Fluxes(mixture.get(), p->eq_system, transport.get(), neq, dim, axisym));
    fluxes->sgs_model_type_ = p->sgs.model_type;  // config.GetSgsModelType() etc., src/fluxes.cpp:66-69
    fluxes->sgs_model_const_ = p->sgs.model_const > 0.0 ? p->sgs.model_const  // defaults: src/M2ulPhyS.cpp:2693-2698
                               : (p->sgs.model_type == 1 ? 0.12 : (p->sgs.model_type == 2 ? 0.135 : 0.0));
    fluxes->sgs_model_floor_ = p->sgs.model_floor;
    fluxes->vsd_ = p->visc_sponge;
    if (fluxes->vsd_.enabled) {  // "ensure normal is actually a unit normal", the host constructor src/fluxes.cpp:77-90
      double Nmag = 0;
      for (int d = 0; d < dim; d++) Nmag += fluxes->vsd_.normal[d] * fluxes->vsd_.normal[d];
      Nmag = std::sqrt(Nmag);
      for (int d = 0; d < dim; d++) fluxes->vsd_.normal[d] /= Nmag;
    }
    if (p->sgs.model_type > 0 && dim != 3) throw std::runtime_error("sgs models index three directions (src/fluxes.cpp:524-529)");
    // elSize (src/rhs_operator.cpp:145-156): Mesh::GetElementSize(e, 1) / order, the smallest singular value of the
    // Jacobian at the element centre [MFEM]; one-sided Jacobi (Hestenes) on the columns of J
    elSize.assign(m->num_elements, 0.0);
    for (int e = 0; e < m->num_elements; e++) {
      double h;
      if (m->elem_size) {
        h = m->elem_size[e];
      } else {
        const double xi[3] = {0.5, 0.5, 0.5};
        double x[3], J[9];
        mesh.transform(e, xi, x, J);
        h = min_singular_value(dim, J);
      }
      elSize[e] = h / d->order;
    }
    if (d->use_roe && (dim != 2 || axisym || p->working_fluid != TPSRHS_DRY_AIR))
      throw std::runtime_error("Eval_Roe: 2-D, single species, not axisymmetric (src/riemann_solver.cpp:117-206)");
    rsolver.reset(new RiemannSolver(neq, mixture.get(), fluxes.get(), d->use_roe != 0));
    if (p->working_fluid == TPSRHS_USER_DEFINED) {  // src/rhs_operator.cpp:125-129 (the table gas: lteSource below)
      source.reset(new SourceTerm(dim, neq, static_cast<PerfectMixture *>(mixture.get()), transport.get(), *p));
    }
    for (int i = 0; i < nbc; i++)
      bcs[bc[i].attribute].reset(
          new BoundaryCondition(bc[i], mixture.get(), fluxes.get(), rsolver.get(), axisym, d->use_bc_in_grad != 0));
    for (const Face &F : mesh.faces)
      if (F.e2 < 0 && bcs.find(F.attr) == bcs.end())
        throw std::runtime_error("no boundary condition for attribute " + std::to_string(F.attr));

    const int p_ = d->order;
    volRule.init(dim, segment_rule(d->int_rule_type, 2 * p_));
    const int orderW = dim * 1 - 1;  // IsoparametricTransformation::OrderW, Qk order-1 geometry
    faceRule.init(dim - 1, segment_rule(d->int_rule_type, orderW + 2 * p_));

    // node coordinates (mesh->GetNodes(*coordsDof), src/rhs_operator.cpp:139-142)
    coords.assign(static_cast<size_t>(ndofs) * dim, 0.0);
    for (int e = 0; e < ne; e++)
      for (int k = 0; k < dof; k++) {
        double xi[3], x[3], J[9];
        fe.node_ref(k, xi);
        mesh.transform(e, xi, x, J);
        for (int dd = 0; dd < dim; dd++) coords[static_cast<size_t>(e) * dof + k + dd * ndofs] = x[dd];
      }

    assemble();
    Up.assign(static_cast<size_t>(neq) * ndofs, 0.0);
    gradUp.assign(static_cast<size_t>(neq) * ndofs * dim, 0.0);
    z.assign(static_cast<size_t>(neq) * ndofs, 0.0);
    flux.assign(static_cast<size_t>(neq) * ndofs * dim, 0.0);
    faceContrib.assign(static_cast<size_t>(neq) * ndofs * dim, 0.0);
  }

  void assemble() {
    Ke.resize(ne);
    MeInv.resize(ne);
    Aflux.resize(ne);
#pragma omp parallel for schedule(static)
    for (int e = 0; e < ne; e++) {
      std::vector<double> shape(dof), dshape(static_cast<size_t>(dof) * dim), dphys(static_cast<size_t>(dof) * dim);
      Dense K(dof, dim * dof), M(dof, dof), A(dof, dim * dof);
      for (int q = 0; q < volRule.npts; q++) {
        const double *xi = &volRule.x[q * dim];
        double x[3], J[9], Ji[9];
        mesh.transform(e, xi, x, J);
        const double det = det_and_inverse(dim, J, Ji);
        fe.calc_shape(xi, shape.data());
        fe.calc_dshape(xi, dshape.data());
        // CalcPhysDShape: dphys(k,d) = sum_m dshape(k,m) Jinv(m,d)
        for (int k = 0; k < dof; k++)
          for (int dd = 0; dd < dim; dd++) {
            double s = 0.0;
            for (int mm = 0; mm < dim; mm++) s += dshape[k + mm * dof] * Ji[mm + dd * dim];
            dphys[k + dd * dof] = s;
          }
        const double detJac = det * volRule.w[q];
        // Ke (src/gradients.cpp:104-121)
        for (int dd = 0; dd < dim; dd++)
          for (int k = 0; k < dof; k++)
            for (int j = 0; j < dof; j++) K(j, k + dd * dof) += shape[j] * dphys[k + dd * dof] * detJac;
        // MassIntegrator at order 2p (src/rhs_operator.cpp:179-185); weight r if axisymmetric (:198-201)
        const double mw = axisym ? detJac * x[0] : detJac;
        for (int j = 0; j < dof; j++)
          for (int k = 0; k < dof; k++) M(k, j) += shape[k] * shape[j] * mw;
        // DomainIntegrator (src/domain_integrator.cpp:71-97): shape*w (*r), dshapedr * adj(J)
        double sw = volRule.w[q];
        if (axisym) sw *= x[0];
        for (int dd = 0; dd < dim; dd++)
          for (int j = 0; j < dof; j++) {
            double dx = 0.0;  // dshapedx(j,dd) = sum_m dshapedr(j,m) adjJ(m,dd), adjJ = det * Jinv
            for (int mm = 0; mm < dim; mm++) dx += dshape[j + mm * dof] * (det * Ji[mm + dd * dim]);
            for (int k = 0; k < dof; k++) A(j, k + dd * dof) += shape[k] * sw * dx;
          }
      }
      invert_dense(M);
      Ke[e] = std::move(K);
      MeInv[e] = std::move(M);
      Aflux[e] = std::move(A);
    }
    if (axisym) {
      // the gradient uses the plain (unweighted) inverse mass matrix (src/rhs_operator.cpp:245-246
      // passes Me_inv); rebuild it separately
      MeInvGrad.resize(ne);
#pragma omp parallel for schedule(static)
      for (int e = 0; e < ne; e++) {
        std::vector<double> shape(dof);
        Dense M(dof, dof);
        for (int q = 0; q < volRule.npts; q++) {
          const double *xi = &volRule.x[q * dim];
          double x[3], J[9], Ji[9];
          mesh.transform(e, xi, x, J);
          const double det = det_and_inverse(dim, J, Ji);
          fe.calc_shape(xi, shape.data());
          const double mw = det * volRule.w[q];
          for (int j = 0; j < dof; j++)
            for (int k = 0; k < dof; k++) M(k, j) += shape[k] * shape[j] * mw;
        }
        invert_dense(M);
        MeInvGrad[e] = std::move(M);
      }
    }
  }
  std::vector<Dense> MeInvGrad;
  const Dense &gradMassInv(int e) const { return axisym ? MeInvGrad[e] : MeInv[e]; }

  // src/rhs_operator.cpp:642-649
  void updatePrimitives(const double *x) {
    OmpGuard guard;
#pragma omp parallel for schedule(static)
    for (int64_t i = 0; i < ndofs; i++) try {
      double s[MAXEQ], pr[MAXEQ];
      for (int eq = 0; eq < neq; eq++) s[eq] = x[i + eq * ndofs];
      mixture->GetPrimitivesFromConservatives(s, pr);
      for (int eq = 0; eq < neq; eq++) Up[i + eq * ndofs] = pr[eq];
    } catch (const std::exception &e) {
      guard.capture(e);
    }
    guard.rethrow();
  }

  // per-face geometric data at a face quadrature point
  struct FacePoint {
    double xi1[3], xi2[3], x[3], nor[3], w;
  };
  void facePoint(const Face &F, int q, FacePoint &fp) const {
    const double *t = &faceRule.x[q * (dim - 1)];
    double J[9];
    mesh.face_ref_point(F.f1, t, fp.xi1);
    mesh.transform(F.e1, fp.xi1, fp.x, J);
    mesh.face_normal(F.f1, J, fp.nor);
    fp.w = faceRule.w[q];
    if (F.e2 >= 0) {
      double t2[2] = {0.0, 0.0};
      Mesh::map_tangent(F, dim, t, t2);
      mesh.face_ref_point(F.f2, t2, fp.xi2);
    }
  }

  // src/gradients.cpp:144-232
  void computeGradients() {
    const int64_t N = ndofs;
    // volume part: gradUp = -(Ke Up)
#pragma omp parallel for schedule(static)
    for (int e = 0; e < ne; e++) {
      const Dense &K = Ke[e];
      for (int eq = 0; eq < neq; eq++)
        for (int d = 0; d < dim; d++)
          for (int j = 0; j < dof; j++) {
            double s = 0.0;
            for (int k = 0; k < dof; k++) s += K(j, k + d * dof) * Up[static_cast<int64_t>(e) * dof + k + eq * N];
            gradUp[static_cast<int64_t>(e) * dof + j + eq * N + d * neq * N] = -s;
          }
    }
    // face part (GradFaceIntegrator::AssembleFaceVector, src/faceGradientIntegration.cpp:40-140)
    std::fill(faceContrib.begin(), faceContrib.end(), 0.0);
    const int nf = static_cast<int>(mesh.faces.size());
    std::vector<double> fbuf(static_cast<size_t>(nf) * 2 * dof * neq * dim, 0.0);
#pragma omp parallel for schedule(dynamic, 16)
    for (int f = 0; f < nf; f++) {
      const Face &F = mesh.faces[f];
      std::vector<double> shape1(dof), shape2(dof);
      double *el1 = &fbuf[static_cast<size_t>(f) * 2 * dof * neq * dim];
      double *el2 = el1 + static_cast<size_t>(dof) * neq * dim;
      for (int q = 0; q < faceRule.npts; q++) {
        FacePoint fp;
        facePoint(F, q, fp);
        fe.calc_shape(fp.xi1, shape1.data());
        double iUp1[MAXEQ], iUp2[MAXEQ], mean[MAXEQ];
        for (int eq = 0; eq < neq; eq++) {
          double s = 0.0;
          for (int k = 0; k < dof; k++) s += Up[static_cast<int64_t>(F.e1) * dof + k + eq * N] * shape1[k];
          iUp1[eq] = s;
        }
        if (F.e2 < 0) {
          shape2 = shape1;
          if (disc.use_bc_in_grad) {
            bcs.at(F.attr)->computeBdrPrimitiveStateForGradient(iUp1, iUp2);
          } else {
            for (int eq = 0; eq < neq; eq++) iUp2[eq] = iUp1[eq];
          }
        } else {
          fe.calc_shape(fp.xi2, shape2.data());
          for (int eq = 0; eq < neq; eq++) {
            double s = 0.0;
            for (int k = 0; k < dof; k++) s += Up[static_cast<int64_t>(F.e2) * dof + k + eq * N] * shape2[k];
            iUp2[eq] = s;
          }
        }
        for (int eq = 0; eq < neq; eq++) mean[eq] = 0.5 * iUp1[eq] + 0.5 * iUp2[eq];
        double nor[3];
        for (int d = 0; d < dim; d++) nor[d] = fp.nor[d] * fp.w;
        for (int d = 0; d < dim; d++)
          for (int eq = 0; eq < neq; eq++) {
            const double du1n = (mean[eq] - iUp1[eq]) * nor[d];
            const double du2n = (iUp2[eq] - mean[eq]) * nor[d];
            for (int k = 0; k < dof; k++) {
              el1[k + (eq + d * neq) * dof] += shape1[k] * du1n;
              el2[k + (eq + d * neq) * dof] += shape2[k] * du2n;
            }
          }
      }
    }
    // scatter (ParNonlinearForm::Mult adds elvect of element 1 and, for interior faces, element 2)
#pragma omp parallel for schedule(static)
    for (int e = 0; e < ne; e++)
      for (int lf = 0; lf < 2 * dim; lf++) {
        const int f = mesh.elem_faces[e][lf];
        const Face &F = mesh.faces[f];
        const bool first = (F.e1 == e && F.f1 == lf);
        const double *el = &fbuf[static_cast<size_t>(f) * 2 * dof * neq * dim] +
                           (first ? 0 : static_cast<size_t>(dof) * neq * dim);
        for (int d = 0; d < dim; d++)
          for (int eq = 0; eq < neq; eq++)
            for (int k = 0; k < dof; k++)
              faceContrib[static_cast<int64_t>(e) * dof + k + eq * N + d * neq * N] += el[k + (eq + d * neq) * dof];
      }
    // inverse mass (src/gradients.cpp:198-229)
#pragma omp parallel for schedule(static)
    for (int e = 0; e < ne; e++) {
      const Dense &Mi = gradMassInv(e);
      std::vector<double> rhs(dof), aux(dof);
      for (int d = 0; d < dim; d++)
        for (int eq = 0; eq < neq; eq++) {
          const int64_t off = static_cast<int64_t>(e) * dof + eq * N + d * neq * N;
          for (int k = 0; k < dof; k++) rhs[k] = -gradUp[off + k] + faceContrib[off + k];
          for (int j = 0; j < dof; j++) {
            double s = 0.0;
            for (int k = 0; k < dof; k++) s += Mi(j, k) * rhs[k];
            aux[j] = s;
          }
          for (int k = 0; k < dof; k++) gradUp[off + k] = aux[k];
        }
    }
  }

  void interpGrad(int e, const double *shape, double *g) const {
    const int64_t N = ndofs;
    for (int eq = 0; eq < neq; eq++)
      for (int d = 0; d < dim; d++) {
        double s = 0.0;
        for (int k = 0; k < dof; k++) s += gradUp[static_cast<int64_t>(e) * dof + k + eq * N + d * neq * N] * shape[k];
        g[eq + d * neq] = s;
      }
  }

  // BCintegrator::updateBCMean -> InletBC/OutletBC::updateMean (src/outletBC.cpp:470-561, src/inletBC.cpp:482-566):
  // unweighted mean of the interpolated primitives over the boundary quadrature points of each non-reflecting
  // patch; on the first call the boundary state is initialised from the same interpolated primitives.
  double bc_dt = 0.0;
  std::vector<int> nrFaceOrdinal;  // face -> ordinal among the faces of its (non-reflecting) patch, -1 otherwise
  void updateBCMean() {
    const int64_t N = ndofs;
    const int nf = static_cast<int>(mesh.faces.size());
    bool any = false;
    for (auto &kv : bcs) any = any || kv.second->nonReflecting;
    if (!any) return;
    if (nrFaceOrdinal.empty()) {
      nrFaceOrdinal.assign(nf, -1);
      std::map<int, int> count;
      for (int f = 0; f < nf; f++) {
        const Face &F = mesh.faces[f];
        if (F.e2 < 0 && bcs.at(F.attr)->nonReflecting) nrFaceOrdinal[f] = count[F.attr]++;
      }
      for (auto &kv : bcs)
        if (kv.second->nonReflecting) {
          kv.second->boundaryU.assign(static_cast<size_t>(count[kv.first]) * faceRule.npts * neq, 0.0);
          kv.second->dt = &bc_dt;
          kv.second->refLength = disc.ref_length > 0.0 ? disc.ref_length : 1.0;
        }
    }
    for (auto &kv : bcs) {
      BoundaryCondition &bc = *kv.second;
      if (!bc.nonReflecting) continue;
      double localMeanUp[MAXEQ + 1] = {0};
      std::vector<double> shape(dof);
      for (int f = 0; f < nf; f++) {
        const Face &F = mesh.faces[f];
        if (F.e2 >= 0 || F.attr != kv.first) continue;
        for (int q = 0; q < faceRule.npts; q++) {
          FacePoint fp;
          facePoint(F, q, fp);
          fe.calc_shape(fp.xi1, shape.data());
          const int Nbdr = nrFaceOrdinal[f] * faceRule.npts + q;
          for (int eq = 0; eq < neq; eq++) {
            double sum = 0.;
            for (int k = 0; k < dof; k++) sum += shape[k] * Up[static_cast<int64_t>(F.e1) * dof + k + eq * N];
            localMeanUp[eq] += sum;
            if (!bc.bdrUInit) bc.boundaryU[eq + Nbdr * neq] = sum;
          }
        }
      }
      const double totNbdr = static_cast<double>(bc.boundaryU.size() / neq);
      for (int eq = 0; eq < neq; eq++) bc.meanUp[eq] = localMeanUp[eq] / totNbdr;  // single rank: no Allreduce
      if (!bc.bdrUInit) {
        for (size_t i = 0; i < bc.boundaryU.size() / neq; i++) {
          double iUp[MAXEQ], iState[MAXEQ];
          for (int eq = 0; eq < neq; eq++) iUp[eq] = bc.boundaryU[eq + i * neq];
          mixture->GetConservativesFromPrimitives(iUp, iState);
          for (int eq = 0; eq < neq; eq++) bc.boundaryU[eq + i * neq] = iState[eq];
        }
        bc.bdrUInit = true;
      }
    }
  }

  // A->Mult: interior faces (src/face_integrator.cpp:194-352) + boundary (src/BCintegrator.cpp:295-441)
  void faceFluxes(const double *x, double /*time*/) {
    updateBCMean();  // src/rhs_operator.cpp:364
    const int64_t N = ndofs;
    const int nf = static_cast<int>(mesh.faces.size());
    std::vector<double> fbuf(static_cast<size_t>(nf) * 2 * dof * neq, 0.0);
    const int nAct = mixture->numActiveSpecies;
    OmpGuard guard;
#pragma omp parallel for schedule(dynamic, 16)
    for (int f = 0; f < nf; f++) try {
      const Face &F = mesh.faces[f];
      std::vector<double> shape1(dof), shape2(dof);
      double *el1 = &fbuf[static_cast<size_t>(f) * 2 * dof * neq];
      double *el2 = el1 + static_cast<size_t>(dof) * neq;
      for (int q = 0; q < faceRule.npts; q++) {
        FacePoint fp;
        facePoint(F, q, fp);
        fe.calc_shape(fp.xi1, shape1.data());
        double u1[MAXEQ], u2[MAXEQ], g1[MAXEQ * MAXDIM], g2[MAXEQ * MAXDIM], fluxN[MAXEQ];
        for (int eq = 0; eq < neq; eq++) {
          double s = 0.0;
          for (int k = 0; k < dof; k++) s += x[static_cast<int64_t>(F.e1) * dof + k + eq * N] * shape1[k];
          u1[eq] = s;
        }
        for (int sp = 0; sp < nAct; sp++) u1[nvel + 2 + sp] = std::max(u1[nvel + 2 + sp], 0.0);
        interpGrad(F.e1, shape1.data(), g1);
        double transip[3] = {fp.x[0], fp.x[1], dim == 3 ? fp.x[2] : 0.0};
        if (F.e2 >= 0) {
          fe.calc_shape(fp.xi2, shape2.data());
          for (int eq = 0; eq < neq; eq++) {
            double s = 0.0;
            for (int k = 0; k < dof; k++) s += x[static_cast<int64_t>(F.e2) * dof + k + eq * N] * shape2[k];
            u2[eq] = s;
          }
          for (int sp = 0; sp < nAct; sp++) u2[nvel + 2 + sp] = std::max(u2[nvel + 2 + sp], 0.0);
          interpGrad(F.e2, shape2.data(), g2);
          rsolver->Eval(u1, u2, fp.nor, fluxN);  // src/face_integrator.cpp:324
          double v1[MAXEQ * MAXDIM], v2[MAXEQ * MAXDIM];
          double d1 = 0, d2 = 0;  // the distance function at the point, each side with its own shape functions (:303-308)
          if (!distance.empty()) {
            for (int k = 0; k < dof; k++) d1 += distance[static_cast<int64_t>(F.e1) * dof + k] * shape1[k];
            for (int k = 0; k < dof; k++) d2 += distance[static_cast<int64_t>(F.e2) * dof + k] * shape2[k];
          }
          fluxes->ComputeViscousFluxes(u1, g1, transip, elSize[F.e1], d1, v1);  // delta1, delta2: face_integrator.cpp:253-276
          fluxes->ComputeViscousFluxes(u2, g2, transip, elSize[F.e2], d2, v2);
          for (int i = 0; i < neq * dim; i++) v1[i] = -0.5 * (v1[i] + v2[i]);
          for (int eq = 0; eq < neq; eq++)
            for (int d = 0; d < dim; d++) fluxN[eq] += v1[eq + d * neq] * fp.nor[d];
          for (int eq = 0; eq < neq; eq++) fluxN[eq] *= fp.w;
          if (axisym)
            for (int eq = 0; eq < neq; eq++) fluxN[eq] *= transip[0];
          for (int eq = 0; eq < neq; eq++)
            for (int k = 0; k < dof; k++) {
              el2[k + eq * dof] += shape2[k] * fluxN[eq];
              el1[k + eq * dof] -= shape1[k] * fluxN[eq];
            }
        } else {
          double d1 = 0;  // src/BCintegrator.cpp:409-412
          if (!distance.empty())
            for (int k = 0; k < dof; k++) d1 += distance[static_cast<int64_t>(F.e1) * dof + k] * shape1[k];
          bcs.at(F.attr)->computeBdrFlux(fp.nor, u1, g1, transip, elSize[F.e1], d1, fluxN,
                                         nrFaceOrdinal.empty() || nrFaceOrdinal[f] < 0 ? -1 : nrFaceOrdinal[f] * faceRule.npts + q);
          for (int eq = 0; eq < neq; eq++) fluxN[eq] *= fp.w;
          if (axisym)
            for (int eq = 0; eq < neq; eq++) fluxN[eq] *= transip[0];
          for (int eq = 0; eq < neq; eq++)
            for (int k = 0; k < dof; k++) el1[k + eq * dof] -= fluxN[eq] * shape1[k];
        }
      }
    } catch (const std::exception &e) {
      guard.capture(e);
    }
    guard.rethrow();
    std::fill(z.begin(), z.end(), 0.0);
#pragma omp parallel for schedule(static)
    for (int e = 0; e < ne; e++)
      for (int lf = 0; lf < 2 * dim; lf++) {
        const int f = mesh.elem_faces[e][lf];
        const Face &F = mesh.faces[f];
        const bool first = (F.e1 == e && F.f1 == lf);
        const double *el = &fbuf[static_cast<size_t>(f) * 2 * dof * neq] + (first ? 0 : static_cast<size_t>(dof) * neq);
        for (int eq = 0; eq < neq; eq++)
          for (int k = 0; k < dof; k++) z[static_cast<int64_t>(e) * dof + k + eq * N] += el[k + eq * dof];
      }
  }

  // src/rhs_operator.cpp:493-559 ; flux(i,d,k) stored [i + d*N + k*N*dim]
  void getFlux(const double *x) {
    const int64_t N = ndofs;
    const int nAct = mixture->numActiveSpecies;
    double mcs_all = 0.0;
    OmpGuard guard;
#pragma omp parallel for schedule(static) reduction(max : mcs_all)
    for (int64_t i = 0; i < N; i++) try {
      double state[MAXEQ], g[MAXEQ * MAXDIM], f[MAXEQ * MAXDIM], fv[MAXEQ * MAXDIM];
      for (int k = 0; k < neq; k++) state[k] = x[i + k * N];
      for (int sp = 0; sp < nAct; sp++) state[nvel + 2 + sp] = std::max(state[nvel + 2 + sp], 0.0);
      for (int eq = 0; eq < neq; eq++)
        for (int d = 0; d < dim; d++) g[eq + d * neq] = gradUp[i + eq * N + d * neq * N];
      double xyz[3] = {0, 0, 0};
      for (int d = 0; d < dim; d++) xyz[d] = coords[i + d * N];
      fluxes->ComputeConvectiveFluxes(state, f);
      if (phys.eq_system != TPSRHS_EULER) {
        fluxes->ComputeViscousFluxes(state, g, xyz, elSize[i / dof], distance.empty() ? 0.0 : distance[i], fv);
        for (int k = 0; k < neq * dim; k++) f[k] -= fv[k];
      }
      for (int d = 0; d < dim; d++)
        for (int k = 0; k < neq; k++) flux[i + d * N + k * N * dim] = f[k + d * neq];
      const double mcs = mixture->ComputeMaxCharSpeed(state);
      if (mcs > mcs_all) mcs_all = mcs;
    } catch (const std::exception &e) {
      guard.capture(e);
    }
    guard.rethrow();
    max_char_speed = mcs_all;
  }

  void mult(const double *x, double *y, double time) {
    const int64_t N = ndofs;
    max_char_speed = 0.0;
    updatePrimitives(x);
    computeGradients();
    faceFluxes(x, time);
    getFlux(x);
    // for eq: Aflux->AddMult(flux(eq), z(eq))  (src/rhs_operator.cpp:379-391)
#pragma omp parallel for schedule(static)
    for (int e = 0; e < ne; e++) {
      const Dense &A = Aflux[e];
      for (int eq = 0; eq < neq; eq++)
        for (int j = 0; j < dof; j++) {
          double s = 0.0;
          for (int d = 0; d < dim; d++)
            for (int k = 0; k < dof; k++)
              s += A(j, k + d * dof) * flux[static_cast<int64_t>(e) * dof + k + d * N + eq * N * dim];
          z[static_cast<int64_t>(e) * dof + j + eq * N] += s;
        }
    }
    // inverse mass (src/rhs_operator.cpp:432-448)
#pragma omp parallel for schedule(static)
    for (int e = 0; e < ne; e++) {
      const Dense &Mi = MeInv[e];
      for (int eq = 0; eq < neq; eq++)
        for (int j = 0; j < dof; j++) {
          double s = 0.0;
          for (int k = 0; k < dof; k++) s += Mi(j, k) * z[static_cast<int64_t>(e) * dof + k + eq * N];
          y[static_cast<int64_t>(e) * dof + j + eq * N] = s;
        }
    }
    // forcing terms (src/rhs_operator.cpp:451-461)
    // order of the forcing array: ConstantPressureGradient, SpongeZone(s), HeatSource(s), SourceTerm,
    // AxisymmetricSource, JouleHeating (src/rhs_operator.cpp:101-166)
    if (has_forcing && forcing_in.has_pressure_gradient) constantPressureGradient(y);
    if (has_forcing) passiveScalar(y);
    if (has_forcing)
      for (size_t zn = 0; zn < spongeSigma.size(); zn++) spongeZone(static_cast<int>(zn), y);
    if (has_forcing)
      for (size_t hs = 0; hs < heatNodes.size(); hs++)  // HeatSource::updateTerms, src/forcing_terms.cpp:923-936
        for (int64_t node : heatNodes[hs]) y[node + (dim + 1) * N] += forcing_in.heat_sources[hs].value;
    if (source) source->updateTerms(x, Up.data(), gradUp.data(), ndofs, y);
    if (phys.working_fluid == TPSRHS_LTE_FLUID) lteSource(y);
    if (axisym) axisymmetricSource(x, y);
    if (!joule.empty()) {  // JouleHeating::updateTerms, src/forcing_terms.cpp:443-471
      if (nvel != 3) throw std::runtime_error("JouleHeating asserts nvel == 3");
      const bool twoT = phys.working_fluid != TPSRHS_DRY_AIR && phys.mixture.two_temperature;
      for (int64_t n = 0; n < N; n++) {
        const double heating = joule[n];
        if (heating > 0.) {
          y[n + (nvel + 1) * N] += heating;
          if (twoT) y[n + (neq - 1) * N] += heating;
        }
      }
    }
  }

  // constructors of SpongeZone (src/forcing_terms.cpp:475-627, USERDEF target) and HeatSource (:873-921)
  void setForcing(const tpsrhs_forcing *f) {
    spongeSigma.clear();
    spongeRadial.clear();
    spongePlaneNodes.clear();
    heatNodes.clear();
    scalarNodes.clear();
    has_forcing = f != nullptr;
    if (!f) return;
    forcing_in = *f;
    const int64_t N = ndofs;
    for (int i = 0; i < f->num_passive_scalars; i++) {  // PassiveScalar constructor, src/forcing_terms.cpp:795-818
      const tpsrhs_passive_scalar &ps = f->passive_scalars[i];
      std::vector<int64_t> list;
      for (int64_t n = 0; n < N; n++) {
        double dist = 0.;
        for (int d = 0; d < dim; d++) dist += (coords[n + d * N] - ps.coords[d]) * (coords[n + d * N] - ps.coords[d]);
        dist = std::sqrt(dist);
        if (dist < ps.radius) list.push_back(n);
      }
      scalarNodes.push_back(list);
    }
    for (int zn = 0; zn < f->num_sponge_zones; zn++) {
      tpsrhs_sponge_zone &sz = forcing_in.sponge_zones[zn];
      double mod = 0.;
      for (int d = 0; d < dim; d++) mod += sz.normal[d] * sz.normal[d];
      mod = std::sqrt(mod);
      for (int d = 0; d < dim; d++) sz.normal[d] /= mod;
      std::vector<double> sigma(N, 0.0), radial;
      std::vector<int64_t> nodesVec;
      if (sz.type == TPSRHS_SPONGE_ANNULUS) radial.assign(static_cast<size_t>(N) * 3, 0.0);
      for (int64_t n = 0; n < N; n++) {
        double Xn[3] = {0, 0, 0};
        for (int d = 0; d < dim; d++) Xn[d] = coords[n + d * N];
        double distInit = 0.;
        for (int d = 0; d < dim; d++) distInit -= sz.normal[d] * (Xn[d] - sz.point_init[d]);
        double distF = 0.;
        for (int d = 0; d < dim; d++) distF += sz.normal[d] * (Xn[d] - sz.point0[d]);
        if (sz.type == TPSRHS_SPONGE_PLANAR) {
          if (std::fabs(distInit) < sz.tol) nodesVec.push_back(n);
          if (distInit > 0. && distF > 0.) {
            const double planeDistance = distF + distInit;
            sigma[n] = distInit / planeDistance / planeDistance;
          }
        } else {
          double R = 0., tmp[3];
          for (int d = 0; d < dim; d++) tmp[d] = Xn[d] - sz.point_init[d] + distInit * sz.normal[d];
          for (int d = 0; d < dim; d++) R += tmp[d] * tmp[d];
          R = std::sqrt(R);
          if (std::fabs(R - sz.r1) < sz.tol) nodesVec.push_back(n);
          if (distInit > 0. && distF > 0. && R - sz.r1 > 0.) {
            const double planeDistance = sz.r2 - sz.r1;
            sigma[n] = (R - sz.r1) / planeDistance / planeDistance;
            for (int d = 0; d < dim; d++) radial[3 * n + d] = tmp[d] / R;
          }
        }
      }
      spongeSigma.push_back(sigma);
      spongeRadial.push_back(radial);
      if (sz.solution_type != TPSRHS_SPONGE_MIXEDOUT) nodesVec.clear();
      spongePlaneNodes.push_back(nodesVec);
    }
    for (int hs = 0; hs < f->num_heat_sources; hs++) {
      const tpsrhs_heat_source &h = f->heat_sources[hs];
      double norm[3] = {0, 0, 0}, mod = 0.;
      for (int d = 0; d < dim; d++) norm[d] = h.point2[d] - h.point1[d];
      for (int d = 0; d < dim; d++) mod += norm[d] * norm[d];
      mod = std::sqrt(mod);
      for (int d = 0; d < dim; d++) norm[d] /= mod;
      std::vector<int64_t> nodes;
      for (int64_t n = 0; n < N; n++) {
        double X[3], proj = 0, normR = 0.;
        for (int d = 0; d < dim; d++) X[d] = coords[n + d * N] - h.point1[d];
        for (int d = 0; d < dim; d++) proj += X[d] * norm[d];
        for (int d = 0; d < dim; d++) {
          const double r = X[d] - proj * norm[d];
          normR += r * r;
        }
        normR = std::sqrt(normR);
        if (normR < h.radius && proj > 0 && proj < mod) nodes.push_back(n);
      }
      heatNodes.push_back(nodes);
    }
  }

  // ConstantPressureGradient::updateTerms (CPU branch), src/forcing_terms.cpp:132-171
  // SourceTerm::updateTerms (src/source_term.cpp:62-256) for the table gas (numSpecies == 1: no chemistry, not
  // two-temperature): what is left is the radiation sink at the node's primitive temperature (:207-209)
  void lteSource(double *y) {
    if (phys.radiation.model != TPSRHS_NET_EMISSION) return;
    const int64_t N = ndofs;
    for (int64_t n = 0; n < N; n++) y[n + (1 + nvel) * N] += -4.0 * PI_ * lteNec.eval(Up[n + (1 + nvel) * N]);
  }
  // PassiveScalar: node lists of the constructor (src/forcing_terms.cpp:795-818), updateTerms (:826-848)
  void passiveScalar(double *y) {
    const int64_t N = ndofs;
    for (int i = 0; i < forcing_in.num_passive_scalars; i++) {
      const tpsrhs_passive_scalar &ps = forcing_in.passive_scalars[i];
      const double Z = ps.value;
      for (int64_t node : scalarNodes[i]) {
        double vel = 0.;
        for (int d = 0; d < dim; d++) vel += Up[node + (1 + d) * N] * Up[node + (1 + d) * N];
        vel = std::sqrt(vel);
        y[node + (neq - 1) * N] -= vel * (Up[node + (neq - 1) * N] - Up[node] * Z) / ps.radius;
      }
    }
  }
  void constantPressureGradient(double *y) {
    const int64_t N = ndofs;
    const double *pressGrad = forcing_in.pressure_gradient;
    for (int64_t index = 0; index < N; index++) {
      double primi[MAXEQ];
      for (int eq = 0; eq < neq; eq++) primi[eq] = Up[index + eq * N];
      double p;
      if (phys.working_fluid == TPSRHS_DRY_AIR)
        p = mixture->GetGasConstant() * primi[0] * primi[nvel + 1];  // DryAir::ComputePressureFromPrimitives :361
      else if (phys.working_fluid == TPSRHS_LTE_FLUID)
        p = static_cast<LteMixture *>(mixture.get())->ComputePressureFromPrimitives(primi);
      else
        p = static_cast<PerfectMixture *>(mixture.get())->ComputePressureFromPrimitives(primi);
      double grad_pV = 0.;
      for (int d = 0; d < dim; d++) {
        const double vel = Up[index + (d + 1) * N];
        y[index + (d + 1) * N] -= pressGrad[d];
        grad_pV -= vel * pressGrad[d];
        grad_pV -= p * gradUp[index + (d + 1) * N + d * N * neq];
      }
      y[index + (1 + nvel) * N] += grad_pV;
    }
  }

  // SpongeZone::computeMixedOutValues, src/forcing_terms.cpp:713-743 (one rank: the MPI_Allreduce is the identity),
  // with DryAir::computeConservedStateFromConvectiveFlux, src/equation_of_state.cpp:414-444
  void computeMixedOutValues(int zn) {
    if (phys.working_fluid != TPSRHS_DRY_AIR || axisym) throw std::runtime_error("mixed-out sponge: dry air, planar / 3-D");
    const int64_t N = ndofs;
    tpsrhs_sponge_zone &sz = forcing_in.sponge_zones[zn];
    double meanNormalFluxes[MAXEQ + 1];
    for (int eq = 0; eq <= neq; eq++) meanNormalFluxes[eq] = 0.;
    for (int64_t node : spongePlaneNodes[zn]) {
      double Upn[MAXEQ], Un[MAXEQ], f[MAXEQ * MAXDIM];
      for (int eq = 0; eq < neq; eq++) Upn[eq] = Up[node + eq * N];
      mixture->GetConservativesFromPrimitives(Upn, Un);
      fluxes->ComputeConvectiveFluxes(Un, f);
      for (int eq = 0; eq < neq; eq++)
        for (int d = 0; d < dim; d++) meanNormalFluxes[eq] += sz.normal[d] * f[eq + d * neq];
    }
    meanNormalFluxes[neq] = static_cast<double>(spongePlaneNodes[zn].size());
    for (int eq = 0; eq < neq; eq++) meanNormalFluxes[eq] /= meanNormalFluxes[neq];
    const double gamma = mixture->GetSpecificHeatRatio();
    const int iTh = 1 + nvel;
    double temp = 0.;
    for (int d = 0; d < dim; d++) temp += meanNormalFluxes[1 + d] * sz.normal[d];
    const double A = 1. - 2. * gamma / (gamma - 1.);
    const double B = 2 * temp / (gamma - 1.);
    double C = -2. * meanNormalFluxes[0] * meanNormalFluxes[iTh];
    for (int d = 0; d < nvel; d++) C += meanNormalFluxes[1 + d] * meanNormalFluxes[1 + d];
    const double p = (-B - std::sqrt(B * B - 4. * A * C)) / (2. * A);
    double Upm[MAXEQ];
    Upm[0] = meanNormalFluxes[0] * meanNormalFluxes[0] / (temp - p);
    Upm[iTh] = p / (mixture->GetGasConstant() * Upm[0]);
    for (int d = 0; d < nvel; d++) Upm[1 + d] = (meanNormalFluxes[1 + d] - p * sz.normal[d]) / meanNormalFluxes[0];
    mixture->GetConservativesFromPrimitives(Upm, sz.target_U);
  }

  // SpongeZone::addSpongeZoneForcing, src/forcing_terms.cpp:637-711
  void spongeZone(int zn, double *y) {
    const int64_t N = ndofs;
    if (forcing_in.sponge_zones[zn].solution_type == TPSRHS_SPONGE_MIXEDOUT) computeMixedOutValues(zn);  // updateTerms, :631-635
    const tpsrhs_sponge_zone &sz = forcing_in.sponge_zones[zn];
    const double *targetU = sz.target_U;
    double Upt[MAXEQ], targetCyl[MAXEQ];
    for (int eq = 0; eq < neq; eq++) targetCyl[eq] = targetU[eq];
    mixture->GetPrimitivesFromConservatives(targetU, Upt);
    double speedSound;  // mixture->ComputeSpeedOfSound(Up, true)
    if (phys.working_fluid == TPSRHS_DRY_AIR)
      speedSound = std::sqrt(mixture->GetSpecificHeatRatio() * mixture->GetGasConstant() * Upt[nvel + 1]);  // :337-348
    else if (phys.working_fluid == TPSRHS_LTE_FLUID)
      speedSound = static_cast<LteMixture *>(mixture.get())->ComputeSpeedOfSound(Upt, true);
    else  // the primitive branch (:1406-1419) rebuilds exactly the quantities of the conserved one
      speedSound = static_cast<PerfectMixture *>(mixture.get())->ComputeSpeedOfSound(targetU);
    for (int64_t n = 0; n < N; n++) {
      double s = spongeSigma[zn][n];
      if (s > 0.) {
        s *= sz.mult_factor;
        double Upn[MAXEQ], Un[MAXEQ];
        for (int eq = 0; eq < neq; eq++) Upn[eq] = Up[n + eq * N];
        mixture->GetConservativesFromPrimitives(Upn, Un);
        if (sz.type == TPSRHS_SPONGE_ANNULUS) {  // :686-705; the block at :667-683 is overwritten by this one
          double ur[3], uz[3], uth[3], MM[9], inv[9];
          for (int d = 0; d < 3; d++) {
            ur[d] = spongeRadial[zn][3 * n + d];
            uz[d] = sz.normal[d];
          }
          uth[0] = uz[1] * ur[2] - ur[1] * uz[2];
          uth[1] = uz[2] * ur[0] - uz[0] * ur[2];
          uth[2] = uz[0] * ur[1] - ur[0] * uz[1];
          for (int d = 0; d < 3; d++) {  // MM(i, d), column-major as DenseMatrix
            MM[0 + 3 * d] = ur[d];
            MM[1 + 3 * d] = uth[d];
            MM[2 + 3 * d] = uz[d];
          }
          invert3(MM, inv);
          for (int i = 0; i < 3; i++) {
            double v = 0.0;
            for (int j = 0; j < 3; j++) v += inv[i + 3 * j] * targetU[1 + j];
            targetCyl[1 + i] = v;
          }
        }
        for (int eq = 0; eq < neq; eq++) y[n + eq * N] -= speedSound * s * (Un[eq] - targetCyl[eq]);
      }
    }
  }
  static void invert3(const double *A, double *B) {  // column-major 3x3 inverse (DenseMatrix::Invert)
    const double c00 = A[4] * A[8] - A[7] * A[5], c10 = A[7] * A[2] - A[1] * A[8], c20 = A[1] * A[5] - A[4] * A[2];
    const double det = A[0] * c00 + A[3] * c10 + A[6] * c20;
    B[0] = c00 / det;
    B[1] = c10 / det;
    B[2] = c20 / det;
    B[3] = (A[6] * A[5] - A[3] * A[8]) / det;
    B[4] = (A[0] * A[8] - A[6] * A[2]) / det;
    B[5] = (A[3] * A[2] - A[0] * A[5]) / det;
    B[6] = (A[3] * A[7] - A[6] * A[4]) / det;
    B[7] = (A[6] * A[1] - A[0] * A[7]) / det;
    B[8] = (A[0] * A[4] - A[3] * A[1]) / det;
  }

  // AxisymmetricSource::updateTerms, src/forcing_terms.cpp:255-382 (CPU branch)
  void axisymmetricSource(const double *x, double *y);

  // L2 norm of a nodal field difference (exact mass; stands in for GridFunction::ComputeLpError)
  double l2_norm(const double *a, const double *b) const {
    RuleND rule;
    rule.init(dim, gauss_legendre(fe.p + 3));
    double tot = 0.0;
#pragma omp parallel for schedule(static) reduction(+ : tot)
    for (int e = 0; e < ne; e++) {
      std::vector<double> shape(dof);
      for (int q = 0; q < rule.npts; q++) {
        const double *xi = &rule.x[q * dim];
        double x[3], J[9], Ji[9];
        mesh.transform(e, xi, x, J);
        const double det = det_and_inverse(dim, J, Ji);
        fe.calc_shape(xi, shape.data());
        double v = 0.0;
        for (int k = 0; k < dof; k++) {
          const int64_t n = static_cast<int64_t>(e) * dof + k;
          v += shape[k] * (a[n] - (b ? b[n] : 0.0));
        }
        tot += v * v * det * rule.w[q];
      }
    }
    return std::sqrt(tot);
  }
};

void Operator::axisymmetricSource(const double *x, double *y) {
  const int64_t N = ndofs;
  OmpGuard guard;
#pragma omp parallel for schedule(static)
  for (int64_t n = 0; n < N; n++) try {
    double U[MAXEQ], prim[MAXEQ], g[MAXEQ * MAXDIM];
    for (int eq = 0; eq < neq; eq++) {
      U[eq] = x[n + eq * N];
      prim[eq] = Up[n + eq * N];
      for (int d = 0; d < dim; d++) g[eq + d * neq] = gradUp[n + eq * N + d * neq * N];
    }
    axisym_source_point(*mixture, *transport, phys.eq_system, neq, dim, coords[n + 0 * N], U, prim, g, n, N, y);
  } catch (const std::exception &e) {
    guard.capture(e);
  }
  guard.rethrow();
}

}  // namespace tpsoracle

// --------------------------------------------------------------------------------------------
// C entry points (ctypes)
// --------------------------------------------------------------------------------------------
using tpsoracle::Operator;
static thread_local std::string g_err;

extern "C" {

const char *tpsoracle_last_error() { return g_err.c_str(); }

int tpsoracle_create(const tpsrhs_mesh *mesh, const tpsrhs_disc *disc, const tpsrhs_physics *phys, int nbc,
                     const tpsrhs_bc *bcs, void **out) {
  try {
    Operator *op = new Operator();
    op->setup(mesh, disc, phys, nbc, bcs);
    *out = op;
    return 0;
  } catch (const std::exception &e) {
    g_err = e.what();
    return 1;
  }
}
int tpsoracle_destroy(void *h) {
  delete static_cast<Operator *>(h);
  return 0;
}
int tpsoracle_rk4_step(void *h, double *x, double *time, double dt, double *max_char_speed, int64_t *nan_count);
// num_steps x M2ulPhyS::solveStep (src/M2ulPhyS.cpp:2004-2019): RK4 step, NaN census, species clamp, then
// dt = CFL * hmin / max_char_speed / dim unless the time step is constant
int tpsoracle_advance(void *h, double *x, double *time, double *dt, int num_steps, int constant_dt, double cfl,
                      double hmin, int64_t *nan_count) {
  Operator *op = static_cast<Operator *>(h);
  int64_t total = 0;
  for (int s = 0; s < num_steps; s++) {
    double speed = 0.0;
    int64_t bad = 0;
    const int st = tpsoracle_rk4_step(h, x, time, *dt, &speed, &bad);
    if (st != 0) return st;
    total += bad;
    if (!constant_dt) *dt = cfl * hmin / speed / static_cast<double>(op->dim);
  }
  if (nan_count) *nan_count = total;
  return 0;
}
int tpsoracle_set_dt(void *h, double dt) {
  static_cast<Operator *>(h)->bc_dt = dt;
  return 0;
}
// boundaryU of a non-reflecting patch ([point][eq], points = faces of the patch in mesh order x quadrature
// points) and its meanUp; returns the number of points
int tpsoracle_get_boundary_state(void *h, int attr, double *boundaryU, double *meanUp) {
  Operator *op = static_cast<Operator *>(h);
  const tpsoracle::BoundaryCondition &bc = *op->bcs.at(attr);
  if (boundaryU) std::memcpy(boundaryU, bc.boundaryU.data(), bc.boundaryU.size() * sizeof(double));
  if (meanUp) std::memcpy(meanUp, bc.meanUp, op->neq * sizeof(double));
  return static_cast<int>(bc.boundaryU.size() / op->neq);
}
int tpsoracle_set_forcing(void *h, const tpsrhs_forcing *f) {
  try {
    static_cast<Operator *>(h)->setForcing(f);
    return 0;
  } catch (const std::exception &e) {
    g_err = e.what();
    return 1;
  }
}
int tpsoracle_set_mixing_length(void *h, const double *dist, const tpsrhs_mixing_length *in) {  // HOST array, copied
  Operator *op = static_cast<Operator *>(h);
  if (dist && in) {
    op->distance.assign(dist, dist + op->ndofs);
    op->mixlen.reset(new tpsoracle::MixingLengthTransport(op->mixture.get(), *in, op->transport.get()));
    op->fluxes->transport = op->mixlen.get();
  } else {
    op->distance.clear();
    op->fluxes->transport = op->transport.get();
    op->mixlen.reset();
  }
  return 0;
}
int tpsoracle_set_joule_heating(void *h, const double *jh) {  // HOST array, copied; NULL disables
  Operator *op = static_cast<Operator *>(h);
  if (jh)
    op->joule.assign(jh, jh + op->ndofs);
  else
    op->joule.clear();
  return 0;
}
int64_t tpsoracle_num_dofs(void *h) { return static_cast<Operator *>(h)->ndofs; }
int tpsoracle_num_equation(void *h) { return static_cast<Operator *>(h)->neq; }
int tpsoracle_set_threads(int n) {
  omp_set_num_threads(n);
  return omp_get_max_threads();
}
int tpsoracle_mult(void *h, const double *x, double *y, double time, double *max_char_speed) {
  try {
    Operator *op = static_cast<Operator *>(h);
    op->mult(x, y, time);
    if (max_char_speed) *max_char_speed = op->max_char_speed;
    return 0;
  } catch (const std::exception &e) {
    g_err = e.what();
    return 1;
  }
}
// Gradients::computeGradients on a caller-supplied primitive field (test/test_gradient.cpp:161-162)
int tpsoracle_compute_gradients(void *h, const double *up_in) {
  try {
    Operator *op = static_cast<Operator *>(h);
    if (up_in) std::memcpy(op->Up.data(), up_in, op->Up.size() * sizeof(double));
    op->computeGradients();
    return 0;
  } catch (const std::exception &e) {
    g_err = e.what();
    return 1;
  }
}
int tpsoracle_update_primitives(void *h, const double *x) {
  static_cast<Operator *>(h)->updatePrimitives(x);
  return 0;
}
int tpsoracle_get_primitives(void *h, double *out) {
  Operator *op = static_cast<Operator *>(h);
  std::memcpy(out, op->Up.data(), op->Up.size() * sizeof(double));
  return 0;
}
// SourceTerm::updateTerms' side output plasma_conductivity_ (src/source_term.cpp:125-199) for the state x: the clamped
// node state and primitives through the transport model's ComputeSourceTransportProperties; stored by the reference for
// single-species fluids, mixtures without reactions and ambipolar mixtures (:178-199)
int tpsoracle_get_plasma_conductivity(void *h, const double *x, double *out) {
  Operator *op = static_cast<Operator *>(h);
  try {
    if (op->phys.working_fluid == TPSRHS_DRY_AIR) throw std::runtime_error("plasma conductivity: dry air has no SourceTerm");
    const bool mix = op->phys.working_fluid == TPSRHS_USER_DEFINED;
    if (mix && !op->phys.mixture.ambipolar && op->phys.chemistry.num_reactions > 0)
      throw std::runtime_error("plasma conductivity: not stored for a reacting mixture that is not ambipolar");
    const int64_t N = op->ndofs;
    const int neq = op->neq, nact = op->mixture->numActiveSpecies;
    for (int64_t n = 0; n < N; n++) {
      double Un[tpsoracle::MAXEQ], upn[tpsoracle::MAXEQ], g[tpsoracle::MAXEQ * tpsoracle::MAXDIM] = {0};
      for (int eq = 0; eq < neq; eq++) Un[eq] = x[n + eq * N];
      op->mixture->GetPrimitivesFromConservatives(Un, upn);  // Up_ of the same state
      for (int sp = 0; sp < nact; sp++) {  // src/source_term.cpp:127-132 (the reference's hard-coded 3 + 2 + sp)
        const int eq = 3 + 2 + sp;
        if (eq < neq) {
          upn[eq] = std::max(upn[eq], 0.0);
          Un[eq] = std::max(Un[eq], 0.0);
        }
      }
      double Efield[tpsoracle::MAXDIM] = {0, 0, 0};
      double gt[tpsoracle::MAXSP] = {0}, st[tpsoracle::MAXSP * 2] = {0}, dv[tpsoracle::MAXSP * tpsoracle::MAXDIM] = {0}, ns[tpsoracle::MAXSP] = {0};
      op->transport->ComputeSourceTransportProperties(Un, upn, g, Efield, 0.0, gt, st, dv, ns);
      out[n] = gt[tpsoracle::ELECTRIC_CONDUCTIVITY];
    }
    return 0;
  } catch (const std::exception &e) {
    g_err = e.what();
    return 1;
  }
}
int tpsoracle_get_gradients(void *h, double *out) {
  Operator *op = static_cast<Operator *>(h);
  std::memcpy(out, op->gradUp.data(), op->gradUp.size() * sizeof(double));
  return 0;
}
int tpsoracle_node_coords(void *h, double *out) {
  Operator *op = static_cast<Operator *>(h);
  std::memcpy(out, op->coords.data(), op->coords.size() * sizeof(double));
  return 0;
}
double tpsoracle_l2_norm(void *h, const double *a, const double *b) { return static_cast<Operator *>(h)->l2_norm(a, b); }
// sum_e 1^T M_e y_e for one scalar field (discrete conservation checks)
double tpsoracle_integral(void *h, const double *a) {
  Operator *op = static_cast<Operator *>(h);
  tpsoracle::RuleND rule;
  rule.init(op->dim, tpsoracle::gauss_legendre(op->fe.p + 2));
  double tot = 0.0;
  std::vector<double> shape(op->dof);
  for (int e = 0; e < op->ne; e++)
    for (int q = 0; q < rule.npts; q++) {
      const double *xi = &rule.x[q * op->dim];
      double x[3], J[9], Ji[9];
      op->mesh.transform(e, xi, x, J);
      const double det = tpsoracle::det_and_inverse(op->dim, J, Ji);
      op->fe.calc_shape(xi, shape.data());
      double v = 0.0;
      for (int k = 0; k < op->dof; k++) v += shape[k] * a[static_cast<int64_t>(e) * op->dof + k];
      tot += v * det * rule.w[q] * (op->axisym ? x[0] : 1.0);
    }
  return tot;
}

// ---- point-wise entry points for unit tests (tests/test_oracle_physics.py) -------------------
int tpsoracle_point_prim(void *h, const double *state, double *prim) {
  static_cast<Operator *>(h)->mixture->GetPrimitivesFromConservatives(state, prim);
  return 0;
}
int tpsoracle_point_cons(void *h, const double *prim, double *state) {
  static_cast<Operator *>(h)->mixture->GetConservativesFromPrimitives(prim, state);
  return 0;
}
double tpsoracle_point_pressure(void *h, const double *state) {
  return static_cast<Operator *>(h)->mixture->ComputePressure(state);
}
double tpsoracle_point_max_char_speed(void *h, const double *state) {
  return static_cast<Operator *>(h)->mixture->ComputeMaxCharSpeed(state);
}
int tpsoracle_point_convective_flux(void *h, const double *state, double *flux) {
  static_cast<Operator *>(h)->fluxes->ComputeConvectiveFluxes(state, flux);
  return 0;
}
// the same at position `x` (dim entries) of an element of grid scale `delta`: the sub-grid scale models and the
// viscous sponge read them (src/fluxes.cpp:223-246)
int tpsoracle_point_viscous_flux_at(void *h, const double *state, const double *gradUp, const double *x, double delta,
                                    double *flux) {
  Operator *op = static_cast<Operator *>(h);
  double transip[3] = {0, 0, 0};
  for (int d = 0; d < op->dim; d++) transip[d] = x[d];
  op->fluxes->ComputeViscousFluxes(state, gradUp, transip, delta, 0.0, flux);
  return 0;
}
int tpsoracle_element_sizes(void *h, double *out) {  // elSize: GetElementSize(e, 1) / order
  Operator *op = static_cast<Operator *>(h);
  for (int e = 0; e < op->ne; e++) out[e] = op->elSize[e];
  return op->ne;
}
// ... at wall distance `distance` (the mixing-length model reads it)
int tpsoracle_point_viscous_flux_dist(void *h, const double *state, const double *gradUp, double radius, double distance,
                                      double *flux) {
  double transip[3] = {radius, 0, 0};
  static_cast<Operator *>(h)->fluxes->ComputeViscousFluxes(state, gradUp, transip, 0.0, distance, flux);
  return 0;
}
int tpsoracle_point_viscous_flux(void *h, const double *state, const double *gradUp, double radius, double *flux) {
  double transip[3] = {radius, 0, 0};
  static_cast<Operator *>(h)->fluxes->ComputeViscousFluxes(state, gradUp, transip, 0.0, 0.0, flux);
  return 0;
}
int tpsoracle_point_bdr_viscous_flux(void *h, const double *state, const double *gradUp, double radius,
                                     const double *normal, const double *primFlux, const int *primFluxIdxs,
                                     double *normalFlux) {
  Operator *op = static_cast<Operator *>(h);
  tpsoracle::BoundaryViscousFluxData bc;
  for (int d = 0; d < 3; d++) bc.normal[d] = d < op->dim ? normal[d] : 0.0;
  for (int i = 0; i < tpsoracle::MAXEQ; i++) {
    bc.primFlux[i] = primFlux ? primFlux[i] : 0.0;
    bc.primFluxIdxs[i] = primFluxIdxs ? primFluxIdxs[i] != 0 : false;
  }
  double transip[3] = {radius, 0, 0};
  op->fluxes->ComputeBdrViscousFluxes(state, gradUp, transip, 0.0, 0.0, bc, normalFlux);
  return 0;
}
int tpsoracle_point_lf(void *h, const double *s1, const double *s2, const double *nor, double *flux) {
  static_cast<Operator *>(h)->rsolver->Eval_LF(s1, s2, nor, flux);
  return 0;
}
int tpsoracle_point_roe(void *h, const double *s1, const double *s2, const double *nor, double *flux) {
  static_cast<Operator *>(h)->rsolver->Eval_Roe(s1, s2, nor, flux);
  return 0;
}
int tpsoracle_point_bdr_flux(void *h, int attr, const double *nor, const double *state, const double *gradUp,
                             double radius, double *flux) {
  try {
    double transip[3] = {radius, 0, 0};
    static_cast<Operator *>(h)->bcs.at(attr)->computeBdrFlux(nor, state, gradUp, transip, 0.0, 0.0, flux);
    return 0;
  } catch (const std::exception &e) {
    g_err = e.what();
    return 1;
  }
}
int tpsoracle_point_flux_transport(void *h, const double *state, const double *gradUp, double *buffer4,
                                   double *diffVel) {
  Operator *op = static_cast<Operator *>(h);
  double E[3] = {0, 0, 0};
  op->transport->ComputeFluxTransportProperties(state, gradUp, E, -1.0, 0.0, buffer4, diffVel);
  return 0;
}
// One RK4 step as M2ulPhyS::solveStep takes it (src/M2ulPhyS.cpp:2004-2008): MFEM's RK4Solver::Step
// [third party, MFEM >= 4.4 linalg/ode.cpp: k1..k4 with y = x + a dt k, z accumulating dt/6, dt/3, dt/3, dt/6],
// then the NaN census of Check_NAN and, for mixtures, the species clamp of Check_Undershoot (:2526-2548).
int tpsoracle_rk4_step(void *h, double *x, double *time, double dt, double *max_char_speed, int64_t *nan_count) {
  Operator *op = static_cast<Operator *>(h);
  const int64_t n = static_cast<int64_t>(op->neq) * op->ndofs;
  std::vector<double> k(n), y(n), z(n);
  try {
    op->bc_dt = dt;
    op->mult(x, k.data(), *time);
    for (int64_t i = 0; i < n; i++) {
      y[i] = x[i] + (dt / 2) * k[i];
      z[i] = x[i] + (dt / 6) * k[i];
    }
    op->mult(y.data(), k.data(), *time + dt / 2);
    for (int64_t i = 0; i < n; i++) {
      y[i] = x[i] + (dt / 2) * k[i];
      z[i] += (dt / 3) * k[i];
    }
    op->mult(y.data(), k.data(), *time + dt / 2);
    for (int64_t i = 0; i < n; i++) {
      y[i] = x[i] + dt * k[i];
      z[i] += (dt / 3) * k[i];
    }
    op->mult(y.data(), k.data(), *time + dt);
  } catch (const std::exception &e) {
    g_err = e.what();
    return 1;
  }
  int64_t bad = 0;
  const int sp0 = op->nvel + 2, sp1 = sp0 + op->mixture->numActiveSpecies;
  for (int64_t i = 0; i < n; i++) {
    double v = z[i] + (dt / 6) * k[i];
    if (v != v) bad++;
    const int64_t eq = i / op->ndofs;
    if (eq >= sp0 && eq < sp1) v = std::max(v, 0.0);
    x[i] = v;
  }
  *time += dt;
  if (max_char_speed) *max_char_speed = op->max_char_speed;
  if (nan_count) *nan_count = bad;
  return 0;
}
// collision-integral fits by name id: 0 att11, 1 att12, 2 att13, 3 att14, 4 att15, 5 rep22, 6 rep23,
// 7 rep24 (argument: nondimensional temperature); 8 ArAr22, 9 ArAr1P11, 10..14 eAr1r r=1..5 (argument: T in K)
// LinearTable::eval / findInterval on host arrays (test/test_table.cpp:26-121)
int tpsoracle_table_eval(const tpsrhs_table *t, int64_t n, const double *x, double *f, int *interval) {
  try {
    tpsoracle::LinearTable tab;
    tab.init(*t);
    for (int64_t i = 0; i < n; i++) {
      if (f) f[i] = tab.eval(x[i]);
      if (interval) interval[i] = tab.findInterval(x[i]);
    }
    return 0;
  } catch (const std::exception &e) {
    g_err = e.what();
    return 1;
  }
}

double tpsoracle_collision_integral(int id, double x) {
  using namespace tpsoracle::collision;
  switch (id) {
    case 0: return charged::att11(x);
    case 1: return charged::att12(x);
    case 2: return charged::att13(x);
    case 3: return charged::att14(x);
    case 4: return charged::att15(x);
    case 5: return charged::rep22(x);
    case 6: return charged::rep23(x);
    case 7: return charged::rep24(x);
    case 8: return argon::ArAr22(x);
    case 9: return argon::ArAr1P11(x);
    default: return (id >= 10 && id <= 14) ? argon::eAr1r(id - 9, x) : 0.0;
  }
}
int tpsoracle_point_source_transport(void *h, const double *state, const double *prim, const double *gradUp,
                                     double *global1, double *species, double *diffVel, double *n_sp) {
  Operator *op = static_cast<Operator *>(h);
  double E[3] = {0, 0, 0};
  op->transport->ComputeSourceTransportProperties(state, prim, gradUp, E, 0.0, global1, species, diffVel, n_sp);
  return 0;
}
int tpsoracle_point_source(void *h, const double *state, const double *prim, const double *gradUp, double *src) {
  Operator *op = static_cast<Operator *>(h);
  if (!op->source) return 1;
  op->source->point(state, prim, gradUp, src);
  return 0;
}
}
