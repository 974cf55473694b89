// TEST INFRASTRUCTURE -- CPU oracle, never shipped, never on the product path.
//
// Point-wise physics of the TPS compressible solver, restated function by function from the
// reference (paths relative to the pecos/tps tree).  Same call structure as the reference (virtual
// mixture / transport objects, raw-pointer state in, raw-pointer result out) so that the timed CPU
// baseline pays what the reference pays.
#ifndef TPS_ORACLE_PHYSICS_HPP_
#define TPS_ORACLE_PHYSICS_HPP_

#include <cmath>
#include <cstring>
#include <memory>
#include <stdexcept>
#include <vector>

#include "../include/tpsrhs.h"

namespace tpsoracle {

// An exception must not leave an OpenMP worksharing loop: bodies that call the physics (which throws on
// inadmissible states, as the reference exits) record the first message here; the caller rethrows after the loop.
struct OmpGuard {
  std::string msg;
  bool failed = false;
  void capture(const std::exception &e) {
#pragma omp critical(tpsoracle_guard)
    {
      if (!failed) {
        failed = true;
        msg = e.what();
      }
    }
  }
  void rethrow() const {
    if (failed) throw std::runtime_error(msg);
  }
};


constexpr int MAXEQ = TPSRHS_MAXEQUATIONS;
constexpr int MAXSP = TPSRHS_MAXSPECIES;
constexpr int MAXDIM = 3;

// src/equation_of_state.hpp:55-67
constexpr double UNIVERSALGASCONSTANT = 8.3144598;
constexpr double AVOGADRONUMBER = 6.0221409e+23;
constexpr double BOLTZMANNCONSTANT = UNIVERSALGASCONSTANT / AVOGADRONUMBER;
constexpr double VACUUMPERMITTIVITY = 8.8541878128e-12;
constexpr double ELECTRONCHARGE = 1.60218e-19;
constexpr double MOLARELECTRONCHARGE = ELECTRONCHARGE * AVOGADRONUMBER;
constexpr double PI_ = 3.14159265358979323846;

enum FluxTrns { VISCOSITY, BULK_VISCOSITY, HEAVY_THERMAL_CONDUCTIVITY, ELECTRON_THERMAL_CONDUCTIVITY, NUM_FLUX_TRANS };
enum SrcTrns { ELECTRIC_CONDUCTIVITY, NUM_SRC_TRANS };
enum SpeciesTrns { MF_FREQUENCY, NUM_SPECIES_COEFFS };

// src/dataStructures.hpp:575-605
struct BoundaryViscousFluxData {
  double normal[MAXDIM];
  double primFlux[MAXEQ];
  bool primFluxIdxs[MAXEQ];
};
struct BoundaryPrimitiveData {
  double prim[MAXEQ];
  bool primIdxs[MAXEQ];
};

// ------------------------------------------------------------------------------------------
// GasMixture (src/equation_of_state.hpp:72-330)
// ------------------------------------------------------------------------------------------
class GasMixture {
 public:
  int dim = 0, nvel = 0, num_equation = 0;
  int numSpecies = 1, numActiveSpecies = 0;
  bool ambipolar = false, twoTemperature = false;
  int iTh = 0;  // index of the heavy-species energy / temperature

  virtual ~GasMixture() {}
  virtual double ComputePressure(const double *state, double *electronPressure = nullptr) const = 0;
  virtual double ComputeTemperature(const double *state) const = 0;
  virtual double ComputeMaxCharSpeed(const double *state) const = 0;
  virtual void GetPrimitivesFromConservatives(const double *conserv, double *primit) const = 0;
  virtual void GetConservativesFromPrimitives(const double *primit, double *conserv) const = 0;
  virtual void computeSpeciesEnthalpies(const double *state, double *speciesEnthalpies) const = 0;
  virtual void computeStagnationState(const double *stateIn, double *stagnationState) const = 0;
  virtual void computeStagnantStateWithTemp(const double *stateIn, double Temp, double *stateOut) const = 0;
  virtual void modifyEnergyForPressure(const double *stateIn, double *stateOut, double p,
                                       bool modifyElectronEnergy = false) const = 0;
  // GasMixture::modifyStateFromPrimitive, src/equation_of_state.cpp:131-140
  void modifyStateFromPrimitive(const double *state, const BoundaryPrimitiveData &bcState, double *out) const {
    double prim[MAXEQ];
    GetPrimitivesFromConservatives(state, prim);
    for (int i = 0; i < num_equation; i++)
      if (bcState.primIdxs[i]) prim[i] = bcState.prim[i];
    GetConservativesFromPrimitives(prim, out);
  }
  virtual void computeSheathBdrFlux(const double *, BoundaryViscousFluxData &) const {
    throw std::runtime_error("sheath boundary flux needs a plasma mixture");
  }
  virtual double ComputePressureFromPrimitives(const double *) const {
    throw std::runtime_error("ComputePressureFromPrimitives: not restated for this mixture through the base class");
  }
  virtual double GetGasConstant() const { return 0.0; }
  virtual double GetSpecificHeatRatio() const { return 0.0; }
  virtual double GetGasParams(int sp, int param) const { return 0.0; }
};

// DryAir: src/equation_of_state.cpp:146-412, inline parts src/equation_of_state.hpp:605-628
class DryAir : public GasMixture {
 public:
  double specific_heat_ratio, gas_constant;
  DryAir(const tpsrhs_dry_air &in, int dim_, int nvel_) {
    dim = dim_;
    nvel = nvel_;
    specific_heat_ratio = in.specific_heat_ratio;
    gas_constant = in.gas_constant;
    numSpecies = 1;
    numActiveSpecies = 0;
    num_equation = nvel + 2;  // setNumEquations, :197-201
    iTh = nvel + 1;
  }
  double GetGasConstant() const override { return gas_constant; }
  double GetSpecificHeatRatio() const override { return specific_heat_ratio; }
  double ComputePressure(const double *state, double *electronPressure = nullptr) const override {
    if (electronPressure != nullptr) *electronPressure = 0.0;
    double den_vel2 = 0;
    for (int d = 0; d < nvel; d++) den_vel2 += state[d + 1] * state[d + 1];
    den_vel2 /= state[0];
    return (specific_heat_ratio - 1.0) * (state[1 + nvel] - 0.5 * den_vel2);
  }
  double ComputeTemperature(const double *state) const override {
    double den_vel2 = 0;
    for (int d = 0; d < nvel; d++) den_vel2 += state[d + 1] * state[d + 1];
    den_vel2 /= state[0];
    return (specific_heat_ratio - 1.0) / gas_constant * (state[1 + nvel] - 0.5 * den_vel2) / state[0];
  }
  double ComputeMaxCharSpeed(const double *state) const override {  // :278-292
    const double den = state[0];
    double den_vel2 = 0;
    for (int d = 0; d < nvel; d++) den_vel2 += state[d + 1] * state[d + 1];
    den_vel2 /= den;
    const double pres = ComputePressure(state);
    const double sound = std::sqrt(specific_heat_ratio * pres / den);
    const double vel = std::sqrt(den_vel2 / den);
    return vel + sound;
  }
  void GetPrimitivesFromConservatives(const double *conserv, double *primit) const override {  // :321-335
    const double T = ComputeTemperature(conserv);
    for (int eq = 0; eq < num_equation; eq++) primit[eq] = conserv[eq];
    for (int d = 0; d < nvel; d++) primit[1 + d] /= conserv[0];
    primit[iTh] = T;
  }
  void GetConservativesFromPrimitives(const double *primit, double *conserv) const override {  // :298-315
    for (int eq = 0; eq < num_equation; eq++) conserv[eq] = primit[eq];
    double v2 = 0.;
    for (int d = 0; d < nvel; d++) {
      v2 += primit[1 + d] * primit[1 + d];
      conserv[1 + d] *= primit[0];
    }
    conserv[iTh] = gas_constant * primit[0] * primit[iTh] / (specific_heat_ratio - 1.) + 0.5 * primit[0] * v2;
  }
  void computeSpeciesEnthalpies(const double *, double *h) const override {
    for (int sp = 0; sp < numSpecies; sp++) h[sp] = 0.0;
  }
  void computeStagnationState(const double *stateIn, double *out) const override {  // :367-378
    const double p = ComputePressure(stateIn);
    for (int eq = 0; eq < num_equation; eq++) out[eq] = stateIn[eq];
    for (int d = 0; d < nvel; d++) out[1 + d] = 0.;
    out[iTh] = p / (specific_heat_ratio - 1.);
  }
  void computeStagnantStateWithTemp(const double *stateIn, double Temp, double *out) const override {  // :380-387
    for (int eq = 0; eq < num_equation; eq++) out[eq] = stateIn[eq];
    for (int d = 0; d < nvel; d++) out[1 + d] = 0.;
    out[iTh] = gas_constant / (specific_heat_ratio - 1.) * stateIn[0] * Temp;
  }
  void modifyEnergyForPressure(const double *stateIn, double *stateOut, double p, bool) const override {  // :402-411
    double tmp[MAXEQ];
    for (int eq = 0; eq < num_equation; eq++) tmp[eq] = stateIn[eq];
    double ke = 0.;
    for (int d = 0; d < nvel; d++) ke += stateIn[1 + d] * stateIn[1 + d];
    ke *= 0.5 / stateIn[0];
    for (int eq = 0; eq < num_equation; eq++) stateOut[eq] = tmp[eq];
    stateOut[iTh] = p / (specific_heat_ratio - 1.) + ke;
  }
};

// ------------------------------------------------------------------------------------------
// TransportProperties (src/transport_properties.hpp:48-260)
// ------------------------------------------------------------------------------------------
class TransportProperties {
 public:
  GasMixture *mixture;
  int numSpecies, dim, nvel, numActiveSpecies, num_equation;
  bool ambipolar, twoTemperature;
  const double Xeps_ = 1.0e-30;
  explicit TransportProperties(GasMixture *m) : mixture(m) {
    numSpecies = m->numSpecies;
    dim = m->dim;
    nvel = m->nvel;
    numActiveSpecies = m->numActiveSpecies;
    ambipolar = m->ambipolar;
    twoTemperature = m->twoTemperature;
    num_equation = m->num_equation;
  }
  virtual ~TransportProperties() {}
  virtual void ComputeFluxTransportProperties(const double *state, const double *gradUp, const double *Efield,
                                              double radius, double distance, double *transportBuffer,
                                              double *diffusionVelocity) = 0;
  virtual void ComputeSourceTransportProperties(const double *state, const double *Up, const double *gradUp,
                                                const double *Efield, double distance, double *globalTransport,
                                                double *speciesTransport, double *diffusionVelocity,
                                                double *n_sp) = 0;
  virtual void GetViscosities(const double *conserved, const double *primitive, double *visc) = 0;

  // src/transport_properties.cpp:59-201
  void correctMassDiffusionFlux(const double *Y_sp, double *diffusionVelocity) const {
    double Vc[MAXDIM];
    for (int v = 0; v < nvel; v++) Vc[v] = 0.0;
    for (int sp = 0; sp < numSpecies; sp++)
      for (int d = 0; d < nvel; d++) Vc[d] += Y_sp[sp] * diffusionVelocity[sp + d * numSpecies];
    for (int sp = 0; sp < numSpecies; sp++)
      for (int d = 0; d < nvel; d++) diffusionVelocity[sp + d * numSpecies] -= Vc[d];
  }
  double computeMixtureElectricConductivity(const double *mobility, const double *n_sp) const {
    double mho = 0.0;
    for (int sp = 0; sp < numSpecies; sp++)
      mho += mobility[sp] * n_sp[sp] * mixture->GetGasParams(sp, TPSRHS_SPECIES_CHARGES);
    return mho;
  }
  void addAmbipolarEfield(const double *mobility, const double *n_sp, double *diffusionVelocity) const {
    const double mho = computeMixtureElectricConductivity(mobility, n_sp);
    double ambE[MAXDIM];
    for (int v = 0; v < nvel; v++) ambE[v] = 0.0;
    for (int sp = 0; sp < numSpecies; sp++)
      for (int d = 0; d < nvel; d++)
        ambE[d] -= diffusionVelocity[sp + d * numSpecies] * n_sp[sp] *
                   mixture->GetGasParams(sp, TPSRHS_SPECIES_CHARGES);
    for (int d = 0; d < nvel; d++) ambE[d] /= (mho + Xeps_);
    for (int sp = 0; sp < numSpecies; sp++)
      for (int d = 0; d < nvel; d++) diffusionVelocity[sp + d * numSpecies] += mobility[sp] * ambE[d];
  }
  void addMixtureDrift(const double *mobility, const double *, const double *Efield,
                       double *diffusionVelocity) const {
    for (int sp = 0; sp < numSpecies; sp++) {
      if (mixture->GetGasParams(sp, TPSRHS_SPECIES_CHARGES) == 0.0) continue;
      for (int d = 0; d < nvel; d++) diffusionVelocity[sp + d * numSpecies] += mobility[sp] * Efield[d];
    }
  }
  double linearAverage(const double *X_sp, const double *speciesTransport) const {
    double average = 0.0;
    for (int sp = 0; sp < numSpecies; sp++) average += X_sp[sp] * speciesTransport[sp];
    return average;
  }
  void CurtissHirschfelder(const double *X_sp, const double *Y_sp, const double *binaryDiff,
                           double *avgDiff) const {
    for (int sp = 0; sp < numSpecies; sp++) avgDiff[sp] = 0.0;
    for (int spI = 0; spI < numSpecies; spI++) {
      for (int spJ = 0; spJ < numSpecies; spJ++) {
        if (spI == spJ) continue;
        avgDiff[spI] += (X_sp[spJ] + Xeps_) / binaryDiff[spI + spJ * numSpecies];
      }
      avgDiff[spI] = (1.0 - Y_sp[spI]) / avgDiff[spI];
    }
  }
};

// DryAirTransport: src/transport_properties.cpp:205-330
class DryAirTransport : public TransportProperties {
 public:
  double visc_mult, bulk_visc_mult, C1_, S0_, Pr_, Sc, gas_constant, cp_div_pr;
  DryAirTransport(GasMixture *m, const tpsrhs_dry_air &in) : TransportProperties(m) {
    visc_mult = in.visc_mult;
    bulk_visc_mult = in.bulk_visc_mult;
    C1_ = in.sutherland_C1;
    S0_ = in.sutherland_S0;
    Pr_ = in.sutherland_Pr;
    Sc = 0.71;
    gas_constant = m->GetGasConstant();
    const double g = m->GetSpecificHeatRatio();
    cp_div_pr = g * gas_constant / (Pr_ * (g - 1.));
  }
  void ComputeFluxTransportProperties(const double *state, const double *, const double *, double, double,
                                      double *transportBuffer, double *diffusionVelocity) override {  // :224-266
    const double p = mixture->ComputePressure(state);
    const double temp = p / gas_constant / state[0];
    for (int i = 0; i < NUM_FLUX_TRANS; i++) transportBuffer[i] = 0.0;
    const double viscosity = (C1_ * visc_mult * std::pow(temp, 1.5) / (temp + S0_));
    transportBuffer[VISCOSITY] = viscosity;
    transportBuffer[BULK_VISCOSITY] = bulk_visc_mult * viscosity;
    transportBuffer[HEAVY_THERMAL_CONDUCTIVITY] = cp_div_pr * transportBuffer[VISCOSITY];
    for (int v = 0; v < nvel; v++)
      for (int sp = 0; sp < numSpecies; sp++) diffusionVelocity[sp + v * numSpecies] = 0.0;
  }
  void ComputeSourceTransportProperties(const double *, const double *, const double *, const double *, double,
                                        double *globalTransport, double *speciesTransport,
                                        double *diffusionVelocity, double *n_sp) override {
    for (int i = 0; i < NUM_SRC_TRANS; i++) globalTransport[i] = 0.0;
    for (int i = 0; i < numSpecies; i++) speciesTransport[i] = 0.0;
    for (int i = 0; i < numSpecies * nvel; i++) diffusionVelocity[i] = 0.0;
    for (int i = 0; i < numSpecies; i++) n_sp[i] = 0.0;
  }
  void GetViscosities(const double *conserved, const double *, double *visc) override {  // :268-276
    const double p = mixture->ComputePressure(conserved);
    const double temp = p / gas_constant / conserved[0];
    visc[0] = (C1_ * visc_mult * std::pow(temp, 1.5) / (temp + S0_));
    visc[1] = bulk_visc_mult * visc[0];
  }
};

// ------------------------------------------------------------------------------------------
// MixingLengthTransport (src/mixing_length_transport.cpp:44-169): the molecular transport plus an algebraic eddy
// viscosity in the flux properties; the source properties are the molecular ones
// ------------------------------------------------------------------------------------------
class MixingLengthTransport : public TransportProperties {
 public:
  TransportProperties *molecular_transport_;
  double max_mixing_length_, Prt_, Let_, bulk_mult_;
  MixingLengthTransport(GasMixture *mix, const tpsrhs_mixing_length &in, TransportProperties *molecular)
      : TransportProperties(mix),
        molecular_transport_(molecular),
        max_mixing_length_(in.max_mixing_length),
        Prt_(in.pr_ratio),
        Let_(in.lewis),
        bulk_mult_(in.bulk_multiplier) {}
  void ComputeFluxTransportProperties(const double *state, const double *gradUp, const double *Efield, double radius,
                                      double distance, double *transportBuffer, double *diffusionVelocity) override {
    molecular_transport_->ComputeFluxTransportProperties(state, gradUp, Efield, radius, distance, transportBuffer,
                                                         diffusionVelocity);
    const double kappa = transportBuffer[HEAVY_THERMAL_CONDUCTIVITY];
    const double mu = transportBuffer[VISCOSITY];
    const double cp_over_Pr = kappa / mu;
    double primitiveState[MAXEQ];
    mixture->GetPrimitivesFromConservatives(state, primitiveState);
    const double rho = state[0];
    double ur = 0;
    if (nvel != dim) ur = primitiveState[1];
    double S = 0;
    for (int i = 0; i < dim; i++)
      for (int j = 0; j < dim; j++) {
        const double ui_xj = gradUp[(1 + i) + j * num_equation];
        const double uj_xi = gradUp[(1 + j) + i * num_equation];
        const double Sij = 0.5 * (ui_xj + uj_xi);
        S += 2 * Sij * Sij;
      }
    if (nvel != dim) {
      const double ut = primitiveState[3];
      const double ut_r = gradUp[3 + 0 * num_equation];
      const double ut_z = gradUp[3 + 1 * num_equation];
      double Szx = 0.5 * ut_r;
      if (radius > 0) Szx -= 0.5 * ut / radius;
      const double Szy = 0.5 * ut_z;
      double Szz = 0.0;
      if (radius > 0) Szz += ur / radius;
      S += 2 * (2 * Szx * Szx + 2 * Szy * Szy + Szz * Szz);
    }
    S = std::sqrt(S);
    double mixing_length = 0.41 * distance;
    if (mixing_length > max_mixing_length_) mixing_length = max_mixing_length_;
    const double mut = rho * mixing_length * mixing_length * S;
    transportBuffer[VISCOSITY] += mut;
    transportBuffer[BULK_VISCOSITY] += bulk_mult_ * mut;
    const double Pr_over_Prt = Prt_;
    const double kappat = mut * cp_over_Pr * Pr_over_Prt;
    transportBuffer[HEAVY_THERMAL_CONDUCTIVITY] += kappat;
  }
  void ComputeSourceTransportProperties(const double *state, const double *Up, const double *gradUp, const double *Efield,
                                        double distance, double *globalTransport, double *speciesTransport,
                                        double *diffusionVelocity, double *n_sp) override {
    molecular_transport_->ComputeSourceTransportProperties(state, Up, gradUp, Efield, distance, globalTransport,
                                                           speciesTransport, diffusionVelocity, n_sp);
  }
  void GetViscosities(const double *conserved, const double *primitive, double *visc) override {
    molecular_transport_->GetViscosities(conserved, primitive, visc);
  }
};

// ------------------------------------------------------------------------------------------
// Fluxes (src/fluxes.cpp)
// ------------------------------------------------------------------------------------------
class Fluxes {
 public:
  GasMixture *mixture;
  TransportProperties *transport;
  int eqSystem, dim, nvel, num_equation, numActiveSpecies;
  bool axisymmetric;
  // sub-grid scale model and viscous sponge (src/fluxes.hpp:77-84); set by the operator from tpsrhs_physics
  int sgs_model_type_ = 0;
  double sgs_model_const_ = 0.0, sgs_model_floor_ = 0.0;
  tpsrhs_visc_sponge vsd_{};
  Fluxes(GasMixture *m, int eqSys, TransportProperties *t, int neq, int dim_, bool axisym)
      : mixture(m), transport(t), eqSystem(eqSys), dim(dim_), num_equation(neq), axisymmetric(axisym) {
    nvel = m->nvel;
    numActiveSpecies = m->numActiveSpecies;
  }

  // src/fluxes.cpp:513-537
  void sgsSmag(const double *state, const double *gradUp, double delta, double &mu) const {
    double Sij[6];
    double Smag = 0.;
    const double Cd = sgs_model_const_;
    Sij[0] = gradUp[1 + 0 * num_equation];
    Sij[1] = gradUp[2 + 1 * num_equation];
    Sij[2] = gradUp[3 + 2 * num_equation];
    Sij[3] = 0.5 * (gradUp[1 + 1 * num_equation] + gradUp[2 + 0 * num_equation]);
    Sij[4] = 0.5 * (gradUp[1 + 2 * num_equation] + gradUp[3 + 0 * num_equation]);
    Sij[5] = 0.5 * (gradUp[2 + 2 * num_equation] + gradUp[3 + 1 * num_equation]);
    for (int i = 0; i < 3; i++) Smag += Sij[i] * Sij[i];
    for (int i = 3; i < 6; i++) Smag += 2.0 * Sij[i] * Sij[i];
    Smag = std::sqrt(2.0 * Smag);
    const double l_floor = sgs_model_floor_;
    const double d_model = Cd * std::max(delta - l_floor, 0.0);
    mu = state[0] * d_model * d_model * Smag;
  }
  // src/fluxes.cpp:543-665, the branch without LAPACK (the one the reference's device build runs)
  void sgsSigma(const double *state, const double *gradUp, double delta, double &mu) const {
    const double Cd = sgs_model_const_;
    const double sml = 1.0e-12;
    const double l_floor = sgs_model_floor_;
    const double d_model = std::max((delta - l_floor), sml);
    double Qij[3][3], B[3][3], ev[3], sigma[3];
    const double pi = 3.14159265359;
    const double onethird = 1. / 3.;
    for (int i = 0; i < dim; i++)
      for (int j = 0; j < dim; j++) {
        Qij[i][j] = 0;
        for (int k = 0; k < dim; k++) Qij[i][j] += gradUp[k + 1 + i * num_equation] * gradUp[k + 1 + j * num_equation];
      }
    const double d4 = std::pow(d_model, 4);
    for (int j = 0; j < dim; j++)
      for (int i = 0; i < dim; i++) Qij[i][j] *= d4;
    const double p1 = Qij[0][1] * Qij[0][1] + Qij[0][2] * Qij[0][2] + Qij[1][2] * Qij[1][2];
    const double q = onethird * (Qij[0][0] + Qij[1][1] + Qij[2][2]);
    const double p2 = (Qij[0][0] - q) * (Qij[0][0] - q) + (Qij[1][1] - q) * (Qij[1][1] - q) +
                      (Qij[2][2] - q) * (Qij[2][2] - q) + 2.0 * p1;
    const double p = std::sqrt(std::max(p2, 0.0) / 6.0);
    for (int j = 0; j < dim; j++)
      for (int i = 0; i < dim; i++) B[i][j] = Qij[i][j];
    for (int i = 0; i < dim; i++) B[i][i] -= q;
    for (int j = 0; j < dim; j++)
      for (int i = 0; i < dim; i++) B[i][j] *= (1.0 / std::max(p, sml));
    const double detB = B[0][0] * (B[1][1] * B[2][2] - B[2][1] * B[1][2]) -
                        B[0][1] * (B[1][0] * B[2][2] - B[2][0] * B[1][2]) +
                        B[0][2] * (B[1][0] * B[2][1] - B[2][0] * B[1][1]);
    const double r = 0.5 * detB;
    double phi;
    if (r <= -1.0) {
      phi = onethird * pi;
    } else if (r >= 1.0) {
      phi = 0.0;
    } else {
      phi = onethird * std::acos(r);
    }
    ev[0] = q + 2.0 * p * std::cos(phi);
    ev[2] = q + 2.0 * p * std::cos(phi + (2.0 * onethird * pi));
    ev[1] = 3.0 * q - ev[0] - ev[2];
    sigma[0] = std::sqrt(std::max(ev[0], sml));
    sigma[1] = std::sqrt(std::max(ev[1], sml));
    sigma[2] = std::sqrt(std::max(ev[2], sml));
    mu = sigma[2] * (sigma[0] - sigma[1]) * (sigma[1] - sigma[2]);
    mu = std::max(mu, 0.0);
    mu /= (sigma[0] * sigma[0]);
    mu *= (Cd * Cd);
    mu *= state[0];
    if (mu != mu) mu = 0.0;
  }
  // src/fluxes.cpp:669-688
  void viscSpongePlanar(const double *x, double &wgt) const {
    const double factor = std::max(vsd_.ratio, 1.0);
    const double width = vsd_.width;
    double dist = 0.;
    for (int d = 0; d < dim; d++) dist += (x[d] - vsd_.point[d]) * vsd_.normal[d];
    wgt = 0.5 * (std::tanh(dist / width - 2.0) + 1.0);
    wgt *= (factor - 1.0);
    wgt += 1.0;
  }
  // src/fluxes.cpp:221-246 (shared by the interior and the boundary routine)
  void sgsAndSponge(const double *state, const double *gradUp, const double *transip, double delta, double &visc,
                    double &bulkViscosity, double &k, double *diffusionVelocity) const {
    const double Pr_Cp = visc / k;
    if (sgs_model_type_ > 0) {
      double mu_sgs = 0.;
      if (sgs_model_type_ == 1) sgsSmag(state, gradUp, delta, mu_sgs);
      if (sgs_model_type_ == 2) sgsSigma(state, gradUp, delta, mu_sgs);
      bulkViscosity *= (1.0 + mu_sgs / visc);
      visc += mu_sgs;
      k += (mu_sgs / Pr_Cp);
    }
    if (vsd_.enabled) {
      double wgt = 0.;
      viscSpongePlanar(transip, wgt);
      visc *= wgt;
      bulkViscosity *= wgt;
      k *= wgt;
      for (int sp = 0; sp < numActiveSpecies; sp++)
        for (int d = 0; d < dim; d++) diffusionVelocity[sp + d * mixture->numSpecies] *= wgt;
    }
  }

  // src/fluxes.cpp:135-170
  void ComputeConvectiveFluxes(const double *state, double *flux) const {
    double Pe = 0.0;
    const double pres = mixture->ComputePressure(state, &Pe);
    for (int d = 0; d < dim; d++) {
      flux[0 + d * num_equation] = state[d + 1];
      for (int i = 0; i < nvel; i++) flux[1 + i + d * num_equation] = state[i + 1] * state[d + 1] / state[0];
      flux[1 + d + d * num_equation] += pres;
    }
    const double H = (state[1 + nvel] + pres) / state[0];
    for (int d = 0; d < dim; d++) flux[1 + nvel + d * num_equation] = state[d + 1] * H;
    for (int sp = 0; sp < numActiveSpecies; sp++)
      for (int d = 0; d < dim; d++)
        flux[nvel + 2 + sp + d * num_equation] = state[nvel + 2 + sp] * state[1 + d] / state[0];
    if (mixture->twoTemperature) {
      const double electronEnthalpy = (state[num_equation - 1] + Pe) / state[0];
      for (int d = 0; d < dim; d++) flux[num_equation - 1 + d * num_equation] = electronEnthalpy * state[1 + d];
    }
  }

  // src/fluxes.cpp:178-335
  void ComputeViscousFluxes(const double *state, const double *gradUp, const double *transip, double delta,
                            double distance, double *flux) const {
    for (int d = 0; d < dim; d++)
      for (int eq = 0; eq < num_equation; eq++) flux[eq + d * num_equation] = 0.;
    if (eqSystem == TPSRHS_EULER) return;

    double radius = -1;
    if (axisymmetric) radius = transip[0];

    double vel[MAXDIM], vtmp[MAXDIM], stress[MAXDIM * MAXDIM];
    double Efield[MAXDIM];
    for (int v = 0; v < nvel; v++) Efield[v] = 0.0;

    const int numSpecies = mixture->numSpecies;
    const bool twoT = mixture->twoTemperature;

    double speciesEnthalpies[MAXSP];
    mixture->computeSpeciesEnthalpies(state, speciesEnthalpies);

    double transportBuffer[NUM_FLUX_TRANS];
    double diffusionVelocity[MAXSP * MAXDIM];
    transport->ComputeFluxTransportProperties(state, gradUp, Efield, radius, distance, transportBuffer,
                                              diffusionVelocity);
    double visc = transportBuffer[VISCOSITY];
    double bulkViscosity = transportBuffer[BULK_VISCOSITY];
    bulkViscosity -= 2. / 3. * visc;
    double k = transportBuffer[HEAVY_THERMAL_CONDUCTIVITY];
    double ke = transportBuffer[ELECTRON_THERMAL_CONDUCTIVITY];
    sgsAndSponge(state, gradUp, transip, delta, visc, bulkViscosity, k, diffusionVelocity);

    if (twoT) {
      for (int d = 0; d < dim; d++) {
        double qeFlux = ke * gradUp[num_equation - 1 + d * num_equation];
        flux[1 + nvel + d * num_equation] += qeFlux;
        flux[num_equation - 1 + d * num_equation] += qeFlux;
        flux[num_equation - 1 + d * num_equation] -=
            speciesEnthalpies[numSpecies - 2] * diffusionVelocity[numSpecies - 2 + d * numSpecies];
      }
    } else {
      k += ke;
    }

    const double ur = (axisymmetric ? state[1] / state[0] : 0);
    const double ut = (axisymmetric ? state[3] / state[0] : 0);

    for (int d = 0; d < dim; d++) flux[0 + d * num_equation] = 0.;

    double divV = 0.;
    for (int i = 0; i < dim; i++) {
      for (int j = 0; j < dim; j++)
        stress[i + j * dim] = gradUp[(1 + j) + i * num_equation] + gradUp[(1 + i) + j * num_equation];
      divV += gradUp[(1 + i) + i * num_equation];
    }
    for (int i = 0; i < dim; i++)
      for (int j = 0; j < dim; j++) stress[i + j * dim] *= visc;
    if (axisymmetric && radius > 0) divV += ur / radius;
    for (int i = 0; i < dim; i++) stress[i + i * dim] += bulkViscosity * divV;
    for (int i = 0; i < dim; i++)
      for (int j = 0; j < dim; j++) flux[(1 + i) + j * num_equation] = stress[i + j * dim];

    double tau_tr = 0, tau_tz = 0;
    if (axisymmetric) {
      const double ut_r = gradUp[3 + 0 * num_equation];
      const double ut_z = gradUp[3 + 1 * num_equation];
      tau_tr = ut_r;
      if (radius > 0) tau_tr -= ut / radius;
      tau_tr *= visc;
      tau_tz = visc * ut_z;
      flux[(1 + 2) + 0 * num_equation] = tau_tr;
      flux[(1 + 2) + 1 * num_equation] = tau_tz;
    }

    for (int d = 0; d < dim; d++) vel[d] = state[1 + d] / state[0];
    for (int i = 0; i < dim; i++) {
      vtmp[i] = 0.0;
      for (int j = 0; j < dim; j++) vtmp[i] += stress[i + j * dim] * vel[j];
    }
    for (int d = 0; d < dim; d++) {
      flux[(1 + nvel) + d * num_equation] += vtmp[d];
      flux[(1 + nvel) + d * num_equation] += k * gradUp[(1 + nvel) + d * num_equation];
      for (int sp = 0; sp < numSpecies; sp++)
        flux[(1 + nvel) + d * num_equation] -= speciesEnthalpies[sp] * diffusionVelocity[sp + d * numSpecies];
    }
    if (axisymmetric) {
      flux[(1 + nvel) + 0 * num_equation] += ut * tau_tr;
      flux[(1 + nvel) + 1 * num_equation] += ut * tau_tz;
    }
    for (int sp = 0; sp < numActiveSpecies; sp++)
      for (int d = 0; d < dim; d++)
        flux[(nvel + 2 + sp) + d * num_equation] = -state[nvel + 2 + sp] * diffusionVelocity[sp + d * numSpecies];
  }

  // src/fluxes.cpp:344-505
  void ComputeBdrViscousFluxes(const double *state, const double *gradUp, const double *transip, double delta,
                               double distance, const BoundaryViscousFluxData &bcFlux, double *normalFlux) const {
    for (int eq = 0; eq < num_equation; eq++) normalFlux[eq] = 0.;
    if (eqSystem == TPSRHS_EULER) return;

    double radius = -1;
    if (axisymmetric) radius = transip[0];

    double stress[MAXDIM * MAXDIM];
    double Efield[MAXDIM];
    for (int v = 0; v < nvel; v++) Efield[v] = 0.0;

    const int numSpecies = mixture->numSpecies;
    const bool twoT = mixture->twoTemperature;

    double speciesEnthalpies[MAXSP];
    mixture->computeSpeciesEnthalpies(state, speciesEnthalpies);

    double transportBuffer[NUM_FLUX_TRANS];
    double diffusionVelocity[MAXSP * MAXDIM];
    transport->ComputeFluxTransportProperties(state, gradUp, Efield, radius, distance, transportBuffer,
                                              diffusionVelocity);
    double visc = transportBuffer[VISCOSITY];
    double bulkViscosity = transportBuffer[BULK_VISCOSITY];
    bulkViscosity -= 2. / 3. * visc;
    double k = transportBuffer[HEAVY_THERMAL_CONDUCTIVITY];
    double ke = transportBuffer[ELECTRON_THERMAL_CONDUCTIVITY];
    sgsAndSponge(state, gradUp, transip, delta, visc, bulkViscosity, k, diffusionVelocity);

    const int primFluxSize = twoT ? numSpecies + nvel + 2 : numSpecies + nvel + 1;
    double normalPrimFlux[MAXEQ + 2];
    for (int eq = 0; eq < primFluxSize; eq++) normalPrimFlux[eq] = 0.0;

    for (int sp = 0; sp < numSpecies; sp++)
      for (int d = 0; d < dim; d++) normalPrimFlux[sp] += diffusionVelocity[sp + d * numSpecies] * bcFlux.normal[d];
    for (int i = 0; i < numSpecies; i++)
      if (bcFlux.primFluxIdxs[i]) normalPrimFlux[i] = bcFlux.primFlux[i];

    const double ur = (axisymmetric ? state[1] / state[0] : 0);
    const double ut = (axisymmetric ? state[3] / state[0] : 0);

    double divV = 0.;
    for (int i = 0; i < dim; i++) {
      for (int j = 0; j < dim; j++)
        stress[i + j * dim] = gradUp[(1 + j) + i * num_equation] + gradUp[(1 + i) + j * num_equation];
      divV += gradUp[(1 + i) + i * num_equation];
    }
    for (int i = 0; i < dim; i++)
      for (int j = 0; j < dim; j++) stress[i + j * dim] *= visc;
    if (axisymmetric && radius > 0) divV += ur / radius;
    for (int i = 0; i < dim; i++) stress[i + i * dim] += bulkViscosity * divV;
    for (int i = 0; i < dim; i++)
      for (int j = 0; j < dim; j++) normalPrimFlux[numSpecies + i] += stress[i + j * dim] * bcFlux.normal[j];

    if (axisymmetric) {
      const double ut_r = gradUp[3 + 0 * num_equation];
      const double ut_z = gradUp[3 + 1 * num_equation];
      double tau_tr = ut_r;
      if (radius > 0) tau_tr -= ut / radius;
      tau_tr *= visc;
      const double tau_tz = visc * ut_z;
      normalPrimFlux[numSpecies + nvel - 1] += tau_tr * bcFlux.normal[0];
      normalPrimFlux[numSpecies + nvel - 1] += tau_tz * bcFlux.normal[1];
    }

    if (twoT) {
      for (int d = 0; d < dim; d++)
        normalPrimFlux[primFluxSize - 1] -= ke * gradUp[(num_equation - 1) + d * num_equation] * bcFlux.normal[d];
      normalPrimFlux[primFluxSize - 1] += speciesEnthalpies[numSpecies - 2] * normalPrimFlux[numSpecies - 2];
    } else {
      k += ke;
    }
    for (int d = 0; d < dim; d++)
      normalPrimFlux[numSpecies + nvel] -= k * gradUp[(1 + nvel) + d * num_equation] * bcFlux.normal[d];
    for (int sp = 0; sp < numSpecies; sp++) {
      if (twoT && (sp == numSpecies - 2)) continue;
      normalPrimFlux[numSpecies + nvel] += speciesEnthalpies[sp] * normalPrimFlux[sp];
    }
    for (int i = numSpecies; i < primFluxSize; i++)
      if (bcFlux.primFluxIdxs[i]) normalPrimFlux[i] = bcFlux.primFlux[i];

    double vel0[MAXDIM];
    for (int d = 0; d < nvel; d++) vel0[d] = state[1 + d] / state[0];

    for (int sp = 0; sp < numActiveSpecies; sp++)
      normalFlux[nvel + 2 + sp] = -state[nvel + 2 + sp] * normalPrimFlux[sp];
    for (int d = 0; d < nvel; d++) normalFlux[d + 1] = normalPrimFlux[numSpecies + d];
    for (int d = 0; d < nvel; d++) normalFlux[nvel + 1] += normalPrimFlux[numSpecies + d] * vel0[d];
    normalFlux[nvel + 1] -= normalPrimFlux[numSpecies + nvel];
    if (twoT) {
      normalFlux[nvel + 1] -= normalPrimFlux[primFluxSize - 1];
      normalFlux[num_equation - 1] = -normalPrimFlux[primFluxSize - 1];
    }
  }
};

// ------------------------------------------------------------------------------------------
// RiemannSolverTPS (src/riemann_solver.cpp:53-115), Lax-Friedrichs only
// ------------------------------------------------------------------------------------------
class RiemannSolver {
 public:
  int num_equation;
  GasMixture *mixture;
  Fluxes *fluxClass;
  bool useRoe = false;
  RiemannSolver(int neq, GasMixture *m, Fluxes *f, bool roe = false)
      : num_equation(neq), mixture(m), fluxClass(f), useRoe(roe) {}
  // src/riemann_solver.cpp:66-72
  void Eval(const double *state1, const double *state2, const double *nor, double *flux, bool LF = false) const {
    if (useRoe && !LF)
      Eval_Roe(state1, state2, nor, flux);
    else
      Eval_LF(state1, state2, nor, flux);
  }
  // src/riemann_solver.cpp:117-206 (Roe, Lohner); as there: 2-D velocity components, gamma - 1 = 0.4
  void Eval_Roe(const double *state1, const double *state2, const double *nor, double *flux) const {
    const int dim = mixture->dim;
    const int NS_eq = 2 + dim;
    double normag = 0;
    for (int i = 0; i < dim; i++) normag += nor[i] * nor[i];
    normag = std::sqrt(normag);
    double unitN[3];
    for (int d = 0; d < dim; d++) unitN[d] = nor[d] / normag;
    double fluxes1[MAXEQ * MAXDIM], fluxes2[MAXEQ * MAXDIM], meanFlux[MAXEQ];
    fluxClass->ComputeConvectiveFluxes(state1, fluxes1);
    fluxClass->ComputeConvectiveFluxes(state2, fluxes2);
    for (int eq = 0; eq < NS_eq; eq++) {
      meanFlux[eq] = 0.;
      for (int d = 0; d < dim; d++)
        meanFlux[eq] += (fluxes1[eq + d * num_equation] + fluxes2[eq + d * num_equation]) * unitN[d];
    }
    const double r = std::sqrt(state1[0] * state2[0]);
    double vel[3];
    for (int i = 0; i < dim; i++) {
      vel[i] = state1[i + 1] / std::sqrt(state1[0]) + state2[i + 1] / std::sqrt(state2[0]);
      vel[i] /= std::sqrt(state1[0]) + std::sqrt(state2[0]);
    }
    double qk = 0.;
    for (int d = 0; d < dim; d++) qk += vel[d] * unitN[d];
    const double p1 = mixture->ComputePressure(state1);
    const double p2 = mixture->ComputePressure(state2);
    double H = (state1[1 + dim] + p1) / std::sqrt(state1[0]) + (state2[1 + dim] + p2) / std::sqrt(state2[0]);
    H /= std::sqrt(state1[0]) + std::sqrt(state2[0]);
    const double a2 = 0.4 * (H - 0.5 * (vel[0] * vel[0] + vel[1] * vel[1]));
    const double a = std::sqrt(a2);
    double lamb[3] = {qk, qk + a, qk - a};
    if (std::fabs(lamb[0]) < 1e-4) lamb[0] = 1e-4;
    const double deltaP = p2 - p1;
    const double deltaU = state2[1] / state2[0] - state1[1] / state1[0];
    const double deltaV = state2[2] / state2[0] - state1[2] / state1[0];
    const double deltaQk = deltaU * unitN[0] + deltaV * unitN[1];
    double DF1[4], DF4[4], DF5[4];
    DF1[0] = 1.;
    DF1[1] = vel[0];
    DF1[2] = vel[1];
    DF1[3] = 0.5 * (vel[0] * vel[0] + vel[1] * vel[1]);
    for (int i = 0; i < 4; i++) DF1[i] *= state2[0] - state1[0] - deltaP / a2;
    DF1[1] += r * (deltaU - unitN[0] * deltaQk);
    DF1[2] += r * (deltaV - unitN[1] * deltaQk);
    DF1[3] += r * (vel[0] * deltaU + vel[1] * deltaV - qk * deltaQk);
    for (int i = 0; i < 4; i++) DF1[i] *= std::fabs(lamb[0]);
    DF4[0] = 1.;
    DF4[1] = vel[0] + unitN[0] * a;
    DF4[2] = vel[1] + unitN[1] * a;
    DF4[3] = H + qk * a;
    for (int i = 0; i < 4; i++) DF4[i] *= std::fabs(lamb[1]) * (deltaP + r * a * deltaQk) * 0.5 / a2;
    DF5[0] = 1.;
    DF5[1] = vel[0] - unitN[0] * a;
    DF5[2] = vel[1] - unitN[1] * a;
    DF5[3] = H - qk * a;
    for (int i = 0; i < 4; i++) DF5[i] *= std::fabs(lamb[2]) * (deltaP - r * a * deltaQk) * 0.5 / a2;
    for (int i = 0; i < NS_eq; i++) flux[i] = (meanFlux[i] - (DF1[i] + DF4[i] + DF5[i])) * 0.5 * normag;
  }
  void ComputeFluxDotN(const double *state, const double *nor, double *fluxN) const {
    const int dim = mixture->dim;
    double fluxes[MAXEQ * MAXDIM];
    fluxClass->ComputeConvectiveFluxes(state, fluxes);
    for (int eq = 0; eq < num_equation; eq++) {
      fluxN[eq] = 0;
      for (int d = 0; d < dim; d++) fluxN[eq] += fluxes[eq + d * num_equation] * nor[d];
    }
  }
  void Eval_LF(const double *state1, const double *state2, const double *nor, double *flux) const {
    const int dim = mixture->dim;
    const double maxE1 = mixture->ComputeMaxCharSpeed(state1);
    const double maxE2 = mixture->ComputeMaxCharSpeed(state2);
    const double maxE = std::fmax(maxE1, maxE2);
    double flux1[MAXEQ], flux2[MAXEQ];
    ComputeFluxDotN(state1, nor, flux1);
    ComputeFluxDotN(state2, nor, flux2);
    double normag = 0;
    for (int i = 0; i < dim; i++) normag += nor[i] * nor[i];
    normag = std::sqrt(normag);
    for (int i = 0; i < num_equation; i++)
      flux[i] = 0.5 * (flux1[i] + flux2[i]) - 0.5 * maxE * (state2[i] - state1[i]) * normag;
  }
};

// ------------------------------------------------------------------------------------------
// Boundary conditions: the types of SURVEY.md 8a(a9)
// ------------------------------------------------------------------------------------------
class BoundaryCondition {
 public:
  int category, type;
  GasMixture *mixture;
  Fluxes *fluxClass;
  RiemannSolver *rsolver;
  int dim, nvel, num_equation, numActiveSpecies;
  bool useBCinGrad;
  double inputState[4 + MAXSP];
  double wallTemp = 0.0;
  int hvyCond = -1, elecCond = -1;
  BoundaryViscousFluxData bcFlux;
  BoundaryPrimitiveData bcState;
  // state of the non-reflecting inlet / outlet types (src/inletBC.cpp:60-160, src/outletBC.cpp:60-210)
  bool nonReflecting = false;
  double tangent1[3] = {0, 0, 0};
  double area_ = 0.0, refLength = 1.0;
  const double *dt = nullptr;            // the reference member `double &dt` (src/BoundaryCondition.hpp:54)
  mutable std::vector<double> boundaryU; // [point][eq], advanced by every flux evaluation
  double meanUp[MAXEQ];
  bool bdrUInit = false;

  BoundaryCondition(const tpsrhs_bc &bc, GasMixture *m, Fluxes *f, RiemannSolver *r, bool axisym, bool bcInGrad)
      : category(bc.category), type(bc.type), mixture(m), fluxClass(f), rsolver(r), useBCinGrad(bcInGrad) {
    dim = m->dim;
    nvel = m->nvel;
    num_equation = m->num_equation;
    numActiveSpecies = m->numActiveSpecies;
    for (int i = 0; i < 4 + MAXSP; i++) inputState[i] = bc.data[i];
    const int numSpecies = m->numSpecies;
    const int primFluxSize = m->twoTemperature ? numSpecies + nvel + 2 : numSpecies + nvel + 1;
    for (int i = 0; i < MAXEQ; i++) {
      bcFlux.primFlux[i] = 0.0;
      bcFlux.primFluxIdxs[i] = false;
      bcState.prim[i] = 0.0;
      bcState.primIdxs[i] = false;
    }
    (void)primFluxSize;
    if (category == TPSRHS_WALL) {  // src/wallBC.cpp:65-148
      switch (type) {
        case TPSRHS_INV:
          for (int i = 0; i < numSpecies; i++) bcFlux.primFluxIdxs[i] = true;
          if (axisym) {
            bcFlux.primFluxIdxs[numSpecies + nvel] = true;
            if (m->twoTemperature) bcFlux.primFluxIdxs[numSpecies + nvel + 1] = true;
          }
          break;
        case TPSRHS_SLIP:  // src/wallBC.cpp:77-85
          for (int i = 0; i < numSpecies; i++) bcFlux.primFluxIdxs[i] = true;
          if (axisym) {
            bcFlux.primFluxIdxs[numSpecies + nvel] = true;
            if (m->twoTemperature) bcFlux.primFluxIdxs[numSpecies + nvel + 1] = true;
          }
          break;
        case TPSRHS_VISC_ADIAB:
          for (int i = 0; i < numSpecies; i++) bcFlux.primFluxIdxs[i] = true;
          bcFlux.primFluxIdxs[numSpecies + nvel] = true;
          if (m->twoTemperature) bcFlux.primFluxIdxs[numSpecies + nvel + 1] = true;
          break;
        case TPSRHS_VISC_ISOTH:
          for (int i = 0; i < numSpecies; i++) bcFlux.primFluxIdxs[i] = true;
          wallTemp = bc.data[0];
          break;
        case TPSRHS_VISC_GNRL: {  // src/wallBC.cpp:112-148
          hvyCond = static_cast<int>(bc.data[2]);
          elecCond = static_cast<int>(bc.data[3]);
          for (int d = 0; d < nvel; d++) bcState.primIdxs[d + 1] = true;
          for (int i = 0; i < numSpecies; i++) bcFlux.primFluxIdxs[i] = true;
          if (hvyCond == TPSRHS_ISOTH) {
            bcState.prim[nvel + 1] = bc.data[0];
            bcState.primIdxs[nvel + 1] = true;
          } else if (hvyCond == TPSRHS_ADIAB) {
            bcFlux.primFluxIdxs[numSpecies + nvel] = true;
          } else {
            throw std::runtime_error("Thermal condition not understood.");
          }
          if (elecCond == TPSRHS_ISOTH) {
            bcState.prim[num_equation - 1] = bc.data[1];
            bcState.primIdxs[num_equation - 1] = true;
          } else if (elecCond == TPSRHS_ADIAB) {
            bcFlux.primFluxIdxs[numSpecies + nvel + 1] = true;
          } else if (elecCond == TPSRHS_SHTH) {
            if (m->twoTemperature) bcFlux.primFluxIdxs[numSpecies + nvel + 1] = true;
          } else {
            throw std::runtime_error("Electron thermal condition not understood.");
          }
        } break;
        default:
          throw std::runtime_error("wall type outside the hot-path scope");
      }
    } else if (category == TPSRHS_INLET) {
      if (type == TPSRHS_SUB_DENS_VEL_NR || type == TPSRHS_SUB_VEL_CONST_ENT)
        nonReflecting = true;
      else if (type >= TPSRHS_SUB_DENS_VEL_FACE_X && type <= TPSRHS_SUB_DENS_VEL_FACE_Z) {
        if (dim != 3 || axisym) throw std::runtime_error("face-relative inlets: dim == 3");
      } else if (type != TPSRHS_SUB_DENS_VEL)
        throw std::runtime_error("inlet type outside the hot-path scope");
    } else if (category == TPSRHS_OUTLET) {
      if (type == TPSRHS_SUB_P_NR || type == TPSRHS_SUB_MF_NR || type == TPSRHS_SUB_MF_NR_PW)
        nonReflecting = true;
      else if (type != TPSRHS_SUB_P)
        throw std::runtime_error("outlet type outside the hot-path scope");
    }
    if (nonReflecting) {
      if (m->numSpecies > 1 || axisym)
        throw std::runtime_error("non-reflecting boundary conditions: perfect gas, not axisymmetric");
      for (int d = 0; d < 3; d++) tangent1[d] = bc.data[4 + d];
      area_ = bc.data[7];
      for (int eq = 0; eq < MAXEQ; eq++) meanUp[eq] = 0.0;
    }
  }

  // Non-reflecting inlet (src/inletBC.cpp:576-727) and outlets (src/outletBC.cpp:573-728, 739-892, 894-1027):
  // characteristic estimate of d(U)/dt at the boundary point from the patch mean `meanUp`, the normal gradient
  // and the target; the boundary state boundaryU[bdrN] is advanced by dt with it; the Riemann solver sees the
  // state BEFORE the update.
  void computeNonReflectingFlux(const double *normal, const double *stateIn, const double *gradState, int bdrN,
                                double *bdrFlux) const {
    const double gamma = mixture->GetSpecificHeatRatio();
    const bool inlet = category == TPSRHS_INLET;
    double unitNorm[3] = {0, 0, 0}, tangent2[3] = {0, 0, 0};
    {
      double mod = 0.;
      for (int d = 0; d < dim; d++) mod += normal[d] * normal[d];
      for (int d = 0; d < dim; d++) unitNorm[d] = normal[d] * ((inlet ? -1. : 1.) / std::sqrt(mod));  // inlet: into the domain
    }
    double meanVel[3] = {0, 0, 0};
    for (int d = 0; d < dim; d++) {
      meanVel[0] += unitNorm[d] * meanUp[d + 1];
      meanVel[1] += tangent1[d] * meanUp[d + 1];
    }
    if (dim == 3) {
      tangent2[0] = unitNorm[1] * tangent1[2] - unitNorm[2] * tangent1[1];
      tangent2[1] = unitNorm[2] * tangent1[0] - unitNorm[0] * tangent1[2];
      tangent2[2] = unitNorm[0] * tangent1[1] - unitNorm[1] * tangent1[0];
      for (int d = 0; d < dim; d++) meanVel[2] += tangent2[d] * meanUp[d + 1];
    }
    double normGrad[MAXEQ];
    for (int eq = 0; eq < num_equation; eq++) {
      normGrad[eq] = 0.;
      for (int d = 0; d < dim; d++) normGrad[eq] += unitNorm[d] * gradState[eq + d * num_equation];
    }
    // DryAir::ComputePressureDerivative(normGrad, stateIn, false), src/equation_of_state.cpp:350-359
    const double Rg = mixture->GetGasConstant();
    const double dpdn = Rg * (mixture->ComputeTemperature(stateIn) * normGrad[0] + stateIn[0] * normGrad[nvel + 1]);
    const double meanP = Rg * meanUp[0] * meanUp[nvel + 1];                      // ComputePressureFromPrimitives :361
    const double speedSound = std::sqrt(gamma * Rg * meanUp[nvel + 1]);          // ComputeSpeedOfSound(meanUp) :337-348
    double meanK = 0.;
    for (int d = 0; d < dim; d++) meanK += meanUp[1 + d] * meanUp[1 + d];
    meanK *= 0.5;
    const double sigma = speedSound / refLength;
    double L1, L2, L3 = 0., L4 = 0., L5;
    if (inlet) {  // src/inletBC.cpp:601-648
      double meanDV[3] = {0, 0, 0};
      for (int d = 0; d < nvel; d++) meanDV[d] = meanUp[1 + d] - inputState[1 + d];
      L1 = 0.;
      for (int d = 0; d < dim; d++) L1 += unitNorm[d] * normGrad[1 + d];
      L1 = dpdn - meanUp[0] * speedSound * L1;
      L1 *= meanVel[0] - speedSound;
      L5 = 0.;
      for (int d = 0; d < dim; d++) L5 += meanDV[d] * unitNorm[d];
      L5 *= sigma * 2. * meanUp[0] * speedSound;
      for (int d = 0; d < dim; d++) L3 += meanDV[d] * tangent1[d];
      L3 *= sigma;
      if (dim == 3) {
        for (int d = 0; d < dim; d++) L4 += meanDV[d] * tangent2[d];
        L4 *= sigma;
      }
      L2 = sigma * speedSound * speedSound * (meanUp[0] - inputState[0]) - 0.5 * L5;
      if (type == TPSRHS_SUB_VEL_CONST_ENT) L2 = 0.;
    } else {  // src/outletBC.cpp:617-646, 781-807, 939-968
      L2 = speedSound * speedSound * normGrad[0] - dpdn;
      L2 *= meanVel[0];
      for (int d = 0; d < dim; d++) L3 += tangent1[d] * normGrad[1 + d];
      L3 *= meanVel[0];
      if (dim == 3) {
        for (int d = 0; d < dim; d++) L4 += tangent2[d] * normGrad[1 + d];
        L4 *= meanVel[0];
      }
      L5 = 0.;
      for (int d = 0; d < dim; d++) L5 += unitNorm[d] * normGrad[1 + d];
      L5 = dpdn + meanUp[0] * speedSound * L5;
      L5 *= meanVel[0] + speedSound;
      if (type == TPSRHS_SUB_P_NR) {
        L1 = sigma * (meanP - inputState[0]);
      } else {
        double vn = meanVel[0];
        if (type == TPSRHS_SUB_MF_NR_PW) {
          vn = 0.;
          for (int d = 0; d < dim; d++) vn += stateIn[1 + d] * unitNorm[d];
          vn /= stateIn[0];
        }
        L1 = -sigma * (vn - inputState[0] / meanUp[0] / area_);
        L1 *= meanUp[0] * speedSound;
      }
    }
    const double d1 = (L2 + 0.5 * (L5 + L1)) / speedSound / speedSound;
    const double d2 = 0.5 * (L5 - L1) / meanUp[0] / speedSound;
    const double d3 = L3, d4 = L4;
    const double d5 = 0.5 * (L5 + L1);
    bdrFlux[0] = d1;
    bdrFlux[1] = meanVel[0] * d1 + meanUp[0] * d2;
    bdrFlux[2] = meanVel[1] * d1 + meanUp[0] * d3;
    if (dim == 3) bdrFlux[3] = meanVel[2] * d1 + meanUp[0] * d4;
    bdrFlux[1 + dim] = meanUp[0] * meanVel[0] * d2;
    bdrFlux[1 + dim] += meanUp[0] * meanVel[1] * d3;
    if (dim == 3) bdrFlux[1 + dim] += meanUp[0] * meanVel[2] * d4;
    bdrFlux[1 + dim] += meanK * d1 + d5 / (gamma - 1.);

    double state2[MAXEQ], stateN[MAXEQ], newU[MAXEQ];
    for (int eq = 0; eq < num_equation; eq++) state2[eq] = boundaryU[eq + bdrN * num_equation];
    for (int eq = 0; eq < num_equation; eq++) stateN[eq] = state2[eq];
    for (int d = 0; d < dim; d++) stateN[1 + d] = 0.;
    for (int d = 0; d < dim; d++) {
      stateN[1] += state2[1 + d] * unitNorm[d];
      stateN[2] += state2[1 + d] * tangent1[d];
      if (dim == 3) stateN[3] += state2[1 + d] * tangent2[d];
    }
    for (int i = 0; i < num_equation; i++) newU[i] = stateN[i] - (*dt) * bdrFlux[i];
    {  // back to Cartesian momentum: M rows = unitNorm, tangent1, tangent2; momX = M^-1 momN
      double M[9] = {0}, invM[9];
      for (int d = 0; d < dim; d++) {  // column-major M(i, d)
        M[0 + dim * d] = unitNorm[d];
        M[1 + dim * d] = tangent1[d];
        if (dim == 3) M[2 + dim * d] = tangent2[d];
      }
      if (dim == 2) {
        const double det = M[0] * M[3] - M[2] * M[1];
        invM[0] = M[3] / det;
        invM[1] = -M[1] / det;
        invM[2] = -M[2] / det;
        invM[3] = M[0] / det;
      } else {
        const double c00 = M[4] * M[8] - M[7] * M[5], c10 = M[7] * M[2] - M[1] * M[8], c20 = M[1] * M[5] - M[4] * M[2];
        const double det = M[0] * c00 + M[3] * c10 + M[6] * c20;
        invM[0] = c00 / det;
        invM[1] = c10 / det;
        invM[2] = c20 / det;
        invM[3] = (M[6] * M[5] - M[3] * M[8]) / det;
        invM[4] = (M[0] * M[8] - M[6] * M[2]) / det;
        invM[5] = (M[3] * M[2] - M[0] * M[5]) / det;
        invM[6] = (M[3] * M[7] - M[6] * M[4]) / det;
        invM[7] = (M[6] * M[1] - M[0] * M[7]) / det;
        invM[8] = (M[0] * M[4] - M[3] * M[1]) / det;
      }
      double momX[3] = {0, 0, 0};
      for (int i = 0; i < dim; i++)
        for (int j = 0; j < dim; j++) momX[i] += invM[i + dim * j] * newU[1 + j];
      for (int d = 0; d < dim; d++) newU[1 + d] = momX[d];
    }
    for (int eq = 0; eq < num_equation; eq++) boundaryU[eq + bdrN * num_equation] = newU[eq];
    rsolver->Eval(stateIn, state2, normal, bdrFlux, true);
  }

  // src/wallBC.cpp:241-266 (only the isothermal wall alters the gradient ghost state)
  void computeBdrPrimitiveStateForGradient(const double *primIn, double *primBC) const {
    for (int eq = 0; eq < num_equation; eq++) primBC[eq] = primIn[eq];
    if (category == TPSRHS_WALL && type == TPSRHS_VISC_ISOTH) {
      for (int i = 0; i < nvel; i++) primBC[1 + i] = 0.0;
      primBC[nvel + 1] = wallTemp;
    }
  }

  void computeBdrFlux(const double *normal, const double *stateIn, const double *gradState, const double *transip,
                      double delta, double distance, double *bdrFlux, int bdrN = -1) const {
    if (nonReflecting) {
      if (bdrN < 0 || !dt) throw std::runtime_error("non-reflecting boundary condition without its boundary state");
      computeNonReflectingFlux(normal, stateIn, gradState, bdrN, bdrFlux);
      return;
    }
    // the reference stores the unit normal in the member bcFlux_ (src/wallBC.cpp:448,492); the
    // oracle's face loop is threaded, so each call works on its own copy
    BoundaryViscousFluxData bcFlux = this->bcFlux;
    if (category == TPSRHS_INLET && type >= TPSRHS_SUB_DENS_VEL_FACE_X && type <= TPSRHS_SUB_DENS_VEL_FACE_Z) {
      // InletBC::subsonicReflectingDensityVelocityFace, src/inletBC.cpp:758-864 (tangentW = the global axis, :453-464)
      const double p = mixture->ComputePressure(stateIn);
      double state2[MAXEQ];
      for (int eq = 0; eq < num_equation; eq++) state2[eq] = stateIn[eq];
      const double wt = 1.0;  // the time ramp of the reference is overwritten by 1 (:770-773)
      const double Un = wt * inputState[1], Ut = wt * inputState[2];
      double unitNorm[3], tangent1[3], tangent2[3] = {0, 0, 0};
      double mod = 0.;
      for (int d = 0; d < dim; d++) mod += normal[d] * normal[d];
      for (int d = 0; d < dim; d++) unitNorm[d] = normal[d] * (-1.0 / std::sqrt(mod));  // inward-facing normal
      tangent2[type - TPSRHS_SUB_DENS_VEL_FACE_X] = 1.0;
      {  // ensure normal is orthogonal to tangent-w
        double tmag = 0.0, tn = 0.0;
        for (int d = 0; d < dim; d++) tmag += tangent2[d] * tangent2[d];
        for (int d = 0; d < dim; d++) tn += tangent2[d] * unitNorm[d];
        for (int d = 0; d < dim; d++) unitNorm[d] -= (tn / tmag) * tangent2[d];
      }
      tangent1[0] = +(unitNorm[1] * tangent2[2] - unitNorm[2] * tangent2[1]);
      tangent1[1] = -(unitNorm[0] * tangent2[2] - unitNorm[2] * tangent2[0]);
      tangent1[2] = +(unitNorm[0] * tangent2[1] - unitNorm[1] * tangent2[0]);
      state2[0] = inputState[0];
      state2[1] = state2[0] * Un;
      state2[2] = state2[0] * Ut;
      state2[3] = state2[0] * 0.0;
      {  // transform from face coords to global: M rows = unitNorm, tangent1, tangent2; momX = M^-1 momN [MFEM CalcInverse]
        const double M[9] = {unitNorm[0], unitNorm[1], unitNorm[2], tangent1[0], tangent1[1], tangent1[2], tangent2[0], tangent2[1], tangent2[2]};
        const double c00 = M[4] * M[8] - M[5] * M[7], c01 = M[5] * M[6] - M[3] * M[8], c02 = M[3] * M[7] - M[4] * M[6];
        const double det = M[0] * c00 + M[1] * c01 + M[2] * c02;
        const double inv[9] = {c00, M[2] * M[7] - M[1] * M[8], M[1] * M[5] - M[2] * M[4],
                               c01, M[0] * M[8] - M[2] * M[6], M[2] * M[3] - M[0] * M[5],
                               c02, M[1] * M[6] - M[0] * M[7], M[0] * M[4] - M[1] * M[3]};
        const double momN[3] = {state2[1], state2[2], state2[3]};
        for (int i = 0; i < 3; i++) state2[1 + i] = (inv[3 * i] * momN[0] + inv[3 * i + 1] * momN[1] + inv[3 * i + 2] * momN[2]) / det;
      }
      for (int sp = 0; sp < numActiveSpecies; sp++) state2[nvel + 2 + sp] = inputState[4 + sp];
      double tmpU[MAXEQ];
      for (int eq = 0; eq < num_equation; eq++) tmpU[eq] = state2[eq];
      for (int eq = 1; eq <= dim; eq++) tmpU[eq] = 2.0 * state2[eq] - stateIn[eq];
      mixture->modifyEnergyForPressure(tmpU, tmpU, p, true);
      rsolver->Eval(stateIn, tmpU, normal, bdrFlux, true);
      return;
    }
    if (category == TPSRHS_INLET) {  // src/inletBC.cpp:729-757
      const double p = mixture->ComputePressure(stateIn);
      double state2[MAXEQ];
      for (int eq = 0; eq < num_equation; eq++) state2[eq] = stateIn[eq];
      state2[0] = inputState[0];
      state2[1] = inputState[0] * inputState[1];
      state2[2] = inputState[0] * inputState[2];
      if (nvel == 3) state2[3] = inputState[0] * inputState[3];
      for (int sp = 0; sp < numActiveSpecies; sp++) state2[nvel + 2 + sp] = inputState[4 + sp];
      mixture->modifyEnergyForPressure(state2, state2, p, true);
      rsolver->Eval_LF(stateIn, state2, normal, bdrFlux);
      return;
    }
    if (category == TPSRHS_OUTLET) {  // src/outletBC.cpp:731-737
      double state2[MAXEQ];
      mixture->modifyEnergyForPressure(stateIn, state2, inputState[0]);
      rsolver->Eval_LF(stateIn, state2, normal, bdrFlux);
      return;
    }
    switch (type) {
      case TPSRHS_SLIP: {  // computeSlipWallFlux, src/wallBC.cpp:326-428: mirror the normal velocity in a wall frame
        double vel[MAXDIM];
        for (int d = 0; d < nvel; d++) vel[d] = stateIn[1 + d] / stateIn[0];
        const double sml = 1.0e-15;
        double unitNorm[3] = {0, 0, 0}, tangent1[3] = {0, 0, 0}, tangent2[3] = {0, 0, 0};
        double normN = 0.;
        for (int d = 0; d < dim; d++) normN += normal[d] * normal[d];
        normN = std::sqrt(std::max(normN, sml));
        for (int d = 0; d < dim; d++) unitNorm[d] = normal[d] / normN;
        int dir = 0;
        if (dim == 3) {
          if (std::abs(unitNorm[0]) >= std::abs(unitNorm[1]) && std::abs(unitNorm[0]) >= std::abs(unitNorm[2])) dir = 0;
          if (std::abs(unitNorm[1]) >= std::abs(unitNorm[0]) && std::abs(unitNorm[1]) >= std::abs(unitNorm[2])) dir = 1;
          if (std::abs(unitNorm[2]) >= std::abs(unitNorm[0]) && std::abs(unitNorm[2]) >= std::abs(unitNorm[1])) dir = 2;
        } else {
          if (std::abs(unitNorm[0]) >= std::abs(unitNorm[1])) dir = 0;
          if (std::abs(unitNorm[1]) >= std::abs(unitNorm[0])) dir = 1;
        }
        const int next_dir = (dir + 1) % dim, previous_dir = (dir + 2) % dim;
        tangent1[next_dir] = +1.;
        tangent1[previous_dir] = -1.;  // (2-D: previous_dir == dir, overwritten below, as in the reference)
        tangent1[dir] = unitNorm[previous_dir] * tangent1[previous_dir] + unitNorm[next_dir] * tangent1[next_dir];
        tangent1[dir] *= -1. / unitNorm[dir];
        double mod = 0.;
        for (int d = 0; d < dim; d++) mod += tangent1[d] * tangent1[d];
        for (int d = 0; d < dim; d++) tangent1[d] *= 1. / std::max(std::sqrt(mod), sml);
        if (dim == 3) {
          tangent2[0] = +(unitNorm[1] * tangent1[2] - unitNorm[2] * tangent1[1]);
          tangent2[1] = -(unitNorm[0] * tangent1[2] - unitNorm[2] * tangent1[0]);
          tangent2[2] = +(unitNorm[0] * tangent1[1] - unitNorm[1] * tangent1[0]);
          mod = 0.;
          for (int d = 0; d < dim; d++) mod += tangent2[d] * tangent2[d];
          for (int d = 0; d < dim; d++) tangent2[d] *= 1. / std::max(std::sqrt(mod), sml);
        }
        double M[9] = {0}, invM[9] = {0}, nVel[3] = {0, 0, 0};  // row-major M(i, d)
        for (int d = 0; d < dim; d++) {
          M[0 * dim + d] = unitNorm[d];
          M[1 * dim + d] = tangent1[d];
          if (dim == 3) M[2 * dim + d] = tangent2[d];
        }
        for (int i = 0; i < dim; i++)
          for (int d = 0; d < dim; d++) nVel[i] += M[i * dim + d] * vel[d];
        nVel[0] = -nVel[0];  // mirror normal component
        if (dim == 2) {
          const double det = M[0] * M[3] - M[1] * M[2];
          invM[0] = M[3] / det;
          invM[1] = -M[1] / det;
          invM[2] = -M[2] / det;
          invM[3] = M[0] / det;
        } else {
          const double c00 = M[4] * M[8] - M[5] * M[7], c01 = M[5] * M[6] - M[3] * M[8], c02 = M[3] * M[7] - M[4] * M[6];
          const double det = M[0] * c00 + M[1] * c01 + M[2] * c02;
          const double inv[9] = {c00, M[2] * M[7] - M[1] * M[8], M[1] * M[5] - M[2] * M[4],
                                 c01, M[0] * M[8] - M[2] * M[6], M[2] * M[3] - M[0] * M[5],
                                 c02, M[1] * M[6] - M[0] * M[7], M[0] * M[4] - M[1] * M[3]};
          for (int k = 0; k < 9; k++) invM[k] = inv[k] / det;
        }
        double gVel[3] = {0, 0, 0};
        for (int i = 0; i < dim; i++)
          for (int d = 0; d < dim; d++) gVel[i] += invM[i * dim + d] * nVel[d];
        double state2[MAXEQ];
        for (int eq = 0; eq < num_equation; eq++) state2[eq] = stateIn[eq];
        state2[1] = stateIn[0] * gVel[0];
        state2[2] = stateIn[0] * gVel[1];
        if (dim == 3) state2[3] = stateIn[0] * gVel[2];
        rsolver->Eval(stateIn, state2, normal, bdrFlux);
      } break;
      case TPSRHS_INV: {  // src/wallBC.cpp:277-320
        double vel[MAXDIM];
        for (int d = 0; d < nvel; d++) vel[d] = stateIn[1 + d] / stateIn[0];
        double norm = 0.;
        for (int d = 0; d < dim; d++) norm += normal[d] * normal[d];
        norm = std::sqrt(norm);
        double unitN[MAXDIM];
        for (int d = 0; d < dim; d++) unitN[d] = normal[d] / norm;
        double vn = 0;
        for (int d = 0; d < dim; d++) vn += vel[d] * unitN[d];
        double stateMirror[MAXEQ];
        for (int eq = 0; eq < num_equation; eq++) stateMirror[eq] = stateIn[eq];
        stateMirror[1] = stateIn[0] * (vel[0] - 2. * vn * unitN[0]);
        stateMirror[2] = stateIn[0] * (vel[1] - 2. * vn * unitN[1]);
        if (dim == 3) stateMirror[3] = stateIn[0] * (vel[2] - 2. * vn * unitN[2]);
        if ((nvel == 3) && (dim == 2)) stateMirror[3] = stateIn[0] * vel[2];
        rsolver->Eval(stateIn, stateMirror, normal, bdrFlux);  // LF = false: Roe when flow/useRoe
        double wallViscF[MAXEQ], viscF[MAXEQ * MAXDIM], viscFw[MAXEQ * MAXDIM];
        fluxClass->ComputeViscousFluxes(stateMirror, gradState, transip, delta, distance, viscFw);
        for (int eq = 0; eq < num_equation; eq++) {
          wallViscF[eq] = 0.0;
          for (int d = 0; d < dim; d++) wallViscF[eq] += viscFw[eq + d * num_equation] * normal[d];
        }
        fluxClass->ComputeViscousFluxes(stateIn, gradState, transip, delta, distance, viscF);
        for (int eq = 1; eq < num_equation; eq++) {
          bdrFlux[eq] -= 0.5 * wallViscF[eq];
          for (int d = 0; d < dim; d++) bdrFlux[eq] -= 0.5 * viscF[eq + d * num_equation] * normal[d];
        }
      } break;
      case TPSRHS_VISC_ADIAB: {  // src/wallBC.cpp:430-469
        double wallState[MAXEQ];
        mixture->computeStagnationState(stateIn, wallState);
        rsolver->Eval_LF(stateIn, wallState, normal, bdrFlux);
        double viscF[MAXEQ * MAXDIM];
        fluxClass->ComputeViscousFluxes(stateIn, gradState, transip, delta, 0.0, viscF);
        double normN = 0.;
        for (int d = 0; d < dim; d++) normN += normal[d] * normal[d];
        for (int d = 0; d < dim; d++) bcFlux.normal[d] = normal[d] * (1. / std::sqrt(normN));
        double wallViscF[MAXEQ];
        fluxClass->ComputeBdrViscousFluxes(wallState, gradState, transip, delta, 0.0, bcFlux, wallViscF);
        for (int eq = 0; eq < num_equation; eq++) wallViscF[eq] *= std::sqrt(normN);
        for (int eq = 1; eq < num_equation; eq++) {
          bdrFlux[eq] -= 0.5 * wallViscF[eq];
          for (int d = 0; d < dim; d++) bdrFlux[eq] -= 0.5 * viscF[eq + d * num_equation] * normal[d];
        }
      } break;
      case TPSRHS_VISC_ISOTH: {  // src/wallBC.cpp:471-510
        double wallState[MAXEQ];
        for (int eq = 0; eq < num_equation; eq++) wallState[eq] = stateIn[eq];
        if (useBCinGrad) {
          for (int i = 0; i < nvel; i++) wallState[i + 1] *= -1.0;
        } else {
          mixture->computeStagnantStateWithTemp(stateIn, wallTemp, wallState);
        }
        rsolver->Eval_LF(stateIn, wallState, normal, bdrFlux);
        double normN = 0.;
        for (int d = 0; d < dim; d++) normN += normal[d] * normal[d];
        for (int d = 0; d < dim; d++) bcFlux.normal[d] = normal[d] * (1. / std::sqrt(normN));
        mixture->computeStagnantStateWithTemp(stateIn, wallTemp, wallState);
        double wallViscF[MAXEQ];
        fluxClass->ComputeBdrViscousFluxes(wallState, gradState, transip, delta, 0.0, bcFlux, wallViscF);
        for (int eq = 0; eq < num_equation; eq++) wallViscF[eq] *= std::sqrt(normN);
        double viscF[MAXEQ * MAXDIM];
        fluxClass->ComputeViscousFluxes(stateIn, gradState, transip, delta, 0.0, viscF);
        for (int eq = 1; eq < num_equation; eq++) {
          bdrFlux[eq] -= 0.5 * wallViscF[eq];
          for (int d = 0; d < dim; d++) bdrFlux[eq] -= 0.5 * viscF[eq + d * num_equation] * normal[d];
        }
      } break;
      case TPSRHS_VISC_GNRL: {  // src/wallBC.cpp:512-543
        double wallState[MAXEQ];
        mixture->modifyStateFromPrimitive(stateIn, bcState, wallState);
        rsolver->Eval_LF(stateIn, wallState, normal, bdrFlux);
        double normN = 0.;
        for (int d = 0; d < dim; d++) normN += normal[d] * normal[d];
        for (int d = 0; d < dim; d++) bcFlux.normal[d] = normal[d] * (1. / std::sqrt(normN));
        if (elecCond == TPSRHS_SHTH) mixture->computeSheathBdrFlux(wallState, bcFlux);
        double wallViscF[MAXEQ];
        fluxClass->ComputeBdrViscousFluxes(wallState, gradState, transip, delta, 0.0, bcFlux, wallViscF);
        for (int eq = 0; eq < num_equation; eq++) wallViscF[eq] *= std::sqrt(normN);
        double viscF[MAXEQ * MAXDIM];
        fluxClass->ComputeViscousFluxes(stateIn, gradState, transip, delta, 0.0, viscF);
        for (int eq = 1; eq < num_equation; eq++) {
          bdrFlux[eq] -= 0.5 * wallViscF[eq];
          for (int d = 0; d < dim; d++) bdrFlux[eq] -= 0.5 * viscF[eq + d * num_equation] * normal[d];
        }
      } break;
      default:
        throw std::runtime_error("wall type outside the hot-path scope");
    }
  }
};

}  // namespace tpsoracle
#endif
