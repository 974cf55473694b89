// CPU oracle (TEST INFRASTRUCTURE, never shipped): the local-thermodynamic-equilibrium table gas of the reference,
// WorkingFluid::LTE_FLUID with one-dimensional tables (`flow/lte/table_dim = 1`: the variant the reference's device
// build runs, src/M2ulPhyS.cpp:164-255).  One species, no species equations: the state is (rho, rho u, rho E); the
// thermodynamics (e(T), R(T), c(T) and the inverse T(e)) and the transport (mu(T), kappa(T), sigma(T)) are
// LinearTables in the temperature.
//   LteMixture    src/lte_mixture.cpp:76-470
//   LteTransport  src/lte_transport_properties.cpp:60-140
//   LinearTable::eval_x  src/table.cpp:103-113
// The two-dimensional (T, rho) tables of the reference's CPU build interpolate with GSL (a third-party dependency
// absent here) and are not restated.
#ifndef TPSORACLE_LTE_HPP_
#define TPSORACLE_LTE_HPP_

#include "plasma.hpp"

namespace tpsoracle {

// LinearTable::eval_x (src/table.cpp:103-113)
inline double table_eval_x(const LinearTable &t, double xEval) {
  const int index = t.findInterval(xEval);
  const double xt = t.xLog ? std::log(xEval) : xEval;
  const double xt_xt = t.xLog ? 1. / xEval : 1.0;
  double ft_x = t.b[index] * xt_xt;
  if (t.fLog) {
    const double ft = t.a[index] + t.b[index] * xt;
    ft_x *= std::exp(ft);
  }
  return ft_x;
}

class LteMixture : public GasMixture {
 public:
  LinearTable energy_table_, R_table_, c_table_, T_table_;
  LteMixture(const tpsrhs_lte &in, int dim_, int nvel_) {
    dim = dim_;
    nvel = nvel_;
    numSpecies = 1;  // src/lte_mixture.cpp:84-90
    ambipolar = false;
    twoTemperature = false;
    numActiveSpecies = 0;
    num_equation = nvel + 2;
    iTh = nvel + 1;
    energy_table_.init(in.energy_table);
    R_table_.init(in.gas_constant_table);
    c_table_.init(in.sound_speed_table);
    // "Construct e -> T table from T -> e" (src/M2ulPhyS.cpp:193-200): abscissae and values of the energy table swapped
    tpsrhs_table rev = in.energy_table;
    rev.x_data = in.energy_table.f_data;
    rev.f_data = in.energy_table.x_data;
    T_table_.init(rev);
  }
  // src/lte_mixture.cpp:161-218
  bool ComputeTemperatureInternal(const double *state, double &T) const {
    const double rho = state[0];
    double den_vel2 = 0;
    for (int d = 0; d < nvel; d++) den_vel2 += state[d + 1] * state[d + 1];
    den_vel2 /= rho;
    const double energy = (state[1 + nvel] - 0.5 * den_vel2) / rho;
    T = T_table_.eval(energy);
    double res = energy - energy_table_.eval(T);
    const double res0 = std::abs(res);
    const double atol = 1e-18, rtol = 1e-12, dT_atol = 1e-12, dT_rtol = 1e-8;
    bool converged = ((std::abs(res) < atol) || (std::abs(res) / std::abs(res0) < rtol));
    const int niter_max = 20;
    int niter = 0;
    while (!converged && (niter < niter_max)) {
      const double dedT = table_eval_x(energy_table_, T);
      const double dT = res / dedT;
      T += dT;
      if (!(T > 0)) throw std::runtime_error("LteMixture: assert(T > 0)");
      res = energy - energy_table_.eval(T);
      converged = ((std::abs(res) < atol) || (std::abs(res) / res0 < rtol) || (std::abs(dT) < dT_atol) ||
                   (std::abs(dT) / T < dT_rtol));
      niter++;
    }
    return converged;
  }
  double ComputeTemperature(const double *state) const override {
    double T;
    if (!ComputeTemperatureInternal(state, T)) throw std::runtime_error("LteMixture: temperature did not converge");
    return T;
  }
  double ComputePressure(const double *state, double *electronPressure = nullptr) const override {  // :119-131
    if (electronPressure != nullptr) *electronPressure = 0.0;
    const double rho = state[0];
    const double T = ComputeTemperature(state);
    const double R = R_table_.eval(T);
    return rho * R * T;
  }
  double ComputePressureFromPrimitives(const double *Up) const override {  // :138-147
    const double rho = Up[0];
    const double T = Up[1 + nvel];
    const double R = R_table_.eval(T);
    return rho * R * T;
  }
  // src/lte_mixture.cpp:236-296
  double ComputeTemperatureFromDensityPressure(double rho, double p) const {
    double T = p / (rho * 208.);
    double R = R_table_.eval(T);
    double res = p - rho * R * T;
    const double res0 = std::abs(res);
    const double atol = 1e-18, rtol = 1e-12, dT_atol = 1e-12, dT_rtol = 1e-8;
    bool converged = ((std::abs(res) < atol) || (std::abs(res) / std::abs(res0) < rtol));
    const int niter_max = 20;
    int niter = 0;
    while (!converged && (niter < niter_max)) {
      const double R_T = table_eval_x(R_table_, T);
      const double dpdT = rho * R + rho * R_T * T;
      const double dT = res / dpdT;
      T += dT;
      if (!(T > 0)) throw std::runtime_error("LteMixture: assert(T > 0)");
      R = R_table_.eval(T);
      res = p - rho * R * T;
      converged = ((std::abs(res) < atol) || (std::abs(res) / res0 < rtol) || (std::abs(dT) < dT_atol) ||
                   (std::abs(dT) / T < dT_rtol));
      niter++;
    }
    return T;  // the reference only prints a warning when the iteration has not converged
  }
  void computeSpeciesEnthalpies(const double *, double *h) const override {
    for (int sp = 0; sp < numSpecies; sp++) h[sp] = 0.0;
  }
  void GetPrimitivesFromConservatives(const double *conserv, double *primit) const override {  // :311-321
    const double T = ComputeTemperature(conserv);
    for (int i = 0; i < num_equation; i++) primit[i] = conserv[i];
    for (int d = 0; d < nvel; d++) primit[1 + d] /= conserv[0];
    primit[nvel + 1] = T;
  }
  void GetConservativesFromPrimitives(const double *primit, double *conserv) const override {  // :329-349
    for (int i = 0; i < num_equation; i++) conserv[i] = primit[i];
    double v2 = 0.;
    for (int d = 0; d < nvel; d++) {
      v2 += primit[1 + d] * primit[1 + d];
      conserv[1 + d] *= primit[0];
    }
    const double T = primit[1 + nvel];
    const double energy = energy_table_.eval(T);
    conserv[1 + nvel] = primit[0] * (energy + 0.5 * v2);
  }
  double ComputeSpeedOfSound(const double *Uin, bool primitive) const {  // :357-372
    double T;
    if (primitive) {
      T = Uin[1 + nvel];
    } else {
      T = ComputeTemperature(Uin);
    }
    return c_table_.eval(T);
  }
  double ComputeMaxCharSpeed(const double *state) const override {  // :378-391
    const double den = state[0];
    double den_vel2 = 0;
    for (int d = 0; d < nvel; d++) den_vel2 += state[d + 1] * state[d + 1];
    den_vel2 /= den;
    const double sound = ComputeSpeedOfSound(state, false);
    const double vel = std::sqrt(den_vel2 / den);
    return vel + sound;
  }
  // LteMixture does not override computeStagnationState: GasMixture's (src/equation_of_state.cpp:100-113) drops the
  // momentum and the bulk kinetic energy
  void computeStagnationState(const double *stateIn, double *out) const override {
    for (int eq = 0; eq < num_equation; eq++) out[eq] = stateIn[eq];
    for (int d = 0; d < nvel; d++) out[1 + d] = 0.;
    double kineticEnergy = 0.0;
    for (int d = 0; d < nvel; d++) kineticEnergy += 0.5 * stateIn[1 + d] * stateIn[1 + d] / stateIn[0];
    out[iTh] = stateIn[iTh] - kineticEnergy;
  }
  void computeStagnantStateWithTemp(const double *stateIn, double Temp, double *stateOut) const override {  // :424-441
    for (int i = 0; i < num_equation; i++) stateOut[i] = stateIn[i];
    for (int d = 0; d < nvel; d++) stateOut[1 + d] = 0.;
    const double rho = stateIn[0];
    const double energy = energy_table_.eval(Temp);
    stateOut[1 + nvel] = rho * energy;
  }
  void modifyEnergyForPressure(const double *stateIn, double *stateOut, double p, bool) const override {  // :448-467
    double tmp[MAXEQ];
    for (int eq = 0; eq < num_equation; eq++) tmp[eq] = stateIn[eq];
    const double rho = tmp[0];
    double ke = 0.;
    for (int d = 0; d < nvel; d++) ke += tmp[1 + d] * tmp[1 + d];
    ke *= 0.5 / rho;
    const double T = ComputeTemperatureFromDensityPressure(rho, p);
    const double energy = energy_table_.eval(T);
    for (int eq = 0; eq < num_equation; eq++) stateOut[eq] = tmp[eq];
    stateOut[1 + nvel] = rho * energy + ke;
  }
};

// LteTransport: src/lte_transport_properties.cpp:84-140
class LteTransport : public TransportProperties {
 public:
  LinearTable mu_table_, kappa_table_, sigma_table_;
  LteTransport(GasMixture *m, const tpsrhs_lte &in) : TransportProperties(m) {
    mu_table_.init(in.viscosity_table);
    kappa_table_.init(in.conductivity_table);
    sigma_table_.init(in.electric_conductivity_table);
  }
  void ComputeFluxTransportProperties(const double *state, const double *, const double *, double, double,
                                      double *transportBuffer, double *diffusionVelocity) override {
    const double T = mixture->ComputeTemperature(state);
    transportBuffer[VISCOSITY] = mu_table_.eval(T);
    transportBuffer[HEAVY_THERMAL_CONDUCTIVITY] = kappa_table_.eval(T);
    transportBuffer[BULK_VISCOSITY] = 0.0;
    transportBuffer[ELECTRON_THERMAL_CONDUCTIVITY] = 0.0;
    for (int v = 0; v < nvel; v++)
      for (int sp = 0; sp < numSpecies; sp++) diffusionVelocity[sp + v * numSpecies] = 0.0;
  }
  void ComputeSourceTransportProperties(const double *, const double *Up, const double *, const double *, double,
                                        double *globalTransport, double *, double *, double *) override {
    const double T = Up[1 + nvel];
    double sigma = sigma_table_.eval(T);
    if (sigma < 1.0) sigma = 1.0;
    globalTransport[ELECTRIC_CONDUCTIVITY] = sigma;
  }
  void GetViscosities(const double *, const double *primitive, double *visc) override {
    const double T = primitive[1 + nvel];
    visc[0] = mu_table_.eval(T);
    visc[1] = 0.;
  }
};

}  // namespace tpsoracle
#endif
