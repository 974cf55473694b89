// TEST INFRASTRUCTURE -- CPU oracle, never shipped, never on the product path.
//
// Multi-species plasma closures of the oracle, restated function by function from the reference:
//   PerfectMixture        src/equation_of_state.cpp:478-1942
//   ConstantTransport     src/transport_properties.cpp:303-449
//   GasMinimalTransport   src/gas_transport.cpp:43-830 (argon ternary branch)
//   collision integrals   src/collision_integrals.cpp:53-201 (charged, argon)
//   Chemistry / Reaction  src/chemistry.cpp:40-299, src/reaction.cpp:38-83, src/table.cpp:39-110
//   NetEmission           src/radiation.hpp:54-69
//   SourceTerm            src/source_term.cpp:62-256
//   AxisymmetricSource    src/forcing_terms.cpp:255-382
#ifndef TPS_ORACLE_PLASMA_HPP_
#define TPS_ORACLE_PLASMA_HPP_

#include <algorithm>

#include "physics.hpp"

namespace tpsoracle {

// ------------------------------------------------------------------------------------------
// collision integrals (fits; coefficients are the published fit data the reference carries)
// ------------------------------------------------------------------------------------------
namespace collision {
namespace charged {  // src/collision_integrals.cpp:53-115; argument: Debye-nondimensional temperature
inline double fit(double c0, double c1, double c2, double c3, double Tp) {
  return c0 * std::pow(std::log(1.0 + c1 * std::pow(Tp, c2)), c3) / Tp / Tp;
}
inline double att11(double Tp) { return fit(0.2150, 5.2194, 1.0472, 1.2435, Tp); }
inline double att12(double Tp) { return fit(0.0991, 7.4684, 1.0155, 1.1536, Tp); }
inline double att13(double Tp) { return fit(0.0616, 7.8271, 0.9452, 1.1105, Tp); }
inline double att14(double Tp) { return fit(0.0308, 13.9567, 0.9511, 1.1803, Tp); }
inline double att15(double Tp) { return fit(0.0232, 13.7888, 0.9148, 1.1532, Tp); }
inline double att22(double Tp) { return fit(0.2423, 4.6796, 1.3290, 1.1279, Tp); }
inline double att23(double Tp) { return fit(0.1221, 8.7542, 1.3875, 1.1110, Tp); }
inline double att24(double Tp) { return fit(0.0619, 18.2538, 1.4341, 1.1618, Tp); }
inline double rep11(double Tp) { return fit(0.3904, 0.9100, 1.1025, 1.0544, Tp); }
inline double rep12(double Tp) { return fit(0.1547, 1.6597, 1.1725, 0.9792, Tp); }
inline double rep13(double Tp) { return fit(0.0814, 2.5815, 1.1948, 0.9570, Tp); }
inline double rep14(double Tp) { return fit(0.0683, 1.9774, 1.2033, 0.8264, Tp); }
inline double rep15(double Tp) { return fit(0.0346, 4.5177, 1.2132, 0.9294, Tp); }
inline double rep22(double Tp) { return fit(0.4128, 1.2436, 1.1830, 1.0123, Tp); }
inline double rep23(double Tp) { return fit(0.2203, 1.8832, 1.2059, 0.9851, Tp); }
inline double rep24(double Tp) { return fit(0.1323, 2.7248, 1.2129, 0.9847, Tp); }
}  // namespace charged
namespace argon {  // src/collision_integrals.cpp:124-201; T in K, result in m^2
inline double ArAr11(double T) { return 2.2910e-18 * std::pow(T, -0.3032); }
inline double ArAr22(double T) { return 1.7e-18 * std::pow(T, -0.25); }
inline double ArAr1P11(double T) { return 4.574321e-18 * std::pow(T, -0.1805); }
inline double logT_fit(double logT, const double c[9]) {
  double pw[9];
  pw[0] = 1. / logT;
  pw[1] = 1.;
  for (int k = 0; k < 7; k++) pw[k + 2] = pw[k + 1] * logT;
  double fit = 0.0;
  for (int k = 0; k < 9; k++) fit += c[k] * pw[k];
  return fit;
}
inline double eAr1r(int r, double T) {
  static const double C[5][9] = {
      {6.36254140e-18, 1.84835040e-18, -5.87727093e-18, 3.20023027e-18, -8.50509054e-19, 1.28163820e-19,
       -1.11712910e-20, 5.25649382e-22, -1.03296658e-23},
      {1.91338172e-17, 5.45418129e-18, -1.78361685e-17, 9.75657946e-18, -2.61115722e-18, 3.98310268e-19,
       -3.53503678e-20, 1.70375066e-21, -3.45211955e-23},
      {3.04685398e-17, 8.39750994e-18, -2.88132528e-17, 1.60147037e-17, -4.34837891e-18, 6.73136845e-19,
       -6.06704580e-20, 2.97216168e-21, -6.12760944e-23},
      {3.90777949e-17, 1.04696956e-17, -3.73774204e-17, 2.10610498e-17, -5.79029566e-18, 9.07573157e-19,
       -8.28466766e-20, 4.11188110e-21, -8.59225098e-23},
      {4.41333290e-17, 1.15696010e-17, -4.25651305e-17, 2.42442440e-17, -6.73359258e-18, 1.06641697e-18,
       -9.83933863e-20, 4.93775812e-21, -1.04362372e-22}};
  return logT_fit(std::log(T), C[r - 1]);
}
inline double eAr11(double T) { return eAr1r(1, T); }
}  // namespace argon
}  // namespace collision

// ------------------------------------------------------------------------------------------
// PerfectMixture
// ------------------------------------------------------------------------------------------
class PerfectMixture : public GasMixture {
 public:
  double gasParams[MAXSP * TPSRHS_NUM_GASPARAMS];
  double molarCV_[MAXSP], molarCP_[MAXSP];
  int iBackground, iElectron, iTe = -1;

  PerfectMixture(const tpsrhs_perfect_mixture &in, int dim_, int nvel_) {
    dim = dim_;
    nvel = nvel_;
    numSpecies = in.num_species;
    ambipolar = false;
    twoTemperature = false;
    if (in.is_electron_included) {
      ambipolar = in.ambipolar != 0;
      twoTemperature = in.two_temperature != 0;
    }
    for (int sp = 0; sp < numSpecies; sp++)
      for (int p = 0; p < TPSRHS_NUM_GASPARAMS; p++)
        gasParams[sp + p * numSpecies] = in.gas_params[sp + p * numSpecies];
    numActiveSpecies = ambipolar ? (numSpecies - 2) : (numSpecies - 1);
    num_equation = twoTemperature ? (nvel + 3 + numActiveSpecies) : (nvel + 2 + numActiveSpecies);
    iBackground = numSpecies - 1;
    iElectron = numSpecies - 2;
    if (twoTemperature) iTe = num_equation - 1;
    iTh = nvel + 1;
    if (!in.is_electron_included) throw std::runtime_error("PerfectMixture requires the electron species");
    if (GetGasParams(iBackground, TPSRHS_SPECIES_CHARGES) != 0.0 ||
        GetGasParams(iElectron, TPSRHS_FORMATION_ENERGY) != 0.0 ||
        GetGasParams(iBackground, TPSRHS_FORMATION_ENERGY) != 0.0)
      throw std::runtime_error("PerfectMixture: background must be neutral; background/electron formation energy 0");
    for (int sp = 0; sp < numSpecies; sp++) {
      molarCV_[sp] = in.molar_cv[sp] * UNIVERSALGASCONSTANT;
      molarCP_[sp] = molarCV_[sp] + UNIVERSALGASCONSTANT;
    }
  }
  double GetGasParams(int sp, int param) const override { return gasParams[sp + param * numSpecies]; }

  double computeHeaviesHeatCapacity(const double *n_sp, double nB) const {  // :576-584
    double c = 0.0;
    for (int sp = 0; sp < numActiveSpecies; sp++) {
      if (sp == iElectron) continue;
      c += n_sp[sp] * molarCV_[sp];
    }
    c += nB * molarCV_[iBackground];
    return c;
  }
  double computeAmbipolarElectronNumberDensity(const double *n_sp) const {  // :607-618
    double n_e = 0.0;
    for (int sp = 0; sp < numActiveSpecies; sp++) n_e += GetGasParams(sp, TPSRHS_SPECIES_CHARGES) * n_sp[sp];
    if (n_e < 0.0) n_e = 0.0;
    return n_e;
  }
  double computeBackgroundMassDensity(double rho, const double *n_sp, double &n_e, bool isElectronComputed) const {
    if ((!isElectronComputed) && ambipolar) n_e = computeAmbipolarElectronNumberDensity(n_sp);  // :620-650
    double rhoB = rho;
    for (int sp = 0; sp < numActiveSpecies; sp++) rhoB -= GetGasParams(sp, TPSRHS_SPECIES_MW) * n_sp[sp];
    if (ambipolar) rhoB -= n_e * GetGasParams(iElectron, TPSRHS_SPECIES_MW);
    if (rhoB < 0.)  // the reference prints and exits here (src/equation_of_state.cpp:641-647)
      throw std::runtime_error("Negative background density: rho = " + std::to_string(rho) +
                               ", n_sp[0] = " + std::to_string(n_sp[0]));
    return rhoB;
  }
  void computeNumberDensities(const double *state, double *n_sp) const {  // :947-961
    for (int sp = 0; sp < numSpecies; sp++) n_sp[sp] = 0.0;
    double n_e = 0.0;
    for (int sp = 0; sp < numActiveSpecies; sp++) n_sp[sp] = state[nvel + 2 + sp] / GetGasParams(sp, TPSRHS_SPECIES_MW);
    if (ambipolar) {
      n_e = computeAmbipolarElectronNumberDensity(n_sp);
      n_sp[iElectron] = n_e;
    }
    const double rhoB = computeBackgroundMassDensity(state[0], n_sp, n_e, true);
    n_sp[iBackground] = rhoB / GetGasParams(iBackground, TPSRHS_SPECIES_MW);
  }
  void computeSpeciesPrimitives(const double *state, double *X_sp, double *Y_sp, double *n_sp) const {  // :882-927
    for (int sp = 0; sp < numSpecies; sp++) X_sp[sp] = Y_sp[sp] = n_sp[sp] = 0.0;
    double n_e = 0.0, n = 0.0;
    for (int sp = 0; sp < numActiveSpecies; sp++) {
      n_sp[sp] = state[nvel + 2 + sp] / GetGasParams(sp, TPSRHS_SPECIES_MW);
      n += n_sp[sp];
      if (ambipolar) n_e += GetGasParams(sp, TPSRHS_SPECIES_CHARGES) * n_sp[sp];
    }
    if (ambipolar) {
      n_sp[iElectron] = n_e;
      n += n_e;
    }
    double Yb = 1.;
    for (int sp = 0; sp < numActiveSpecies; sp++) {
      Y_sp[sp] = state[nvel + 2 + sp] / state[0];
      Yb -= Y_sp[sp];
    }
    if (ambipolar) {
      Y_sp[iElectron] = n_e * GetGasParams(iElectron, TPSRHS_SPECIES_MW) / state[0];
      Yb -= Y_sp[iElectron];
    }
    if (Yb < 0.0) throw std::runtime_error("negative background mass fraction");
    Y_sp[iBackground] = Yb;
    n_sp[iBackground] = Y_sp[iBackground] * state[0] / GetGasParams(iBackground, TPSRHS_SPECIES_MW);
    n += n_sp[iBackground];
    for (int sp = 0; sp < numSpecies; sp++) X_sp[sp] = n_sp[sp] / n;
  }
  void computeTemperaturesBase(const double *state, const double *n_sp, double n_e, double n_B, double &T_h,
                               double &T_e) const {  // :1141-1172
    double totalHeatCapacity = computeHeaviesHeatCapacity(n_sp, n_B);
    if (!twoTemperature) totalHeatCapacity += n_e * molarCV_[iElectron];
    double totalEnergy = state[iTh];
    for (int sp = 0; sp < numSpecies - 2; sp++) totalEnergy -= n_sp[sp] * GetGasParams(sp, TPSRHS_FORMATION_ENERGY);
    T_h = 0.0;
    for (int d = 0; d < nvel; d++) T_h -= state[d + 1] * state[d + 1];
    T_h *= 0.5 / state[0];
    T_h += totalEnergy;
    if (twoTemperature) T_h -= state[iTe];
    T_h /= totalHeatCapacity;
    T_e = twoTemperature ? state[iTe] / n_e / molarCV_[iElectron] : T_h;
  }
  double computePressureBase(const double *n_sp, double n_e, double n_B, double T_h, double T_e) const {  // :1044-1062
    double n_h = 0.0;
    for (int sp = 0; sp < numActiveSpecies; sp++) {
      if (sp == iElectron) continue;
      n_h += n_sp[sp];
    }
    n_h += n_B;
    double p = n_h * T_h;
    p += twoTemperature ? n_e * T_e : n_e * T_h;
    return p * UNIVERSALGASCONSTANT;
  }
  double ComputePressure(const double *state, double *electronPressure = nullptr) const override {  // :1029-1042
    double n_sp[MAXSP];
    computeNumberDensities(state, n_sp);
    double T_h, T_e;
    computeTemperaturesBase(state, n_sp, n_sp[iElectron], n_sp[iBackground], T_h, T_e);
    if (electronPressure != nullptr) *electronPressure = n_sp[iElectron] * UNIVERSALGASCONSTANT * T_e;
    return computePressureBase(n_sp, n_sp[iElectron], n_sp[iBackground], T_h, T_e);
  }
  double ComputePressureFromPrimitives(const double *Up) const {  // :988-1010
    double n_e = ambipolar ? computeAmbipolarElectronNumberDensity(&Up[nvel + 2]) : Up[nvel + 2 + iElectron];
    const double rhoB = computeBackgroundMassDensity(Up[0], &Up[nvel + 2], n_e, true);
    const double nB = rhoB / GetGasParams(iBackground, TPSRHS_SPECIES_MW);
    const double T_h = Up[iTh];
    const double T_e = twoTemperature ? Up[iTe] : Up[iTh];
    return computePressureBase(&Up[nvel + 2], n_e, nB, T_h, T_e);
  }
  double ComputeTemperature(const double *state) const override {
    double n_sp[MAXSP], T_h, T_e;
    computeNumberDensities(state, n_sp);
    computeTemperaturesBase(state, n_sp, n_sp[iElectron], n_sp[iBackground], T_h, T_e);
    return T_h;
  }
  void GetPrimitivesFromConservatives(const double *conserv, double *primit) const override {  // :679-700
    double n_sp[MAXSP];
    computeNumberDensities(conserv, n_sp);
    for (int sp = 0; sp < numActiveSpecies; sp++) primit[nvel + 2 + sp] = n_sp[sp];
    primit[0] = conserv[0];
    for (int d = 0; d < nvel; d++) primit[d + 1] = conserv[d + 1] / conserv[0];
    double T_h, T_e;
    computeTemperaturesBase(conserv, n_sp, n_sp[iElectron], n_sp[iBackground], T_h, T_e);
    primit[iTh] = T_h;
    if (twoTemperature) primit[iTe] = T_e;
  }
  void GetConservativesFromPrimitives(const double *primit, double *conserv) const override {  // :744-783
    conserv[0] = primit[0];
    for (int d = 0; d < nvel; d++) conserv[d + 1] = primit[d + 1] * primit[0];
    for (int sp = 0; sp < numActiveSpecies; sp++)
      conserv[nvel + 2 + sp] = primit[nvel + 2 + sp] * GetGasParams(sp, TPSRHS_SPECIES_MW);
    double n_e = ambipolar ? computeAmbipolarElectronNumberDensity(&primit[nvel + 2]) : primit[nvel + 2 + iElectron];
    const double rhoB = computeBackgroundMassDensity(primit[0], &primit[nvel + 2], n_e, true);
    const double nB = rhoB / GetGasParams(iBackground, TPSRHS_SPECIES_MW);
    if (twoTemperature) conserv[iTe] = n_e * molarCV_[iElectron] * primit[iTe];
    double totalHeatCapacity = computeHeaviesHeatCapacity(&primit[nvel + 2], nB);
    if (!twoTemperature) totalHeatCapacity += n_e * molarCV_[iElectron];
    double totalEnergy = 0.0;
    for (int d = 0; d < nvel; d++) totalEnergy += primit[d + 1] * primit[d + 1];
    totalEnergy *= 0.5 * primit[0];
    totalEnergy += totalHeatCapacity * primit[iTh];
    if (twoTemperature) totalEnergy += conserv[iTe];
    for (int sp = 0; sp < numSpecies - 2; sp++)
      totalEnergy += primit[nvel + 2 + sp] * GetGasParams(sp, TPSRHS_FORMATION_ENERGY);
    conserv[iTh] = totalEnergy;
  }
  void computeSpeciesEnthalpies(const double *state, double *h) const override {  // :1192-1207
    double n_sp[MAXSP], T_h, T_e;
    computeNumberDensities(state, n_sp);
    computeTemperaturesBase(state, n_sp, n_sp[iElectron], n_sp[iBackground], T_h, T_e);
    for (int sp = 0; sp < numSpecies; sp++) {
      const double temp = (sp == iElectron) ? T_e : T_h;
      h[sp] = n_sp[sp] * (molarCP_[sp] * temp + GetGasParams(sp, TPSRHS_FORMATION_ENERGY));
    }
  }
  double ComputeSpeedOfSound(const double *U) const {  // :1405-1434 (conserved branch), :1311-1340
    double n_sp[MAXSP], T_h, T_e;
    computeNumberDensities(U, n_sp);
    computeTemperaturesBase(U, n_sp, n_sp[iElectron], n_sp[iBackground], T_h, T_e);
    const double p = computePressureBase(n_sp, n_sp[iElectron], n_sp[iBackground], T_h, T_e);
    double cv = 0.0, n_h = n_sp[iBackground];
    for (int sp = 0; sp < numActiveSpecies; sp++) {
      if (sp == iElectron) continue;
      cv += n_sp[sp] * molarCV_[sp];
      n_h += n_sp[sp];
    }
    cv += n_sp[iBackground] * molarCV_[iBackground];
    const double gamma = 1.0 + n_h * UNIVERSALGASCONSTANT / cv;
    return std::sqrt(gamma * p / U[0]);
  }
  double ComputeMaxCharSpeed(const double *state) const override {  // :1359-1373
    const double den = state[0];
    double den_vel2 = 0;
    for (int d = 0; d < nvel; d++) den_vel2 += state[d + 1] * state[d + 1];
    den_vel2 /= den;
    const double sound = ComputeSpeedOfSound(state);
    const double vel = std::sqrt(den_vel2 / den);
    return vel + sound;
  }
  void ComputeMoleFractionGradient(const double *numberDensities, const double *gradUp, double *gradX) const {
    for (int d = 0; d < dim; d++)  // :1534-1592
      for (int sp = 0; sp < numSpecies; sp++) gradX[sp + d * numSpecies] = 0.0;
    double totalN = 0.0;
    for (int sp = 0; sp < numSpecies; sp++) totalN += numberDensities[sp];
    double neGrad[MAXDIM] = {0, 0, 0};
    if (ambipolar)
      for (int sp = 0; sp < numActiveSpecies; sp++)
        for (int d = 0; d < dim; d++)
          neGrad[d] += gradUp[(nvel + 2 + sp) + d * num_equation] * GetGasParams(sp, TPSRHS_SPECIES_CHARGES);
    double nBGrad[MAXDIM];
    for (int d = 0; d < dim; d++) nBGrad[d] = gradUp[0 + d * num_equation];
    for (int sp = 0; sp < numActiveSpecies; sp++)
      for (int d = 0; d < dim; d++)
        nBGrad[d] -= gradUp[(nvel + 2 + sp) + d * num_equation] * GetGasParams(sp, TPSRHS_SPECIES_MW);
    if (ambipolar)
      for (int d = 0; d < dim; d++) nBGrad[d] += -GetGasParams(iElectron, TPSRHS_SPECIES_MW) * neGrad[d];
    for (int d = 0; d < dim; d++) nBGrad[d] /= GetGasParams(iBackground, TPSRHS_SPECIES_MW);
    double totalNGrad[MAXDIM] = {0, 0, 0};
    for (int sp = 0; sp < numActiveSpecies; sp++)
      for (int d = 0; d < dim; d++) totalNGrad[d] += gradUp[(nvel + 2 + sp) + d * num_equation];
    if (ambipolar)
      for (int d = 0; d < dim; d++) totalNGrad[d] += neGrad[d];
    for (int d = 0; d < dim; d++) totalNGrad[d] += nBGrad[d];
    for (int sp = 0; sp < numActiveSpecies; sp++)
      for (int d = 0; d < dim; d++)
        gradX[sp + d * numSpecies] = gradUp[(nvel + 2 + sp) + d * num_equation] / totalN -
                                     numberDensities[sp] / totalN / totalN * totalNGrad[d];
    if (ambipolar) {
      const int sp = iElectron;
      for (int d = 0; d < dim; d++)
        gradX[sp + d * numSpecies] = neGrad[d] / totalN - numberDensities[sp] / totalN / totalN * totalNGrad[d];
    }
    const int sp = iBackground;
    for (int d = 0; d < dim; d++)
      gradX[sp + d * numSpecies] = nBGrad[d] / totalN - numberDensities[sp] / totalN / totalN * totalNGrad[d];
  }
  void computeStagnationState(const double *stateIn, double *out) const override {  // GasMixture, :100-115
    for (int eq = 0; eq < num_equation; eq++) out[eq] = stateIn[eq];
    for (int d = 0; d < nvel; d++) out[1 + d] = 0.;
    double ke = 0.0;
    for (int d = 0; d < nvel; d++) ke += 0.5 * stateIn[1 + d] * stateIn[1 + d] / stateIn[0];
    out[iTh] = stateIn[iTh] - ke;
  }
  void computeStagnantStateWithTemp(const double *stateIn, double Temp, double *out) const override {  // :1596-1620
    for (int eq = 0; eq < num_equation; eq++) out[eq] = stateIn[eq];
    for (int d = 0; d < nvel; d++) out[1 + d] = 0.;
    double n_sp[MAXSP];
    computeNumberDensities(stateIn, n_sp);
    const double Ch = computeHeaviesHeatCapacity(n_sp, n_sp[iBackground]);
    const double Ue = n_sp[iElectron] * molarCV_[iElectron] * Temp;
    out[iTh] = Ch * Temp + Ue;
    if (twoTemperature) out[iTe] = Ue;
    for (int sp = 0; sp < numSpecies - 2; sp++) out[iTh] += n_sp[sp] * GetGasParams(sp, TPSRHS_FORMATION_ENERGY);
  }
  void modifyEnergyForPressure(const double *stateIn, double *stateOut, double p,
                               bool modifyElectronEnergy = false) const override {  // :1698-1742
    double in[MAXEQ];
    for (int eq = 0; eq < num_equation; eq++) in[eq] = stateIn[eq];
    for (int eq = 0; eq < num_equation; eq++) stateOut[eq] = in[eq];
    double n_sp[MAXSP];
    computeNumberDensities(in, n_sp);
    double Th = 0., pe = 0.0;
    if (twoTemperature && (!modifyElectronEnergy)) {
      const double Xeps = 1.0e-30;
      const double Te = in[iTe] / (n_sp[iElectron] + Xeps) / molarCV_[iElectron];
      pe = n_sp[iElectron] * UNIVERSALGASCONSTANT * Te;
    }
    for (int sp = 0; sp < numSpecies; sp++) {
      if (twoTemperature && (!modifyElectronEnergy) && (sp == iElectron)) continue;
      Th += n_sp[sp];
    }
    Th = (p - pe) / (Th * UNIVERSALGASCONSTANT);
    const double totalHeatCapacity = computeHeaviesHeatCapacity(n_sp, n_sp[iBackground]);
    double rE = totalHeatCapacity * Th;
    double electronEnergy = 0.0;
    if (twoTemperature) {
      electronEnergy = modifyElectronEnergy ? n_sp[iElectron] * molarCV_[iElectron] * Th : in[iTe];
      stateOut[iTe] = electronEnergy;
    } else {
      electronEnergy = n_sp[iElectron] * molarCV_[iElectron] * Th;
    }
    rE += electronEnergy;
    for (int d = 0; d < nvel; d++) rE += 0.5 * in[d + 1] * in[d + 1] / in[0];
    for (int sp = 0; sp < numSpecies - 2; sp++) rE += n_sp[sp] * GetGasParams(sp, TPSRHS_FORMATION_ENERGY);
    stateOut[iTh] = rE;
  }
  // Bohm fluxes of a sheath edge on a fully catalytic wall, src/equation_of_state.cpp:1909-1942
  void computeSheathBdrFlux(const double *state, BoundaryViscousFluxData &bcFlux) const override {
    double n_sp[MAXSP], T_h, T_e;
    computeNumberDensities(state, n_sp);
    computeTemperaturesBase(state, n_sp, n_sp[iElectron], n_sp[iBackground], T_h, T_e);
    for (int sp = 0; sp < numSpecies; sp++) bcFlux.primFlux[sp] = 0.0;
    for (int sp = 0; sp < numSpecies; sp++) {
      const double Zsp = GetGasParams(sp, TPSRHS_SPECIES_CHARGES);
      if (Zsp > 0.0) {
        const double msp = GetGasParams(sp, TPSRHS_SPECIES_MW);
        const double VB = std::sqrt((T_h + Zsp * T_e) * UNIVERSALGASCONSTANT / msp);
        bcFlux.primFlux[sp] = VB;
        bcFlux.primFlux[iElectron] += Zsp * n_sp[sp] * VB;
        bcFlux.primFlux[iBackground] -= msp * n_sp[sp] * VB;
      }
    }
    bcFlux.primFlux[iElectron] /= n_sp[iElectron];
    bcFlux.primFlux[iBackground] -= GetGasParams(iElectron, TPSRHS_SPECIES_MW) * n_sp[iElectron] * bcFlux.primFlux[iElectron];
    bcFlux.primFlux[iBackground] /= GetGasParams(iBackground, TPSRHS_SPECIES_MW) * n_sp[iBackground];
    if (twoTemperature) {
      const double vTe = std::sqrt(8.0 * UNIVERSALGASCONSTANT * T_e / PI_ / GetGasParams(iElectron, TPSRHS_SPECIES_MW));
      const double gamma = -std::log(4.0 / vTe * bcFlux.primFlux[iElectron]);
      bcFlux.primFlux[numSpecies + nvel + 1] =
          bcFlux.primFlux[iElectron] * (gamma + 2.0) * n_sp[iElectron] * UNIVERSALGASCONSTANT * T_e;
    }
  }
  void computeElectronPressureGrad(double n_e, double T_e, const double *gradUp, double *gradPe) const {  // :1847-1870
    double neGrad[MAXDIM] = {0, 0, 0};
    if (ambipolar) {
      for (int sp = 0; sp < numActiveSpecies; sp++)
        for (int d = 0; d < dim; d++)
          neGrad[d] += gradUp[(nvel + 2 + sp) + d * num_equation] * GetGasParams(sp, TPSRHS_SPECIES_CHARGES);
    } else {
      for (int d = 0; d < dim; d++) neGrad[d] = gradUp[(nvel + numSpecies) + d * num_equation];
    }
    for (int d = 0; d < dim; d++)
      gradPe[d] = (neGrad[d] * T_e + n_e * gradUp[iTe + d * num_equation]) * UNIVERSALGASCONSTANT;
  }
};

// ------------------------------------------------------------------------------------------
// ConstantTransport (src/transport_properties.cpp:303-449)
// ------------------------------------------------------------------------------------------
class ConstantTransport : public TransportProperties {
 public:
  PerfectMixture *pm;
  tpsrhs_constant_transport in;
  const double qeOverkB_ = ELECTRONCHARGE / BOLTZMANNCONSTANT;
  ConstantTransport(PerfectMixture *m, const tpsrhs_constant_transport &inputs) : TransportProperties(m), pm(m), in(inputs) {
    if (m->twoTemperature && in.electron_index < 0) throw std::runtime_error("constant transport: electron index");
  }
  void common(const double *state, const double *gradUp, const double *Efield, double *diffusionVelocity, double *n_sp,
              double *mobility) {
    double prim[MAXEQ];
    mixture->GetPrimitivesFromConservatives(state, prim);
    const double Te = twoTemperature ? prim[num_equation - 1] : prim[nvel + 1];
    const double Th = prim[nvel + 1];
    double X_sp[MAXSP], Y_sp[MAXSP];
    pm->computeSpeciesPrimitives(state, X_sp, Y_sp, n_sp);
    for (int v = 0; v < nvel; v++)
      for (int sp = 0; sp < numSpecies; sp++) diffusionVelocity[sp + v * numSpecies] = 0.0;
    double gradX[MAXSP * MAXDIM];
    pm->ComputeMoleFractionGradient(n_sp, gradUp, gradX);
    for (int sp = 0; sp < numSpecies; sp++)
      for (int d = 0; d < dim; d++)
        diffusionVelocity[sp + d * numSpecies] = -in.diffusivity[sp] * gradX[sp + d * numSpecies] / (X_sp[sp] + Xeps_);
    for (int sp = 0; sp < numSpecies; sp++) {
      const double temp = (sp == in.electron_index) ? Te : Th;
      mobility[sp] = qeOverkB_ * mixture->GetGasParams(sp, TPSRHS_SPECIES_CHARGES) / temp * in.diffusivity[sp];
    }
    sigma_ = computeMixtureElectricConductivity(mobility, n_sp) * MOLARELECTRONCHARGE;
    if (ambipolar) addAmbipolarEfield(mobility, n_sp, diffusionVelocity);
    addMixtureDrift(mobility, n_sp, Efield, diffusionVelocity);
    correctMassDiffusionFlux(Y_sp, diffusionVelocity);
  }
  double sigma_ = 0.0;
  void ComputeFluxTransportProperties(const double *state, const double *gradUp, const double *Efield, double, double,
                                      double *tb, double *diffusionVelocity) override {
    tb[VISCOSITY] = in.viscosity;
    tb[BULK_VISCOSITY] = in.bulk_viscosity;
    tb[HEAVY_THERMAL_CONDUCTIVITY] = in.thermal_conductivity;
    tb[ELECTRON_THERMAL_CONDUCTIVITY] = in.electron_thermal_conductivity;
    double n_sp[MAXSP], mob[MAXSP];
    common(state, gradUp, Efield, diffusionVelocity, n_sp, mob);
  }
  void ComputeSourceTransportProperties(const double *state, const double *, const double *gradUp, const double *Efield,
                                        double, double *globalTransport, double *speciesTransport,
                                        double *diffusionVelocity, double *n_sp) override {
    double mob[MAXSP];
    common(state, gradUp, Efield, diffusionVelocity, n_sp, mob);
    globalTransport[ELECTRIC_CONDUCTIVITY] = sigma_;
    for (int sp = 0; sp < numSpecies; sp++) speciesTransport[sp + MF_FREQUENCY * numSpecies] = in.mt_freq[sp];
  }
  void GetViscosities(const double *, const double *, double *visc) override {
    visc[0] = in.viscosity;
    visc[1] = in.bulk_viscosity;
  }
};

// ------------------------------------------------------------------------------------------
// GasMinimalTransport, argon ternary (src/gas_transport.cpp:43-830)
// ------------------------------------------------------------------------------------------
class GasMinimalTransport : public TransportProperties {
 public:
  PerfectMixture *pm;
  int electronIndex_, ionIndex_, neutralIndex_;
  const double kB_ = BOLTZMANNCONSTANT;
  const double debyeFactor_ = BOLTZMANNCONSTANT * VACUUMPERMITTIVITY / ELECTRONCHARGE / ELECTRONCHARGE;
  const double qeOverkB_ = ELECTRONCHARGE / BOLTZMANNCONSTANT;
  double viscosityFactor_, kOverEtaFactor_, diffusivityFactor_, mfFreqFactor_;
  double mw_[MAXSP], muw_[MAXSP * MAXSP];
  bool thirdOrderkElectron_, multiply_;
  double fluxTrnsMultiplier_[4], spcsTrnsMultiplier_[1], diffMult_, mobilMult_;

  GasMinimalTransport(PerfectMixture *m, const tpsrhs_gas_transport &in, bool ternary = true)
      : TransportProperties(m), pm(m) {
    viscosityFactor_ = 5. / 16. * std::sqrt(PI_ * kB_);
    kOverEtaFactor_ = 15. / 4. * kB_;
    diffusivityFactor_ = 3. / 16. * std::sqrt(2.0 * PI_ * kB_) / AVOGADRONUMBER;
    mfFreqFactor_ = 4. / 3. * AVOGADRONUMBER * std::sqrt(8. * kB_ / PI_);
    if (ternary && numSpecies != 3) throw std::runtime_error("argon ternary transport supports Ar, Ar.+1, E only");
    if (numSpecies > 7)  // the reference's assert for Gas:Ar, src/gas_transport.cpp:905-911
      throw std::runtime_error("argon mixture transport supports at most 7 species");
    neutralIndex_ = in.neutral_index;
    ionIndex_ = in.ion_index;
    electronIndex_ = in.electron_index;
    if (neutralIndex_ < 0 || ionIndex_ < 0 || electronIndex_ < 0) throw std::runtime_error("argon transport indices");
    for (int sp = 0; sp < numSpecies; sp++) mw_[sp] = m->GetGasParams(sp, TPSRHS_SPECIES_MW);
    if (std::fabs(mw_[neutralIndex_] - mw_[electronIndex_] - mw_[ionIndex_]) >= 1.0e-12)
      throw std::runtime_error("argon transport: inconsistent species masses");
    for (int sp = 0; sp < numSpecies; sp++) mw_[sp] /= AVOGADRONUMBER;
    for (int i = 0; i < numSpecies; i++)
      for (int j = i; j < numSpecies; j++) {
        muw_[i + j * numSpecies] = mw_[i] * mw_[j] / (mw_[i] + mw_[j]);
        if (i != j) muw_[j + i * numSpecies] = muw_[i + j * numSpecies];
      }
    thirdOrderkElectron_ = in.third_order_k_electron != 0;
    multiply_ = in.multiply != 0;
    for (int t = 0; t < 4; t++) fluxTrnsMultiplier_[t] = in.flux_trns_multiplier[t];
    spcsTrnsMultiplier_[0] = in.spcs_trns_multiplier[0];
    diffMult_ = in.diff_mult;
    mobilMult_ = in.mobil_mult;
  }
  double getMuw(int i, int j) const { return muw_[i + j * numSpecies]; }

  // L polynomials of Devoto's third-order electron conductivity, src/gas_transport.hpp:148-158
  static double L11ee(const double *Q2) { return Q2[0]; }
  static double L11ea(const double *Q1) { return 6.25 * Q1[0] - 15. * Q1[1] + 12. * Q1[2]; }
  static double L12ee(const double *Q2) { return 1.75 * Q2[0] - 2.0 * Q2[1]; }
  static double L12ea(const double *Q1) { return 10.9375 * Q1[0] - 39.375 * Q1[1] + 57. * Q1[2] - 30. * Q1[3]; }
  static double L22ee(const double *Q2) { return 4.8125 * Q2[0] - 7.0 * Q2[1] + 5. * Q2[2]; }
  static double L22ea(const double *Q1) {
    return 19.140625 * Q1[0] - 91.875 * Q1[1] + 199.5 * Q1[2] - 210. * Q1[3] + 90. * Q1[4];
  }
  double thirdOrderKe(const double *X_sp, double debyeLength, double Te, double nondimTe) const {  // :400-489
    const double debyeCircle = PI_ * debyeLength * debyeLength;
    double Q2[3] = {debyeCircle * collision::charged::rep22(nondimTe), debyeCircle * collision::charged::rep23(nondimTe),
                    debyeCircle * collision::charged::rep24(nondimTe)};
    double Q1Ion[5] = {debyeCircle * collision::charged::att11(nondimTe), debyeCircle * collision::charged::att12(nondimTe),
                       debyeCircle * collision::charged::att13(nondimTe), debyeCircle * collision::charged::att14(nondimTe),
                       debyeCircle * collision::charged::att15(nondimTe)};
    double Q1N[5];
    for (int r = 1; r <= 5; r++) Q1N[r - 1] = collision::argon::eAr1r(r, Te);
    double L11 = std::sqrt(2.0) * X_sp[electronIndex_] * L11ee(Q2);
    L11 += X_sp[ionIndex_] * L11ea(Q1Ion);
    L11 += X_sp[neutralIndex_] * L11ea(Q1N);
    double L12 = std::sqrt(2.0) * X_sp[electronIndex_] * L12ee(Q2);
    L12 += X_sp[ionIndex_] * L12ea(Q1Ion);
    L12 += X_sp[neutralIndex_] * L12ea(Q1N);
    double L22 = std::sqrt(2.0) * X_sp[electronIndex_] * L22ee(Q2);
    L22 += X_sp[ionIndex_] * L22ea(Q1Ion);
    L22 += X_sp[neutralIndex_] * L22ea(Q1N);
    return viscosityFactor_ * kOverEtaFactor_ * std::sqrt(2.0 * Te / mw_[electronIndex_]) * X_sp[electronIndex_] /
           (L11 - L12 * L12 / L22);
  }

  // shared by the flux and source variants: diffusivity, mobility, diffusion velocity
  void diffusion(const double *X_sp, const double *Y_sp, const double *n_sp, double nTotal, double Te, double Th,
                 double nondimTe, double debyeCircle, const double *gradUp, const double *Efield, bool source,
                 double *diffusivity, double *mobility, double *diffusionVelocity) {
    double binaryDiff[9];
    for (int i = 0; i < 9; i++) binaryDiff[i] = 0.0;
    binaryDiff[electronIndex_ + neutralIndex_ * numSpecies] = diffusivityFactor_ *
                                                              std::sqrt(Te / getMuw(electronIndex_, neutralIndex_)) /
                                                              nTotal / collision::argon::eAr11(Te);
    binaryDiff[neutralIndex_ + electronIndex_ * numSpecies] = binaryDiff[electronIndex_ + neutralIndex_ * numSpecies];
    binaryDiff[neutralIndex_ + ionIndex_ * numSpecies] =
        diffusivityFactor_ * std::sqrt(Th / getMuw(neutralIndex_, ionIndex_)) / nTotal / collision::argon::ArAr1P11(Th);
    binaryDiff[ionIndex_ + neutralIndex_ * numSpecies] = binaryDiff[neutralIndex_ + ionIndex_ * numSpecies];
    binaryDiff[electronIndex_ + ionIndex_ * numSpecies] = diffusivityFactor_ *
                                                          std::sqrt(Te / getMuw(ionIndex_, electronIndex_)) / nTotal /
                                                          (collision::charged::att11(nondimTe) * debyeCircle);
    binaryDiff[ionIndex_ + electronIndex_ * numSpecies] = binaryDiff[electronIndex_ + ionIndex_ * numSpecies];
    CurtissHirschfelder(X_sp, Y_sp, binaryDiff, diffusivity);
    for (int sp = 0; sp < numSpecies; sp++) {
      const double temp = (sp == electronIndex_) ? Te : Th;
      mobility[sp] = qeOverkB_ * mixture->GetGasParams(sp, TPSRHS_SPECIES_CHARGES) / temp * diffusivity[sp];
    }
    (void)source;
  }
  void velocities(const double *X_sp, const double *Y_sp, const double *n_sp, const double *gradUp, const double *Efield,
                  const double *diffusivity, const double *mobility, double *diffusionVelocity) {
    double gradX[MAXEQ * MAXDIM];
    pm->ComputeMoleFractionGradient(n_sp, gradUp, gradX);
    for (int v = 0; v < nvel; v++)
      for (int sp = 0; sp < numSpecies; sp++) diffusionVelocity[sp + v * numSpecies] = 0.0;
    for (int sp = 0; sp < numSpecies; sp++)
      for (int d = 0; d < dim; d++)
        diffusionVelocity[sp + d * numSpecies] = -diffusivity[sp] * gradX[sp + d * numSpecies] / (X_sp[sp] + Xeps_);
    if (ambipolar) addAmbipolarEfield(mobility, n_sp, diffusionVelocity);
    addMixtureDrift(mobility, n_sp, Efield, diffusionVelocity);
    correctMassDiffusionFlux(Y_sp, diffusionVelocity);
  }

  void ComputeFluxTransportProperties(const double *state, const double *gradUp, const double *Efield, double, double,
                                      double *tb, double *diffusionVelocity) override {  // :206-398
    for (int p = 0; p < NUM_FLUX_TRANS; p++) tb[p] = 0.0;
    double prim[MAXEQ];
    mixture->GetPrimitivesFromConservatives(state, prim);
    double n_sp[MAXSP], X_sp[MAXSP], Y_sp[MAXSP];
    pm->computeSpeciesPrimitives(state, X_sp, Y_sp, n_sp);
    double nTotal = 0.0;
    for (int sp = 0; sp < numSpecies; sp++) nTotal += n_sp[sp];
    const double Te = twoTemperature ? prim[num_equation - 1] : prim[nvel + 1];
    const double Th = prim[nvel + 1];
    const double nOverT = (n_sp[electronIndex_] + Xeps_) / Te + (n_sp[ionIndex_] + Xeps_) / Th;
    const double debyeLength = std::sqrt(debyeFactor_ / AVOGADRONUMBER / nOverT);
    const double debyeCircle = PI_ * debyeLength * debyeLength;
    const double nondimTe = debyeLength * 4.0 * PI_ * debyeFactor_ * Te;
    const double nondimTh = debyeLength * 4.0 * PI_ * debyeFactor_ * Th;
    double speciesViscosity[MAXSP], speciesHvyThrmCnd[MAXSP];
    speciesViscosity[ionIndex_] =
        viscosityFactor_ * std::sqrt(mw_[ionIndex_] * Th) / (collision::charged::rep22(nondimTh) * debyeCircle);
    speciesViscosity[neutralIndex_] = viscosityFactor_ * std::sqrt(mw_[neutralIndex_] * Th) / collision::argon::ArAr22(Th);
    speciesViscosity[electronIndex_] = 0.0;
    for (int sp = 0; sp < numSpecies; sp++) speciesHvyThrmCnd[sp] = speciesViscosity[sp] * kOverEtaFactor_ / mw_[sp];
    tb[VISCOSITY] = linearAverage(X_sp, speciesViscosity);
    tb[HEAVY_THERMAL_CONDUCTIVITY] = linearAverage(X_sp, speciesHvyThrmCnd);
    tb[BULK_VISCOSITY] = 0.0;
    if (thirdOrderkElectron_) {
      tb[ELECTRON_THERMAL_CONDUCTIVITY] = thirdOrderKe(X_sp, debyeLength, Te, nondimTe);
    } else {
      tb[ELECTRON_THERMAL_CONDUCTIVITY] = viscosityFactor_ * kOverEtaFactor_ * std::sqrt(Te / mw_[electronIndex_]) *
                                          X_sp[electronIndex_] / (collision::charged::rep22(nondimTe) * debyeCircle);
    }
    double diffusivity[MAXSP], mobility[MAXSP];
    diffusion(X_sp, Y_sp, n_sp, nTotal, Te, Th, nondimTe, debyeCircle, gradUp, Efield, false, diffusivity, mobility,
              diffusionVelocity);
    if (multiply_) {
      for (int t = 0; t < NUM_FLUX_TRANS; t++) tb[t] *= fluxTrnsMultiplier_[t];
      for (int sp = 0; sp < numSpecies; sp++) {
        diffusivity[sp] *= diffMult_;
        mobility[sp] *= mobilMult_;
      }
    }
    velocities(X_sp, Y_sp, n_sp, gradUp, Efield, diffusivity, mobility, diffusionVelocity);
  }
  void ComputeSourceTransportProperties(const double *state, const double *Up, const double *gradUp,
                                        const double *Efield, double, double *globalTransport, double *speciesTransport,
                                        double *diffusionVelocity, double *n_sp) override {  // :592-773
    for (int p = 0; p < NUM_SRC_TRANS; p++) globalTransport[p] = 0.0;
    for (int sp = 0; sp < numSpecies; sp++) speciesTransport[sp] = 0.0;
    double X_sp[3], Y_sp[3];
    pm->computeSpeciesPrimitives(state, X_sp, Y_sp, n_sp);
    double nTotal = 0.0;
    for (int sp = 0; sp < numSpecies; sp++) nTotal += n_sp[sp];
    const double Te = twoTemperature ? Up[num_equation - 1] : Up[nvel + 1];
    const double Th = Up[nvel + 1];
    const double nOverT = (n_sp[electronIndex_] + Xeps_) / Te + (n_sp[ionIndex_] + Xeps_) / Th;
    const double debyeLength = std::sqrt(debyeFactor_ / AVOGADRONUMBER / nOverT);
    const double debyeCircle = PI_ * debyeLength * debyeLength;
    const double nondimTe = debyeLength * 4.0 * PI_ * debyeFactor_ * Te;
    const double Qea = collision::argon::eAr11(Te);
    const double Qie = collision::charged::att11(nondimTe) * debyeCircle;
    double diffusivity[MAXSP], mobility[MAXSP];
    diffusion(X_sp, Y_sp, n_sp, nTotal, Te, Th, nondimTe, debyeCircle, gradUp, Efield, true, diffusivity, mobility,
              diffusionVelocity);
    speciesTransport[ionIndex_ + MF_FREQUENCY * numSpecies] =
        mfFreqFactor_ * std::sqrt(Te / mw_[electronIndex_]) * n_sp[ionIndex_] * Qie;
    speciesTransport[neutralIndex_ + MF_FREQUENCY * numSpecies] =
        mfFreqFactor_ * std::sqrt(Te / mw_[electronIndex_]) * n_sp[neutralIndex_] * Qea;
    // (the reference also writes speciesTransport[neutralIndex2_ ...] with neutralIndex2_ == -1 for
    //  argon, src/gas_transport.cpp:713-714: an out-of-bounds store that no result depends on)
    if (multiply_) {
      for (int sp = 0; sp < numSpecies; sp++) {
        diffusivity[sp] *= diffMult_;
        mobility[sp] *= mobilMult_;
        speciesTransport[sp + MF_FREQUENCY * numSpecies] *= spcsTrnsMultiplier_[MF_FREQUENCY];
      }
    }
    globalTransport[ELECTRIC_CONDUCTIVITY] = computeMixtureElectricConductivity(mobility, n_sp) * MOLARELECTRONCHARGE;
    velocities(X_sp, Y_sp, n_sp, gradUp, Efield, diffusivity, mobility, diffusionVelocity);
  }
  void GetViscosities(const double *conserved, const double *primitive, double *visc) override {  // :775-822
    double n_sp[3], X_sp[3], Y_sp[3];
    pm->computeSpeciesPrimitives(conserved, X_sp, Y_sp, n_sp);
    const double Te = twoTemperature ? primitive[num_equation - 1] : primitive[nvel + 1];
    const double Th = primitive[nvel + 1];
    const double nOverT = (n_sp[electronIndex_] + Xeps_) / Te + (n_sp[ionIndex_] + Xeps_) / Th;
    const double debyeLength = std::sqrt(debyeFactor_ / AVOGADRONUMBER / nOverT);
    const double debyeCircle = PI_ * debyeLength * debyeLength;
    const double nondimTh = debyeLength * 4.0 * PI_ * debyeFactor_ * Th;
    double sv[3];
    sv[ionIndex_] = viscosityFactor_ * std::sqrt(mw_[ionIndex_] * Th) / (collision::charged::rep22(nondimTh) * debyeCircle);
    sv[neutralIndex_] = viscosityFactor_ * std::sqrt(mw_[neutralIndex_] * Th) / collision::argon::ArAr22(Th);
    sv[electronIndex_] = 0.0;
    visc[0] = linearAverage(X_sp, sv);
    visc[1] = 0.0;
    if (multiply_) {
      visc[0] *= fluxTrnsMultiplier_[VISCOSITY];
      visc[1] *= fluxTrnsMultiplier_[BULK_VISCOSITY];
    }
  }
};

// ------------------------------------------------------------------------------------------
// GasMixtureTransport, argon (src/gas_transport.cpp:870-1560): any number of species, every pair's
// collision integral picked from the input table (src/gas_transport.cpp:995-1283)
// ------------------------------------------------------------------------------------------
class GasMixtureTransport : public GasMinimalTransport {
 public:
  int collisionIndex_[MAXSP * MAXSP];
  struct CollisionInputs {
    double Te, Th, debyeCircle, ndimTe, ndimTh;
  };
  GasMixtureTransport(PerfectMixture *m, const tpsrhs_gas_transport &in) : GasMinimalTransport(m, in, /*ternary*/ false) {
    for (int i = 0; i < numSpecies; i++)
      for (int j = i; j < numSpecies; j++) {
        const int c = in.collision_index[i + j * numSpecies];
        if (c < TPSRHS_CLMB_ATT || c > TPSRHS_AR_AR) throw std::runtime_error("collision type outside the argon set");
        collisionIndex_[i + j * numSpecies] = c;
      }
  }
  CollisionInputs computeCollisionInputs(const double *primitive, const double *n_sp) const {  // :185-204
    CollisionInputs c;
    c.Te = twoTemperature ? primitive[num_equation - 1] : primitive[nvel + 1];
    c.Th = primitive[nvel + 1];
    double nOverT = 0.0;
    for (int sp = 0; sp < numSpecies; sp++) {
      const double q = mixture->GetGasParams(sp, TPSRHS_SPECIES_CHARGES);
      nOverT += (n_sp[sp] + Xeps_) / c.Te * q * q;
    }
    const double debyeLength = std::sqrt(debyeFactor_ / AVOGADRONUMBER / nOverT);
    c.debyeCircle = PI_ * debyeLength * debyeLength;
    c.ndimTe = debyeLength * 4.0 * PI_ * debyeFactor_ * c.Te;
    c.ndimTh = debyeLength * 4.0 * PI_ * debyeFactor_ * c.Th;
    return c;
  }
  double collisionIntegral(int a, int b, int l, int r, const CollisionInputs &c) const {
    const int spI = std::min(a, b), spJ = std::max(a, b);
    const int idx = collisionIndex_[spI + spJ * numSpecies];
    const bool withE = (spI == electronIndex_) || (spJ == electronIndex_);
    namespace ch = collision::charged;
    namespace ar = collision::argon;
    if (idx == TPSRHS_CLMB_ATT || idx == TPSRHS_CLMB_REP) {
      const double t = withE ? c.ndimTe : c.ndimTh;
      typedef double (*F)(double);
      static const F att1[5] = {ch::att11, ch::att12, ch::att13, ch::att14, ch::att15};
      static const F att2[3] = {ch::att22, ch::att23, ch::att24};
      static const F rep1[5] = {ch::rep11, ch::rep12, ch::rep13, ch::rep14, ch::rep15};
      static const F rep2[3] = {ch::rep22, ch::rep23, ch::rep24};
      if (l == 1 && r >= 1 && r <= 5) return c.debyeCircle * (idx == TPSRHS_CLMB_ATT ? att1 : rep1)[r - 1](t);
      if (l == 2 && r >= 2 && r <= 4) return c.debyeCircle * (idx == TPSRHS_CLMB_ATT ? att2 : rep2)[r - 2](t);
      throw std::runtime_error("Coulomb collision integral order not supported");
    }
    const double t = withE ? c.Te : c.Th;
    switch (idx) {
      case TPSRHS_AR_AR1P:
        if (l == 1 && r == 1) return ar::ArAr1P11(t);
        break;
      case TPSRHS_AR_E:
        if (l == 1 && r >= 1 && r <= 5) return ar::eAr1r(r, t);
        break;
      case TPSRHS_AR_AR:
        if (l == 1 && r == 1) return ar::ArAr11(t);
        if (l == 2 && r == 2) return ar::ArAr22(t);
        break;
    }
    throw std::runtime_error("collision integral not supported for this pair");
  }
  double thirdOrderKe(const double *X_sp, const CollisionInputs &c) const {  // :1388-1407
    double Q2[3];
    for (int r = 0; r < 3; r++) Q2[r] = collisionIntegral(electronIndex_, electronIndex_, 2, r + 2, c);
    double L11 = std::sqrt(2.0) * X_sp[electronIndex_] * L11ee(Q2);
    double L12 = std::sqrt(2.0) * X_sp[electronIndex_] * L12ee(Q2);
    double L22 = std::sqrt(2.0) * X_sp[electronIndex_] * L22ee(Q2);
    for (int sp = 0; sp < numSpecies; sp++) {
      if (sp == electronIndex_) continue;
      double Q1[5];
      for (int r = 0; r < 5; r++) Q1[r] = collisionIntegral(sp, electronIndex_, 1, r + 1, c);
      L11 += X_sp[sp] * L11ea(Q1);
      L12 += X_sp[sp] * L12ea(Q1);
      L22 += X_sp[sp] * L22ea(Q1);
    }
    return viscosityFactor_ * kOverEtaFactor_ * std::sqrt(2.0 * c.Te / mw_[electronIndex_]) * X_sp[electronIndex_] /
           (L11 - L12 * L12 / L22);
  }
  void mixtureDiffusion(const double *X_sp, const double *Y_sp, double nTotal, const CollisionInputs &c,
                        double *diffusivity, double *mobility) const {
    double binaryDiff[MAXSP * MAXSP];
    for (int i = 0; i < MAXSP * MAXSP; i++) binaryDiff[i] = 0.0;
    for (int spI = 0; spI < numSpecies - 1; spI++)
      for (int spJ = spI + 1; spJ < numSpecies; spJ++) {
        const double temp = ((spI == electronIndex_) || (spJ == electronIndex_)) ? c.Te : c.Th;
        binaryDiff[spI + spJ * numSpecies] =
            diffusivityFactor_ * std::sqrt(temp / getMuw(spI, spJ)) / nTotal / collisionIntegral(spI, spJ, 1, 1, c);
        binaryDiff[spJ + spI * numSpecies] = binaryDiff[spI + spJ * numSpecies];
      }
    CurtissHirschfelder(X_sp, Y_sp, binaryDiff, diffusivity);
    for (int sp = 0; sp < numSpecies; sp++) {
      const double temp = (sp == electronIndex_) ? c.Te : c.Th;
      mobility[sp] = qeOverkB_ * mixture->GetGasParams(sp, TPSRHS_SPECIES_CHARGES) / temp * diffusivity[sp];
    }
  }
  void ComputeFluxTransportProperties(const double *state, const double *gradUp, const double *Efield, double, double,
                                      double *tb, double *diffusionVelocity) override {  // :1285-1386
    for (int p = 0; p < NUM_FLUX_TRANS; p++) tb[p] = 0.0;
    double prim[MAXEQ];
    mixture->GetPrimitivesFromConservatives(state, prim);
    double n_sp[MAXSP], X_sp[MAXSP], Y_sp[MAXSP];
    pm->computeSpeciesPrimitives(state, X_sp, Y_sp, n_sp);
    double nTotal = 0.0;
    for (int sp = 0; sp < numSpecies; sp++) nTotal += n_sp[sp];
    const CollisionInputs c = computeCollisionInputs(prim, n_sp);
    double sv[MAXSP], sk[MAXSP];
    for (int sp = 0; sp < numSpecies; sp++) {
      if (sp == electronIndex_) {
        sv[sp] = sk[sp] = 0.0;
        continue;
      }
      sv[sp] = viscosityFactor_ * std::sqrt(mw_[sp] * c.Th) / collisionIntegral(sp, sp, 2, 2, c);
      sk[sp] = sv[sp] * kOverEtaFactor_ / mw_[sp];
    }
    tb[VISCOSITY] = linearAverage(X_sp, sv);
    tb[HEAVY_THERMAL_CONDUCTIVITY] = linearAverage(X_sp, sk);
    tb[BULK_VISCOSITY] = 0.0;
    if (thirdOrderkElectron_) {
      tb[ELECTRON_THERMAL_CONDUCTIVITY] = thirdOrderKe(X_sp, c);
    } else {
      tb[ELECTRON_THERMAL_CONDUCTIVITY] = viscosityFactor_ * kOverEtaFactor_ * std::sqrt(c.Te / mw_[electronIndex_]) *
                                          X_sp[electronIndex_] / collisionIntegral(electronIndex_, electronIndex_, 2, 2, c);
    }
    double diffusivity[MAXSP], mobility[MAXSP];
    mixtureDiffusion(X_sp, Y_sp, nTotal, c, diffusivity, mobility);
    if (multiply_) {
      for (int t = 0; t < NUM_FLUX_TRANS; t++) tb[t] *= fluxTrnsMultiplier_[t];
      for (int sp = 0; sp < numSpecies; sp++) {
        diffusivity[sp] *= diffMult_;
        mobility[sp] *= mobilMult_;
      }
    }
    velocities(X_sp, Y_sp, n_sp, gradUp, Efield, diffusivity, mobility, diffusionVelocity);
  }
  void ComputeSourceTransportProperties(const double *state, const double *Up, const double *gradUp,
                                        const double *Efield, double, double *globalTransport, double *speciesTransport,
                                        double *diffusionVelocity, double *n_sp) override {  // :1409-1497
    for (int p = 0; p < NUM_SRC_TRANS; p++) globalTransport[p] = 0.0;
    for (int sp = 0; sp < numSpecies; sp++) speciesTransport[sp] = 0.0;
    double X_sp[MAXSP], Y_sp[MAXSP];
    pm->computeSpeciesPrimitives(state, X_sp, Y_sp, n_sp);
    double nTotal = 0.0;
    for (int sp = 0; sp < numSpecies; sp++) nTotal += n_sp[sp];
    const CollisionInputs c = computeCollisionInputs(Up, n_sp);
    double diffusivity[MAXSP], mobility[MAXSP];
    mixtureDiffusion(X_sp, Y_sp, nTotal, c, diffusivity, mobility);
    for (int sp = 0; sp < numSpecies; sp++) {
      if (sp == electronIndex_) continue;
      speciesTransport[sp + MF_FREQUENCY * numSpecies] = mfFreqFactor_ * std::sqrt(c.Te / mw_[electronIndex_]) * n_sp[sp] *
                                                         collisionIntegral(sp, electronIndex_, 1, 1, c);
    }
    if (multiply_) {
      for (int sp = 0; sp < numSpecies; sp++) {
        diffusivity[sp] *= diffMult_;
        mobility[sp] *= mobilMult_;
        speciesTransport[sp + MF_FREQUENCY * numSpecies] *= spcsTrnsMultiplier_[MF_FREQUENCY];
      }
    }
    globalTransport[ELECTRIC_CONDUCTIVITY] = computeMixtureElectricConductivity(mobility, n_sp) * MOLARELECTRONCHARGE;
    velocities(X_sp, Y_sp, n_sp, gradUp, Efield, diffusivity, mobility, diffusionVelocity);
  }
  void GetViscosities(const double *conserved, const double *primitive, double *visc) override {  // :1499-1535
    double n_sp[MAXSP], X_sp[MAXSP], Y_sp[MAXSP];
    pm->computeSpeciesPrimitives(conserved, X_sp, Y_sp, n_sp);
    const CollisionInputs c = computeCollisionInputs(primitive, n_sp);
    double sv[MAXSP];
    for (int sp = 0; sp < numSpecies; sp++)
      sv[sp] = (sp == electronIndex_) ? 0.0
                                      : viscosityFactor_ * std::sqrt(mw_[sp] * c.Th) / collisionIntegral(sp, sp, 2, 2, c);
    visc[0] = linearAverage(X_sp, sv);
    visc[1] = 0.0;
    if (multiply_) {
      visc[0] *= fluxTrnsMultiplier_[VISCOSITY];
      visc[1] *= fluxTrnsMultiplier_[BULK_VISCOSITY];
    }
  }
};

inline TransportProperties *make_transport(PerfectMixture *pm, const tpsrhs_physics &p) {
  switch (p.transport_model) {
    case TPSRHS_CONSTANT:
      return new ConstantTransport(pm, p.constant_transport);
    case TPSRHS_ARGON_MINIMAL:
      return new GasMinimalTransport(pm, p.gas_transport);
    case TPSRHS_ARGON_MIXTURE:
      return new GasMixtureTransport(pm, p.gas_transport);
    default:
      throw std::runtime_error("transport model outside the hot-path scope");
  }
}

// ------------------------------------------------------------------------------------------
// LinearTable (src/table.cpp:39-110)
// ------------------------------------------------------------------------------------------
class LinearTable {
 public:
  int N = 0;
  bool xLog = false, fLog = false;
  std::vector<double> x, f, a, b;
  void init(const tpsrhs_table &t) {
    N = t.n_data;
    xLog = t.x_log_scale != 0;
    fLog = t.f_log_scale != 0;
    x.assign(t.x_data, t.x_data + N);
    f.assign(t.f_data, t.f_data + N);
    a.assign(N, 0.0);
    b.assign(N, 0.0);
    for (int k = 0; k < N - 1; k++) {
      a[k] = fLog ? std::log(f[k]) : f[k];
      const double df = fLog ? (std::log(f[k + 1]) - std::log(f[k])) : (f[k + 1] - f[k]);
      b[k] = xLog ? df / (std::log(x[k + 1]) - std::log(x[k])) : df / (x[k + 1] - x[k]);
      a[k] -= xLog ? b[k] * std::log(x[k]) : b[k] * x[k];
    }
  }
  int findInterval(double xEval) const {
    int count = N, first = 0;
    while (count > 0) {
      int it = first;
      const int step = count / 2;
      it += step;
      if (xEval > x[it]) {
        first = ++it;
        count -= step + 1;
      } else {
        count = step;
      }
    }
    first = std::max(1, std::min(N - 1, first));
    return first - 1;
  }
  double eval(double xEval) const {
    const int index = findInterval(xEval);
    const double xt = xLog ? std::log(xEval) : xEval;
    double ft = a[index] + b[index] * xt;
    if (fLog) ft = std::exp(ft);
    return ft;
  }
};

// ------------------------------------------------------------------------------------------
// Chemistry (src/chemistry.cpp:40-299) with Arrhenius / HoffertLien / Tabulated (src/reaction.cpp:38-83)
// ------------------------------------------------------------------------------------------
class Chemistry {
 public:
  PerfectMixture *mixture;
  tpsrhs_chemistry in;
  int numSpecies, numReactions;
  std::vector<LinearTable> tables;
  Chemistry(PerfectMixture *m, const tpsrhs_chemistry &c) : mixture(m), in(c) {
    numSpecies = m->numSpecies;
    numReactions = c.num_reactions;
    tables.resize(numReactions);
    for (int r = 0; r < numReactions; r++) {
      if (c.reaction_models[r] == TPSRHS_TABULATED_RXN) tables[r].init(c.rate_tables[r]);
      if (c.reaction_models[r] > TPSRHS_TABULATED_RXN) throw std::runtime_error("reaction model outside the scope");
    }
  }
  bool isElectronInvolvedAt(int r) const {
    return (in.electron_index < 0) ? false : (in.reactant_stoich[in.electron_index + r * numSpecies] != 0);
  }
  void computeForwardRateCoeffs(double T_h, double T_e, double *kfwd) const {
    const double Thlim = std::max(T_h, in.minimum_temperature);
    const double Telim = std::max(T_e, in.minimum_temperature);
    for (int r = 0; r < numReactions; r++) {
      const double temp = isElectronInvolvedAt(r) ? Telim : Thlim;
      const double A = in.rate_params[0 + r * TPSRHS_MAXCHEMPARAMS], b = in.rate_params[1 + r * TPSRHS_MAXCHEMPARAMS],
                   E = in.rate_params[2 + r * TPSRHS_MAXCHEMPARAMS];
      switch (in.reaction_models[r]) {
        case TPSRHS_ARRHENIUS:
          kfwd[r] = A * std::pow(temp, b) * std::exp(-E / UNIVERSALGASCONSTANT / temp);
          break;
        case TPSRHS_HOFFERTLIEN: {
          const double tf = E / BOLTZMANNCONSTANT / temp;
          kfwd[r] = A * std::pow(temp, b) * (tf + 2.0) * std::exp(-tf);
        } break;
        default:
          kfwd[r] = tables[r].eval(temp);
      }
    }
  }
  void computeEquilibriumConstants(double T_h, double T_e, double *kC) const {
    const double Thlim = std::max(T_h, in.minimum_temperature);
    const double Telim = std::max(T_e, in.minimum_temperature);
    for (int r = 0; r < numReactions; r++) {
      kC[r] = 0.0;
      const double temp = isElectronInvolvedAt(r) ? Telim : Thlim;
      if (in.detailed_balance[r])
        kC[r] = in.equilibrium_constant_params[0 + r * TPSRHS_MAXCHEMPARAMS] *
                std::pow(temp, in.equilibrium_constant_params[1 + r * TPSRHS_MAXCHEMPARAMS]) *
                std::exp(-in.equilibrium_constant_params[2 + r * TPSRHS_MAXCHEMPARAMS] / temp);
    }
  }
  void computeProgressRate(const double *ns, const double *kfwd, const double *keq, double *progressRate) const {
    for (int r = 0; r < numReactions; r++) {
      double rate = 1.;
      for (int sp = 0; sp < numSpecies; sp++) rate *= std::pow(ns[sp], in.reactant_stoich[sp + r * numSpecies]);
      if (in.detailed_balance[r]) {
        double rateBWD = 1.;
        for (int sp = 0; sp < numSpecies; sp++) rateBWD *= std::pow(ns[sp], in.product_stoich[sp + r * numSpecies]);
        rate -= rateBWD / keq[r];
      }
      progressRate[r] = kfwd[r] * rate;
    }
  }
  void computeCreationRate(const double *progressRate, double *creationRate) const {
    for (int sp = 0; sp < numSpecies; sp++) {
      creationRate[sp] = 0.;
      for (int r = 0; r < numReactions; r++)
        creationRate[sp] +=
            progressRate[r] * (in.product_stoich[sp + r * numSpecies] - in.reactant_stoich[sp + r * numSpecies]);
      creationRate[sp] *= mixture->GetGasParams(sp, TPSRHS_SPECIES_MW);
    }
  }
};

// ------------------------------------------------------------------------------------------
// SourceTerm::updateTerms (src/source_term.cpp:62-256)
// ------------------------------------------------------------------------------------------
class SourceTerm {
 public:
  int dim, num_equation, nvel, numSpecies, numActiveSpecies, numReactions;
  PerfectMixture *mixture;
  TransportProperties *transport;
  Chemistry chemistry;
  bool enableRadiation;
  LinearTable nec;
  SourceTerm(int dim_, int neq, PerfectMixture *m, TransportProperties *t, const tpsrhs_physics &p)
      : dim(dim_), num_equation(neq), mixture(m), transport(t), chemistry(m, p.chemistry) {
    nvel = m->nvel;
    numSpecies = m->numSpecies;
    numActiveSpecies = m->numActiveSpecies;
    numReactions = p.chemistry.num_reactions;
    enableRadiation = p.radiation.model == TPSRHS_NET_EMISSION;
    if (enableRadiation) nec.init(p.radiation.nec_table);
  }
  void point(const double *Uin, const double *Upin, const double *gradUpn, double *srcTerm) const {
    double upn[MAXEQ], Un[MAXEQ];
    for (int eq = 0; eq < num_equation; eq++) {
      upn[eq] = Upin[eq];
      Un[eq] = Uin[eq];
      srcTerm[eq] = 0.0;
    }
    for (int sp = 0; sp < numActiveSpecies; sp++) {
      const int eq = 3 + 2 + sp;  // hard-coded nvel = 3 in the reference (src/source_term.cpp:129)
      if (eq < num_equation) {
        upn[eq] = std::max(upn[eq], 0.0);
        Un[eq] = std::max(Un[eq], 0.0);
      }
    }
    double Efield[MAXDIM] = {0, 0, 0};
    double globalTransport[MAXSP], speciesTransport[MAXSP], diffusionVelocity[MAXSP * MAXDIM], ns[MAXSP];
    for (int i = 0; i < MAXSP * MAXDIM; i++) diffusionVelocity[i] = 0.0;
    transport->ComputeSourceTransportProperties(Un, upn, gradUpn, Efield, 0.0, globalTransport, speciesTransport,
                                                diffusionVelocity, ns);
    const double Th = upn[1 + nvel];
    const double Te = mixture->twoTemperature ? upn[num_equation - 1] : Th;
    double progressRates[TPSRHS_MAXREACTIONS], creationRates[MAXSP];
    for (int r = 0; r < TPSRHS_MAXREACTIONS; r++) progressRates[r] = 0.0;
    if (numSpecies > 1 && numReactions > 0) {
      double kfwd[TPSRHS_MAXREACTIONS], kC[TPSRHS_MAXREACTIONS];
      chemistry.computeForwardRateCoeffs(Th, Te, kfwd);
      chemistry.computeEquilibriumConstants(Th, Te, kC);
      chemistry.computeProgressRate(ns, kfwd, kC, progressRates);
      chemistry.computeCreationRate(progressRates, creationRates);
      for (int sp = 0; sp < numActiveSpecies; sp++) srcTerm[2 + nvel + sp] += creationRates[sp];
    }
    if (enableRadiation) srcTerm[1 + nvel] += -4.0 * PI_ * nec.eval(Th);  // src/radiation.hpp:68
    if (mixture->twoTemperature) {
      for (int r = 0; r < numReactions; r++)
        if (chemistry.isElectronInvolvedAt(r))
          srcTerm[num_equation - 1] -= chemistry.in.reaction_energies[r] * progressRates[r];
      double gradPe[MAXDIM];
      mixture->computeElectronPressureGrad(ns[numSpecies - 2], Te, gradUpn, gradPe);
      for (int d = 0; d < dim; d++) srcTerm[num_equation - 1] += gradPe[d] * upn[d + 1];
      const double me = mixture->GetGasParams(numSpecies - 2, TPSRHS_SPECIES_MW);
      const double ne = ns[numSpecies - 2];
      for (int sp = 0; sp < numSpecies; sp++) {
        if (sp == numSpecies - 2) continue;
        const double m_sp = mixture->GetGasParams(sp, TPSRHS_SPECIES_MW);
        double energy = 1.5 * UNIVERSALGASCONSTANT * (Te - Th);
        energy *= 2.0 * me * m_sp / (m_sp + me) / (m_sp + me) * ne * speciesTransport[sp + MF_FREQUENCY * numSpecies];
        srcTerm[num_equation - 1] -= energy;
      }
    }
  }
  void updateTerms(const double *U, const double *Up, const double *gradUp, int64_t N, double *y) const {
    OmpGuard guard;
#pragma omp parallel for schedule(static)
    for (int64_t n = 0; n < N; n++) try {
      double upn[MAXEQ], Un[MAXEQ], g[MAXEQ * MAXDIM], src[MAXEQ];
      for (int eq = 0; eq < num_equation; eq++) {
        upn[eq] = Up[n + eq * N];
        Un[eq] = U[n + eq * N];
        for (int d = 0; d < dim; d++) g[eq + d * num_equation] = gradUp[n + eq * N + d * num_equation * N];
      }
      point(Un, upn, g, src);
      for (int eq = 0; eq < num_equation; eq++) y[n + eq * N] += src[eq];
    } catch (const std::exception &e) {
      guard.capture(e);
    }
    guard.rethrow();
  }
};

// AxisymmetricSource::updateTerms at one node (src/forcing_terms.cpp:293-380)
inline void axisym_source_point(GasMixture &mix, TransportProperties &trans, int eqSys, int neqn, int sdim,
                                double radius, const double *Uin, const double *Upin, const double *gradUp, int64_t n,
                                int64_t dof, double *y) {
  double U[MAXEQ], Up[MAXEQ];
  for (int eq = 0; eq < neqn; eq++) {
    U[eq] = Uin[eq];
    Up[eq] = Upin[eq];
  }
  for (int sp = 0; sp < mix.numActiveSpecies; sp++) {
    const int eq = 3 + 2 + sp;
    U[eq] = std::max(U[eq], 0.0);
    Up[eq] = std::max(Up[eq], 0.0);
  }
  const double rho = Up[0], ur = Up[1], ut = Up[3];
  double pressure;
  if (PerfectMixture *pm = dynamic_cast<PerfectMixture *>(&mix)) {
    pressure = pm->ComputePressureFromPrimitives(Up);
  } else if (mix.GetGasConstant() == 0.0) {  // the table gas (oracle/lte.hpp): rho R(T) T, src/lte_mixture.cpp:138-147
    pressure = mix.ComputePressureFromPrimitives(Up);
  } else {
    pressure = mix.GetGasConstant() * Up[0] * Up[mix.iTh];  // DryAir::ComputePressureFromPrimitives
  }
  const double rurut = rho * ur * ut, rutut = rho * ut * ut;
  double tau_tt = 0.0, tau_tr = 0.0;
  if (eqSys != TPSRHS_EULER) {
    const double ur_r = gradUp[1 + 0 * neqn], uz_z = gradUp[2 + 1 * neqn], ut_r = gradUp[3 + 0 * neqn];
    double visc_vec[2];
    trans.GetViscosities(U, Up, visc_vec);
    const double visc = visc_vec[0];
    const double bulkVisc = visc_vec[1] - 2. / 3. * visc;
    double divV = ur_r + uz_z;
    if (radius > 0) divV += ur / radius;
    tau_tt = (radius > 0) ? 2.0 * ur / radius * visc : 0.0;
    tau_tt += bulkVisc * divV;
    tau_tr = ut_r;
    if (radius > 0) tau_tr -= ut / radius;
    tau_tr *= visc;
  }
  (void)sdim;
  y[n + 1 * dof] += (pressure + rutut - tau_tt) / radius;
  y[n + 3 * dof] += (-rurut + tau_tr) / radius;
}

}  // namespace tpsoracle
#endif
