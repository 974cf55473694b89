// TEST INFRASTRUCTURE -- CPU oracle, never shipped, never on the product path.
//
// Multi-species plasma closures of the oracle: PerfectMixture, ConstantTransport,
// GasMinimalTransport (argon), Chemistry/Reaction, NetEmission, SourceTerm, AxisymmetricSource.
#ifndef TPS_ORACLE_PLASMA_HPP_
#define TPS_ORACLE_PLASMA_HPP_

#include "physics.hpp"

namespace tpsoracle {

class PerfectMixture : public GasMixture {
 public:
  PerfectMixture(const tpsrhs_perfect_mixture &, int, int) {
    throw std::runtime_error("PerfectMixture: not built yet in the oracle");
  }
  double ComputePressure(const double *, double * = nullptr) const override { return 0; }
  double ComputeTemperature(const double *) const override { return 0; }
  double ComputeMaxCharSpeed(const double *) const override { return 0; }
  void GetPrimitivesFromConservatives(const double *, double *) const override {}
  void GetConservativesFromPrimitives(const double *, double *) const override {}
  void computeSpeciesEnthalpies(const double *, double *) const override {}
  void computeStagnationState(const double *, double *) const override {}
  void computeStagnantStateWithTemp(const double *, double, double *) const override {}
  void modifyEnergyForPressure(const double *, double *, double, bool = false) const override {}
};

inline TransportProperties *make_transport(PerfectMixture *, const tpsrhs_physics &) {
  throw std::runtime_error("plasma transport: not built yet in the oracle");
}

class SourceTerm {
 public:
  SourceTerm(int, int, PerfectMixture *, TransportProperties *, const tpsrhs_physics &) {}
  void point(const double *, const double *, const double *, double *) const {}
  void updateTerms(const double *, const double *, const double *, int64_t, double *) const {}
};

inline void axisym_source_point(GasMixture &, TransportProperties &, int, int, int, double, const double *,
                                const double *, const double *, int64_t, int64_t, double *) {
  throw std::runtime_error("axisymmetric source: not built yet in the oracle");
}

}  // namespace tpsoracle
#endif
