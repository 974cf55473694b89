// Host side of the plasma parameter block: tpsrhs_physics (the PODs of src/dataStructures.hpp:537-729) -> PlasmaParams / ChemDev
// as the kernels read them, with the reference's constructor checks (PerfectMixture, GasMinimalTransport,
// GasMixtureTransport, Chemistry).  No HIP call in here: libtpsrhs.so (tpsrhs.hip) uploads the result, and the host-side
// sanitizer build of the point physics (tests/host_physics/) fills the same blocks with the same code.
#ifndef TPSRHS_PLASMA_PARAMS_HOST_HPP_
#define TPSRHS_PLASMA_PARAMS_HOST_HPP_

#include <cmath>
#include <cstring>
#include <stdexcept>
#include <string>
#include <vector>

#include "physics_plasma.hpp"

namespace tpsrhs {

struct UnsupportedConfig : std::runtime_error {  // -> TPSRHS_ERR_UNSUPPORTED
  explicit UnsupportedConfig(const std::string &s) : std::runtime_error(s) {}
};

// LinearTable::LinearTable (src/table.cpp:39-50): abscissae and interval coefficients [x | a | b] in `buf`
inline TableDev table_coeffs(const tpsrhs_table &t, std::vector<double> &buf) {
  if (t.n_data < 2 || !t.x_data || !t.f_data) throw std::invalid_argument("table needs >= 2 points");
  const int N = t.n_data;
  buf.assign(3 * static_cast<size_t>(N), 0.0);
  double *x = buf.data(), *a = x + N, *b = a + N;
  for (int k = 0; k < N; k++) x[k] = t.x_data[k];
  for (int k = 0; k < N - 1; k++) {
    const double f0 = t.f_data[k], f1 = t.f_data[k + 1];
    a[k] = t.f_log_scale ? std::log(f0) : f0;
    const double df = t.f_log_scale ? (std::log(f1) - std::log(f0)) : (f1 - f0);
    b[k] = t.x_log_scale ? df / (std::log(x[k + 1]) - std::log(x[k])) : df / (x[k + 1] - x[k]);
    a[k] -= t.x_log_scale ? b[k] * std::log(x[k]) : b[k] * x[k];
  }
  TableDev td;
  td.n = N;
  td.x_log = t.x_log_scale;
  td.f_log = t.f_log_scale;
  td.pad = 0;
  td.x = td.a = td.b = nullptr;  // set by the caller once the coefficients have a home (device buffer)
  // uniformly spaced abscissae?  (the device then starts its interval search from the spacing)
  td.x0 = x[0];
  td.inv_dx = 0.0;
  const double dx = (x[N - 1] - x[0]) / (N - 1);
  if (dx > 0.0) {
    bool uniform = true;
    for (int k = 0; k < N && uniform; k++) uniform = std::fabs(x[k] - (x[0] + k * dx)) <= 0.25 * dx;
    if (uniform) td.inv_dx = 1.0 / dx;
  }
  return td;
}


// `place_table(const tpsrhs_table &) -> TableDev` gives a table's coefficients a home (the library: a device buffer) and
// returns the record that points there.  p.chem is left for the caller (the library uploads `c` and stores its address).
template <int NSP, class PlaceTable>
void fill_plasma_params(PlasmaParams<NSP> &p, ChemDev &chem, const tpsrhs_disc *disc, const tpsrhs_physics *phys, int num_bcs,
                        const tpsrhs_bc *bcs, PlaceTable &&place_table) {
  const tpsrhs_perfect_mixture &mx = phys->mixture;
  std::memset(static_cast<void *>(&p), 0, sizeof(p));
  for (int sp = 0; sp < NSP; sp++) {
    p.mw[sp] = mx.gas_params[sp + TPSRHS_SPECIES_MW * NSP];
    p.charge[sp] = mx.gas_params[sp + TPSRHS_SPECIES_CHARGES * NSP];
    p.eform[sp] = mx.gas_params[sp + TPSRHS_FORMATION_ENERGY * NSP];
    p.cv[sp] = mx.molar_cv[sp] * kRgas;
    p.cp[sp] = p.cv[sp] + kRgas;
    p.imw[sp] = 1.0 / p.mw[sp];
    p.mwp[sp] = p.mw[sp] / kAvogadro;
    p.sq_mwp[sp] = std::sqrt(p.mwp[sp]);
    p.kf_imwp[sp] = (15. / 4. * kBoltz) / p.mwp[sp];
    p.vf_sq_mwp[sp] = 5. / 16. * std::sqrt(kPi * kBoltz) * p.sq_mwp[sp];
    p.qkb_charge[sp] = (kQe / kBoltz) * p.charge[sp];
    p.rg_imw[sp] = kRgas / p.mw[sp];
  }
  p.ke_fac = 5. / 16. * std::sqrt(kPi * kBoltz) * (15. / 4. * kBoltz) / std::sqrt(p.mwp[NSP - 2]);
  p.ke_fac3 = std::sqrt(2.0) * p.ke_fac;
  p.icv_e = 1.0 / p.cv[NSP - 2];
  {  // sqrt(m_i m_j / (m_i + m_j)) / d_fc of the binary diffusivities (src/gas_transport.cpp:291-310,1353-1365)
    const double dfc = 3. / 16. * std::sqrt(2.0 * kPi * kBoltz) / kAvogadro;
    for (int i = 0; i < NSP; i++)
      for (int j = 0; j < NSP; j++)
        p.sq_muw_idfc[i + j * NSP] = std::sqrt(p.mwp[i] * p.mwp[j] / (p.mwp[i] + p.mwp[j])) / dfc;
  }
  // PerfectMixture::PerfectMixture consistency checks, src/equation_of_state.cpp:505-530
  if (p.charge[NSP - 1] != 0.0 || p.eform[NSP - 2] != 0.0 || p.eform[NSP - 1] != 0.0)
    throw std::invalid_argument("mixture: background must be neutral; background/electron formation energy must be 0");
  const tpsrhs_constant_transport &ct = phys->constant_transport;
  p.c_visc = ct.viscosity;
  p.c_bulk = ct.bulk_viscosity;
  p.c_k = ct.thermal_conductivity;
  p.c_ke = ct.electron_thermal_conductivity;
  for (int sp = 0; sp < NSP; sp++) {
    p.c_diff[sp] = ct.diffusivity[sp];
    p.c_mtfreq[sp] = ct.mt_freq[sp];
  }
  p.c_eidx = ct.electron_index;
  if (phys->transport_model == TPSRHS_CONSTANT && mx.two_temperature && ct.electron_index < 0)
    throw std::invalid_argument("constant transport: electron index required for two-temperature plasma");
  const tpsrhs_gas_transport &gt = phys->gas_transport;
  if (phys->transport_model == TPSRHS_ARGON_MINIMAL) {
    // GasMinimalTransport::GasMinimalTransport, src/gas_transport.cpp:43-128
    if (NSP != 3) throw UnsupportedConfig("argon_minimal transport is the ternary (Ar, Ar.+1, E) model");
    if (gt.electron_index != NSP - 2 || gt.neutral_index != NSP - 1 || gt.ion_index != 0)
      throw std::invalid_argument("argon transport: species must be ordered (ion, electron, neutral background)");
    if (std::fabs(p.mw[gt.neutral_index] - p.mw[gt.electron_index] - p.mw[gt.ion_index]) >= 1.0e-12)
      throw std::invalid_argument("argon transport: inconsistent species masses");
  }
  if (phys->transport_model == TPSRHS_ARGON_MIXTURE) {
    // GasMixtureTransport::GasMixtureTransport + the (pair, l, r) requests of its closures
    // (src/gas_transport.cpp:870-990, 1285-1497): every request must have a fit
    if (gt.electron_index != NSP - 2) throw std::invalid_argument("argon mixture transport: electron index");
    for (int i = 0; i < NSP; i++)
      for (int j = i; j < NSP; j++) {
        const int c = gt.collision_index[i + j * NSP];
        const bool e_i = (i == NSP - 2), e_j = (j == NSP - 2);
        bool ok;
        if (i == j)  // (2,2) for the viscosity / k_e
          ok = e_i ? (c == TPSRHS_CLMB_REP) : (c == TPSRHS_CLMB_REP || c == TPSRHS_AR_AR);
        else if (e_i || e_j)  // (1,1..5) against electrons
          ok = (c == TPSRHS_CLMB_ATT || c == TPSRHS_CLMB_REP || c == TPSRHS_AR_E);
        else  // (1,1) between heavy species
          ok = (c == TPSRHS_CLMB_ATT || c == TPSRHS_CLMB_REP || c == TPSRHS_AR_AR1P || c == TPSRHS_AR_AR);
        if (!ok) throw UnsupportedConfig("argon mixture transport: no collision integral for a species pair of this type");
        p.coll[i + j * NSP] = c;
        // the collision types that occur, by group (PlasmaPhys::coll_table evaluates each of them once per point)
        if (i == j) {
          if (!e_i) p.cmask_diag |= 1 << c;
        } else if (e_i || e_j) {
          p.cmask_e |= 1 << c;
        } else {
          p.cmask_h |= 1 << c;
        }
      }
  }
  p.third_order = gt.third_order_k_electron;
  p.multiply = gt.multiply;
  for (int k = 0; k < 4; k++) p.mult_flux[k] = gt.flux_trns_multiplier[k];
  p.mult_spcs = gt.spcs_trns_multiplier[0];
  p.mult_diff = gt.diff_mult;
  p.mult_mobil = gt.mobil_mult;
  p.eq_system = phys->eq_system;
  p.use_bc_in_grad = disc->use_bc_in_grad;
  p.axisymmetric = disc->axisymmetric;
  p.num_bcs = num_bcs;
  for (int i = 0; i < num_bcs; i++) {
    p.bc[i].category = bcs[i].category;
    p.bc[i].type = bcs[i].type;
    for (int k = 0; k < 4 + TPSRHS_MAXSPECIES; k++) p.bc[i].data[k] = bcs[i].data[k];
  }
  // chemistry + radiation block
  const tpsrhs_chemistry &ch = phys->chemistry;
  if (ch.num_reactions < 0 || ch.num_reactions > TPSRHS_MAXREACTIONS) throw std::invalid_argument("num_reactions");
  ChemDev *c = &chem;
  std::memset(static_cast<void *>(c), 0, sizeof(ChemDev));
  c->num_reactions = ch.num_reactions;
  c->electron_index = ch.electron_index;
  c->min_temperature = ch.minimum_temperature;
  for (int r = 0; r < ch.num_reactions; r++) {
    c->energy[r] = ch.reaction_energies[r];
    c->model[r] = static_cast<signed char>(ch.reaction_models[r]);
    c->detailed_balance[r] = ch.detailed_balance[r] != 0;
    if (ch.reaction_models[r] > TPSRHS_TABULATED_RXN || ch.reaction_models[r] < 0)
      throw UnsupportedConfig("reaction model outside the built scope (Arrhenius, Hoffert-Lien, tabulated)");
    for (int k = 0; k < TPSRHS_MAXCHEMPARAMS; k++) {
      c->rate[k + r * TPSRHS_MAXCHEMPARAMS] = ch.rate_params[k + r * TPSRHS_MAXCHEMPARAMS];
      c->keq[k + r * TPSRHS_MAXCHEMPARAMS] = ch.equilibrium_constant_params[k + r * TPSRHS_MAXCHEMPARAMS];
    }
    for (int sp = 0; sp < NSP; sp++) {
      const int a = ch.reactant_stoich[sp + r * NSP], b = ch.product_stoich[sp + r * NSP];
      if (a < 0 || a > 8 || b < 0 || b > 8) throw std::invalid_argument("stoichiometric coefficient out of range");
      c->reactant[sp + r * NSP] = static_cast<signed char>(a);
      c->product[sp + r * NSP] = static_cast<signed char>(b);
    }
    c->table_share[r] = static_cast<signed char>(r);
    if (ch.reaction_models[r] == TPSRHS_TABULATED_RXN) {
      c->table[r] = place_table(ch.rate_tables[r]);
      const tpsrhs_table &t = ch.rate_tables[r];
      for (int q = 0; q < r; q++) {  // an earlier tabulated reaction on the same abscissae (same scale flag: the same search)?
        if (ch.reaction_models[q] != TPSRHS_TABULATED_RXN || c->table_share[q] != q) continue;
        const tpsrhs_table &o = ch.rate_tables[q];
        if (o.n_data != t.n_data) continue;
        bool same = true;
        for (int k = 0; k < t.n_data && same; k++) same = (o.x_data[k] == t.x_data[k]);
        if (same) {
          c->table_share[r] = static_cast<signed char>(q);
          break;
        }
      }
    }
  }
  c->radiation = phys->radiation.model;
  if (c->radiation == TPSRHS_NET_EMISSION) c->nec = place_table(phys->radiation.nec_table);
}

}  // namespace tpsrhs
#endif
