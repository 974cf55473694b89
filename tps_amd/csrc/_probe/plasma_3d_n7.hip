// PROBE build (not shipped): only the instantiation that tools/sweep_instantiations.py found wrong in round 3 --
// 3-D, Gauss-Lobatto pair, p = 1, seven species, not ambipolar, single temperature, argon mixture transport.
#include "../operator.hpp"
#include "../physics_plasma.hpp"

extern "C" int pick_plasma_3d_n7(tpsrhs_operator *op, int two_temperature, int transport, char *, int) {
  typedef PlasmaPhys<3, 3, 7, false, false, TRANSPORT_ARGON_MIXTURE> PH;
  if (two_temperature || transport != TRANSPORT_ARGON_MIXTURE || !op->nc || op->order != 1) throw Unsupported("probe build");
  upload_tables(3, op->order, 1);
  op->point_eval = &launch_point_eval<PH>;
  op->launch = &launch_all<3, 1, PH, 1>;
  return TPSRHS_OK;
}
