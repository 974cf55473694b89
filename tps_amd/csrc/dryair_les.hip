// Kernel instantiations of dry air with the sub-grid scale models and the planar viscous sponge of the reference's
// Fluxes (src/fluxes.cpp:223-246, 513-688): planar 2-D and 3-D, Gauss-Legendre pair (collocated kernels only).
#include "operator.hpp"
#include "physics_dryair.hpp"

template <int DIM>
static void pick_les_order(tpsrhs_operator *op) {
  typedef DryAirPhys<DIM, false, true> PH;
  upload_tables(DIM, op->order);
  op->point_eval = &launch_point_eval<PH>;
  switch (op->order) {
    case 1: op->launch = &launch_all<DIM, 1, PH>; break;
    case 2: op->launch = &launch_all<DIM, 2, PH>; break;
    case 3: op->launch = &launch_all<DIM, 3, PH>; break;
    case 4: op->launch = &launch_all<DIM, 4, PH>; break;
    case 5: op->launch = &launch_all<DIM, 5, PH>; break;
    default: throw Unsupported("polynomial order " + std::to_string(op->order) + " is not built (1..5)");
  }
}

void pick_dryair_les(tpsrhs_operator *op) {
  if (op->dim == 3)
    pick_les_order<3>(op);
  else
    pick_les_order<2>(op);
}
