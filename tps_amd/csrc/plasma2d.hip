// Kernel instantiations of the ambipolar ternary plasma, planar 2-D.
#include "operator.hpp"
#include "physics_plasma.hpp"

template <bool TWOT, int TR>
static void pick(tpsrhs_operator *op) {
  typedef PlasmaPhys<2, 2, 3, true, TWOT, TR> PH;
  upload_tables(2, op->order);
  switch (op->order) {
    case 1: op->launch = &launch_all<2, 1, PH>; break;
    case 2: op->launch = &launch_all<2, 2, PH>; break;
    case 3: op->launch = &launch_all<2, 3, PH>; break;
    default: throw Unsupported("plasma kernels are built for polynomial orders 1..3");
  }
}

void pick_plasma2d(tpsrhs_operator *op, bool two_temperature, int transport) {
  if (transport == TRANSPORT_CONSTANT) {
    if (two_temperature) pick<true, TRANSPORT_CONSTANT>(op); else pick<false, TRANSPORT_CONSTANT>(op);
  } else if (transport == TRANSPORT_ARGON_MINIMAL) {
    if (two_temperature) pick<true, TRANSPORT_ARGON_MINIMAL>(op); else pick<false, TRANSPORT_ARGON_MINIMAL>(op);
  } else {
    if (two_temperature) pick<true, TRANSPORT_ARGON_MIXTURE>(op); else pick<false, TRANSPORT_ARGON_MIXTURE>(op);
  }
}
