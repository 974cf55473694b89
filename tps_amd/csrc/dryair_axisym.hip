// Kernel instantiations of dry air, axisymmetric (dim 2, velocity components r, z, theta).
#include "operator.hpp"
#include "physics_dryair_axisym.hpp"

void pick_dryair_axisym(tpsrhs_operator *op) { pick_order<2, DryAirAxiPhys>(op); }
