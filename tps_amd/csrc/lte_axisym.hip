// Kernel instantiations of the table gas (WorkingFluid::LTE_FLUID, one-dimensional tables), axisymmetric
// (dim 2, velocity components r, z, theta).
#include "operator.hpp"
#include "physics_dryair_axisym.hpp"

void pick_lte_axisym(tpsrhs_operator *op) { pick_order<2, LteAxiPhys>(op); }
