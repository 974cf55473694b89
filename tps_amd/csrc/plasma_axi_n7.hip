// Plasma kernel family: dim 2, 3 velocity components, 7 species, ambipolar = false.
#include "plasma_family.hpp"
TPSRHS_PLASMA_FAMILY(pick_plasma_axi_n7, 2, 3, 7, false)
