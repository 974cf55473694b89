// Plasma kernel family: dim 2, 3 velocity components, 4 species, ambipolar = false.
#include "plasma_family.hpp"
TPSRHS_PLASMA_FAMILY(pick_plasma_axi_n4, 2, 3, 4, false)
