// Plasma kernel family: dim 2, 3 velocity components, 8 species, ambipolar = false.
#include "plasma_family.hpp"
TPSRHS_PLASMA_FAMILY(pick_plasma_axi_n8, 2, 3, 8, false)
