// Plasma kernel family: dim 2, 3 velocity components, 8 species, ambipolar = true.
#include "plasma_family.hpp"
TPSRHS_PLASMA_FAMILY(pick_plasma_axi_n8a, 2, 3, 8, true)
