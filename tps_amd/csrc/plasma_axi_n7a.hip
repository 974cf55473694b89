// Plasma kernel family: dim 2, 3 velocity components, 7 species, ambipolar = true.
#include "plasma_family.hpp"
TPSRHS_PLASMA_FAMILY(pick_plasma_axi_n7a, 2, 3, 7, true)
