// Plasma kernel family: dim 2, 3 velocity components, 5 species, ambipolar = true.
#include "plasma_family.hpp"
TPSRHS_PLASMA_FAMILY(pick_plasma_axi_n5a, 2, 3, 5, true)
