// Plasma kernel family: dim 2, 3 velocity components, 6 species, ambipolar = true.
#include "plasma_family.hpp"
TPSRHS_PLASMA_FAMILY(pick_plasma_axi_n6a, 2, 3, 6, true)
