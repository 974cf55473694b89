// Plasma kernel family: dim 2, 3 velocity components, 6 species, ambipolar = false.
#include "plasma_family.hpp"
TPSRHS_PLASMA_FAMILY(pick_plasma_axi_n6, 2, 3, 6, false)
