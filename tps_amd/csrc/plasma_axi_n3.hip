// Plasma kernel family: dim 2, 3 velocity components, 3 species, ambipolar = false.
#include "plasma_family.hpp"
TPSRHS_PLASMA_FAMILY(pick_plasma_axi_n3, 2, 3, 3, false)
