// Plasma kernel family: dim 2, 3 velocity components, 3 species, ambipolar = true.
#include "plasma_family.hpp"
TPSRHS_PLASMA_FAMILY(pick_plasma_axi_n3a, 2, 3, 3, true)
