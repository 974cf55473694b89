// Plasma kernel family: dim 3, 3 velocity components, 8 species, ambipolar = false.
#include "plasma_family.hpp"
TPSRHS_PLASMA_FAMILY(pick_plasma_3d_n8, 3, 3, 8, false)
