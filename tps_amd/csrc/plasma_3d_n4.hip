// Plasma kernel family: dim 3, 3 velocity components, 4 species, ambipolar = false.
#include "plasma_family.hpp"
TPSRHS_PLASMA_FAMILY(pick_plasma_3d_n4, 3, 3, 4, false)
