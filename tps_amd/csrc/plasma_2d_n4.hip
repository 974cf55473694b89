// Plasma kernel family: dim 2, 2 velocity components, 4 species, ambipolar = false.
#include "plasma_family.hpp"
TPSRHS_PLASMA_FAMILY(pick_plasma_2d_n4, 2, 2, 4, false)
