// Plasma kernel family: dim 2, 2 velocity components, 5 species, ambipolar = false.
#include "plasma_family.hpp"
TPSRHS_PLASMA_FAMILY(pick_plasma_2d_n5, 2, 2, 5, false)
