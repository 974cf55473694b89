// Plasma kernel family: dim 2, 2 velocity components, 7 species, ambipolar = true.
#include "plasma_family.hpp"
TPSRHS_PLASMA_FAMILY(pick_plasma_2d_n7a, 2, 2, 7, true)
