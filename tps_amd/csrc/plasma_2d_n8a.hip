// Plasma kernel family: dim 2, 2 velocity components, 8 species, ambipolar = true.
#include "plasma_family.hpp"
TPSRHS_PLASMA_FAMILY(pick_plasma_2d_n8a, 2, 2, 8, true)
