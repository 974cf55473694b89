// Plasma kernel family: dim 2, 2 velocity components, 6 species, ambipolar = true.
#include "plasma_family.hpp"
TPSRHS_PLASMA_FAMILY(pick_plasma_2d_n6a, 2, 2, 6, true)
