// Plasma kernel family: dim 3, 3 velocity components, 7 species, ambipolar = true.
#include "plasma_family.hpp"
TPSRHS_PLASMA_FAMILY(pick_plasma_3d_n7a, 3, 3, 7, true)
