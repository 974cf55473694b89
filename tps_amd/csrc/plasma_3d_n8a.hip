// Plasma kernel family: dim 3, 3 velocity components, 8 species, ambipolar = true.
#include "plasma_family.hpp"
TPSRHS_PLASMA_FAMILY(pick_plasma_3d_n8a, 3, 3, 8, true)
