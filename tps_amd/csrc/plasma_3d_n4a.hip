// Plasma kernel family: dim 3, 3 velocity components, 4 species, ambipolar = true.
#include "plasma_family.hpp"
TPSRHS_PLASMA_FAMILY(pick_plasma_3d_n4a, 3, 3, 4, true)
