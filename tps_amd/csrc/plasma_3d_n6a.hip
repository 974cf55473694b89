// Plasma kernel family: dim 3, 3 velocity components, 6 species, ambipolar = true.
#include "plasma_family.hpp"
TPSRHS_PLASMA_FAMILY(pick_plasma_3d_n6a, 3, 3, 6, true)
