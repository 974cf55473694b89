// Plasma kernel family: dim 3, 3 velocity components, 6 species, ambipolar = false.
#include "plasma_family.hpp"
TPSRHS_PLASMA_FAMILY(pick_plasma_3d_n6, 3, 3, 6, false)
