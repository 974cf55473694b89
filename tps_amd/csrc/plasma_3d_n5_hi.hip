// Plasma kernel family: dim 3, 3 velocity components, 5 species, ambipolar = false; polynomial orders 4 and 5.
#define TPSRHS_PLASMA_HIGH_ORDERS 1
#include "plasma_family.hpp"
TPSRHS_PLASMA_FAMILY(pick_plasma_3d_n5_hi, 3, 3, 5, false)
