// Plasma kernel family: dim 2, 2 velocity components, 5 species, ambipolar = true; polynomial orders 4 and 5.
#define TPSRHS_PLASMA_HIGH_ORDERS 1
#include "plasma_family.hpp"
TPSRHS_PLASMA_FAMILY(pick_plasma_2d_n5a_hi, 2, 2, 5, true)
