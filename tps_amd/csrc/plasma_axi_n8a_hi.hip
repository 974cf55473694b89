// Plasma kernel family: dim 2, 3 velocity components, 8 species, ambipolar = true; polynomial orders 4 and 5.
#define TPSRHS_PLASMA_HIGH_ORDERS 1
#include "plasma_family.hpp"
TPSRHS_PLASMA_FAMILY(pick_plasma_axi_n8a_hi, 2, 3, 8, true)
