// development probe: the kernels of bench.py's cfg5 (axisymmetric p = 3, ternary ambipolar two-temperature, argon mixture transport)
#include "../plasma_family.hpp"
extern "C" void cfg5mix_probe(tpsrhs_operator *op) { op->launch = &launch_all<2, 3, PlasmaPhys<2, 3, 3, true, true, TRANSPORT_ARGON_MIXTURE>>; }
