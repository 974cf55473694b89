// development probe: only the kernels of the metric's workload (3-D p = 3, ternary ambipolar single-temperature argon,
// argon-minimal transport), for quick register / ISA experiments:
//   hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC -c tps_amd/csrc/_probe/headline.hip -o /tmp/headline.o
#include "../plasma_family.hpp"
extern "C" void headline_probe(tpsrhs_operator *op) { op->launch = &launch_all<3, 3, PlasmaPhys<3, 3, 3, true, false, TRANSPORT_ARGON_MINIMAL>>; }
