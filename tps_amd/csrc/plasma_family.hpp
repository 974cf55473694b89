// Shared by the plasma translation units: one TU = one (dim, nvel, species count, ambipolar) family, so
// that the families compile in parallel.  Each TU defines TPSRHS_PLASMA_FAMILY(name, DIM, NVEL, NSP, AMBI)
// through this header and exports `void name(tpsrhs_operator *, bool two_temperature, int transport)`.
#ifndef TPSRHS_PLASMA_FAMILY_HPP_
#define TPSRHS_PLASMA_FAMILY_HPP_

#include "operator.hpp"
#include "physics_plasma.hpp"

template <int DIM, int NVEL, int NSP, bool AMBI, bool TWOT, int TR>
static void pick_plasma_orders(tpsrhs_operator *op) {
  typedef PlasmaPhys<DIM, NVEL, NSP, AMBI, TWOT, TR> PH;
  if (op->nc) {
    // the Gauss-Lobatto pair (orders 1..3, planar / 3-D): every species count
#ifndef TPSRHS_PLASMA_HIGH_ORDERS
    if constexpr (NVEL == DIM) {
      pick_order_nc<DIM, PH>(op);
      return;
    } else
#endif
    {
      throw Unsupported("Gauss-Lobatto basis + rule: planar 2-D and 3-D, polynomial orders 1..3");
    }
  }
  upload_tables(DIM, op->order);
  op->point_eval = &launch_point_eval<PH>;
  switch (op->order) {
#ifndef TPSRHS_PLASMA_HIGH_ORDERS
    case 1: op->launch = &launch_all<DIM, 1, PH>; break;
    case 2: op->launch = &launch_all<DIM, 2, PH>; break;
    case 3: op->launch = &launch_all<DIM, 3, PH>; break;
#else  // the `_hi` translation units: orders 4 and 5 (MAXDOFS = 216 = hex p=5, src/dataStructures.hpp:41-65)
    case 4: op->launch = &launch_all<DIM, 4, PH>; break;
    case 5: op->launch = &launch_all<DIM, 5, PH>; break;
#endif
    default: throw Unsupported("plasma kernels of this family: polynomial order " + std::to_string(op->order) + " is not built");
  }
}

// which transport models a family instantiates: the ternary collision model needs three species
template <int DIM, int NVEL, int NSP, bool AMBI>
static void pick_plasma_family(tpsrhs_operator *op, bool two_temperature, int transport) {
  if (transport == TRANSPORT_CONSTANT) {
    if (two_temperature)
      pick_plasma_orders<DIM, NVEL, NSP, AMBI, true, TRANSPORT_CONSTANT>(op);
    else
      pick_plasma_orders<DIM, NVEL, NSP, AMBI, false, TRANSPORT_CONSTANT>(op);
  } else if (transport == TRANSPORT_ARGON_MIXTURE) {
    if constexpr (NSP <= 7) {  // the reference's argon mixture transport asserts at most 7 species (src/gas_transport.cpp:905-911)
      if (two_temperature)
        pick_plasma_orders<DIM, NVEL, NSP, AMBI, true, TRANSPORT_ARGON_MIXTURE>(op);
      else
        pick_plasma_orders<DIM, NVEL, NSP, AMBI, false, TRANSPORT_ARGON_MIXTURE>(op);
    } else {
      throw Unsupported("argon_mixture transport supports at most 7 species");
    }
  } else {
    if constexpr (NSP == 3) {
      if (two_temperature)
        pick_plasma_orders<DIM, NVEL, NSP, AMBI, true, TRANSPORT_ARGON_MINIMAL>(op);
      else
        pick_plasma_orders<DIM, NVEL, NSP, AMBI, false, TRANSPORT_ARGON_MINIMAL>(op);
    } else {
      throw Unsupported("argon_minimal transport is the ternary (Ar, Ar.+1, E) model");
    }
  }
}

#if TPSRHS_STAMP
// diagnostic builds (one translation unit compiled with -DTPSRHS_STAMP=1): the phase cycles of the first
// `nblocks` blocks of the last stamped kernel, [nblocks][NSTAMP]
extern "C" int tpsrhs_debug_stamps(unsigned int *out, int nblocks) {
  const size_t bytes = static_cast<size_t>(nblocks) * tpsrhs::NSTAMP * sizeof(unsigned int);
  return hipMemcpyFromSymbol(out, HIP_SYMBOL(tpsrhs::g_stamp), bytes) == hipSuccess ? 0 : 1;
}
#endif

#if TPSRHS_DUMPF
// diagnostic builds (one translation unit compiled with -DTPSRHS_DUMPF=1): what the last k_flux of this unit left
extern "C" int tpsrhs_debug_dumpf(double *out, int n) {
  return hipMemcpyFromSymbol(out, HIP_SYMBOL(tpsrhs::g_dumpf), static_cast<size_t>(n) * sizeof(double)) == hipSuccess ? 0 : 1;
}
#endif

// One family = one shared object, libtpsrhs_<unit>.so, that the core library loads on demand (tpsrhs.hip::load_family).
// The entry point is C: no exception crosses the boundary, a refusal comes back as a status code and a message.
#define TPSRHS_PLASMA_FAMILY(name, DIM, NVEL, NSP, AMBI)                                                              \
  extern "C" int name(tpsrhs_operator *op, int two_temperature, int transport, char *err, int errlen) {               \
    auto say = [&](const char *what) {                                                                                \
      if (err && errlen > 0) {                                                                                        \
        std::strncpy(err, what, static_cast<size_t>(errlen) - 1);                                                     \
        err[errlen - 1] = 0;                                                                                          \
      }                                                                                                               \
    };                                                                                                                \
    try {                                                                                                             \
      pick_plasma_family<DIM, NVEL, NSP, AMBI>(op, two_temperature != 0, transport);                                  \
      return TPSRHS_OK;                                                                                               \
    } catch (const Unsupported &e) {                                                                                  \
      say(e.what());                                                                                                  \
      return TPSRHS_ERR_UNSUPPORTED;                                                                                  \
    } catch (const DeviceError &e) {                                                                                  \
      say(e.what());                                                                                                  \
      return TPSRHS_ERR_DEVICE;                                                                                       \
    } catch (const std::invalid_argument &e) {                                                                        \
      say(e.what());                                                                                                  \
      return TPSRHS_ERR_INVALID_ARGUMENT;                                                                             \
    } catch (const std::exception &e) {                                                                               \
      say(e.what());                                                                                                  \
      return TPSRHS_ERR_DEVICE;                                                                                       \
    }                                                                                                                 \
  }
#endif
