// MIXED-precision instantiation of the time-stepper (TRPL_FLAG_MIXED): fp64 state, BDF history, assembly,
// residual norms, PL and likelihood; the tridiagonal CORRECTION solves A delta = b - A c in fp32
// (stepper_impl.hpp: correct_mixed).  FMA contraction on.  Built for L = 128 / 256 / 512 (configs[4]).
#include "stepper_impl.hpp"

namespace trpl {
hipError_t launch_stepper_mixed(const StepArgs &a, hipStream_t stream) { return launch_stepper_mixed_impl(a, stream); }
}  // namespace trpl
