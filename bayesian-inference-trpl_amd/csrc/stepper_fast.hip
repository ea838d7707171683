// FAST arithmetic instantiation of the time-stepper (default): FMA contraction on.
#include "stepper_impl.hpp"

namespace trpl {
hipError_t launch_stepper_fast(const StepArgs &a, hipStream_t stream) { return launch_stepper<false>(a, stream); }
}  // namespace trpl
