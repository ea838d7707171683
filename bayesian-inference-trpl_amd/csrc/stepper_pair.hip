// FAST arithmetic, two L = 128 systems per wavefront (stepper_pair_impl.hpp): FMA contraction on.
#include "stepper_pair_impl.hpp"

namespace trpl {
// Only the isolated variant is shipped: without it a system's last bits depend on its partner (the
// paired reciprocal of update_field2) and a NaN could cross the seam; it costs 2.7 % (measured).
hipError_t launch_stepper_pair(const StepArgs &a, hipStream_t stream) { return launch_stepper_pair_t<true>(a, stream); }
}  // namespace trpl
