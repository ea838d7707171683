// FAST arithmetic, two L = 128 systems per wavefront (stepper_pair_impl.hpp): FMA contraction on.
#include "stepper_pair_impl.hpp"

namespace trpl {
// Only the isolated variant is shipped: without isolation a system's last bits depend on its partner (the
// paired reciprocal of update_field2) and a NaN could cross the seam.  Since round 4 the isolation is optimistic
// (stepper_pair_impl.hpp): the always-voiding selects it replaced cost 2.3 % (measured).
hipError_t launch_stepper_pair(const StepArgs &a, hipStream_t stream)
{
    // The solver's exchanges: PCR strides 2..8 and the pair step on ds_swizzle rotates, stride-1 fetches on DPP.  Measured in
    // round 1 on one box (system-timesteps/s): LDS-staged 3.436e8, this 3.535e8, stride 1 on swizzles too 3.535e8 / 3.50e8.
    return launch_stepper_pair_t<true>(a, stream);
}
}  // namespace trpl
