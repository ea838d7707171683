// FAST arithmetic, two L = 128 systems per wavefront (stepper_pair_impl.hpp): FMA contraction on.
#include "stepper_pair_impl.hpp"

namespace trpl {
// Only the isolated variant is shipped: without it a system's last bits depend on its partner (the
// paired reciprocal of update_field2) and a NaN could cross the seam; it costs 2.7 % (measured).
hipError_t launch_stepper_pair(const StepArgs &a, hipStream_t stream)
{
    // XM = 1: PCR strides 2..8 and the pair step on ds_swizzle rotates, stride-1 fetches on DPP.  Measured
    // on one box (system-timesteps/s, default bench): XM 0 (LDS-staged) 3.436e8, 1: 3.535e8, 3: 3.535e8,
    // 7: 3.50e8.
    return launch_stepper_pair_t<true, 1>(a, stream);
}
}  // namespace trpl
