// FAST arithmetic, two L = 128 systems per wavefront (stepper_pair_impl.hpp): FMA contraction on.
#include "stepper_pair_impl.hpp"

#ifndef TRPL_PAIR_XM
#define TRPL_PAIR_XM 1            // exchange mode of the paired kernel's solver (pcr.hpp: pcr64_levels), see below
#endif

namespace trpl {
// Only the isolated variant is shipped: without isolation a system's last bits depend on its partner (the
// paired reciprocal of update_field2) and a NaN could cross the seam.  Since round 4 the isolation is optimistic
// (stepper_pair_impl.hpp): the always-voiding selects it replaced cost 2.3 % (measured).
hipError_t launch_stepper_pair(const StepArgs &a, hipStream_t stream)
{
    // XM = 1: PCR strides 2..8 and the pair step on ds_swizzle rotates, stride-1 fetches on DPP.  Measured
    // on one box (system-timesteps/s, default bench): XM 0 (LDS-staged) 3.436e8, 1: 3.535e8, 3: 3.535e8,
    // 7: 3.50e8.
    return launch_stepper_pair_t<true, TRPL_PAIR_XM>(a, stream);
}
}  // namespace trpl
