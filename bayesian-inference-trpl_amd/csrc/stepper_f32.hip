// fp32-state instantiation of the time-stepper (TRPL_FLAG_FP32): see stepper_f32_impl.hpp.
#include "stepper_f32_impl.hpp"

namespace trpl {
hipError_t launch_stepper_f32(const StepArgs &a, hipStream_t stream) { return launch_stepper_f32_impl(a, stream); }
}  // namespace trpl
