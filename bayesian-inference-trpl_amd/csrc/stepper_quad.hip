// FAST arithmetic, four L = 128 systems per wavefront (stepper_quad_impl.hpp): FMA contraction on (see the Makefile).
#include "stepper_quad_impl.hpp"

namespace trpl {
hipError_t launch_stepper_quad(const StepArgs &a, hipStream_t stream) { return launch_stepper_quad_t(a, stream); }
}  // namespace trpl
