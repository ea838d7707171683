// STRICT arithmetic instantiation of the time-stepper: this translation unit is compiled with
// -ffp-contract=off so no multiply-add is fused; together with IEEE divides and the reference's
// operation order the N/P/E state and every convergence decision are bit-identical to the
// sequentially executed reference (pvSimPCR.py:93-306).  Selected by TRPL_FLAG_STRICT.
#include "stepper_impl.hpp"

namespace trpl {
hipError_t launch_stepper_strict(const StepArgs &a, hipStream_t stream) { return launch_stepper<true>(a, stream); }
}  // namespace trpl
