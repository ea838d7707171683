// STRICT arithmetic instantiations: this translation unit is compiled with -ffp-contract=off so
// no multiply-add is fused; together with IEEE divides and the reference's operation order the
// N/P/E state, every convergence decision (stepper) and the PCR solution (batched solve) are
// bit-identical to the sequentially executed reference (pvSimPCR.py:42-306).
// Selected by TRPL_FLAG_STRICT.
#include "stepper_impl.hpp"
#include "pcr_batched_impl.hpp"

namespace trpl {
hipError_t launch_stepper_strict(const StepArgs &a, hipStream_t stream) { return launch_stepper<true>(a, stream); }
hipError_t launch_pcr_batched_strict(const void *ld, const void *d, const void *ud, const void *b, void *x,
                                     int64_t S, int L, int elem_bytes, hipStream_t stream)
{
    return elem_bytes == 8 ? launch_pcr_batched_t<double, true>(ld, d, ud, b, x, S, L, stream)
                           : launch_pcr_batched_t<float, true>(ld, d, ud, b, x, S, L, stream);
}
}  // namespace trpl
