// FAST arithmetic instantiations (default): FMA contraction on.
#include "stepper_impl.hpp"
#include "pcr_batched_impl.hpp"

namespace trpl {
hipError_t launch_stepper_fast(const StepArgs &a, hipStream_t stream) { return launch_stepper<false>(a, stream); }
hipError_t launch_pcr_batched_fast(const void *ld, const void *d, const void *ud, const void *b, void *x,
                                   int64_t S, int L, int elem_bytes, hipStream_t stream)
{
    return elem_bytes == 8 ? launch_pcr_batched_t<double, false>(ld, d, ud, b, x, S, L, stream)
                           : launch_pcr_batched_t<float, false>(ld, d, ud, b, x, S, L, stream);
}
}  // namespace trpl
