// MIXED-precision instantiation of the time-stepper (TRPL_FLAG_MIXED): fp64 state, BDF history, assembly,
// residual norms, PL and likelihood; the tridiagonal CORRECTION solves A delta = b - A c in fp32
// (stepper_impl.hpp: correct_mixed).  FMA contraction on.  Built for L = 128 / 256 / 512 (configs[4]).
#include "stepper_impl.hpp"

namespace trpl {

// MIXED: fp64 state, history, assembly and residuals; fp32 correction solves (L >= 128, FAST arithmetic)
static hipError_t launch_stepper_mixed_impl(const StepArgs &a, hipStream_t stream)
{
    const int64_t nsys = a.S * a.C;
    if (nsys <= 0) return hipSuccess;
    dim3 grid((unsigned)nsys), block(64);
    const bool snap = a.n_snap > 0 || a.resN != nullptr;
    switch (a.L) {
#define TRPL_CASE(LL) \
    case LL: \
        if (snap) hipLaunchKernelGGL((stepper_kernel<LL, false, true, true>), grid, block, 0, stream, a); \
        else      hipLaunchKernelGGL((stepper_kernel<LL, false, false, true>), grid, block, 0, stream, a); \
        break;
        TRPL_CASE(128) TRPL_CASE(256) TRPL_CASE(512)
#undef TRPL_CASE
    default: return hipErrorInvalidValue;
    }
    return hipGetLastError();
}

hipError_t launch_stepper_mixed(const StepArgs &a, hipStream_t stream) { return launch_stepper_mixed_impl(a, stream); }
}  // namespace trpl
