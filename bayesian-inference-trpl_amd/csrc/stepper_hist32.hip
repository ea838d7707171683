// HIST32 instantiation of the one-system time-stepper (TRPL_FLAG_HIST32): fp64 state, assembly, solves, residual norms,
// PL and likelihood; the BDF history in difference form with the older differences stored in fp32 (stepper_impl.hpp,
// comment at stepper_kernel).  FMA contraction on.  Built for L = 256 / 512 (the grids whose history pins the occupancy).
#include "stepper_impl.hpp"

namespace trpl {

hipError_t launch_stepper_hist32(const StepArgs &a, hipStream_t stream)
{
    const int64_t nsys = a.S * a.C;
    if (nsys <= 0) return hipSuccess;
    if (a.n_snap > 0 || a.resN != nullptr || a.bundle > 1) return hipErrorInvalidValue;
    dim3 grid((unsigned)nsys), block(64);
    switch (a.L) {
    case 256: hipLaunchKernelGGL((stepper_kernel<256, false, false, false, false, true>), grid, block, 0, stream, a); break;
    case 512: hipLaunchKernelGGL((stepper_kernel<512, false, false, false, false, true>), grid, block, 0, stream, a); break;
    default: return hipErrorInvalidValue;
    }
    return hipGetLastError();
}

}  // namespace trpl
