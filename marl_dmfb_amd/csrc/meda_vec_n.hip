// meda_vec_n.hip -- instantiates the MEDA transition and reset kernels for ONE droplet count
// (compile with -DMEDA_TU_N=<n>); see meda_kernels.h.
#include "meda_kernels.h"

#ifndef MEDA_TU_N
#error "compile with -DMEDA_TU_N=<droplet count>"
#endif

namespace medak {

template <>
hipError_t launch_meda_step_n<MEDA_TU_N>(const MCfg &c, const MPtrs &p, const MStepArgs &a, hipStream_t s) {
    (void)hipGetLastError();  // drop stale errors left by other users of the runtime
    hipLaunchKernelGGL((k_meda_step<MEDA_TU_N>), dim3((c.E + kBlock - 1) / kBlock), dim3(kBlock), 0, s, c, p, a);
    return hipGetLastError();
}

template <>
hipError_t launch_meda_reset_n<MEDA_TU_N>(const MCfg &c, const MPtrs &p, const uint8_t *mask, int mode, hipStream_t s) {
    (void)hipGetLastError();
    hipLaunchKernelGGL((k_meda_reset<MEDA_TU_N>), dim3((c.E + kBlock - 1) / kBlock), dim3(kBlock), 0, s, c, p, mask, mode);
    return hipGetLastError();
}

}  // namespace medak
