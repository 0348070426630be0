// dmfb_vec_n.hip -- instantiates the transition and reset kernels for ONE droplet count
// (compile with -DDMFB_TU_N=<n>); see dmfb_kernels.h.
#include <hip/hip_ext.h>

#include "dmfb_kernels.h"
#include "dmfb_step_lanes.h"

#ifndef DMFB_TU_N
#error "compile with -DDMFB_TU_N=<droplet count>"
#endif

namespace dmfbk {

template <>
hipError_t launch_step_n<DMFB_TU_N>(const DevCfg &c, const DevPtrs &p, const StepArgs &a, int grid, size_t lds,
                                    hipStream_t s) {
    (void)hipGetLastError();  // drop stale errors left by other users of the runtime
    const bool obs = a.out.d_obs != nullptr;
    if (p.health) {
        if (obs) hipLaunchKernelGGL((k_step<DMFB_TU_N, true, true>), dim3(grid), dim3(kBlock), lds, s, c, p, a);
        else hipLaunchKernelGGL((k_step<DMFB_TU_N, true, false>), dim3(grid), dim3(kBlock), lds, s, c, p, a);
    } else {
        if (obs) hipLaunchKernelGGL((k_step<DMFB_TU_N, false, true>), dim3(grid), dim3(kBlock), lds, s, c, p, a);
        else hipLaunchKernelGGL((k_step<DMFB_TU_N, false, false>), dim3(grid), dim3(kBlock), lds, s, c, p, a);
    }
    return hipGetLastError();
}

#if DMFB_TU_N >= 8
template <>
hipError_t launch_step_lanes_n<DMFB_TU_N>(const DevCfg &c, const DevPtrs &p, const StepArgs &a, int grid, size_t lds,
                                          hipStream_t s) {
    (void)hipGetLastError();
    if (p.health) hipLaunchKernelGGL((k_step_lanes<DMFB_TU_N, true>), dim3(grid), dim3(kBlock), lds, s, c, p, a);
    else hipLaunchKernelGGL((k_step_lanes<DMFB_TU_N, false>), dim3(grid), dim3(kBlock), lds, s, c, p, a);
    return hipGetLastError();
}
#endif

template <>
hipError_t launch_reset_n<DMFB_TU_N>(const DevCfg &c, const DevPtrs &p, const uint8_t *mask, int mode, int grid,
                                     size_t lds, hipStream_t s) {
    (void)hipGetLastError();
    hipLaunchKernelGGL((k_reset<DMFB_TU_N>), dim3(grid), dim3(kBlock), lds, s, c, p, mask, mode);
    return hipGetLastError();
}

template <>
hipError_t launch_observe_n<DMFB_TU_N>(const DevCfg &c, const DevPtrs &p, const uint8_t *mask, int8_t *obs, int grid,
                                       size_t lds, hipStream_t s, hipEvent_t t0, hipEvent_t t1) {
    (void)hipGetLastError();
    if (t0 && t1)  // the events receive the dispatch's own start/end time stamps (what rocprofv3 --kernel-trace reports)
        hipExtLaunchKernelGGL((k_observe<DMFB_TU_N>), dim3(grid), dim3(kObsBlock), lds, s, t0, t1, 0, c, p, mask, obs);
    else
        hipLaunchKernelGGL((k_observe<DMFB_TU_N>), dim3(grid), dim3(kObsBlock), lds, s, c, p, mask, obs);
    return hipGetLastError();
}

}  // namespace dmfbk
