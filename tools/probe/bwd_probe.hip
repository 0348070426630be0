// Timing probe for k_conv9_bwd_mfma<24>: build with -DCRNN_PROBE_SKIP_A/B/C to price the phases.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <vector>
#include "../../marl_dmfb_amd/csrc/crnn_mfma_bwd.h"
#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("hip error %s at %d\n", hipGetErrorString(e), __LINE__); return 1; } } while (0)
int main(int argc, char **argv) {
    constexpr int OD = 24;
    using G = crnn_mfma::GeoMB<OD>;
    const long rows = argc > 1 ? atol(argv[1]) : 81920;
    int8_t *obs; float *w1, *b1, *w2, *x, *g, *part;
    CK(hipMalloc(&obs, rows * 245)); CK(hipMalloc(&w1, OD * 27 * 4)); CK(hipMalloc(&b1, OD * 4)); CK(hipMalloc(&w2, OD * OD * 9 * 4));
    CK(hipMalloc(&x, rows * 610 * 4)); CK(hipMalloc(&g, rows * 610 * 4)); CK(hipMalloc(&part, 256 * G::PART * 4));
    std::vector<int8_t> h(rows * 245); for (auto &v : h) v = rand() % 5;
    CK(hipMemcpy(obs, h.data(), h.size(), hipMemcpyHostToDevice));
    std::vector<float> hw(OD * OD * 9); for (auto &v : hw) v = (rand() % 2001 - 1000) * 1e-4f;
    CK(hipMemcpy(w2, hw.data(), hw.size() * 4, hipMemcpyHostToDevice)); CK(hipMemcpy(w1, hw.data(), OD * 27 * 4, hipMemcpyHostToDevice));
    CK(hipMemcpy(b1, hw.data(), OD * 4, hipMemcpyHostToDevice));
    std::vector<float> hx(rows * 610); for (auto &v : hx) v = (rand() % 2001 - 1000) * 1e-3f;
    CK(hipMemcpy(x, hx.data(), hx.size() * 4, hipMemcpyHostToDevice)); CK(hipMemcpy(g, hx.data(), hx.size() * 4, hipMemcpyHostToDevice));
    const size_t lds = G::LDS_FLOATS * 4;
    CK(hipFuncSetAttribute((const void *)crnn_mfma::k_conv9_bwd_mfma<OD>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
    hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
    for (int it = 0; it < 2; ++it)
        hipLaunchKernelGGL((crnn_mfma::k_conv9_bwd_mfma<OD>), dim3(256), dim3(crnn_mfma::kBlockB), lds, 0, obs, 245L, rows, x, 610L, g, 610L, w1, b1, w2, part);
    CK(hipDeviceSynchronize());
    CK(hipEventRecord(e0));
    for (int it = 0; it < 10; ++it)
        hipLaunchKernelGGL((crnn_mfma::k_conv9_bwd_mfma<OD>), dim3(256), dim3(crnn_mfma::kBlockB), lds, 0, obs, 245L, rows, x, 610L, g, 610L, w1, b1, w2, part);
    CK(hipEventRecord(e1)); CK(hipEventSynchronize(e1));
    float ms; CK(hipEventElapsedTime(&ms, e0, e1));
    printf("%s rows %ld: %.1f us/launch\n", argv[0], rows, ms * 1e3 / 10);
    return 0;
}
