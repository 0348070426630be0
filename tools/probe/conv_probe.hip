// Standalone timing probe for k_conv9_mfma<24>: build variants with -DCRNN_PROBE_SKIP_{CONV1,CONV2,OUT}, -DCRNN_PROBE_NO_GATHER,
// -DCRNN_PROBE_NO_EPI to see which phase costs what; -DCRNN_PROBE_TS prints per-wave phase stamps of one row block.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <vector>
#include <algorithm>
#include "../../marl_dmfb_amd/csrc/crnn_mfma.h"
#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("hip error %s at %d\n", hipGetErrorString(e), __LINE__); return 1; } } while (0)
int main(int argc, char **argv) {
    constexpr int OD = 24;
#ifdef CRNN_PROBE_NW4   // two 4-wave workgroups per CU, 8 rows each
    constexpr int RBV = 8, NW = 4, PER_CU = 2;
#else
    constexpr int RBV = 0, NW = 8, PER_CU = 1;
#endif
    using G = crnn_mfma::GeoM<OD, RBV, NW>;
    const long rows = argc > 1 ? atol(argv[1]) : 81920;
    int8_t *obs; float *w1, *b1, *w2, *b2, *out;
    CK(hipMalloc(&obs, rows * 245)); CK(hipMalloc(&w1, OD * 27 * 4)); CK(hipMalloc(&b1, OD * 4));
    CK(hipMalloc(&w2, OD * OD * 9 * 4)); CK(hipMalloc(&b2, OD * 4)); CK(hipMalloc(&out, rows * 600 * 4));
    std::vector<int8_t> h(rows * 245); for (auto &v : h) v = rand() % 5;
    CK(hipMemcpy(obs, h.data(), h.size(), hipMemcpyHostToDevice));
    std::vector<float> hw(OD * OD * 9); for (auto &v : hw) v = (rand() % 2001 - 1000) * 1e-4f;
    CK(hipMemcpy(w2, hw.data(), hw.size() * 4, hipMemcpyHostToDevice));
    CK(hipMemcpy(w1, hw.data(), OD * 27 * 4, hipMemcpyHostToDevice));
    CK(hipMemcpy(b1, hw.data(), OD * 4, hipMemcpyHostToDevice)); CK(hipMemcpy(b2, hw.data(), OD * 4, hipMemcpyHostToDevice));
    const size_t lds = G::LDS_FLOATS * 4;
    CK(hipFuncSetAttribute((const void *)crnn_mfma::k_conv9_mfma<OD, RBV, NW>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
    const long nb = (rows + G::RB - 1) / G::RB;
    const int grid = nb < 256 * PER_CU ? nb : 256 * PER_CU;
    hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
    for (int it = 0; it < 3; ++it)
        hipLaunchKernelGGL((crnn_mfma::k_conv9_mfma<OD, RBV, NW>), dim3(grid), dim3(G::BLOCK), lds, 0, obs, 245L, rows, w1, b1, w2, b2, out, 600L, 0,
                           (const int8_t *)nullptr, 0, (const float *)nullptr, (const float *)nullptr, (const int32_t *)nullptr, (const int32_t *)nullptr, 1);
    CK(hipDeviceSynchronize());
    CK(hipEventRecord(e0));
    for (int it = 0; it < 20; ++it)
        hipLaunchKernelGGL((crnn_mfma::k_conv9_mfma<OD, RBV, NW>), dim3(grid), dim3(G::BLOCK), lds, 0, obs, 245L, rows, w1, b1, w2, b2, out, 600L, 0,
                           (const int8_t *)nullptr, 0, (const float *)nullptr, (const float *)nullptr, (const int32_t *)nullptr, (const int32_t *)nullptr, 1);
    CK(hipEventRecord(e1)); CK(hipEventSynchronize(e1));
    float ms; CK(hipEventElapsedTime(&ms, e0, e1));
    printf("%s rows %ld: %.1f us/launch\n", argv[0], rows, ms * 1e3 / 20);
#ifdef CRNN_PROBE_TS
    {   // phase stamps of the second row block of workgroups 0 and 100: counter ticks relative to the workgroup's earliest loop top
        unsigned long long *ts; CK(hipMalloc(&ts, (size_t)grid * 8 * 16 * 8)); CK(hipMemset(ts, 0, (size_t)grid * 8 * 16 * 8));
        CK(hipMemcpyToSymbol(HIP_SYMBOL(crnn_mfma::g_crnn_ts), &ts, sizeof(ts)));
        hipLaunchKernelGGL((crnn_mfma::k_conv9_mfma<OD, RBV, NW>), dim3(grid), dim3(G::BLOCK), lds, 0, obs, 245L, rows, w1, b1, w2, b2, out, 600L, 0,
                           (const int8_t *)nullptr, 0, (const float *)nullptr, (const float *)nullptr, (const int32_t *)nullptr, (const int32_t *)nullptr, 1);
        CK(hipDeviceSynchronize());
        std::vector<unsigned long long> hts((size_t)grid * 8 * 16); CK(hipMemcpy(hts.data(), ts, hts.size() * 8, hipMemcpyDeviceToHost));
        const char *names[11] = {"top", "fetch issued", "conv1 done", "B2 passed", "parked", "conv2 done", "B3 passed", "streamed out", "conv2 pass 1", "conv2 pass 2", "conv2 pass 3"};
        for (int wg : {0, 100}) {
            if (wg >= grid) continue;
            unsigned long long t0 = ~0ull; for (int w = 0; w < 8; ++w) t0 = std::min(t0, hts[((size_t)wg * 8 + w) * 16]);
            printf("workgroup %d, waves 0..7 (wave w runs on SIMD w & 3)\n", wg);
            for (int k = 0; k < 11; ++k) {
                printf("  %-13s", names[k]);
                for (int w = 0; w < 8; ++w) printf(" %7lld", (long long)(hts[((size_t)wg * 8 + w) * 16 + k] - t0));
                printf("\n");
            }
        }
    }
#endif
    return 0;
}
