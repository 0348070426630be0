// Timing probe for the conv backward (k_conv9_bwd<24>): -DCRNN_PROBE_SKIP_P1/P2/P3 price the phases.
#include "../../marl_dmfb_amd/csrc/crnn_ops.hip"
#include <vector>
#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("hip error %s at %d\n", hipGetErrorString(e), __LINE__); return 1; } } while (0)
int main(int argc, char **argv) {
    constexpr int OD = 24;
    const long rows = argc > 1 ? atol(argv[1]) : 81920;
    int8_t *obs; float *w2, *x, *g, *part, *grads, *w1, *b1;
    const int plen = crnn_conv9_backward_parts(OD);
    CK(hipMalloc(&w1, OD * 27 * 4)); CK(hipMalloc(&b1, OD * 4)); CK(hipMemset(w1, 0, OD * 27 * 4)); CK(hipMemset(b1, 0, OD * 4));
    CK(hipMalloc(&obs, rows * 245)); CK(hipMalloc(&w2, OD * OD * 9 * 4));
    CK(hipMalloc(&x, rows * 610 * 4)); CK(hipMalloc(&g, rows * 610 * 4)); CK(hipMalloc(&part, 256 * (size_t)plen * 4)); CK(hipMalloc(&grads, 8192 * 4));
    std::vector<float> hx(rows * 610); for (auto &v : hx) v = (rand() % 2001 - 1000) * 1e-3f;
    CK(hipMemcpy(x, hx.data(), hx.size() * 4, hipMemcpyHostToDevice)); CK(hipMemcpy(g, hx.data(), hx.size() * 4, hipMemcpyHostToDevice));
    CK(hipMemset(obs, 1, rows * 245)); CK(hipMemset(w2, 0, OD * OD * 9 * 4));
    hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
    for (int it = 0; it < 2; ++it) if (crnn_conv9_backward(obs, 245, rows, x, 610, g, 610, w1, b1, w2, OD, part, 256, grads, nullptr)) return 2;
    CK(hipDeviceSynchronize());
    CK(hipEventRecord(e0));
    for (int it = 0; it < 10; ++it) crnn_conv9_backward(obs, 245, rows, x, 610, g, 610, w1, b1, w2, OD, part, 256, grads, nullptr);
    CK(hipEventRecord(e1)); CK(hipEventSynchronize(e1));
    float ms; CK(hipEventElapsedTime(&ms, e0, e1));
    printf("%s rows %ld: %.1f us/launch (kernel + reduce)\n", argv[0], rows, ms * 1e3 / 10);
    return 0;
}
