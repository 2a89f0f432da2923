// ucf_peak.hip -- measured fp64 vector-FMA rate of the device: the denominator of the
// roofline fraction for this path (the binding roof is fp64 VALU,
// not HBM and not MFMA).
#include <hip/hip_runtime.h>
#include "../../include/ucf.h"

namespace {
constexpr int ITER = 4096;
__global__ void __launch_bounds__(256) fma_chain(double* out, double seed)
{
    double a0 = seed + threadIdx.x, a1 = a0 + 1, a2 = a0 + 2, a3 = a0 + 3, a4 = a0 + 4, a5 = a0 + 5, a6 = a0 + 6, a7 = a0 + 7;
    const double m = 0.999999, c = 1e-9;
#pragma unroll 8
    for (int i = 0; i < ITER; i++) {
        a0 = __builtin_fma(a0, m, c); a1 = __builtin_fma(a1, m, c); a2 = __builtin_fma(a2, m, c); a3 = __builtin_fma(a3, m, c);
        a4 = __builtin_fma(a4, m, c); a5 = __builtin_fma(a5, m, c); a6 = __builtin_fma(a6, m, c); a7 = __builtin_fma(a7, m, c);
    }
    out[blockIdx.x * blockDim.x + threadIdx.x] = ((a0 + a1) + (a2 + a3)) + ((a4 + a5) + (a6 + a7));
}
}  // namespace

extern "C" int ucf_fp64_fma_peak(double* tflops)
{
    if (!tflops) return UCF_ERR_BAD_ARGUMENT;
    int ndev = 0;
    if (hipGetDeviceCount(&ndev) != hipSuccess || ndev <= 0) return UCF_ERR_NO_DEVICE;
    const int blocks = 256 * 16, threads = 256;
    double* d = nullptr;
    if (hipMalloc((void**)&d, sizeof(double) * blocks * threads) != hipSuccess) return UCF_ERR_NOMEM;
    hipEvent_t e0, e1;
    (void)hipEventCreate(&e0); (void)hipEventCreate(&e1);
    hipLaunchKernelGGL(fma_chain, dim3(blocks), dim3(threads), 0, 0, d, 1.0);      // warm-up
    double best = 0.0;
    for (int rep = 0; rep < 5; rep++) {
        (void)hipEventRecord(e0, 0);
        hipLaunchKernelGGL(fma_chain, dim3(blocks), dim3(threads), 0, 0, d, 1.0 + rep);
        (void)hipEventRecord(e1, 0);
        (void)hipEventSynchronize(e1);
        float ms = 0.f;
        (void)hipEventElapsedTime(&ms, e0, e1);
        const double flops = 2.0 * 8.0 * ITER * (double)blocks * threads;
        const double tf = flops / (ms * 1e-3) * 1e-12;
        if (tf > best) best = tf;
    }
    (void)hipEventDestroy(e0); (void)hipEventDestroy(e1);
    (void)hipFree(d);
    *tflops = best;
    return hipGetLastError() == hipSuccess ? UCF_OK : UCF_ERR_HIP;
}
