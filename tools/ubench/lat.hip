// Microbenchmarks of the gfx950 VALU issue model for fp64: dependent-chain latency, ILP, mixing with
// SALU / v_mov, waves per SIMD.  Prints cycles per VALU instruction per wave (s_memtime).
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>

template <int K, int MODE>
__global__ void chain(double* out, long long* cyc, int n, double a, double b)
{
    double x[K];
    for (int k = 0; k < K; k++) x[k] = threadIdx.x * 1e-3 + k;
    __syncthreads();
    long long t0 = clock64();
    for (int i = 0; i < n; i++) {
#pragma unroll
        for (int u = 0; u < 8; u++) {
#pragma unroll
            for (int k = 0; k < K; k++) {
                if (MODE == 0) x[k] = __builtin_fma(x[k], a, b);                       // v_fma_f64 v,v,s,s? (a,b in SGPR)
                if (MODE == 1) asm volatile("v_fma_f64 %0, %0, %1, %2" : "+v"(x[k]) : "v"(a), "v"(b));
                if (MODE == 2) { asm volatile("v_fma_f64 %0, %0, %1, %2\n s_mov_b32 s20, 0x12345\n s_mov_b32 s21, 0x54321" : "+v"(x[k]) : "v"(a), "v"(b) : "s20", "s21"); }
                if (MODE == 3) { int t; asm volatile("v_fma_f64 %0, %0, %2, %3\n v_mov_b32 %1, 0x12345" : "+v"(x[k]), "=v"(t) : "v"(a), "v"(b)); }
                if (MODE == 4) asm volatile("v_mul_f64 %0, %0, %1" : "+v"(x[k]) : "v"(a));
                if (MODE == 5) asm volatile("v_add_f64 %0, %0, %1" : "+v"(x[k]) : "v"(a));
                if (MODE == 6) { float f = (float)x[k]; asm volatile("v_fma_f32 %0, %0, %0, %0" : "+v"(f)); x[k] = f; }
            }
        }
    }
    long long t1 = clock64();
    double s = 0;
    for (int k = 0; k < K; k++) s += x[k];
    out[blockIdx.x * blockDim.x + threadIdx.x] = s;
    if (threadIdx.x == 0) cyc[blockIdx.x] = t1 - t0;
}

template <int K, int MODE>
void run(const char* name, int threads, int blocks)
{
    double* out; long long* cyc;
    hipMalloc(&out, sizeof(double) * threads * blocks);
    hipMalloc(&cyc, sizeof(long long) * blocks);
    const int n = 2000;
    chain<K, MODE><<<blocks, threads>>>(out, cyc, 10, 0.999, 1e-3);
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    hipEventRecord(e0);
    chain<K, MODE><<<blocks, threads>>>(out, cyc, n, 0.999, 1e-3);
    hipEventRecord(e1);
    hipDeviceSynchronize();
    float ms; hipEventElapsedTime(&ms, e0, e1);
    std::vector<long long> h(blocks);
    hipMemcpy(h.data(), cyc, sizeof(long long) * blocks, hipMemcpyDeviceToHost);
    double avg = 0; for (auto v : h) avg += v; avg /= blocks;
    const double instr = (double)n * 8 * K;
    printf("%-28s K=%d threads=%4d blocks=%5d: %7.2f memtime-ticks/instr/wave  (%.3f ms, %.2f ns/instr/wave)\n", name, K, threads, blocks, avg / instr,
           ms, ms * 1e6 / instr);
    hipFree(out); hipFree(cyc);
}

int main()
{
    // one wave alone: dependent latency and ILP
    run<1, 1>("fma dep", 64, 1);
    run<2, 1>("fma 2 chains", 64, 1);
    run<3, 1>("fma 3 chains", 64, 1);
    run<4, 1>("fma 4 chains", 64, 1);
    run<8, 1>("fma 8 chains", 64, 1);
    run<1, 4>("mul dep", 64, 1);
    run<1, 5>("add dep", 64, 1);
    run<1, 2>("fma dep + 2 s_mov", 64, 1);
    run<1, 3>("fma dep + v_mov", 64, 1);
    run<2, 3>("fma 2ch + v_mov", 64, 1);
    run<4, 3>("fma 4ch + v_mov", 64, 1);
    run<4, 2>("fma 4ch + 2 s_mov", 64, 1);
    // waves per SIMD: one workgroup of 256*W threads on one CU
    run<1, 1>("fma dep 1w/simd", 256, 1);
    run<1, 1>("fma dep 2w/simd", 512, 1);
    run<1, 1>("fma dep 4w/simd", 1024, 1);
    run<2, 1>("fma 2ch 2w/simd", 512, 1);
    run<4, 1>("fma 4ch 2w/simd", 512, 1);
    run<1, 2>("fma dep+2smov 2w/simd", 512, 1);
    run<4, 2>("fma 4ch+2smov 2w/simd", 512, 1);
    run<1, 3>("fma dep+vmov 2w/simd", 512, 1);
    // whole chip
    run<4, 1>("fma 4ch chip 2w/simd", 512, 256);
    run<4, 1>("fma 4ch chip 2w/simd x4", 512, 1024);
    return 0;
}
