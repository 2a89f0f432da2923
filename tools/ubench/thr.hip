// Throughput cost of single instruction kinds on gfx950 with the SIMDs full of waves (the regime integrate_kernel runs in):
// 4 independent chains per wave, W waves per SIMD on every CU; prints SIMD cycles per wave-instruction (wall time x clock
// from s_memtime over the same launch).  Answers "what does one more v_rcp_f64 / v_and_b32 / ds_read_b128 cost".
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>

#define REP 8
template <int MODE>
__global__ void __launch_bounds__(256) body(double* out, long long* cyc, int n, double a, double b, const double2* tab)
{
    __shared__ double2 lds[384];
    for (int k = threadIdx.x; k < 384; k += blockDim.x) lds[k] = tab[k];
    __syncthreads();
    double x[4];
    for (int k = 0; k < 4; k++) x[k] = 1.0 + threadIdx.x * 1e-3 + k * 0.25;
    int iv = threadIdx.x * 2654435761u;
    long long t0 = clock64();
    for (int i = 0; i < n; i++) {
#pragma unroll
        for (int u = 0; u < REP; u++) {
#pragma unroll
            for (int k = 0; k < 4; k++) {
                if (MODE == 0) asm volatile("v_fma_f64 %0, %0, %1, %2" : "+v"(x[k]) : "v"(a), "v"(b));
                if (MODE == 1) asm volatile("v_rcp_f64 %0, %0" : "+v"(x[k]));
                if (MODE == 2) asm volatile("v_rsq_f64 %0, %0" : "+v"(x[k]));
                if (MODE == 3) asm volatile("v_ldexp_f64 %0, %0, %1" : "+v"(x[k]) : "v"(k & 1 ? 1 : -1));
                if (MODE == 4) { int t; asm volatile("v_and_b32 %0, 0x7f, %1" : "=v"(t) : "v"(iv)); iv += t; }     // and + add: 2 instr
                if (MODE == 5) { asm volatile("v_add_u32 %0, %0, %1" : "+v"(iv) : "v"(k)); }
                if (MODE == 6) { unsigned long long m; asm volatile("v_cmp_lt_f64 %0, %1, %2" : "=s"(m) : "v"(x[k]), "v"(a)); asm volatile("" :: "s"(m)); }
                if (MODE == 7) { asm volatile("v_cndmask_b32 %0, %0, %1, vcc" : "+v"(iv) : "v"(k) : ); }
                if (MODE == 8) asm volatile("v_mul_f64 %0, %0, %1" : "+v"(x[k]) : "v"(a));
                if (MODE == 9) asm volatile("v_add_f64 %0, %0, %1" : "+v"(x[k]) : "v"(b));
                if (MODE == 10) { asm volatile("v_lshl_add_u32 %0, %0, 4, %1" : "+v"(iv) : "v"(k)); }
                if (MODE == 11) { double2 v = lds[(iv >> 8) & 255]; iv += (int)__double2loint(v.x) + 7; }          // table read (random index) + add + shift/and
                if (MODE == 12) { double2 v = lds[(threadIdx.x + k) & 255]; iv += (int)__double2loint(v.x) + 7; }   // table read, conflict-free
                if (MODE == 13) asm volatile("v_mov_b32 %0, %1" : "=v"(iv) : "v"(iv));
                if (MODE == 14) asm volatile("v_fma_f64 %0, %0, %1, %2" : "+v"(x[k]) : "s"(a), "v"(b));
                if (MODE == 15) asm volatile("v_mul_f64 %0, %0, 0.5" : "+v"(x[k]));
            }
        }
    }
    long long t1 = clock64();
    double s = iv;
    for (int k = 0; k < 4; k++) s += x[k];
    out[blockIdx.x * blockDim.x + threadIdx.x] = s;
    if (threadIdx.x == 0) cyc[blockIdx.x] = t1 - t0;
}

template <int MODE>
void run(const char* name, int wps, const double2* tab)
{
    const int threads = 256, blocks = 256 * wps;      // wps workgroups of 4 waves per CU = wps waves per SIMD
    double* out; long long* cyc;
    hipMalloc(&out, sizeof(double) * threads * blocks);
    hipMalloc(&cyc, sizeof(long long) * blocks);
    const int n = 4000;
    body<MODE><<<blocks, threads>>>(out, cyc, 10, 0.999, 1e-3, tab);
    hipDeviceSynchronize();
    body<MODE><<<blocks, threads>>>(out, cyc, n, 0.999, 1e-3, tab);
    hipDeviceSynchronize();
    std::vector<long long> h(blocks);
    hipMemcpy(h.data(), cyc, sizeof(long long) * blocks, hipMemcpyDeviceToHost);
    double avg = 0; for (auto v : h) avg += v; avg /= blocks;
    const double instr = (double)n * REP * 4;
    // a wave's own span / its instructions / waves that shared the SIMD = SIMD cycles per wave-instruction
    printf("%-34s %d waves/SIMD: %6.2f cycles per wave-instruction (one wave sees %6.2f)\n", name, wps, avg / instr / wps, avg / instr);
    hipFree(out); hipFree(cyc);
}

int main()
{
    std::vector<double2> t(384);
    for (int i = 0; i < 384; i++) t[i] = make_double2(1.0 + i * 1e-3, 0.5);
    double2* tab; hipMalloc(&tab, sizeof(double2) * 384); hipMemcpy(tab, t.data(), sizeof(double2) * 384, hipMemcpyHostToDevice);
    for (int wps : {1, 4, 6}) {
        run<0>("v_fma_f64 (v,v,v)", wps, tab);
        run<14>("v_fma_f64 (v,s,v)", wps, tab);
        run<8>("v_mul_f64", wps, tab);
        run<15>("v_mul_f64 by 0.5", wps, tab);
        run<9>("v_add_f64", wps, tab);
        run<1>("v_rcp_f64", wps, tab);
        run<2>("v_rsq_f64", wps, tab);
        run<3>("v_ldexp_f64", wps, tab);
        run<6>("v_cmp_lt_f64 -> sgpr", wps, tab);
        run<5>("v_add_u32", wps, tab);
        run<4>("v_and_b32 + v_add_u32 (2 instr)", wps, tab);
        run<7>("v_cndmask_b32", wps, tab);
        run<10>("v_lshl_add_u32", wps, tab);
        run<13>("v_mov_b32", wps, tab);
        run<11>("ds_read_b128 random + ~4 int ops", wps, tab);
        run<12>("ds_read_b128 linear + ~3 int ops", wps, tab);
    }
    return 0;
}
