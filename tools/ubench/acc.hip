// accuracy of the gfx950 fp64 seeds: v_rcp_f64, v_rsq_f64 (max relative error over random inputs)
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cmath>
#include <vector>
#include <random>
__global__ void k(const double* x, double* r, double* q, int n)
{
    int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i < n) { r[i] = __builtin_amdgcn_rcp(x[i]); q[i] = __builtin_amdgcn_rsq(x[i]); }
}
int main()
{
    const int n = 1 << 22;
    std::vector<double> x(n), r(n), q(n);
    std::mt19937_64 g(1);
    std::uniform_real_distribution<double> u(-300.0, 300.0), m(1.0, 2.0);
    for (int i = 0; i < n; i++) x[i] = m(g) * std::pow(2.0, std::floor(u(g)));
    double *dx, *dr, *dq;
    (void)hipMalloc(&dx, n * 8); (void)hipMalloc(&dr, n * 8); (void)hipMalloc(&dq, n * 8);
    (void)hipMemcpy(dx, x.data(), n * 8, hipMemcpyHostToDevice);
    k<<<n / 256, 256>>>(dx, dr, dq, n);
    (void)hipMemcpy(r.data(), dr, n * 8, hipMemcpyDeviceToHost);
    (void)hipMemcpy(q.data(), dq, n * 8, hipMemcpyDeviceToHost);
    long double er = 0, eq = 0;
    for (int i = 0; i < n; i++) {
        long double tr = 1.0L / x[i], tq = 1.0L / sqrtl((long double)x[i]);
        er = fmaxl(er, fabsl((r[i] - tr) / tr)); eq = fmaxl(eq, fabsl((q[i] - tq) / tq));
    }
    printf("v_rcp_f64 max rel err %.3Le = 2^%.1Lf ; v_rsq_f64 max rel err %.3Le = 2^%.1Lf\n", er, log2l(er), eq, log2l(eq));
    return 0;
}
