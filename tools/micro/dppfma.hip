// Development micro-benchmark: the DP-ALU form of data-parallel primitives on gfx950 -- v_fmac_f64_dpp with row_newbcast:k multiplies
// by lane k's operand of each row of 16 lanes, i.e. a broadcast inside the multiply-add.  Checks the semantics and times 48 of them per
// lane against 48 plain multiply-adds (one wavefront, and four wavefronts of one workgroup).
//   hipcc --offload-arch=gfx950 -O3 -o /tmp/dppfma tools/micro/dppfma.hip && /tmp/dppfma
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>
#define FM(k) asm volatile("v_fmac_f64_dpp %0, %1, %2 row_newbcast:" #k " row_mask:0xf bank_mask:0xf" : "+v"(acc[(k) & 3]) : "v"(pv), "v"(a[k + 16 * b]))
__global__ void k_dpp(double* out, const double* A, const double* p, int reps, unsigned long long* clk)
{
    double a[48];
    for (int j = 0; j < 48; ++j) a[j] = A[j * blockDim.x + threadIdx.x];
    double pv = p[threadIdx.x], tot = 0.0;
    const unsigned long long t0 = __builtin_amdgcn_s_memrealtime();
    for (int r = 0; r < reps; ++r) {
        double acc[4] = {0.0, 0.0, 0.0, 0.0};
#pragma unroll
        for (int b = 0; b < 3; ++b) {
            FM(0); FM(1); FM(2); FM(3); FM(4); FM(5); FM(6); FM(7); FM(8); FM(9); FM(10); FM(11); FM(12); FM(13); FM(14); FM(15);
        }
        tot += (acc[0] + acc[1]) + (acc[2] + acc[3]);
        pv += 1e-9 * tot;
    }
    const unsigned long long t1 = __builtin_amdgcn_s_memrealtime();
    out[threadIdx.x] = tot;
    if (threadIdx.x == 0) clk[0] = t1 - t0;
}
__global__ void k_plain(double* out, const double* A, const double* p, int reps, unsigned long long* clk)
{
    double a[48];
    for (int j = 0; j < 48; ++j) a[j] = A[j * blockDim.x + threadIdx.x];
    double pv = p[threadIdx.x], tot = 0.0;
    const unsigned long long t0 = __builtin_amdgcn_s_memrealtime();
    for (int r = 0; r < reps; ++r) {
        double acc[4] = {0.0, 0.0, 0.0, 0.0};
#pragma unroll
        for (int j = 0; j < 48; ++j) acc[j & 3] = __builtin_fma(a[j], pv, acc[j & 3]);
        tot += (acc[0] + acc[1]) + (acc[2] + acc[3]);
        pv += 1e-9 * tot;
    }
    const unsigned long long t1 = __builtin_amdgcn_s_memrealtime();
    out[threadIdx.x] = tot;
    if (threadIdx.x == 0) clk[0] = t1 - t0;
}
int main()
{
    for (int nt : {64, 256}) {
        std::vector<double> A(48 * nt), p(nt), o(nt);
        for (size_t i = 0; i < A.size(); ++i) A[i] = 1.0 + (double)(i % 97) / 97.0;
        for (int i = 0; i < nt; ++i) p[i] = 0.5 + i;
        double *dA, *dp, *dout; unsigned long long* dclk;
        hipMalloc(&dA, A.size() * 8); hipMalloc(&dp, nt * 8); hipMalloc(&dout, nt * 8); hipMalloc(&dclk, 8);
        hipMemcpy(dA, A.data(), A.size() * 8, hipMemcpyHostToDevice); hipMemcpy(dp, p.data(), nt * 8, hipMemcpyHostToDevice);
        // semantics: one repetition, compare with the host
        hipLaunchKernelGGL(k_dpp, dim3(1), dim3(nt), 0, 0, dout, dA, dp, 1, dclk);
        hipMemcpy(o.data(), dout, nt * 8, hipMemcpyDeviceToHost);
        double worst = 0.0;
        for (int t = 0; t < nt; ++t) {
            double acc[4] = {0, 0, 0, 0};
            for (int b = 0; b < 3; ++b) for (int k = 0; k < 16; ++k) acc[k & 3] = __builtin_fma(p[(t & ~15) + k], A[(k + 16 * b) * nt + t], acc[k & 3]);
            const double ref = (acc[0] + acc[1]) + (acc[2] + acc[3]);
            worst = std::max(worst, std::abs(ref - o[t]) / std::abs(ref));
        }
        std::printf("%d threads: row_newbcast semantics, worst relative difference %.2e\n", nt, worst);
        unsigned long long c1, c2; const int reps = 20000;
        for (int rep = 0; rep < 2; ++rep) {
            hipLaunchKernelGGL(k_dpp, dim3(1), dim3(nt), 0, 0, dout, dA, dp, reps, dclk); hipMemcpy(&c1, dclk, 8, hipMemcpyDeviceToHost);
            hipLaunchKernelGGL(k_plain, dim3(1), dim3(nt), 0, 0, dout, dA, dp, reps, dclk); hipMemcpy(&c2, dclk, 8, hipMemcpyDeviceToHost);
        }
        std::printf("%d threads: 48 multiply-adds per lane: dpp %.1f ns, plain %.1f ns per repetition (100 MHz clock)\n", nt, c1 * 10.0 / reps, c2 * 10.0 / reps);
    }
    return 0;
}
