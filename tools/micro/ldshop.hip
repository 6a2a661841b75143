// One-way latency of a value handed from wave to wave through LDS (the chain link of k_tri_flow, seq_split.hip.h):
// NW waves of one workgroup, wave w waits for slot[i - 1] (i = w, w + NW, ...), adds one, writes slot[i]; time / hops.
// variant 0: poll + store only; 1: + a dependent chain of 8 f64 multiply-adds; 2: + a 6-step DPP sum
// hipcc --offload-arch=gfx950 -O3 -o ldshop ldshop.hip && ./ldshop
#include <hip/hip_runtime.h>
#include <cstdio>
constexpr unsigned long long SENT = 0x7FF4DEADBEEF0001ull;
template <int VAR>
__global__ __launch_bounds__(1024) void k_hop(int n, int nw, double* out, long long* cyc)
{
    extern __shared__ double slot[];
    const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
    for (int i = threadIdx.x; i <= n; i += blockDim.x) slot[i] = __longlong_as_double((long long)SENT);
    __syncthreads();
    if (threadIdx.x == 0) slot[0] = 1.0;
    const long long t0 = clock64();
    if (wave < nw)
        for (int i = 1 + wave; i <= n; i += nw) {
            double v;
            do { v = __hip_atomic_load(&slot[i - 1], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP); } while ((unsigned long long)__double_as_longlong(v) == SENT);
            if (VAR >= 1) {
#pragma unroll
                for (int q = 0; q < 8; ++q) v = v * 0.999 + 1e-3 * lane;
            }
            if (VAR >= 2) {
                for (int o = 1; o < 64; o <<= 1) v += __shfl_xor(v, o) * 1e-9;
            }
            if (lane == 0) __hip_atomic_store(&slot[i], v + 1.0, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
        }
    __syncthreads();
    if (threadIdx.x == 0) { *out = slot[n]; *cyc = clock64() - t0; }
}
int main()
{
    double* out; long long* cyc;
    hipMalloc(&out, 8); hipMalloc(&cyc, 8);
    const int n = 16000;
    for (int var = 0; var < 3; ++var)
        for (int nw : {2, 4, 8, 16}) {
            hipEvent_t a, b; hipEventCreate(&a); hipEventCreate(&b);
            for (int rep = 0; rep < 2; ++rep) {
                hipEventRecord(a);
                if (var == 0) hipLaunchKernelGGL(k_hop<0>, dim3(1), dim3(1024), (n + 1) * 8, 0, n, nw, out, cyc);
                else if (var == 1) hipLaunchKernelGGL(k_hop<1>, dim3(1), dim3(1024), (n + 1) * 8, 0, n, nw, out, cyc);
                else hipLaunchKernelGGL(k_hop<2>, dim3(1), dim3(1024), (n + 1) * 8, 0, n, nw, out, cyc);
                hipEventRecord(b); hipEventSynchronize(b);
            }
            float ms; hipEventElapsedTime(&ms, a, b);
            long long c; hipMemcpy(&c, cyc, 8, hipMemcpyDeviceToHost);
            printf("variant %d, %2d waves: %.1f ns per hop, %.0f clock64 ticks per hop\n", var, nw, ms * 1e6 / n, (double)c / n);
        }
    return 0;
}
