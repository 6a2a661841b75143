// Cost of a barrier among a few workgroups placed on ONE XCD (blocks 0, 8, 16, ... of a grid under round-robin placement):
// arrival = one atomic add, release = polling the counter; with a 2 KB vector written before and read (L1 bypassed) after
// every barrier, as a dependency class of a triangular solve would.   hipcc --offload-arch=gfx950 -O3 xcdbar.hip -o xcdbar
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
typedef __attribute__((address_space(1))) unsigned gu32;
typedef __attribute__((address_space(1))) unsigned long long gu64;
template <int SCOPE>
__global__ __launch_bounds__(512) void k_bar(unsigned* cnt, double* buf, int nb, int iters, unsigned* xcc, int stride)
{
    if (blockIdx.x % stride) return;
    const int b = blockIdx.x / stride;
    if (threadIdx.x == 0) {
        unsigned id;
        asm volatile("s_getreg_b32 %0, hwreg(HW_REG_XCC_ID)" : "=s"(id));
        xcc[b] = id & 0xf;
    }
    double acc = 0.0;
    for (int it = 0; it < iters; ++it) {
        // "compute": every thread writes one value others will read after the barrier
        buf[(size_t)(it & 1) * nb * 512 + b * 512 + threadIdx.x] = acc + it;
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        __syncthreads();
        if (threadIdx.x == 0) {
            __hip_atomic_fetch_add((gu32*)cnt, 1u, __ATOMIC_RELAXED, SCOPE);
            const unsigned want = (unsigned)nb * (unsigned)(it + 1);
            while (__hip_atomic_load((gu32*)cnt, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) < want) {}
        }
        __syncthreads();
        const int nbr = (b + 1) % nb;
        acc += __longlong_as_double((long long)__hip_atomic_load((gu64*)(buf + (size_t)(it & 1) * nb * 512 + nbr * 512 + threadIdx.x), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT));
    }
    if (acc == 12345.678) buf[0] = acc;
}
int main()
{
    unsigned *cnt, *xcc; double* buf;
    hipMalloc(&cnt, 64); hipMalloc(&xcc, 256); hipMalloc(&buf, sizeof(double) * 2 * 32 * 512);
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    const int iters = 2000;
    for (int stride : {8, 1})
        for (int nb : {2, 4, 8})
            for (int scope = 0; scope < 2; ++scope) {
                float best = 1e9f;
                for (int rep = 0; rep < 3; ++rep) {
                    hipMemset(cnt, 0, 64);
                    hipEventRecord(e0);
                    if (scope == 0) hipLaunchKernelGGL(k_bar<__HIP_MEMORY_SCOPE_WORKGROUP>, dim3(nb * stride), dim3(512), 0, 0, cnt, buf, nb, iters, xcc, stride);
                    else hipLaunchKernelGGL(k_bar<__HIP_MEMORY_SCOPE_AGENT>, dim3(nb * stride), dim3(512), 0, 0, cnt, buf, nb, iters, xcc, stride);
                    hipEventRecord(e1); hipEventSynchronize(e1);
                    float ms; hipEventElapsedTime(&ms, e0, e1); best = ms < best ? ms : best;
                }
                unsigned h[32]; hipMemcpy(h, xcc, sizeof(unsigned) * nb, hipMemcpyDeviceToHost);
                bool same = true; for (int i = 1; i < nb; ++i) same = same && h[i] == h[0];
                printf("blocks %d at stride %d (%s XCD), arrival atomic scope %s: %.2f us per barrier round\n", nb, stride, same ? "one" : "several",
                       scope ? "agent" : "workgroup", best * 1e3 / iters);
            }
    return 0;
}
