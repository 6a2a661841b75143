// Micro-benchmark: cost of a grid-wide barrier (one block per CU, cooperative launch) with the
// release / acquire pattern the persistent coarse solver would use.  hipcc --offload-arch=gfx950 -O3
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>

__device__ __forceinline__ bool grid_barrier(unsigned* counter, unsigned target, int* err)
{
    __syncthreads();
    if (threadIdx.x == 0) {
        __threadfence();  // release: this block's stores are visible device-wide before the arrival
        atomicAdd(counter, 1u);
        unsigned spins = 0;
        while (__hip_atomic_load(counter, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) < target) {
            __builtin_amdgcn_s_sleep(1);
            if (++spins > (1u << 24)) { *err = 1; break; }
        }
        __threadfence();  // acquire: later loads of this CU see the other blocks' stores
    }
    __syncthreads();
    return true;
}

__global__ __launch_bounds__(256) void k_bar(unsigned* counter, int nbar, double* data, double* out, int* err)
{
    const int nb = gridDim.x;
    double acc = 0.0;
    for (int it = 0; it < nbar; ++it) {
        // every block publishes one value, then everybody sums all values
        if (threadIdx.x == 0) data[(it & 1) * nb + blockIdx.x] = (double)(it + blockIdx.x);
        grid_barrier(counter, (unsigned)(it + 1) * nb, err);
        if (*err) return;
        double s = 0.0;
        for (int i = 0; i < nb; ++i) s += data[(it & 1) * nb + i];
        acc += s;
    }
    if (threadIdx.x == 0) out[blockIdx.x] = acc;
}

int main()
{
    hipDeviceProp_t prop; hipGetDeviceProperties(&prop, 0);
    const int nb = prop.multiProcessorCount;
    unsigned* counter; double *data, *out; int* err;
    hipMalloc(&counter, 4); hipMalloc(&data, sizeof(double) * 2 * nb); hipMalloc(&out, sizeof(double) * nb); hipMalloc(&err, 4);
    for (int nbar : {1, 100, 1000}) {
        hipMemset(counter, 0, 4); hipMemset(err, 0, 4);
        hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
        void* args[] = {&counter, (void*)&nbar, &data, &out, &err};
        hipEventRecord(e0);
        hipError_t st = hipLaunchCooperativeKernel((const void*)k_bar, dim3(nb), dim3(256), args, 0, 0);
        hipEventRecord(e1); hipEventSynchronize(e1);
        float ms; hipEventElapsedTime(&ms, e0, e1);
        std::vector<double> h(nb); int herr;
        hipMemcpy(h.data(), out, sizeof(double) * nb, hipMemcpyDeviceToHost); hipMemcpy(&herr, err, 4, hipMemcpyDeviceToHost);
        double expect = 0; for (int it = 0; it < nbar; ++it) for (int b = 0; b < nb; ++b) expect += it + b;
        int bad = 0; for (int b = 0; b < nb; ++b) bad += h[b] != expect;
        printf("blocks %d barriers %d: launch %s, %.1f us total, %.2f us per barrier, wrong sums %d, timeout flag %d\n",
               nb, nbar, hipGetErrorString(st), ms * 1e3, ms * 1e3 / nbar, bad, herr);
    }
    return 0;
}
