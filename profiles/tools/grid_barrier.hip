// Cost of a grid-wide barrier on gfx950, for the one-kernel-per-step path of launch-bound grids:
//   hipcc --offload-arch=gfx950 -O3 -o grid_barrier profiles/tools/grid_barrier.hip && ./grid_barrier
// (a) sense-reversing counter barrier with agent-scope atomics, workgroups spinning on one word;
// (b) cooperative groups grid.sync(); (c) for scale: a chain of dependent empty launches.
// Each barrier separates a store from a load of another workgroup's value (so the memory ordering is exercised).
#include <hip/hip_runtime.h>
#include <hip/hip_cooperative_groups.h>
#include <cstdio>
#include <vector>
namespace cg = cooperative_groups;

__device__ __forceinline__ void grid_barrier(unsigned *count, unsigned *gen, unsigned nblocks)
{
    __syncthreads();
    if (threadIdx.x == 0) {
        const unsigned g = __hip_atomic_load(gen, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        __threadfence();
        if (__hip_atomic_fetch_add(count, 1u, __ATOMIC_ACQ_REL, __HIP_MEMORY_SCOPE_AGENT) == nblocks - 1) {
            __hip_atomic_store(count, 0u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            __hip_atomic_store(gen, g + 1, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_AGENT);
        } else {
            while (__hip_atomic_load(gen, __ATOMIC_ACQUIRE, __HIP_MEMORY_SCOPE_AGENT) == g)
                __builtin_amdgcn_s_sleep(1);
        }
        __threadfence();
    }
    __syncthreads();
}

__global__ void k_own(double *buf, unsigned *sync, int rounds)
{
    const unsigned nb = gridDim.x;
    double acc = 0.0;
    for (int r = 0; r < rounds; ++r) {
        buf[(r & 1) * nb * 256 + blockIdx.x * 256 + threadIdx.x] = acc + r;
        grid_barrier(sync, sync + 32, nb);
        acc += buf[(r & 1) * nb * 256 + ((blockIdx.x + 1) % nb) * 256 + threadIdx.x];
    }
    if (acc == -1.0)
        buf[0] = acc;
}
__global__ void k_coop(double *buf, int rounds)
{
    cg::grid_group g = cg::this_grid();
    const unsigned nb = gridDim.x;
    double acc = 0.0;
    for (int r = 0; r < rounds; ++r) {
        buf[(r & 1) * nb * 256 + blockIdx.x * 256 + threadIdx.x] = acc + r;
        g.sync();
        acc += buf[(r & 1) * nb * 256 + ((blockIdx.x + 1) % nb) * 256 + threadIdx.x];
    }
    if (acc == -1.0)
        buf[0] = acc;
}
__global__ void k_empty(double *buf) { if (buf[0] == -1.0) buf[1] = 0.0; }

int main()
{
    double *buf;
    unsigned *sync;
    hipMalloc(&buf, 2 * 1024 * 256 * sizeof(double));
    hipMalloc(&sync, 256);
    hipMemset(buf, 0, 2 * 1024 * 256 * sizeof(double));
    hipMemset(sync, 0, 256);
    hipEvent_t e0, e1;
    hipEventCreate(&e0);
    hipEventCreate(&e1);
    const int rounds = 2000;
    for (int nb : {8, 32, 64, 128, 256, 512}) {
        float ms_own = 0, ms_coop = 0;
        for (int rep = 0; rep < 3; ++rep) {
            hipEventRecord(e0);
            hipLaunchKernelGGL(k_own, dim3(nb), dim3(256), 0, 0, buf, sync, rounds);
            hipEventRecord(e1);
            hipEventSynchronize(e1);
            hipEventElapsedTime(&ms_own, e0, e1);
        }
        int rr = rounds;
        void *args[] = {&buf, &rr};
        hipError_t err = hipSuccess;
        for (int rep = 0; rep < 3; ++rep) {
            hipEventRecord(e0);
            err = hipLaunchCooperativeKernel((void *)k_coop, dim3(nb), dim3(256), args, 0, 0);
            hipEventRecord(e1);
            hipEventSynchronize(e1);
            hipEventElapsedTime(&ms_coop, e0, e1);
        }
        printf("%4d workgroups: own barrier %.3f us, grid.sync() %.3f us per round (%s)\n", nb, 1e3 * ms_own / rounds,
               1e3 * ms_coop / rounds, hipGetErrorString(err));
    }
    // dependent launches, plain and as a replayed graph
    hipStream_t st;
    hipStreamCreate(&st);
    for (int i = 0; i < 100; ++i)
        hipLaunchKernelGGL(k_empty, dim3(64), dim3(256), 0, st, buf);
    hipStreamSynchronize(st);
    hipEventRecord(e0, st);
    for (int i = 0; i < rounds; ++i)
        hipLaunchKernelGGL(k_empty, dim3(64), dim3(256), 0, st, buf);
    hipEventRecord(e1, st);
    hipEventSynchronize(e1);
    float ms = 0;
    hipEventElapsedTime(&ms, e0, e1);
    printf("dependent launches: %.3f us each\n", 1e3 * ms / rounds);
    hipGraph_t g;
    hipGraphExec_t ge;
    hipStreamBeginCapture(st, hipStreamCaptureModeThreadLocal);
    for (int i = 0; i < 20; ++i)
        hipLaunchKernelGGL(k_empty, dim3(64), dim3(256), 0, st, buf);
    hipStreamEndCapture(st, &g);
    hipGraphInstantiate(&ge, g, nullptr, nullptr, 0);
    for (int i = 0; i < 10; ++i)
        hipGraphLaunch(ge, st);
    hipStreamSynchronize(st);
    hipEventRecord(e0, st);
    for (int i = 0; i < 100; ++i)
        hipGraphLaunch(ge, st);
    hipEventRecord(e1, st);
    hipEventSynchronize(e1);
    hipEventElapsedTime(&ms, e0, e1);
    printf("graph of 20 dependent launches: %.3f us per launch\n", 1e3 * ms / 2000);
    return 0;
}
