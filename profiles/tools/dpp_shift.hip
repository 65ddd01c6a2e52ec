// Checks the semantics of the gfx9 whole-wavefront DPP shifts on gfx950 and times them against
// ds_bpermute (what __shfl compiles to).  hipcc --offload-arch=gfx950 -O3 dpp_shift.hip -o dpp_shift
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>

template <int CTRL> __device__ __forceinline__ double dpp_mov(double x)
{
    int lo = __double2loint(x), hi = __double2hiint(x);
    lo = __builtin_amdgcn_update_dpp(lo, lo, CTRL, 0xf, 0xf, false);
    hi = __builtin_amdgcn_update_dpp(hi, hi, CTRL, 0xf, 0xf, false);
    return __hiloint2double(hi, lo);
}
__global__ void k_sem(double *out)
{
    const double x = (double)threadIdx.x;
    out[threadIdx.x] = dpp_mov<0x138>(x);       // wave_shr:1
    out[64 + threadIdx.x] = dpp_mov<0x130>(x);  // wave_shl:1
    out[128 + threadIdx.x] = dpp_mov<0x13C>(x); // wave_ror:1
    out[192 + threadIdx.x] = dpp_mov<0x134>(x); // wave_rol:1
}
template <int MODE> __global__ void k_time(const double *in, double *out, int iters)
{
    const int lane = threadIdx.x & 63;
    const int l = lane > 0 ? lane - 1 : 0, r = lane < 63 ? lane + 1 : 63;
    double x = in[blockIdx.x * blockDim.x + threadIdx.x], acc = 0.0;
    for (int i = 0; i < iters; ++i) {
        double a, b;
        if (MODE == 0) {
            a = __shfl(x, l, 64);
            b = __shfl(x, r, 64);
        } else {
            a = dpp_mov<0x138>(x);
            b = dpp_mov<0x130>(x);
        }
        acc = fma(a - x, b - x, acc);
        x = fma(x, 1.0000001, 0.5 * (a + b) * 1e-9);
    }
    out[blockIdx.x * blockDim.x + threadIdx.x] = acc + x;
}
int main()
{
    double *d;
    hipMalloc(&d, 256 * sizeof(double));
    k_sem<<<1, 64>>>(d);
    std::vector<double> h(256);
    hipMemcpy(h.data(), d, 256 * sizeof(double), hipMemcpyDeviceToHost);
    const char *names[4] = {"wave_shr:1", "wave_shl:1", "wave_ror:1", "wave_rol:1"};
    for (int m = 0; m < 4; ++m) {
        printf("%s: lane0<-%g lane1<-%g lane15<-%g lane16<-%g lane31<-%g lane32<-%g lane62<-%g lane63<-%g\n", names[m],
               h[64 * m], h[64 * m + 1], h[64 * m + 15], h[64 * m + 16], h[64 * m + 31], h[64 * m + 32], h[64 * m + 62],
               h[64 * m + 63]);
        int prev_ok = 1, next_ok = 1;
        for (int l = 1; l < 63; ++l) {
            prev_ok &= h[64 * m + l] == l - 1;
            next_ok &= h[64 * m + l] == l + 1;
        }
        printf("   interior lanes: from lane-1 %d, from lane+1 %d\n", prev_ok, next_ok);
    }
    const int n = 256 * 4 * 256 * 4;
    double *in, *out;
    hipMalloc(&in, n * sizeof(double));
    hipMalloc(&out, n * sizeof(double));
    hipMemset(in, 0, n * sizeof(double));
    hipEvent_t e0, e1;
    hipEventCreate(&e0);
    hipEventCreate(&e1);
    for (int mode = 0; mode < 2; ++mode) {
        for (int rep = 0; rep < 2; ++rep) {
            hipEventRecord(e0);
            if (mode == 0)
                k_time<0><<<n / 256, 256>>>(in, out, 2000);
            else
                k_time<1><<<n / 256, 256>>>(in, out, 2000);
            hipEventRecord(e1);
            hipEventSynchronize(e1);
            float ms;
            hipEventElapsedTime(&ms, e0, e1);
            if (rep)
                printf("%s: %.3f ms for 2000 iterations x 2 shifts, %d waves\n", mode ? "dpp" : "ds_bpermute", ms, n / 64);
        }
    }
    return 0;
}
