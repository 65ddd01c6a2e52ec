// Issue cost of the instruction classes of the marching kernels on gfx950, in SIMD cycles per wavefront instruction:
// independent chains (throughput) of v_fma_f64, v_mov_b32, v_cndmask_b32, v_mov_b32 with DPP row_shr:1 / wave_shr:1 /
// row_bcast:15, v_rcp_f64, and of the same as one dependent chain (latency).  4 wavefronts per SIMD, all SIMDs.
//   hipcc --offload-arch=gfx950 -O3 valu_rates.hip -o valu_rates && ./valu_rates
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>

#define NCH 8 /* independent chains */
template <int OP> __device__ __forceinline__ void step(double (&x)[NCH], int (&y)[2 * NCH], double c)
{
#pragma unroll
    for (int k = 0; k < NCH; ++k) {
        if (OP == 0) x[k] = fma(x[k], c, c);
        if (OP == 1) { asm volatile("v_mov_b32 %0, %1" : "=v"(y[2 * k]) : "v"(y[2 * k + 1])); asm volatile("v_mov_b32 %0, %1" : "=v"(y[2 * k + 1]) : "v"(y[2 * k])); }
        if (OP == 2) { y[2 * k] = __builtin_amdgcn_update_dpp(y[2 * k], y[2 * k + 1], 0x111, 0xf, 0xf, false); y[2 * k + 1] = __builtin_amdgcn_update_dpp(y[2 * k + 1], y[2 * k], 0x111, 0xf, 0xf, false); }
        if (OP == 3) { y[2 * k] = __builtin_amdgcn_update_dpp(y[2 * k], y[2 * k + 1], 0x138, 0xf, 0xf, false); y[2 * k + 1] = __builtin_amdgcn_update_dpp(y[2 * k + 1], y[2 * k], 0x138, 0xf, 0xf, false); }
        if (OP == 4) { y[2 * k] = __builtin_amdgcn_update_dpp(y[2 * k], y[2 * k + 1], 0x142, 0xa, 0xf, false); y[2 * k + 1] = __builtin_amdgcn_update_dpp(y[2 * k + 1], y[2 * k], 0x142, 0xa, 0xf, false); }
        if (OP == 5) x[k] = __builtin_amdgcn_rcp(x[k]);
        if (OP == 6) { x[k] = x[k] > c ? c : x[k] + 1.0; } // cmp + 2 cndmask + add
        if (OP == 7) { y[2 * k] = __builtin_amdgcn_mov_dpp(y[2 * k + 1], 0x138, 0xf, 0xf, true); y[2 * k + 1] = __builtin_amdgcn_mov_dpp(y[2 * k], 0x138, 0xf, 0xf, true); }
        if (OP == 8) x[k] = x[k] * c;
        if (OP == 9) x[k] = x[k] + c;
    }
}
// clk[0..1]: shader-clock (s_memtime) and 100 MHz wall-clock (s_memrealtime) ticks wavefront 0 spent in the loop: the
// quotient is the frequency the SIMDs actually ran at under this instruction mix
template <int OP> __global__ void __launch_bounds__(256) k_rate(const double *in, double *out, int iters, long long *clk)
{
    double x[NCH];
    int y[2 * NCH];
    const double c = in[0];
#pragma unroll
    for (int k = 0; k < NCH; ++k) {
        x[k] = in[1 + k] + threadIdx.x;
        y[2 * k] = threadIdx.x + k;
        y[2 * k + 1] = threadIdx.x * 3 + k;
    }
    const long long c0 = clock64(), w0 = wall_clock64();
    for (int i = 0; i < iters; ++i) {
        step<OP>(x, y, c);
        step<OP>(x, y, c);
        step<OP>(x, y, c);
        step<OP>(x, y, c);
    }
    const long long c1 = clock64(), w1 = wall_clock64();
    if (clk && blockIdx.x == 0 && threadIdx.x == 0) {
        clk[0] = c1 - c0;
        clk[1] = w1 - w0;
    }
    double s = 0;
#pragma unroll
    for (int k = 0; k < NCH; ++k)
        s += x[k] + y[2 * k] + y[2 * k + 1];
    out[blockIdx.x * blockDim.x + threadIdx.x] = s;
}
template <int OP> void run(const char *name, int instr_per_step, const double *din, double *dout, double ghz)
{
    const int blocks = 256 * 4, iters = 40000; // 4 wavefronts per SIMD on 256 CUs; ~10 ms per launch
    static long long *dclk = nullptr;
    if (!dclk)
        hipMalloc(&dclk, 2 * sizeof(long long));
    hipEvent_t a, b;
    hipEventCreate(&a);
    hipEventCreate(&b);
    for (int w = 0; w < 5; ++w) // the clocks settle over tens of milliseconds
        k_rate<OP><<<blocks, 256>>>(din, dout, iters, nullptr);
    hipEventRecord(a);
    k_rate<OP><<<blocks, 256>>>(din, dout, iters, dclk);
    hipEventRecord(b);
    hipEventSynchronize(b);
    float ms;
    hipEventElapsedTime(&ms, a, b);
    long long clk[2];
    hipMemcpy(clk, dclk, sizeof(clk), hipMemcpyDeviceToHost);
    const double wave_instr_per_simd = 4.0 /*waves*/ * iters * 4.0 * instr_per_step;
    const double f_ghz = (double)clk[0] / ((double)clk[1] * 10.0); // ticks per 10 ns
    (void)ghz;
    // (the launch is one round of 4 wavefronts per SIMD: launch time x measured frequency / instructions per SIMD)
    printf("%-34s %8.3f ms  shader clock %5.3f GHz (s_memtime / s_memrealtime)  -> %6.2f cycles per wavefront instruction\n", name, ms,
           f_ghz, ms * 1e-3 * f_ghz * 1e9 / wave_instr_per_simd);
}
int main()
{
    std::vector<double> h(64, 1.0000001);
    double *din, *dout;
    hipMalloc(&din, 64 * sizeof(double));
    hipMalloc(&dout, 256 * 4 * 256 * sizeof(double));
    hipMemcpy(din, h.data(), 64 * sizeof(double), hipMemcpyHostToDevice);
    const double ghz = 2.4;
    run<0>("v_fma_f64", NCH, din, dout, ghz);
    run<8>("v_mul_f64", NCH, din, dout, ghz);
    run<9>("v_add_f64", NCH, din, dout, ghz);
    run<1>("v_mov_b32", 2 * NCH, din, dout, ghz);
    run<2>("v_mov_b32 dpp row_shr:1", 2 * NCH, din, dout, ghz);
    run<3>("v_mov_b32 dpp wave_shr:1", 2 * NCH, din, dout, ghz);
    run<7>("v_mov_b32 dpp wave_shr:1 bound_ctrl", 2 * NCH, din, dout, ghz);
    run<4>("v_mov_b32 dpp row_bcast:15", 2 * NCH, din, dout, ghz);
    run<5>("v_rcp_f64", NCH, din, dout, ghz);
    run<6>("cmp + 2 cndmask + add_f64 (4 instr)", 4 * NCH, din, dout, ghz);
    return 0;
}
