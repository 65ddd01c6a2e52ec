// Part of fcpt_kernels.hip (one translation unit, namespace fcpt): index macros, launch geometry, reciprocal / DPP / XCD helpers shared by all kernels.
// Not a stand-alone header: included once, in the order given there.

// 32-bit cell index: fcpt_create rejects grids with (nr+1)*nphi >= 2^31
#define IDX(i, j) ((i) * P.nphi + (j))

// One thread per cell; a 256-thread block is bx (phi) x by (rings), bx = the
// smallest power of two >= nphi capped at 256, so narrow pseudo-1D grids
// (Nphi = 2, 4) still fill their wavefronts with consecutive rings.
struct Launch2D {
    dim3 grid, block;
};
static inline Launch2D launch2d(int nrows, int nphi)
{
    int bx = 1;
    while (bx < nphi && bx < 256)
        bx <<= 1;
    const int by = 256 / bx;
    Launch2D l;
    l.block = dim3(bx, by, 1);
    l.grid = dim3((nphi + bx - 1) / bx, (nrows + by - 1) / by, 1);
    return l;
}
// ROWU (template parameter of every 2-D kernel): the block is at least one wavefront wide in
// phi, so all lanes of a wavefront share the ring index.  Promoting it to a scalar register
// turns every per-ring geometry access (Rmed[i], InvSurf[i], ...) into a scalar-cache load
// instead of a 64-lane vector load with full memory latency.
#define CELL(row0, nrows)                                            \
    const int j = blockIdx.x * blockDim.x + threadIdx.x;             \
    const int i_ = (row0) + blockIdx.y * blockDim.y + threadIdx.y;   \
    if (j >= P.nphi || i_ >= (row0) + (nrows))                       \
        return;                                                      \
    const int i = ROWU ? __builtin_amdgcn_readfirstlane(i_) : i_;
#define JNEXT (j == P.nphi - 1 ? 0 : j + 1)
#define JPREV (j == 0 ? P.nphi - 1 : j - 1)

// two adjacent doubles moved as one 16-byte access (the address is only 8-byte aligned)
typedef double D2v __attribute__((ext_vector_type(2)));
typedef D2v __attribute__((aligned(8))) D2;
#ifdef EXP_NT
#define LD2(p_) __builtin_nontemporal_load((const D2 *)(p_))
#define ST2(p_, v_) __builtin_nontemporal_store((v_), (D2 *)(p_))
#else
#define LD2(p_) (*(const D2 *)(p_))
#define ST2(p_, v_) (*(D2 *)(p_) = (v_))
#endif

// packed per-ring rows through the constant address space (wide scalar loads)
template <class T> __device__ __forceinline__ T crow_load(const T *tab, int i)
{
    static_assert(sizeof(T) % 8 == 0, "rows are made of 8-byte fields");
    typedef const unsigned long long __attribute__((address_space(4))) *cptr;
    cptr src = (cptr)__builtin_assume_aligned((const void *)(tab + i), alignof(T));
    T out;
    unsigned long long *dst = (unsigned long long *)&out;
#pragma unroll
    for (int n = 0; n < (int)(sizeof(T) / 8); ++n)
        dst[n] = src[n];
    return out;
}

// Workgroups are dealt round-robin to the 8 XCDs, each with its own L2.  The marching kernels
// give every XCD a contiguous range of logical blocks, so that neighbouring phi tiles and ring
// chunks -- which read the same halo cells -- meet in one L2 instead of fetching them twice from
// HBM.  Bijective for any block count (blockIdx % 8 only labels blocks that share an XCD).
__device__ __forceinline__ int xcd_block(int b, int nb)
{
#ifdef FCPT_NO_XCD_REMAP
    return b;
#else
    const int q = nb >> 3, r = nb & 7, x = b & 7;
    return (x < r ? x * (q + 1) : r * (q + 1) + (x - r) * q) + (b >> 3);
#endif
}

// Reciprocal from v_rcp_f64 (~2^-26) refined by two Newton steps: ~1 ulp, less than half the
// issue cost of the IEEE division sequence.  Used only where the result feeds a limited
// slope or a specific quantity (errors of a few ulp there are far inside the 1e-10 parity bar).
__device__ __forceinline__ double fast_rcp(double d)
{
    double x = __builtin_amdgcn_rcp(d);
    double e = fma(-d, x, 1.0);
    x = fma(x, e, x);
    e = fma(-d, x, 1.0);
    x = fma(x, e, x);
    return x;
}

// One Newton step: relative error <= 2e-15 (measured on MI355X, profiles/tools/rcp_accuracy.hip);
// used for the van Leer slope, whose error enters the state scaled by (dx - v dt) dq / Q << 1.
__device__ __forceinline__ double fast_rcp1(double d)
{
    const double x = __builtin_amdgcn_rcp(d);
    return fma(x, fma(-d, x, 1.0), x);
}

// The reciprocals of the transport (1 / Sigma of the specific quantities, velocities from momenta) act on
// full-magnitude quantities and keep both Newton steps: with one (-0.4 % per step) the 96 x 64 three-slab ideal-EOS
// case reads 1.14e-10 on v_r after 20 steps and the 110 240-step accretion run shifts its deviation by 1.8e-6 relative
// (profiles/r03_ab_rcp_one_newton_step.txt).
#ifndef FAST_RCP_TR
#define FAST_RCP_TR fast_rcp
#endif
// 1/sqrt(x) from v_rsq_f64 (~2^-26) refined by two Newton steps (~1 ulp): a third of the issue cost of
// sqrt followed by the IEEE division sequence
__device__ __forceinline__ double fast_rsqrt(double x)
{
    double y = __builtin_amdgcn_rsq(x);
    const double h = 0.5 * x;
    y = y * fma(-h * y, y, 1.5);
    y = y * fma(-h * y, y, 1.5);
    return y;
}
// One Newton step on v_rsq_f64: ~1e-15 relative, for roots whose error reaches the state scaled by dt (the smoothed
// distance of the inline potential, the sound speed that enters nu and H of the viscous terms)
__device__ __forceinline__ double fast_rsqrt1(double x)
{
    const double y = __builtin_amdgcn_rsq(x);
    return y * fma(-(0.5 * x) * y, y, 1.5);
}
// exp(x) for the compression heating of the marching source step (SourceEuler.cpp:459-493), where the argument
// -(gamma - 1) dt div v is bounded by the CFL condition and almost always tiny.  While every lane of the wavefront
// has |x| < 1/16 the degree-9 Taylor polynomial is exact to 2.5e-19 relative (nine FMAs; the library routine costs
// 32 vector instructions and parks its nine coefficients in 18 VGPRs of a kernel that has none to spare).  Larger
// arguments (strong shocks) are halved until they fit and the result squared back: 2^s ulp after s squarings.
__device__ __forceinline__ double exp_small(double x)
{
    int s = 0;
    while (__builtin_amdgcn_ballot_w64(fabs(x) >= 0.0625) != 0 && s < 16) { // wave-uniform; NaN compares false
        x *= 0.5;
        ++s;
    }
    double p = 1.0 / 362880.0;
    p = fma(p, x, 1.0 / 40320.0);
    p = fma(p, x, 1.0 / 5040.0);
    p = fma(p, x, 1.0 / 720.0);
    p = fma(p, x, 1.0 / 120.0);
    p = fma(p, x, 1.0 / 24.0);
    p = fma(p, x, 1.0 / 6.0);
    p = fma(p, x, 0.5);
    p = fma(p, x, 1.0);
    p = fma(p, x, 1.0);
    for (; s > 0; --s)
        p *= p;
    return p;
}
__device__ __forceinline__ double dmin(double a, double b) { return b < a ? b : a; } // std::min
__device__ __forceinline__ double dmax(double a, double b) { return a < b ? b : a; } // std::max

// Whole-wavefront shifts by one lane as DPP moves (gfx9 wave_shr:1 / wave_shl:1): the value of lane-1 / lane+1.  Two
// VALU moves instead of two ds_bpermute round trips through the LDS crossbar (profiles/tools/dpp_shift.hip).
// Two forms.  bound_ctrl (KEEP = false, the default): lanes 0 / 63, which have no source lane, read 0 -- parity-clean,
// those lanes are halo in every marching kernel -- and the moves need no prior copy of the destination.  Read-modify-
// write (KEEP = true): lanes 0 / 63 keep their own value, at the price of a register copy per half (90 of the 770
// vector instructions per ring of k_transport_fused).  History of the choice: bound_ctrl measured 2-4.5 % SLOWER
// per step as long as the transport kernel's time was set by its tail of slow wavefronts; with the chunks dealt slow
// ones first (transport_fused.h) the kernel sits at its issue bound and the same change is 1.2 % FASTER (isothermal,
// three A/B pairs; the transport of the ideal EOS -9 us).  The ideal-EOS source march keeps the copies: without them
// it runs 16 us longer (128 VGPRs, a different spill).
// (Tried and dropped: the copy as one v_mov_b64, + 9 %; shifts through the LDS, + 16 %: 124 LDS operations of 512 B
//  per ring and wavefront are as long on the CU's LDS port as the vector moves they replace.)
template <int CTRL, bool KEEP> __device__ __forceinline__ double dpp_shift(double x)
{
    if (KEEP) {
        int lo = __double2loint(x), hi = __double2hiint(x);
        lo = __builtin_amdgcn_update_dpp(lo, lo, CTRL, 0xf, 0xf, false);
        hi = __builtin_amdgcn_update_dpp(hi, hi, CTRL, 0xf, 0xf, false);
        return __hiloint2double(hi, lo);
    }
    const int lo = __builtin_amdgcn_mov_dpp(__double2loint(x), CTRL, 0xf, 0xf, true);
    const int hi = __builtin_amdgcn_mov_dpp(__double2hiint(x), CTRL, 0xf, 0xf, true);
    return __hiloint2double(hi, lo);
}
__device__ __forceinline__ double lane_prev(double x) { return dpp_shift<0x138, false>(x); }
__device__ __forceinline__ double lane_next(double x) { return dpp_shift<0x130, false>(x); }
__device__ __forceinline__ double lane_prev_keep(double x) { return dpp_shift<0x138, true>(x); }
__device__ __forceinline__ double lane_next_keep(double x) { return dpp_shift<0x130, true>(x); }
// Sum over the wavefront in a fixed tree order (deterministic), all in the VALU: DPP row shifts
// build the 16-lane row sums, row_bcast:15 / row_bcast:31 fold the four rows.  The total is valid
// in lane 63.  (The ds_bpermute butterfly costs six dependent LDS round trips per call.)
template <int CTRL, int ROW_MASK> __device__ __forceinline__ double dpp_add(double x)
{
    int lo = __double2loint(x), hi = __double2hiint(x);
    // bound_ctrl: lanes without a source (and rows masked out) contribute 0
    const int slo = __builtin_amdgcn_update_dpp(0, lo, CTRL, ROW_MASK, 0xf, true);
    const int shi = __builtin_amdgcn_update_dpp(0, hi, CTRL, ROW_MASK, 0xf, true);
    return x + __hiloint2double(shi, slo);
}
__device__ __forceinline__ double wave_sum(double x)
{
    x = dpp_add<0x111, 0xf>(x); // row_shr:1
    x = dpp_add<0x112, 0xf>(x); // row_shr:2
    x = dpp_add<0x114, 0xf>(x); // row_shr:4
    x = dpp_add<0x118, 0xf>(x); // row_shr:8  -> lane 15 of each row holds the row sum
    x = dpp_add<0x142, 0xa>(x); // row_bcast:15 into rows 1 and 3
    x = dpp_add<0x143, 0xc>(x); // row_bcast:31 into rows 2 and 3 -> lane 63 holds the total
    return x;
}

// The shift-jump stamps of a transport (k_ring_mean, kernels/transport.h): sequence number of the current step ...
#define SHIFT_SEQ(clk_) ((int)(((clk_)->n_hydro_iter + 1ull) & 0x7fffffffull))
// true if the transport whose k_ring_mean ran last found a ring pair beyond the one-lane shift of k_transport_fused
// (written by an earlier kernel, never by the reading one: scalar loads through the constant address space)
__device__ __forceinline__ bool shift_jump_raised(const int *flag)
{
    const int __attribute__((address_space(4))) *f = (const int __attribute__((address_space(4))) *)flag;
    return f[0] == f[2];
}

// Addresses as 32-bit BYTE offsets from a grid's base: the launchers only take the marching kernels for grids below
// 4 GiB each (2^29 cells), so base (scalar registers) + zext(offset) is the saddr form of global_load / global_store -- one VGPR
// per address instead of two, and no 64-bit address arithmetic in the vector unit (nine v_lshl_add_u64 per ring before).
__device__ __forceinline__ double ld_off(const double *base, unsigned off) { return *(const double *)((const char *)base + off); }
__device__ __forceinline__ void st_off(double *base, unsigned off, double v) { *(double *)((char *)base + off) = v; }
__device__ __forceinline__ D2 ld2_off(const double *base, unsigned off) { return LD2((const double *)((const char *)base + off)); }
__device__ __forceinline__ void st2_off(double *base, unsigned off, D2 v) { ST2((double *)((char *)base + off), v); }


// wavefronts per workgroup of the marching kernels: always launched with 256 threads (a constant instead of
// blockDim.x, which is a load from the dispatch packet and a wait at the top of every wavefront)
#define MARCH_WAVES 4

// (cfl.h) the last stage of the CFL reduction and the CalculateTimeStep policy, evaluated by every workgroup of the
// marching source kernel for itself -- see there
__device__ __forceinline__ double cfl_fold_in_step(const Dev &P);
