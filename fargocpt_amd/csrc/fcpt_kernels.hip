// HIP kernels of the gas update for gfx950 (CDNA4).  FP64 throughout, phi is the
// contiguous (coalesced) axis of every grid, no MFMA (there is no contraction).
//
// Each kernel cites the reference loop nest it restates (paths relative to the
// reference's src/).  Operand order follows the reference so results agree with
// the CPU path to rounding.
#include <hip/hip_runtime.h>

#include <cstdlib>

#include "fcpt_kernels.h"

namespace fcpt {

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
__device__ __forceinline__ double dmin(double a, double b) { return b < a ? b : a; } // std::min
__device__ __forceinline__ double dmax(double a, double b) { return a < b ? b : a; } // std::max

// Whole-wavefront shifts by one lane as DPP moves (gfx9 wave_shr:1 / wave_shl:1): the value of
// lane-1 / lane+1, lanes 0 / 63 keep their own value.  Two VALU moves instead of two
// ds_bpermute round trips through the LDS crossbar (profiles/tools/dpp_shift.hip).
template <int CTRL> __device__ __forceinline__ double dpp_shift(double x)
{
    int lo = __double2loint(x), hi = __double2hiint(x);
    lo = __builtin_amdgcn_update_dpp(lo, lo, CTRL, 0xf, 0xf, false);
    hi = __builtin_amdgcn_update_dpp(hi, hi, CTRL, 0xf, 0xf, false);
    return __hiloint2double(hi, lo);
}
__device__ __forceinline__ double lane_prev(double x) { return dpp_shift<0x138>(x); }
__device__ __forceinline__ double lane_next(double x) { return dpp_shift<0x130>(x); }
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

// ---------------------------------------------------------------------------
// Pframeforce.cpp:21-94 CalculateNbodyPotential (+ Force.cpp:124-159 smoothing)
template <bool ROWU> __global__ void k_potential(const Dev P)
{
    CELL(0, P.nr);
    const double x = P.Rmed[i] * P.cosphi[j];
    const double y = P.Rmed[i] * P.sinphi[j];
    double H;
    if (P.adiabatic && P.lazy_derived) { // k_adi_cs_h in registers
        const double cs = sqrt(P.gamma * (P.gamma - 1.0) * P.energy[IDX(i, j)] * fast_rcp(P.sigma[IDX(i, j)]));
        H = cs * (1.0 / sqrt(P.gamma)) * P.g_inv_omk[i];
    } else {
        H = P.scale_height[IDX(i, j)];
    }
    const double smooth = P.thickness_smoothing * H;
    double pot = 0.0;
    for (int k = 0; k < P.nbodies; ++k) {
        const double dx = x - P.bx[k];
        const double dy = y - P.by[k];
        const double dist_2 = dx * dx + dy * dy;
        const double d2s = dist_2 + smooth * smooth;
        const double inv_d = fast_rsqrt(d2s); // 1 / d_smoothed
        double klahr = 1.0;
        const double r_sm = P.brsm[k];
        if (r_sm > 0.0) {
            const double d_smoothed = d2s * inv_d;
            if (d_smoothed < r_sm) {
                const double q = d_smoothed / r_sm;
                klahr = ((q * q) * (q * q) - 2.0 * (q * q * q) + 2.0 * d_smoothed / r_sm);
            }
        }
        pot += -P.G * P.bm[k] * inv_d * klahr;
    }
    pot += -P.indirect_x * x - P.indirect_y * y;
    P.potential[IDX(i, j)] = pot;
}

// SourceEuler.cpp:325-372 momentum_update_radial
template <bool ROWU> __global__ void k_source_vr(const Dev P)
{
    CELL(P.one_no_ghost_vr, P.maxmo_no_ghost_vr - P.one_no_ghost_vr);
    const double dt = P.clk->dt;
    const int jn = JNEXT;
    double gradp = 2.0 / (P.sigma[IDX(i, j)] + P.sigma[IDX(i - 1, j)]);
    gradp *= (P.pressure[IDX(i, j)] - P.pressure[IDX(i - 1, j)]);
    gradp *= P.InvDiffRmed[i];
    const double gradphi = (P.potential[IDX(i, j)] - P.potential[IDX(i - 1, j)]) * P.InvDiffRmed[i];
    const double vsum =
        P.vazi[IDX(i, j)] + P.vazi[IDX(i, jn)] + P.vazi[IDX(i - 1, j)] + P.vazi[IDX(i - 1, jn)];
    const double vt = 0.25 * vsum + P.Rinf[i] * P.omega_frame;
    const double vt2 = vt * vt;
    const double centrifugal_accel = vt2 * P.InvRinf[i];
    P.vrad[IDX(i, j)] += dt * (-gradp - gradphi + centrifugal_accel);
}

// SourceEuler.cpp:375-428 momentum_update_azimuthal
template <bool ROWU> __global__ void k_source_va(const Dev P)
{
    CELL(P.zero_no_ghost, P.max_no_ghost - P.zero_no_ghost);
    const double dt = P.clk->dt;
    const int jp = JPREV;
    const double invdxtheta = 2.0 / (P.dphi * (P.Rsup[i] + P.Rinf[i]));
    const double gradp = 2.0 / (P.sigma[IDX(i, j)] + P.sigma[IDX(i, jp)]) *
                         (P.pressure[IDX(i, j)] - P.pressure[IDX(i, jp)]) * invdxtheta;
    const double gradphi = (P.potential[IDX(i, j)] - P.potential[IDX(i, jp)]) * invdxtheta;
    P.vazi[IDX(i, j)] = P.vazi[IDX(i, j)] + dt * (-gradp - gradphi);
}

// SourceEuler.cpp:459-493 compression_heating
template <bool ROWU> __global__ void k_compression_heating(const Dev P)
{
    CELL(0, P.nr - 1);
    const double dt = P.clk->dt;
    const int jn = JNEXT;
    const double DIV_V =
        (P.vrad[IDX(i + 1, j)] * P.Rinf[i + 1] - P.vrad[IDX(i, j)] * P.Rinf[i]) * P.InvDiffRsupRb[i] +
        (P.vazi[IDX(i, jn)] - P.vazi[IDX(i, j)]) * P.invdphi * P.InvRmed[i];
    const double e_old = P.energy[IDX(i, j)];
    P.energy[IDX(i, j)] = e_old * exp(-(P.gamma - 1.0) * dt * DIV_V);
}

// viscosity/artificial_viscosity.cpp:48-88 TW: Q_rr, Q_pp (+ dissipation)
template <bool ROWU> __global__ void k_tw_q(const Dev P)
{
    CELL(0, P.nr);
    const double dt = P.clk->dt;
    const int jn = JNEXT;
    const double vr0 = P.vrad[IDX(i, j)], vr1 = P.vrad[IDX(i + 1, j)];
    const double eps_rr = (vr1 - vr0) * P.InvDiffRsup[i];
    const double eps_pp =
        P.InvRmed[i] * ((P.vazi[IDX(i, jn)] - P.vazi[IDX(i, j)]) * P.invdphi + 0.5 * (vr1 + vr0));
    const double div_V = dmin(eps_rr + eps_pp, 0.0);
    const double Dr = P.Rinf[i + 1] - P.Rinf[i];
    const double rDphi = P.Rmed[i] * P.dphi;
    const double dx = P.nphi <= 16 ? dmin(Dr, rDphi) : dmax(Dr, rDphi);
    const double l_sq = (P.art_visc_factor * P.art_visc_factor) * (dx * dx);
    const double rho = P.sigma[IDX(i, j)];
    P.qr[IDX(i, j)] = l_sq * rho * -div_V * (eps_rr - 1.0 / 3.0 * div_V);
    P.qphi[IDX(i, j)] = l_sq * rho * -div_V * (eps_pp - 1.0 / 3.0 * div_V);
    if (P.adiabatic && P.art_visc_dissipation) {
        if (i > P.zero_no_ghost && i < P.max_no_ghost) {
            const double Qplus = -l_sq * div_V * rho * 1.0 / 3.0 *
                                 (eps_rr * eps_rr + eps_pp * eps_pp + (eps_rr - eps_pp) * (eps_rr - eps_pp));
            P.energy[IDX(i, j)] += Qplus * dt;
        }
    }
}
// viscosity/artificial_viscosity.cpp:90-117 TW: v_phi
template <bool ROWU> __global__ void k_tw_va(const Dev P)
{
    CELL(1, P.nr - 2);
    const double dt = P.clk->dt;
    const int jp = JPREV;
    const double sigma_phi_avg = 0.5 * (P.sigma[IDX(i, j)] + P.sigma[IDX(i, jp)]);
    const double dVp = 2.0 * dt / ((P.Rsup[i] + P.Rinf[i]) * sigma_phi_avg) *
                       (P.qphi[IDX(i, j)] - P.qphi[IDX(i, jp)]) * P.invdphi;
    P.vazi[IDX(i, j)] += dVp;
}
// viscosity/artificial_viscosity.cpp:119-139 TW: v_r
template <bool ROWU> __global__ void k_tw_vr(const Dev P)
{
    CELL(P.one_no_ghost_vr, P.maxmo_no_ghost_vr - P.one_no_ghost_vr);
    const double dt = P.clk->dt;
    const double sigma_r_avg = 0.5 * (P.sigma[IDX(i, j)] + P.sigma[IDX(i - 1, j)]);
    const double rm = P.Rmed[i], rmm = P.Rmed[i - 1];
    const double dVr = P.radial_viscosity_factor * dt / sigma_r_avg * 2.0 / (rm * rm - rmm * rmm) *
                       ((P.qr[IDX(i, j)] * rm - P.qr[IDX(i - 1, j)] * rmm) -
                        0.5 * (P.qphi[IDX(i, j)] + P.qphi[IDX(i - 1, j)]) * (rm - rmm));
    P.vrad[IDX(i, j)] += dVr;
}
// viscosity/artificial_viscosity.cpp:165-189 SN: q_r, q_phi
template <bool ROWU> __global__ void k_sn_q(const Dev P)
{
    CELL(0, P.nr);
    const int jn = JNEXT;
    const double C2 = P.art_visc_factor * P.art_visc_factor;
    const double rho = P.sigma[IDX(i, j)];
    const double dv_r = P.vrad[IDX(i + 1, j)] - P.vrad[IDX(i, j)];
    P.qr[IDX(i, j)] = dv_r < 0.0 ? C2 * rho * (dv_r * dv_r) : 0.0;
    const double dv_phi = P.vazi[IDX(i, jn)] - P.vazi[IDX(i, j)];
    P.qphi[IDX(i, j)] = dv_phi < 0.0 ? C2 * rho * (dv_phi * dv_phi) : 0.0;
}
// viscosity/artificial_viscosity.cpp:194-218 SN: energy dissipation
template <bool ROWU> __global__ void k_sn_e(const Dev P)
{
    CELL(P.zero_no_ghost, P.max_no_ghost - P.zero_no_ghost);
    const double dt = P.clk->dt;
    const int jn = JNEXT;
    const double invdxtheta = 1.0 / (P.dphi * P.Rmed[i]);
    const double dv_r = P.vrad[IDX(i + 1, j)] - P.vrad[IDX(i, j)];
    const double dv_phi = P.vazi[IDX(i, jn)] - P.vazi[IDX(i, j)];
    P.energy[IDX(i, j)] = P.energy[IDX(i, j)] - dt * P.qr[IDX(i, j)] * dv_r * P.InvDiffRsup[i] -
                          dt * P.qphi[IDX(i, j)] * dv_phi * invdxtheta;
}
// viscosity/artificial_viscosity.cpp:220-230 SN: v_r
template <bool ROWU> __global__ void k_sn_vr(const Dev P)
{
    CELL(P.one_no_ghost_vr, P.maxmo_no_ghost_vr - P.one_no_ghost_vr);
    const double dt = P.clk->dt;
    P.vrad[IDX(i, j)] = P.vrad[IDX(i, j)] - dt * 2.0 / (P.sigma[IDX(i, j)] + P.sigma[IDX(i - 1, j)]) *
                                                (P.qr[IDX(i, j)] - P.qr[IDX(i - 1, j)]) * P.InvDiffRmed[i];
}
// viscosity/artificial_viscosity.cpp:232-248 SN: v_phi
template <bool ROWU> __global__ void k_sn_va(const Dev P)
{
    CELL(P.zero_no_ghost, P.max_no_ghost - P.zero_no_ghost);
    const double dt = P.clk->dt;
    const int jp = JPREV;
    const double invdxtheta = 1.0 / (P.dphi * P.Rmed[i]);
    P.vazi[IDX(i, j)] = P.vazi[IDX(i, j)] - dt * 2.0 / (P.sigma[IDX(i, j)] + P.sigma[IDX(i, jp)]) *
                                                (P.qphi[IDX(i, j)] - P.qphi[IDX(i, jp)]) * invdxtheta;
}

// SourceEuler.cpp:136-202 assure_temperature_range
__device__ __forceinline__ double clamp_energy(const Dev &P, double e, double rho)
{
    const double e_min = P.tmin * rho / P.mu * P.Rgas / (P.gamma - 1.0);
    const double e_max = P.tmax * rho / P.mu * P.Rgas / (P.gamma - 1.0);
    if (!(e > e_min))
        e = e_min;
    if (!(e < e_max))
        e = e_max;
    return e;
}
// the same with the per-unit-Sigma bounds formed once on the host (the quotient chain above costs
// four IEEE divisions per cell; the bounds only matter where they bind, to 1 ulp)
__device__ __forceinline__ double clamp_energy_fast(const Dev &P, double e, double rho)
{
    const double e_min = P.emin_fac * rho;
    const double e_max = P.emax_fac * rho;
    if (!(e > e_min))
        e = e_min;
    if (!(e < e_max))
        e = e_max;
    return e;
}
template <bool ROWU> __global__ void k_temperature_range(const Dev P)
{
    CELL(0, P.nr);
    P.energy[IDX(i, j)] = clamp_energy(P, P.energy[IDX(i, j)], P.sigma[IDX(i, j)]);
}

// SourceEuler.cpp:1054-1092 compute_sound_speed_normal + :1218-1251 compute_scale_height_old
// (adiabatic branch; the isothermal values are set once by k_iso_cs_h)
template <bool ROWU> __global__ void k_adi_cs_h(const Dev P)
{
    CELL(0, P.nr);
    const double cs = sqrt(P.gamma * (P.gamma - 1.0) * P.energy[IDX(i, j)] / P.sigma[IDX(i, j)]);
    P.soundspeed[IDX(i, j)] = cs;
    const double r = P.Rmed[i];
    const double inv_omega_kepler = 1.0 / sqrt(P.G * P.Mc / (r * r * r));
    P.scale_height[IDX(i, j)] = cs / (sqrt(P.gamma)) * inv_omega_kepler;
}
template <bool ROWU> __global__ void k_iso_cs_h(const Dev P, const double *cs_ring)
{
    CELL(0, P.nr);
    const double cs = cs_ring[i]; // h0 r^beta sqrt(GM/r), evaluated on the host (libm pow)
    P.soundspeed[IDX(i, j)] = cs;
    const double r = P.Rmed[i];
    const double inv_omega_kepler = 1.0 / sqrt(P.G * P.Mc / (r * r * r));
    P.scale_height[IDX(i, j)] = cs * inv_omega_kepler;
}
// viscosity/viscosity.cpp:98-137 update_viscosity
template <bool ROWU> __global__ void k_viscosity(const Dev P)
{
    CELL(0, P.nr);
    if (P.alpha_viscosity)
        P.viscosity[IDX(i, j)] = P.alpha * P.scale_height[IDX(i, j)] * P.soundspeed[IDX(i, j)];
    else
        P.viscosity[IDX(i, j)] = P.nu_const;
}
// SourceEuler.cpp:1442-1473 compute_pressure
template <bool ROWU> __global__ void k_pressure(const Dev P)
{
    CELL(0, P.nr);
    if (P.adiabatic) {
        P.pressure[IDX(i, j)] = (P.gamma - 1.0) * P.energy[IDX(i, j)];
    } else {
        const double cs = P.soundspeed[IDX(i, j)];
        P.pressure[IDX(i, j)] = P.sigma[IDX(i, j)] * (cs * cs);
    }
}
// SourceEuler.cpp:1475-1505 compute_temperature
template <bool ROWU> __global__ void k_temperature(const Dev P)
{
    CELL(0, P.nr);
    if (P.adiabatic) {
        const double c_v_inv = P.mu / P.Rgas * (P.gamma - 1.0);
        P.temperature[IDX(i, j)] = c_v_inv * P.energy[IDX(i, j)] / P.sigma[IDX(i, j)];
    } else {
        P.temperature[IDX(i, j)] = P.mu / P.Rgas * P.pressure[IDX(i, j)] / P.sigma[IDX(i, j)];
    }
}

// viscosity/viscosity.cpp:149-209: div v, tau_rr, tau_phiphi
template <bool ROWU> __global__ void k_stress_diag(const Dev P)
{
    CELL(0, P.nr);
    const int jn = JNEXT;
    const double vr0 = P.vrad[IDX(i, j)], vr1 = P.vrad[IDX(i + 1, j)];
    const double dva = P.vazi[IDX(i, jn)] - P.vazi[IDX(i, j)];
    const double divv =
        (vr1 * P.Rinf[i + 1] - vr0 * P.Rinf[i]) * P.InvDiffRsupRb[i] + dva * P.invdphi * P.InvRmed[i];
    P.divv[IDX(i, j)] = divv;
    const double nu = P.viscosity[IDX(i, j)], sigma = P.sigma[IDX(i, j)];
    const double drr = (vr1 - vr0) * P.InvDiffRsup[i];
    P.trr[IDX(i, j)] = 2.0 * nu * sigma * (drr - 1.0 / 3.0 * divv);
    const double dpp = dva * P.invdphi * P.InvRmed[i] + 0.5 * (vr1 + vr0) * P.InvRmed[i];
    P.tpp[IDX(i, j)] = 2.0 * nu * sigma * (dpp - 1.0 / 3.0 * divv);
}
// viscosity/viscosity.cpp:211-254: tau_rphi on rows 1..Nr-1 (rows 0 and Nr stay 0)
template <bool ROWU> __global__ void k_stress_rphi(const Dev P)
{
    CELL(1, P.nr - 1);
    const int jp = JPREV;
    const double dvazirdr =
        (P.vazi[IDX(i, j)] * P.InvRmed[i] - P.vazi[IDX(i - 1, j)] * P.InvRmed[i - 1]) * P.InvDiffRmed[i];
    const double dvrdphi = (P.vrad[IDX(i, j)] - P.vrad[IDX(i, jp)]) * P.invdphi;
    const double drp = P.Rinf[i] * dvazirdr + dvrdphi * P.InvRinf[i];
    const double nu = 0.25 * (P.viscosity[IDX(i, j)] + P.viscosity[IDX(i - 1, j)] +
                              P.viscosity[IDX(i, jp)] + P.viscosity[IDX(i - 1, jp)]);
    const double sigma = 0.25 * (P.sigma[IDX(i, j)] + P.sigma[IDX(i - 1, j)] + P.sigma[IDX(i, jp)] +
                                 P.sigma[IDX(i - 1, jp)]);
    P.trp[IDX(i, j)] = nu * sigma * drp;
}
// viscosity/viscosity.cpp:368-394: v_phi update
template <bool ROWU> __global__ void k_visc_va(const Dev P)
{
    CELL(1, P.nr - 2);
    const double dt = P.clk->dt;
    const int jp = JPREV;
    const double sigma_avg = 0.5 * (P.sigma[IDX(i, j)] + P.sigma[IDX(i, jp)]);
    const double ra1 = P.Rinf[i + 1], ra0 = P.Rinf[i];
    const double dVp = dt * P.InvRmed[i] / (sigma_avg) *
                       ((2.0 / (ra1 * ra1 - ra0 * ra0)) *
                            (ra1 * ra1 * P.trp[IDX(i + 1, j)] - ra0 * ra0 * P.trp[IDX(i, j)]) +
                        (P.tpp[IDX(i, j)] - P.tpp[IDX(i, jp)]) * P.invdphi);
    P.vazi[IDX(i, j)] += dVp;
}
// viscosity/viscosity.cpp:396-421: v_r update
template <bool ROWU> __global__ void k_visc_vr(const Dev P)
{
    CELL(P.one_no_ghost_vr, P.maxmo_no_ghost_vr - P.one_no_ghost_vr);
    const double dt = P.clk->dt;
    const int jn = JNEXT;
    const double sigma_avg = 0.5 * (P.sigma[IDX(i, j)] + P.sigma[IDX(i - 1, j)]);
    const double dVr = dt / (sigma_avg)*P.radial_viscosity_factor * 2.0 / (P.Rmed[i] + P.Rmed[i - 1]) *
                       ((P.Rmed[i] * P.trr[IDX(i, j)] - P.Rmed[i - 1] * P.trr[IDX(i - 1, j)]) * P.InvDiffRmed[i] +
                        (P.trp[IDX(i, jn)] - P.trp[IDX(i, j)]) * P.invdphi -
                        0.5 * (P.tpp[IDX(i, j)] + P.tpp[IDX(i - 1, j)]));
    P.vrad[IDX(i, j)] += dVr;
}

// SourceEuler.cpp:614-630 calculate_qplus + :496-536 viscous_heating and
// :931-950 calculate_qminus (all cooling terms are out of scope: Q- = 0)
template <bool ROWU> __global__ void k_qplus_qminus(const Dev P)
{
    CELL(0, P.nr);
    double qplus = 0.0;
    if (P.heating_viscous && i >= 1 && i < P.nr - 1) {
        const double nu = P.viscosity[IDX(i, j)];
        if (nu != 0.0) {
            const int jn = JNEXT;
            const double tau_r_phi = 0.25 * (P.trp[IDX(i, j)] + P.trp[IDX(i + 1, j)] + P.trp[IDX(i, jn)] +
                                             P.trp[IDX(i + 1, jn)]);
            const double trr = P.trr[IDX(i, j)], tpp = P.tpp[IDX(i, j)], dv = P.divv[IDX(i, j)];
            const double sigma = P.sigma[IDX(i, j)];
            double q = 1.0 / (2.0 * nu * sigma) * (trr * trr + 2 * (tau_r_phi * tau_r_phi) + tpp * tpp);
            q += (2.0 / 9.0) * nu * sigma * (dv * dv);
            q *= P.heating_viscous_factor;
            qplus += q;
        }
    }
    P.qplus[IDX(i, j)] = qplus;
    P.qminus[IDX(i, j)] = 0.0;
}
__device__ __forceinline__ double substep3_alpha(const Dev &P, double H, double sigma, double energy)
{
    const double b = P.mu * (P.gamma - 1.0) / (P.Rgas * sigma);
    const double b2 = b * b;
    return 1.0 + 2.0 * H * 4.0 * P.sigma_sb / P.c_light * (b2 * b2) * (energy * energy * energy);
}
// opacity.cpp:45-168 lin(): Lin & Papaloizou (1985), cgs in / cgs out
__device__ double opacity_lin(double density, double temperature)
{
    const double power1 = 4.44444444e-2, power2 = 2.381e-2, power3 = 2.267e-1;
    const double t234 = 1.6e3, t456 = 5.7e3, t678 = 2.28e6;
    const double ak1 = 2.e-4, ak2 = 2.e16, ak3 = 5.e-3;
    const double bk3 = 50., bk4 = 2.e-2, bk5 = 2.e4, bk6 = 1.e4, bk7 = 1.5e10, bk8 = 0.348;
    if (temperature > t234 * pow(density, power1)) {
        const double ts4 = 1.e-4 * temperature;
        const double density13 = pow(density, 1.0 / 3.0);
        const double density23 = density13 * density13;
        const double ts42 = ts4 * ts4;
        const double ts44 = ts42 * ts42;
        const double ts48 = ts44 * ts44;
        if (temperature > t456 * pow(density, power2)) {
            if ((temperature < t678 * pow(density, power3)) || (density <= 1e-10)) {
                const double o5 = bk5 * density23 * ts42 * ts4;
                const double o6 = bk6 * density13 * ts48 * ts42;
                const double o7 = bk7 * density / (ts42 * sqrt(ts4));
                const double o6an = o6 * o6, o7an = o7 * o7;
                return pow(pow(o6an * o7an / (o6an + o7an), 2.0) +
                               pow(o5 / (1.0 + pow(ts4 / (1.1 * pow(density, 0.04762)), 10.0)), 4.0),
                           0.25);
            } else {
                const double o7 = bk7 * density / (ts42 * sqrt(ts4));
                const double o8 = bk8;
                const double o7an = o7 * o7, o8an = o8 * o8;
                return pow(o7an * o7an + o8an * o8an, 0.25);
            }
        } else {
            const double o3 = bk3 * ts4;
            const double o4 = bk4 * density23 / (ts48 * ts4);
            const double o5 = bk5 * density23 * ts42 * ts4;
            const double o4an = pow(o4, 4.0), o3an = pow(o3, 4.0);
            return pow((o4an * o3an / (o4an + o3an)) + pow(o5 / (1.0 + 6.561e-5 / ts48), 4.0), 0.25);
        }
    } else {
        const double t2 = temperature * temperature;
        const double t4 = t2 * t2;
        const double t8 = t4 * t4;
        const double t10 = t8 * t2;
        const double o1 = ak1 * t2;
        const double o2 = ak2 * temperature / t8;
        const double o3 = ak3 * temperature;
        const double o1an = o1 * o1, o2an = o2 * o2;
        return pow(pow(o1an * o2an / (o1an + o2an), 2.0) + pow(o3 / (1 + 1.e22 / t10), 4.0), 0.25);
    }
}
// midplane_density + kappa_eff at one cell (compute.cpp:17-87): the effective optical depth
__device__ __forceinline__ double tau_eff_of(const Dev &P, double sigma, double H, double temperature)
{
    const double rho = sigma / (P.density_factor * H);
    const double temperatureCGS = temperature * P.temperature_cgs;
    double kappa;
    if (P.opacity == FCPT_OPACITY_LIN)
        kappa = opacity_lin(rho * P.density_cgs, temperatureCGS) * (1.0 / P.opacity_cgs);
    else if (P.opacity == FCPT_OPACITY_CONST)
        kappa = P.kappa_const;
    else
        kappa = P.kappa_const * (temperatureCGS * temperatureCGS);
    kappa = P.kappa_factor * kappa;
    const double tau = P.tau_factor * (1.0 / P.density_factor) * kappa * sigma;
    if (P.opacity == FCPT_OPACITY_SIMPLE)
        return 3.0 / 8.0 * tau; // D'Angelo et al. 2003 eq. (28)
    return 3.0 / 8.0 * tau + sqrt(3.0) / 4.0 + 1.0 / (4.0 * tau + P.tau_min);
}
// calculate_qminus (SourceEuler.cpp:931-950) at one cell of rows [1, Nr-1): beta cooling
// (thermal_relaxation :632-786, without the opacity-based Ziampras variants) and thermal surface
// cooling (:790-820).  tau_eff is returned for SubStep3's low-density branch (0 without surface cooling).
struct Cooling {
    double qminus, tau_eff;
};
__device__ __forceinline__ Cooling cooling_terms(const Dev &P, int i, int cell, double sigma, double energy, double H)
{
    Cooling c = {0.0, 0.0};
    if (P.cooling_beta && !(P.cooling_at_init && P.cooling_beta_reference == FCPT_BETAREF_REFERENCE)) {
        double beta_inv = 1 / P.cooling_beta_value;
        if (P.cooling_beta_ramp_up > 0.0) {
            const double t = P.clk->time - (P.kick_time_shift ? P.clk->dt : 0.0);
            const double x = 2 * t / P.cooling_beta_ramp_up;
            beta_inv = beta_inv * (1 - exp(-(x * x)));
        }
        double delta_E = energy;
        if (P.cooling_beta_reference == FCPT_BETAREF_REFERENCE) {
            delta_E -= P.energy0[cell] / P.sigma0[cell] * sigma;
        } else if (P.cooling_beta_reference == FCPT_BETAREF_MODEL) {
            const double E0 = 1.0 / (P.gamma - 1.0) * (P.aspect_ratio * P.aspect_ratio) *
                              pow(P.Rmed[i], 2.0 * P.flaring_index - 1.0) * P.G * P.Mc * sigma;
            delta_E -= E0;
        } else if (P.cooling_beta_reference == FCPT_BETAREF_FLOOR) {
            delta_E -= P.tmin * sigma / P.mu * P.Rgas / (P.gamma - 1.0);
        }
        c.qminus += delta_E * P.g_omk[i] * beta_inv;
    }
    if (P.cooling_surface) {
        const double T = P.mu / P.Rgas * (P.gamma - 1.0) * energy / sigma; // compute_temperature
        c.tau_eff = tau_eff_of(P, sigma, H, T);
        const double T2 = T * T, Tm2 = P.tmin * P.tmin;
        c.qminus += P.cooling_radiative_factor * 2 * P.sigma_sb * (T2 * T2 - Tm2 * Tm2) / c.tau_eff;
    }
    return c;
}
// SourceEuler.cpp:1000-1048: energy update of SubStep3 (update_energy != 0) or only the
// alpha rescaling of compute_heating_cooling_for_CFL (:1520-1545)
template <bool ROWU> __global__ void k_substep3(const Dev P, int update_energy)
{
    CELL(1, P.nr - 2);
    const double dt = P.clk->dt;
    const double H = P.scale_height[IDX(i, j)];
    const double sigma = P.sigma[IDX(i, j)];
    const double energy = P.energy[IDX(i, j)];
    const double alpha = substep3_alpha(P, H, sigma, energy);
    const Cooling cool = cooling_terms(P, i, IDX(i, j), sigma, energy, H);
    const double Qplus = P.qplus[IDX(i, j)] / alpha;
    double Qminus = (P.qminus[IDX(i, j)] + cool.qminus) / alpha;
    if (update_energy) {
        double energy_new = energy + dt * (Qplus - Qminus);
        const double SigmaFloor = 10.0 * P.sigma0_val * P.sigma_floor_rel;
        if (sigma < SigmaFloor) {
            // the energy at which the current heating and cooling balance (0 without surface cooling: tau_eff = 0)
            const double e4 = Qplus * cool.tau_eff / (2.0 * P.sigma_sb);
            energy_new = sqrt(sqrt(e4)) * (P.Rgas / P.mu * sigma / (P.gamma - 1.0));
            Qminus = Qplus;
        }
        P.energy[IDX(i, j)] = energy_new;
    }
    P.qplus[IDX(i, j)] = Qplus;
    P.qminus[IDX(i, j)] = Qminus;
}


// ===========================================================================
// Fused source step (default path).  The reference's source / artificial-viscosity /
// viscous-stress substeps are 13 loop nests that stream ~45 grids; here they are three
// out-of-place kernels that stream 17: intermediate tensors (Q_rr, Q_pp, div v, tau_*) are
// re-evaluated from the velocities in registers instead of being stored.
//   k_src_fused : (v_r, v_phi)   -> (v_r_b, v_phi_b)   S1 + S2
//   k_av_fused  : (v_r_b,v_phi_b)-> (v_r, v_phi) [,e]  S3 + artificial viscosity (+ T range)
//   k_visc_fused: (v_r, v_phi)   -> (v_r_b, v_phi_b)   stress tensor + viscous update [+ Q+]
// Row ranges are those of the individual loops; rows outside a range are copied through.

// SourceEuler.cpp:325-428 momentum_update_radial + momentum_update_azimuthal
template <bool ROWU> __global__ void k_src_fused(const Dev P)
{
    CELL(0, P.nr + 1);
    const double dt = P.clk->dt;
    const int jn = JNEXT, jp = JPREV;
    double vr = P.vrad[IDX(i, j)];
    if (i >= P.one_no_ghost_vr && i < P.maxmo_no_ghost_vr) {
        double gradp = 2.0 / (P.sigma[IDX(i, j)] + P.sigma[IDX(i - 1, j)]);
        gradp *= (P.pressure[IDX(i, j)] - P.pressure[IDX(i - 1, j)]);
        gradp *= P.InvDiffRmed[i];
        const double gradphi = (P.potential[IDX(i, j)] - P.potential[IDX(i - 1, j)]) * P.InvDiffRmed[i];
        const double vsum =
            P.vazi[IDX(i, j)] + P.vazi[IDX(i, jn)] + P.vazi[IDX(i - 1, j)] + P.vazi[IDX(i - 1, jn)];
        const double vt = 0.25 * vsum + P.Rinf[i] * P.omega_frame;
        const double vt2 = vt * vt;
        vr += dt * (-gradp - gradphi + vt2 * P.InvRinf[i]);
    }
    P.vrad_b[IDX(i, j)] = vr;
    if (i < P.nr) {
        double va = P.vazi[IDX(i, j)];
        if (i >= P.zero_no_ghost && i < P.max_no_ghost) {
            const double invdxtheta = 2.0 / (P.dphi * (P.Rsup[i] + P.Rinf[i]));
            const double gradp = 2.0 / (P.sigma[IDX(i, j)] + P.sigma[IDX(i, jp)]) *
                                 (P.pressure[IDX(i, j)] - P.pressure[IDX(i, jp)]) * invdxtheta;
            const double gradphi = (P.potential[IDX(i, j)] - P.potential[IDX(i, jp)]) * invdxtheta;
            va = va + dt * (-gradp - gradphi);
        }
        P.vazi_b[IDX(i, j)] = va;
    }
}

struct TwQ {
    double qrr, qpp, eps_rr, eps_pp, div_V, l_sq;
};
// artificial_viscosity.cpp:48-77 at cell (i, j), velocities from the *_b buffers
__device__ __forceinline__ TwQ tw_q_at(const Dev &P, int i, int j)
{
    const int jn = JNEXT;
    const double vr0 = P.vrad_b[IDX(i, j)], vr1 = P.vrad_b[IDX(i + 1, j)];
    TwQ q;
    q.eps_rr = (vr1 - vr0) * P.InvDiffRsup[i];
    q.eps_pp = P.InvRmed[i] * ((P.vazi_b[IDX(i, jn)] - P.vazi_b[IDX(i, j)]) * P.invdphi + 0.5 * (vr1 + vr0));
    q.div_V = dmin(q.eps_rr + q.eps_pp, 0.0);
    const double Dr = P.Rinf[i + 1] - P.Rinf[i];
    const double rDphi = P.Rmed[i] * P.dphi;
    const double dx = P.nphi <= 16 ? dmin(Dr, rDphi) : dmax(Dr, rDphi);
    q.l_sq = (P.art_visc_factor * P.art_visc_factor) * (dx * dx);
    const double rho = P.sigma[IDX(i, j)];
    q.qrr = q.l_sq * rho * -q.div_V * (q.eps_rr - 1.0 / 3.0 * q.div_V);
    q.qpp = q.l_sq * rho * -q.div_V * (q.eps_pp - 1.0 / 3.0 * q.div_V);
    return q;
}
// artificial_viscosity.cpp:165-189 at cell (i, j)
__device__ __forceinline__ void sn_q_at(const Dev &P, int i, int j, double &qr, double &qphi)
{
    const int jn = JNEXT;
    const double C2 = P.art_visc_factor * P.art_visc_factor;
    const double rho = P.sigma[IDX(i, j)];
    const double dv_r = P.vrad_b[IDX(i + 1, j)] - P.vrad_b[IDX(i, j)];
    qr = dv_r < 0.0 ? C2 * rho * (dv_r * dv_r) : 0.0;
    const double dv_phi = P.vazi_b[IDX(i, jn)] - P.vazi_b[IDX(i, j)];
    qphi = dv_phi < 0.0 ? C2 * rho * (dv_phi * dv_phi) : 0.0;
}

// compression_heating (SourceEuler.cpp:459-493) + update_with_artificial_viscosity
// (artificial_viscosity.cpp:11-250) incl. the temperature floor/ceiling
template <bool ROWU> __global__ void k_av_fused(const Dev P)
{
    CELL(0, P.nr + 1);
    const double dt = P.clk->dt;
    const int jn = JNEXT, jp = JPREV;
    const int nr = P.nr;
    double vr = P.vrad_b[IDX(i, j)];
    if (i == nr) {
        P.vrad[IDX(i, j)] = vr;
        return;
    }
    double va = P.vazi_b[IDX(i, j)];
    double e = P.adiabatic ? P.energy[IDX(i, j)] : 0.0;
    if (P.adiabatic && i < nr - 1) { // compression heating, rows [0, Nr-1)
        const double DIV_V =
            (P.vrad_b[IDX(i + 1, j)] * P.Rinf[i + 1] - vr * P.Rinf[i]) * P.InvDiffRsupRb[i] +
            (P.vazi_b[IDX(i, jn)] - va) * P.invdphi * P.InvRmed[i];
        e = e * exp(-(P.gamma - 1.0) * dt * DIV_V);
    }
    const bool upd_vr = i >= P.one_no_ghost_vr && i < P.maxmo_no_ghost_vr;
    if (P.art_visc == FCPT_ARTVISC_TW) {
        const TwQ q = tw_q_at(P, i, j);
        if (P.adiabatic && P.art_visc_dissipation && i > P.zero_no_ghost && i < P.max_no_ghost) {
            const double Qplus = -q.l_sq * q.div_V * P.sigma[IDX(i, j)] * 1.0 / 3.0 *
                                 (q.eps_rr * q.eps_rr + q.eps_pp * q.eps_pp +
                                  (q.eps_rr - q.eps_pp) * (q.eps_rr - q.eps_pp));
            e += Qplus * dt;
        }
        if (i >= 1 && i < nr - 1) {
            const TwQ qm = tw_q_at(P, i, jp);
            const double sigma_phi_avg = 0.5 * (P.sigma[IDX(i, j)] + P.sigma[IDX(i, jp)]);
            va += 2.0 * dt / ((P.Rsup[i] + P.Rinf[i]) * sigma_phi_avg) * (q.qpp - qm.qpp) * P.invdphi;
        }
        if (upd_vr) {
            const TwQ qi = tw_q_at(P, i - 1, j);
            const double sigma_r_avg = 0.5 * (P.sigma[IDX(i, j)] + P.sigma[IDX(i - 1, j)]);
            const double rm = P.Rmed[i], rmm = P.Rmed[i - 1];
            vr += P.radial_viscosity_factor * dt / sigma_r_avg * 2.0 / (rm * rm - rmm * rmm) *
                  ((q.qrr * rm - qi.qrr * rmm) - 0.5 * (q.qpp + qi.qpp) * (rm - rmm));
        }
    } else if (P.art_visc == FCPT_ARTVISC_SN) {
        double qr, qphi;
        sn_q_at(P, i, j, qr, qphi);
        const double invdxtheta = 1.0 / (P.dphi * P.Rmed[i]);
        const bool row_va = i >= P.zero_no_ghost && i < P.max_no_ghost;
        if (P.adiabatic && P.art_visc_dissipation && row_va) {
            const double dv_r = P.vrad_b[IDX(i + 1, j)] - vr;
            const double dv_phi = P.vazi_b[IDX(i, jn)] - va;
            e = e - dt * qr * dv_r * P.InvDiffRsup[i] - dt * qphi * dv_phi * invdxtheta;
        }
        if (upd_vr) {
            double qr_m, qphi_m;
            sn_q_at(P, i - 1, j, qr_m, qphi_m);
            vr = vr - dt * 2.0 / (P.sigma[IDX(i, j)] + P.sigma[IDX(i - 1, j)]) * (qr - qr_m) * P.InvDiffRmed[i];
        }
        if (row_va) {
            double qr_p, qphi_p;
            sn_q_at(P, i, jp, qr_p, qphi_p);
            va = va - dt * 2.0 / (P.sigma[IDX(i, j)] + P.sigma[IDX(i, jp)]) * (qphi - qphi_p) * invdxtheta;
        }
    }
    P.vrad[IDX(i, j)] = vr;
    P.vazi[IDX(i, j)] = va;
    if (P.adiabatic) {
        if (P.art_visc_dissipation)
            e = clamp_energy(P, e, P.sigma[IDX(i, j)]);
        P.energy[IDX(i, j)] = e;
    }
}

struct TauDiag {
    double divv, trr, tpp;
};
// viscosity.cpp:149-209 at cell (i, j), 0 <= i < Nr
__device__ __forceinline__ TauDiag tau_diag_at(const Dev &P, int i, int j)
{
    const int jn = JNEXT;
    const double vr0 = P.vrad[IDX(i, j)], vr1 = P.vrad[IDX(i + 1, j)];
    const double dva = P.vazi[IDX(i, jn)] - P.vazi[IDX(i, j)];
    TauDiag t;
    t.divv = (vr1 * P.Rinf[i + 1] - vr0 * P.Rinf[i]) * P.InvDiffRsupRb[i] + dva * P.invdphi * P.InvRmed[i];
    const double nu = P.viscosity[IDX(i, j)], sigma = P.sigma[IDX(i, j)];
    const double drr = (vr1 - vr0) * P.InvDiffRsup[i];
    t.trr = 2.0 * nu * sigma * (drr - 1.0 / 3.0 * t.divv);
    const double dpp = dva * P.invdphi * P.InvRmed[i] + 0.5 * (vr1 + vr0) * P.InvRmed[i];
    t.tpp = 2.0 * nu * sigma * (dpp - 1.0 / 3.0 * t.divv);
    return t;
}
// viscosity.cpp:211-254 at corner (i, j); rows 0 and Nr are never written (stay 0)
__device__ __forceinline__ double tau_rp_at(const Dev &P, int i, int j)
{
    if (i < 1 || i > P.nr - 1)
        return 0.0;
    const int jp = JPREV;
    const double dvazirdr =
        (P.vazi[IDX(i, j)] * P.InvRmed[i] - P.vazi[IDX(i - 1, j)] * P.InvRmed[i - 1]) * P.InvDiffRmed[i];
    const double dvrdphi = (P.vrad[IDX(i, j)] - P.vrad[IDX(i, jp)]) * P.invdphi;
    const double drp = P.Rinf[i] * dvazirdr + dvrdphi * P.InvRinf[i];
    const double nu = 0.25 * (P.viscosity[IDX(i, j)] + P.viscosity[IDX(i - 1, j)] + P.viscosity[IDX(i, jp)] +
                              P.viscosity[IDX(i - 1, jp)]);
    const double sigma =
        0.25 * (P.sigma[IDX(i, j)] + P.sigma[IDX(i - 1, j)] + P.sigma[IDX(i, jp)] + P.sigma[IDX(i - 1, jp)]);
    return nu * sigma * drp;
}
// compute_viscous_stress_tensor + update_velocities_with_viscosity (viscosity.cpp:139-426)
// and, for the energy equation, viscous_heating (SourceEuler.cpp:496-536) into QPLUS
template <bool ROWU> __global__ void k_visc_fused(const Dev P)
{
    CELL(0, P.nr + 1);
    const double dt = P.clk->dt;
    const int jn = JNEXT, jp = JPREV;
    const int nr = P.nr;
    double vr = P.vrad[IDX(i, j)];
    if (i == nr) {
        P.vrad_b[IDX(i, j)] = vr;
        return;
    }
    double va = P.vazi[IDX(i, j)];
    const TauDiag t = tau_diag_at(P, i, j);
    const double trp = tau_rp_at(P, i, j);
    double trp_ip = 0.0;
    const bool row_va = i >= 1 && i < nr - 1;
    if (row_va) {
        trp_ip = tau_rp_at(P, i + 1, j);
        const TauDiag tjp = tau_diag_at(P, i, jp);
        const double sigma_avg = 0.5 * (P.sigma[IDX(i, j)] + P.sigma[IDX(i, jp)]);
        const double ra1 = P.Rinf[i + 1], ra0 = P.Rinf[i];
        va += dt * P.InvRmed[i] / (sigma_avg) *
              ((2.0 / (ra1 * ra1 - ra0 * ra0)) * (ra1 * ra1 * trp_ip - ra0 * ra0 * trp) +
               (t.tpp - tjp.tpp) * P.invdphi);
    }
    double trp_jn = 0.0;
    if (i >= P.one_no_ghost_vr && i < P.maxmo_no_ghost_vr) {
        trp_jn = tau_rp_at(P, i, jn);
        const TauDiag tim = tau_diag_at(P, i - 1, j);
        const double sigma_avg = 0.5 * (P.sigma[IDX(i, j)] + P.sigma[IDX(i - 1, j)]);
        vr += dt / (sigma_avg)*P.radial_viscosity_factor * 2.0 / (P.Rmed[i] + P.Rmed[i - 1]) *
              ((P.Rmed[i] * t.trr - P.Rmed[i - 1] * tim.trr) * P.InvDiffRmed[i] + (trp_jn - trp) * P.invdphi -
               0.5 * (t.tpp + tim.tpp));
    }
    P.vrad_b[IDX(i, j)] = vr;
    P.vazi_b[IDX(i, j)] = va;
    if (P.adiabatic) {
        double qplus = 0.0;
        if (P.heating_viscous && row_va) {
            const double nu = P.viscosity[IDX(i, j)];
            if (nu != 0.0) {
                if (!(i >= P.one_no_ghost_vr && i < P.maxmo_no_ghost_vr))
                    trp_jn = tau_rp_at(P, i, jn);
                const double tau_r_phi = 0.25 * (trp + trp_ip + trp_jn + tau_rp_at(P, i + 1, jn));
                const double sigma = P.sigma[IDX(i, j)];
                double q = 1.0 / (2.0 * nu * sigma) * (t.trr * t.trr + 2 * (tau_r_phi * tau_r_phi) + t.tpp * t.tpp);
                q += (2.0 / 9.0) * nu * sigma * (t.divv * t.divv);
                q *= P.heating_viscous_factor;
                qplus += q;
            }
        }
        P.qplus[IDX(i, j)] = qplus;
        P.qminus[IDX(i, j)] = 0.0;
    }
}


// ===========================================================================
// Wave-marching source step (isothermal EOS): the whole chain
//   S1+S2 -> artificial viscosity -> stress tensor -> viscous update
// in ONE pass over memory.  A wavefront owns 64 consecutive phi cells (phi neighbours by
// wavefront shuffle) and marches outward ring by ring; every intermediate (v after the
// source terms, Q_rr/Q_pp, v after artificial viscosity, div v, tau_*) lives in a rolling
// register window of 2-4 rings, so each input ring (Sigma, Phi, v_r, v_phi) is read once
// and each output ring (v_r, v_phi) written once: 6 doubles per cell.
// Stage lags for the newest loaded ring m:
//   A  v1(m)            source terms                      (SourceEuler.cpp:325-428)
//   B  Q(m-1)           TW / SN artificial pressure        (artificial_viscosity.cpp:48-77,165-189)
//   C  v2(m-1)          artificial-viscosity update        (artificial_viscosity.cpp:90-139,220-248)
//   D  tau_diag(m-2), tau_rphi(m-1)                        (viscosity.cpp:149-254)
//   E  v3(m-2) -> out   viscous update                     (viscosity.cpp:368-421)
// Lane validity erodes by one cell per phi-coupled stage: lanes 3..61 of a segment are
// final, segments advance by MARCH_VALID = 59 cells.  A chunk of MARCH_ROWS output rings
// needs 5 extra input rings of warm-up.
#define MARCH_VALID 59
#define MARCH_LO 3

template <int AV> // 0: none, 1: TW, 2: SN
__global__ void __launch_bounds__(256) k_source_march(const Dev P, int segs, int rows_per_chunk, int ring_sums)
{
    const int lane = threadIdx.x & 63;
    const int wave = __builtin_amdgcn_readfirstlane(xcd_block(blockIdx.x, gridDim.x) * (blockDim.x >> 6) + (threadIdx.x >> 6));
    const int chunk = wave / segs;
    const int seg = wave - chunk * segs;
    const int nr = P.nr, nphi = P.nphi;
    const int k0 = chunk * rows_per_chunk;
    if (k0 > nr)
        return;
    const int k1 = (k0 + rows_per_chunk < nr + 1) ? k0 + rows_per_chunk : nr + 1; // v_r has rows 0..nr
    const int jraw = seg * MARCH_VALID - MARCH_LO + lane;
    const int j = jraw < 0 ? jraw + nphi : (jraw >= nphi ? jraw - nphi : jraw);
    const bool store_lane = lane >= MARCH_LO && lane < MARCH_LO + MARCH_VALID && jraw < nphi;
    const double dt = P.clk->dt;
    const double C2 = P.art_visc_factor * P.art_visc_factor;

#define NEXT(x) lane_next(x) /* value of cell j+1 */
#define PREV(x) lane_prev(x) /* value of cell j-1 */
    auto crow = [nr](int r) { return r < 0 ? 0 : (r > nr - 1 ? nr - 1 : r); };   // cell rows
    auto vrow = [nr](int r) { return r < 0 ? 0 : (r > nr ? nr : r); };           // v_r rows

    // rolling state (suffix _1.._3 = rings m-1..m-3)
    double S_m = 0, S_1 = 0, S_2 = 0, S_3 = 0, Sp_m = 0, Sp_1 = 0, Sp_2 = 0; // Sigma and Sigma(j-1)
    double F_m = 0, F_1 = 0;                                               // potential
    double va0_m = 0, va0_1 = 0, va0n_m = 0, va0n_1 = 0;                   // v_phi (input) and (j+1)
    double vr1_m = 0, vr1_1 = 0, va1_m = 0, va1_1 = 0;                      // after source terms
    double qr_1 = 0, qr_2 = 0, qp_1 = 0, qp_2 = 0;                          // Q_rr/Q_pp (TW) or q_r/q_phi (SN)
    double vr2_1 = 0, vr2_2 = 0, va2_1 = 0, va2_2 = 0;                      // after artificial viscosity
    double trr_2 = 0, trr_3 = 0, tpp_2 = 0, tpp_3 = 0, trp_1 = 0, trp_2 = 0;

    // ring k0-3 is the "previous" ring of the first iteration
    {
        const int r = crow(k0 - 3);
        S_m = P.sigma[IDX(r, j)];
        F_m = P.potential[IDX(r, j)];
        va0_m = P.vazi[IDX(r, j)];
        Sp_m = PREV(S_m);
        va0n_m = NEXT(va0_m);
    }
    // software prefetch of the next input ring
    int rn = k0 - 2;
    double pS = P.sigma[IDX(crow(rn), j)], pF = P.potential[IDX(crow(rn), j)];
    double pVa = P.vazi[IDX(crow(rn), j)], pVr = P.vrad[IDX(vrow(rn), j)];

    for (int m = k0 - 2; m <= k1 + 1; ++m) {
        const SrcRow R = crow_load(P.src_tab, m + 2); // every per-ring factor of this iteration, one batch
        // ---- shift the window, take the prefetched ring m, prefetch ring m+1 ------------
        S_3 = S_2; S_2 = S_1; S_1 = S_m; Sp_2 = Sp_1; Sp_1 = Sp_m;
        F_1 = F_m;
        va0_1 = va0_m; va0n_1 = va0n_m;
        vr1_1 = vr1_m; va1_1 = va1_m;
        S_m = pS; F_m = pF; va0_m = pVa;
        const double vr0_m = pVr;
        {
            const int r = m + 1;
            pS = P.sigma[IDX(crow(r), j)];
            pF = P.potential[IDX(crow(r), j)];
            pVa = P.vazi[IDX(crow(r), j)];
            pVr = P.vrad[IDX(vrow(r), j)];
        }
        Sp_m = PREV(S_m);
        va0n_m = NEXT(va0_m);
        const double Fp_m = PREV(F_m);

        // ---- A: source terms on ring m ---------------------------------------------------
        {
            const int r = m;
            const double P_m = S_m * R.cs2_m, P_1 = S_1 * R.cs2_m1, Pp_m = Sp_m * R.cs2_m;
            vr1_m = vr0_m;
            if (r >= P.one_no_ghost_vr && r < P.maxmo_no_ghost_vr) {
                double gradp = 2.0 * fast_rcp(S_m + S_1);
                gradp *= (P_m - P_1);
                gradp *= R.idr_m;
                const double gradphi = (F_m - F_1) * R.idr_m;
                const double vsum = va0_m + va0n_m + va0_1 + va0n_1;
                const double vt = 0.25 * vsum + R.rinf_om_m;
                const double vt2 = vt * vt;
                vr1_m = vr0_m + dt * (-gradp - gradphi + vt2 * R.inv_rinf_m);
            }
            va1_m = va0_m;
            if (r >= P.zero_no_ghost && r < P.max_no_ghost) {
                const double invdxtheta = R.inv_dxt_m; // 2 / (dphi (Rsup + Rinf))
                const double gradp = 2.0 * fast_rcp(S_m + Sp_m) * (P_m - Pp_m) * invdxtheta;
                const double gradphi = (F_m - Fp_m) * invdxtheta;
                va1_m = va0_m + dt * (-gradp - gradphi);
            }
        }
        // ---- B: artificial pressure on ring m-1 ------------------------------------------
        qr_2 = qr_1; qp_2 = qp_1;
        {
            const double va1n_1 = NEXT(va1_1);
            if (AV == 1) {
                const double eps_rr = (vr1_m - vr1_1) * R.inv_drsup_b;
                const double eps_pp = R.inv_rmed_b * ((va1n_1 - va1_1) * P.invdphi + 0.5 * (vr1_m + vr1_1));
                const double div_V = dmin(eps_rr + eps_pp, 0.0);
                const double l_sq = R.lsq_b;
                qr_1 = l_sq * S_1 * -div_V * (eps_rr - 1.0 / 3.0 * div_V);
                qp_1 = l_sq * S_1 * -div_V * (eps_pp - 1.0 / 3.0 * div_V);
            } else if (AV == 2) {
                const double dv_r = vr1_m - vr1_1;
                qr_1 = dv_r < 0.0 ? C2 * S_1 * (dv_r * dv_r) : 0.0;
                const double dv_phi = va1n_1 - va1_1;
                qp_1 = dv_phi < 0.0 ? C2 * S_1 * (dv_phi * dv_phi) : 0.0;
            }
        }
        // ---- C: artificial-viscosity update of ring m-1 -----------------------------------
        vr2_2 = vr2_1; va2_2 = va2_1;
        {
            const int r = m - 1;
            vr2_1 = vr1_1;
            va2_1 = va1_1;
            const bool upd_vr = r >= P.one_no_ghost_vr && r < P.maxmo_no_ghost_vr;
            if (AV == 1) {
                const double qpp_p = PREV(qp_1);
                if (r >= 1 && r < nr - 1) {
                    const double sigma_phi_avg = 0.5 * (S_1 + Sp_1);
                    va2_1 = va1_1 + 2.0 * dt * (R.inv_rsum_c * fast_rcp(sigma_phi_avg)) * (qp_1 - qpp_p) * P.invdphi;
                }
                if (upd_vr) {
                    const double sigma_r_avg = 0.5 * (S_1 + S_2);
                    const double rm = R.rmed_c, rmm = R.rmed_cm1;
                    vr2_1 = vr1_1 + P.radial_viscosity_factor * dt * fast_rcp(sigma_r_avg) * 2.0 * R.inv_drmed2_c *
                                        ((qr_1 * rm - qr_2 * rmm) - 0.5 * (qp_1 + qp_2) * (rm - rmm));
                }
            } else if (AV == 2) {
                const double qphi_p = PREV(qp_1);
                if (upd_vr)
                    vr2_1 = vr1_1 - dt * 2.0 * fast_rcp(S_1 + S_2) * (qr_1 - qr_2) * R.idr_c;
                if (r >= P.zero_no_ghost && r < P.max_no_ghost) {
                    const double invdxtheta = R.inv_dxtheta_c;
                    va2_1 = va1_1 - dt * 2.0 * fast_rcp(S_1 + Sp_1) * (qp_1 - qphi_p) * invdxtheta;
                }
            }
        }
        // ---- D: stress tensor: diagonal on ring m-2, r-phi on ring m-1 --------------------
        trr_3 = trr_2; tpp_3 = tpp_2; trp_2 = trp_1;
        {
            const double va2n_2 = NEXT(va2_2);
            const double dva = va2n_2 - va2_2;
            const double divv =
                (vr2_1 * R.rinf_d1 - vr2_2 * R.rinf_d0) * R.inv_drsuprb_d + dva * P.invdphi * R.inv_rmed_d;
            const double nu = R.nu_d;
            const double drr = (vr2_1 - vr2_2) * R.inv_drsup_d;
            trr_2 = 2.0 * nu * S_2 * (drr - 1.0 / 3.0 * divv);
            const double dpp = dva * P.invdphi * R.inv_rmed_d + 0.5 * (vr2_1 + vr2_2) * R.inv_rmed_d;
            tpp_2 = 2.0 * nu * S_2 * (dpp - 1.0 / 3.0 * divv);
        }
        {
            const int r = m - 1;
            const double vr2p_1 = PREV(vr2_1);
            trp_1 = 0.0;
            if (r >= 1 && r <= nr - 1) {
                const double dvazirdr = (va2_1 * R.inv_rmed_r - va2_2 * R.inv_rmed_rm1) * R.idr_r;
                const double dvrdphi = (vr2_1 - vr2p_1) * P.invdphi;
                const double drp = R.rinf_r * dvazirdr + dvrdphi * R.inv_rinf_r;
                const double nu = R.nu_avg_r;
                const double sigma = 0.25 * (S_1 + S_2 + Sp_1 + Sp_2);
                trp_1 = nu * sigma * drp;
            }
        }
        // ---- E: viscous update of ring k = m-2 and store ----------------------------------
        {
            const int k = m - 2;
            const double tpp_p = PREV(tpp_2);
            const double trp_n = NEXT(trp_2);
            if (k >= k0 && k < k1) {
                double vr3 = vr2_2, va3 = va2_2;
                if (k >= 1 && k < nr - 1) {
                    const double sigma_avg = 0.5 * (S_2 + Sp_2);
                    va3 = va2_2 + dt * R.inv_rmed_k * fast_rcp(sigma_avg) *
                                      (R.two_inv_dra2_k * (R.ra1sq_k * trp_1 - R.ra0sq_k * trp_2) +
                                       (tpp_2 - tpp_p) * P.invdphi);
                }
                if (k >= P.one_no_ghost_vr && k < P.maxmo_no_ghost_vr) {
                    const double sigma_avg = 0.5 * (S_2 + S_3);
                    vr3 = vr2_2 + dt * fast_rcp(sigma_avg) * P.radial_viscosity_factor * 2.0 * R.inv_rmsum_k *
                                      ((R.rmed_k * trr_2 - R.rmed_km1 * trr_3) * R.idr_k +
                                       (trp_n - trp_2) * P.invdphi - 0.5 * (tpp_2 + tpp_3));
                }
                if (store_lane) {
                    P.vrad_b[IDX(k, j)] = vr3;
                    if (k < nr)
                        P.vazi_b[IDX(k, j)] = va3;
                }
                if (ring_sums && k < nr) { // this segment's share of sum_j v_phi(k, j) for the transport's <v_phi>
                    const double part = wave_sum(store_lane ? va3 : 0.0);
                    if (lane == 63)
                        P.ring_part[k * P.ring_pstride + seg] = part;
                }
            }
        }
    }
#undef NEXT
#undef PREV
}

// ===========================================================================
// The same march for the energy equation (EquationOfState: ideal).  On top of k_source_march:
//   A   pressure P = (gamma-1) e                                     (SourceEuler.cpp:1442-1473)
//   S3  compression heating of ring m-1 with the velocities of A        (:459-493)
//   B   dissipation of the artificial viscosity into e, temperature floor/ceiling
//                                                                     (artificial_viscosity.cpp:79-88,191-218)
//   V0  c_s, H and the alpha viscosity of ring m-1 from the new e       (:1054-1092,1218-1251, viscosity.cpp:98-137)
//   D   stress tensor with the per-cell viscosity (4-cell average at the corners)
//   E   viscous heating Q+ (:496-536), SubStep3's energy update (:956-1051), floor/ceiling
// Reads Sigma, Phi, v_r, v_phi, e once and writes v_r, v_phi, e, Q+, Q- once (10 doubles per
// cell); the c_s / H / nu / T grids of the step are not written: fcpt_post recomputes them from
// the final state, as recalculate_derived_disk_quantities does.
// COOL: the cooling terms of SubStep3 are compiled in (their opacity laws would otherwise cost the
// common no-cooling case 70 registers: 134 -> 208 VGPRs)
template <int AV, bool COOL> // AV 0: none, 1: TW, 2: SN
__global__ void __launch_bounds__(256) k_source_march_adi(const Dev P, int segs, int rows_per_chunk)
{
    const int lane = threadIdx.x & 63;
    const int wave = __builtin_amdgcn_readfirstlane(xcd_block(blockIdx.x, gridDim.x) * (blockDim.x >> 6) + (threadIdx.x >> 6));
    const int chunk = wave / segs;
    const int seg = wave - chunk * segs;
    const int nr = P.nr, nphi = P.nphi;
    const int k0 = chunk * rows_per_chunk;
    if (k0 > nr)
        return;
    const int k1 = (k0 + rows_per_chunk < nr + 1) ? k0 + rows_per_chunk : nr + 1; // v_r has rows 0..nr
    const int jraw = seg * MARCH_VALID - MARCH_LO + lane;
    const int j = jraw < 0 ? jraw + nphi : (jraw >= nphi ? jraw - nphi : jraw);
    const bool store_lane = lane >= MARCH_LO && lane < MARCH_LO + MARCH_VALID && jraw < nphi;
    const double dt = P.clk->dt;
    const double C2 = P.art_visc_factor * P.art_visc_factor;
    const double gm1 = P.gamma - 1.0;
    const double inv_sqrt_gamma = 1.0 / sqrt(P.gamma);
    const bool dissipate = P.art_visc_dissipation != 0;
    constexpr bool cooling = COOL;

#define NEXT(x) lane_next(x) /* value of cell j+1 */
#define PREV(x) lane_prev(x) /* value of cell j-1 */
    auto crow = [nr](int r) { return r < 0 ? 0 : (r > nr - 1 ? nr - 1 : r); };   // cell rows
    auto vrow = [nr](int r) { return r < 0 ? 0 : (r > nr ? nr : r); };           // v_r rows

    // rolling state (suffix _1.._3 = rings m-1..m-3)
    double S_m = 0, S_1 = 0, S_2 = 0, S_3 = 0, Sp_m = 0, Sp_1 = 0, Sp_2 = 0; // Sigma and Sigma(j-1)
    double F_m = 0, F_1 = 0;                                               // potential
    double Pr_m = 0, Pr_1 = 0;                                             // pressure
    double e0_m = 0, e0_1 = 0;                                             // energy as loaded
    double e2_1 = 0, e2_2 = 0;                                             // after S3 + dissipation + floor
    double nu_1 = 0, nu_2 = 0, nup_1 = 0, nup_2 = 0, H_1 = 0, H_2 = 0;     // viscosity (and at j-1), scale height
    double va0_m = 0, va0_1 = 0, va0n_m = 0, va0n_1 = 0;                   // v_phi (input) and (j+1)
    double vr1_m = 0, vr1_1 = 0, va1_m = 0, va1_1 = 0;                      // after source terms
    double qr_1 = 0, qr_2 = 0, qp_1 = 0, qp_2 = 0;                          // Q_rr/Q_pp (TW) or q_r/q_phi (SN)
    double vr2_1 = 0, vr2_2 = 0, va2_1 = 0, va2_2 = 0;                      // after artificial viscosity
    double trr_2 = 0, trr_3 = 0, tpp_2 = 0, tpp_3 = 0, trp_1 = 0, trp_2 = 0;

    // ring k0-3 is the "previous" ring of the first iteration
    {
        const int r = crow(k0 - 3);
        S_m = P.sigma[IDX(r, j)];
        F_m = P.potential[IDX(r, j)];
        va0_m = P.vazi[IDX(r, j)];
        e0_m = P.energy[IDX(r, j)];
        Pr_m = gm1 * e0_m;
        Sp_m = PREV(S_m);
        va0n_m = NEXT(va0_m);
    }
    // software prefetch of the next input ring
    int rn = k0 - 2;
    double pS = P.sigma[IDX(crow(rn), j)], pF = P.potential[IDX(crow(rn), j)], pE = P.energy[IDX(crow(rn), j)];
    double pVa = P.vazi[IDX(crow(rn), j)], pVr = P.vrad[IDX(vrow(rn), j)];

    for (int m = k0 - 2; m <= k1 + 1; ++m) {
        const SrcRow R = crow_load(P.src_tab, m + 2); // every per-ring factor of this iteration, one batch
        // ---- shift the window, take the prefetched ring m, prefetch ring m+1 ------------
        S_3 = S_2; S_2 = S_1; S_1 = S_m; Sp_2 = Sp_1; Sp_1 = Sp_m;
        F_1 = F_m; Pr_1 = Pr_m; e0_1 = e0_m;
        va0_1 = va0_m; va0n_1 = va0n_m;
        vr1_1 = vr1_m; va1_1 = va1_m;
        e2_2 = e2_1; nu_2 = nu_1; nup_2 = nup_1; H_2 = H_1;
        S_m = pS; F_m = pF; va0_m = pVa; e0_m = pE;
        const double vr0_m = pVr;
        {
            const int r = m + 1;
            pS = P.sigma[IDX(crow(r), j)];
            pF = P.potential[IDX(crow(r), j)];
            pE = P.energy[IDX(crow(r), j)];
            pVa = P.vazi[IDX(crow(r), j)];
            pVr = P.vrad[IDX(vrow(r), j)];
        }
        Pr_m = gm1 * e0_m;
        Sp_m = PREV(S_m);
        va0n_m = NEXT(va0_m);
        const double Fp_m = PREV(F_m);
        const double Prp_m = PREV(Pr_m);

        // ---- A: source terms on ring m ---------------------------------------------------
        {
            const int r = m;
            vr1_m = vr0_m;
            if (r >= P.one_no_ghost_vr && r < P.maxmo_no_ghost_vr) {
                double gradp = 2.0 * fast_rcp(S_m + S_1);
                gradp *= (Pr_m - Pr_1);
                gradp *= R.idr_m;
                const double gradphi = (F_m - F_1) * R.idr_m;
                const double vsum = va0_m + va0n_m + va0_1 + va0n_1;
                const double vt = 0.25 * vsum + R.rinf_om_m;
                const double vt2 = vt * vt;
                vr1_m = vr0_m + dt * (-gradp - gradphi + vt2 * R.inv_rinf_m);
            }
            va1_m = va0_m;
            if (r >= P.zero_no_ghost && r < P.max_no_ghost) {
                const double invdxtheta = R.inv_dxt_m; // 2 / (dphi (Rsup + Rinf))
                const double gradp = 2.0 * fast_rcp(S_m + Sp_m) * (Pr_m - Prp_m) * invdxtheta;
                const double gradphi = (F_m - Fp_m) * invdxtheta;
                va1_m = va0_m + dt * (-gradp - gradphi);
            }
        }
        // ---- S3 + B: compression heating, artificial pressure and its dissipation, ring m-1 --
        qr_2 = qr_1; qp_2 = qp_1;
        {
            const int r = m - 1;
            const double va1n_1 = NEXT(va1_1);
            double e = e0_1;
            if (r < nr - 1) { // compression_heating, rows [0, Nr-1)
                const double DIV_V = (vr1_m * R.rinf_b1 - vr1_1 * R.rinf_b0) * R.inv_drsuprb_b +
                                     (va1n_1 - va1_1) * P.invdphi * R.inv_rmed_b;
                e = e * exp(-gm1 * dt * DIV_V);
            }
            if (AV == 1) {
                const double eps_rr = (vr1_m - vr1_1) * R.inv_drsup_b;
                const double eps_pp = R.inv_rmed_b * ((va1n_1 - va1_1) * P.invdphi + 0.5 * (vr1_m + vr1_1));
                const double div_V = dmin(eps_rr + eps_pp, 0.0);
                const double l_sq = R.lsq_b;
                qr_1 = l_sq * S_1 * -div_V * (eps_rr - 1.0 / 3.0 * div_V);
                qp_1 = l_sq * S_1 * -div_V * (eps_pp - 1.0 / 3.0 * div_V);
                if (dissipate && r > P.zero_no_ghost && r < P.max_no_ghost) {
                    const double Qplus = -l_sq * div_V * S_1 * 1.0 / 3.0 *
                                         (eps_rr * eps_rr + eps_pp * eps_pp + (eps_rr - eps_pp) * (eps_rr - eps_pp));
                    e += Qplus * dt;
                }
            } else if (AV == 2) {
                const double dv_r = vr1_m - vr1_1;
                qr_1 = dv_r < 0.0 ? C2 * S_1 * (dv_r * dv_r) : 0.0;
                const double dv_phi = va1n_1 - va1_1;
                qp_1 = dv_phi < 0.0 ? C2 * S_1 * (dv_phi * dv_phi) : 0.0;
                if (dissipate && r >= P.zero_no_ghost && r < P.max_no_ghost)
                    e = e - dt * qr_1 * dv_r * R.inv_drsup_b - dt * qp_1 * dv_phi * R.inv_dxtheta_b;
            }
            if (dissipate) // update_with_artificial_viscosity ends with the temperature floor/ceiling
                e = clamp_energy_fast(P, e, S_1);
            e2_1 = e;
            // V0: recalculate_viscosity on ring m-1
            const double cs = sqrt(P.gamma * gm1 * e * fast_rcp(S_1));
            H_1 = cs * inv_sqrt_gamma * R.inv_omk_b;
            nu_1 = P.alpha_viscosity ? P.alpha * H_1 * cs : P.nu_const;
            nup_1 = PREV(nu_1);
        }
        // ---- C: artificial-viscosity update of ring m-1 -----------------------------------
        vr2_2 = vr2_1; va2_2 = va2_1;
        {
            const int r = m - 1;
            vr2_1 = vr1_1;
            va2_1 = va1_1;
            const bool upd_vr = r >= P.one_no_ghost_vr && r < P.maxmo_no_ghost_vr;
            if (AV == 1) {
                const double qpp_p = PREV(qp_1);
                if (r >= 1 && r < nr - 1) {
                    const double sigma_phi_avg = 0.5 * (S_1 + Sp_1);
                    va2_1 = va1_1 + 2.0 * dt * (R.inv_rsum_c * fast_rcp(sigma_phi_avg)) * (qp_1 - qpp_p) * P.invdphi;
                }
                if (upd_vr) {
                    const double sigma_r_avg = 0.5 * (S_1 + S_2);
                    const double rm = R.rmed_c, rmm = R.rmed_cm1;
                    vr2_1 = vr1_1 + P.radial_viscosity_factor * dt * fast_rcp(sigma_r_avg) * 2.0 * R.inv_drmed2_c *
                                        ((qr_1 * rm - qr_2 * rmm) - 0.5 * (qp_1 + qp_2) * (rm - rmm));
                }
            } else if (AV == 2) {
                const double qphi_p = PREV(qp_1);
                if (upd_vr)
                    vr2_1 = vr1_1 - dt * 2.0 * fast_rcp(S_1 + S_2) * (qr_1 - qr_2) * R.idr_c;
                if (r >= P.zero_no_ghost && r < P.max_no_ghost)
                    va2_1 = va1_1 - dt * 2.0 * fast_rcp(S_1 + Sp_1) * (qp_1 - qphi_p) * R.inv_dxtheta_b;
            }
        }
        // ---- D: stress tensor: diagonal on ring m-2, r-phi on ring m-1 --------------------
        trr_3 = trr_2; tpp_3 = tpp_2; trp_2 = trp_1;
        double divv_2;
        {
            const double va2n_2 = NEXT(va2_2);
            const double dva = va2n_2 - va2_2;
            divv_2 = (vr2_1 * R.rinf_d1 - vr2_2 * R.rinf_d0) * R.inv_drsuprb_d + dva * P.invdphi * R.inv_rmed_d;
            const double drr = (vr2_1 - vr2_2) * R.inv_drsup_d;
            trr_2 = 2.0 * nu_2 * S_2 * (drr - 1.0 / 3.0 * divv_2);
            const double dpp = dva * P.invdphi * R.inv_rmed_d + 0.5 * (vr2_1 + vr2_2) * R.inv_rmed_d;
            tpp_2 = 2.0 * nu_2 * S_2 * (dpp - 1.0 / 3.0 * divv_2);
        }
        {
            const int r = m - 1;
            const double vr2p_1 = PREV(vr2_1);
            trp_1 = 0.0;
            if (r >= 1 && r <= nr - 1) {
                const double dvazirdr = (va2_1 * R.inv_rmed_r - va2_2 * R.inv_rmed_rm1) * R.idr_r;
                const double dvrdphi = (vr2_1 - vr2p_1) * P.invdphi;
                const double drp = R.rinf_r * dvazirdr + dvrdphi * R.inv_rinf_r;
                const double nu = 0.25 * (nu_1 + nu_2 + nup_1 + nup_2);
                const double sigma = 0.25 * (S_1 + S_2 + Sp_1 + Sp_2);
                trp_1 = nu * sigma * drp;
            }
        }
        // ---- E: viscous update, viscous heating and SubStep3 of ring k = m-2, store -------
        {
            const int k = m - 2;
            const double tpp_p = PREV(tpp_2);
            const double trp_n = NEXT(trp_2);
            const double trp_1n = NEXT(trp_1);
            if (k >= k0 && k < k1) {
                double vr3 = vr2_2, va3 = va2_2;
                const bool row_va = k >= 1 && k < nr - 1;
                if (row_va) {
                    const double sigma_avg = 0.5 * (S_2 + Sp_2);
                    va3 = va2_2 + dt * R.inv_rmed_k * fast_rcp(sigma_avg) *
                                      (R.two_inv_dra2_k * (R.ra1sq_k * trp_1 - R.ra0sq_k * trp_2) +
                                       (tpp_2 - tpp_p) * P.invdphi);
                }
                if (k >= P.one_no_ghost_vr && k < P.maxmo_no_ghost_vr) {
                    const double sigma_avg = 0.5 * (S_2 + S_3);
                    vr3 = vr2_2 + dt * fast_rcp(sigma_avg) * P.radial_viscosity_factor * 2.0 * R.inv_rmsum_k *
                                      ((R.rmed_k * trr_2 - R.rmed_km1 * trr_3) * R.idr_k +
                                       (trp_n - trp_2) * P.invdphi - 0.5 * (tpp_2 + tpp_3));
                }
                double qplus = 0.0, qminus = 0.0, e = e2_2;
                if (k < nr) {
                    if (P.heating_viscous && row_va && nu_2 != 0.0) { // viscous_heating
                        const double tau_r_phi = 0.25 * (trp_2 + trp_1 + trp_n + trp_1n);
                        double q = fast_rcp(2.0 * nu_2 * S_2) * (trr_2 * trr_2 + 2 * (tau_r_phi * tau_r_phi) + tpp_2 * tpp_2);
                        q += (2.0 / 9.0) * nu_2 * S_2 * (divv_2 * divv_2);
                        q *= P.heating_viscous_factor;
                        qplus += q;
                    }
                    if (row_va) { // SubStep3, rows [1, Nr-1)
                        const double bb = P.b_fac * fast_rcp(S_2), b2 = bb * bb; // substep3_alpha
                        const double alpha = 1.0 + 2.0 * H_2 * 4.0 * P.sigma_sb / P.c_light * (b2 * b2) * (e * e * e);
                        const double ralpha = fast_rcp(alpha);
                        double tau_eff = 0.0;
                        if (cooling) { // calculate_qminus
                            const Cooling cool = cooling_terms(P, k, IDX(k, j), S_2, e, H_2);
                            qminus = cool.qminus * ralpha;
                            tau_eff = cool.tau_eff;
                        }
                        qplus = qplus * ralpha;
                        double energy_new = e + dt * (qplus - qminus);
                        const double SigmaFloor = 10.0 * P.sigma0_val * P.sigma_floor_rel;
                        if (S_2 < SigmaFloor) {
                            const double e4 = qplus * tau_eff / (2.0 * P.sigma_sb);
                            energy_new = sqrt(sqrt(e4)) * (P.Rgas / P.mu * S_2 / (P.gamma - 1.0));
                            qminus = qplus;
                        }
                        e = energy_new;
                    }
                    e = clamp_energy_fast(P, e, S_2); // SetTemperatureFloorCeilValues
                }
                if (store_lane) {
                    P.vrad_b[IDX(k, j)] = vr3;
                    if (k < nr) {
                        P.vazi_b[IDX(k, j)] = va3;
                        P.energy_b[IDX(k, j)] = e;
                        P.qplus[IDX(k, j)] = qplus;
                        P.qminus[IDX(k, j)] = qminus;
                        // step_LeapFrog evaluates the mid-step potential with the scale height this kick's
                        // recalculate_viscosity left behind (simulation.cpp:340-378), not with that of the
                        // transported state: keep the grid for it
                        if (P.leapfrog)
                            P.scale_height[IDX(k, j)] = H_2;
                    }
                }
            }
        }
    }
#undef NEXT
#undef PREV
}

// ---------------------------------------------------------------------------
// ComputeDiskOnPlanetAccel (Force.cpp:23-122): specific force of the slab's gas on an object.
// A thread owns a phi column over DOB_ROWS active rings; block sums in a fixed order into
// part[block][4] = {inner a_x, inner a_y, outer a_x, outer a_y}, folded by k_disk_on_body_final
// (two fixed-order stages: the result is deterministic, unlike an atomic accumulation).
#define DOB_ROWS 8
__global__ void __launch_bounds__(256) k_disk_on_body(const Dev P, double x, double y, double r_object,
                                                     double smoothing_fixed, double r_sm, double *part)
{
    const int j = blockIdx.x * blockDim.x + threadIdx.x;
    const int r0 = P.first_active + blockIdx.y * DOB_ROWS;
    const int r1 = r0 + DOB_ROWS < P.active_size ? r0 + DOB_ROWS : P.active_size;
    double a[4] = {0.0, 0.0, 0.0, 0.0};
    if (j < P.nphi) {
        const double cj = P.cosphi[j], sj = P.sinphi[j];
        for (int i = r0; i < r1; ++i) {
            const double rm = P.Rmed[i];
            double smooth = smoothing_fixed;
            if (smoothing_fixed < 0.0) { // compute_smoothing_scaleheight (Force.cpp:124-131)
                double H;
                if (!P.adiabatic) {
                    H = P.cs_ring[i] * P.g_inv_omk[i];
                } else if (P.lazy_derived) {
                    const double cs = sqrt(P.gamma * (P.gamma - 1.0) * P.energy[IDX(i, j)] / P.sigma[IDX(i, j)]);
                    H = cs / (sqrt(P.gamma)) * P.g_inv_omk[i];
                } else {
                    H = P.scale_height[IDX(i, j)];
                }
                smooth = P.thickness_smoothing * H;
            }
            const double cellmass = P.Surf[i] * P.sigma[IDX(i, j)];
            const double dx = rm * cj - x;
            const double dy = rm * sj - y;
            const double dist_sm_2 = dx * dx + dy * dy + smooth * smooth;
            const double dist_sm = sqrt(dist_sm_2);
            const double inv_dist_sm_3 = 1.0 / (dist_sm_2 * dist_sm);
            double klahr = 1.0;
            if (r_sm > 0.0 && dist_sm < r_sm) {
                const double q = dist_sm / r_sm;
                klahr = -(3.0 * ((q * q) * (q * q)) - 4.0 * (q * q * q));
            }
            const double fx = P.G * cellmass * dx * inv_dist_sm_3 * klahr;
            const double fy = P.G * cellmass * dy * inv_dist_sm_3 * klahr;
            const int o = rm < r_object ? 0 : 2;
            a[o] += fx;
            a[o + 1] += fy;
        }
    }
    __shared__ double s_a[4][4];
#pragma unroll
    for (int q = 0; q < 4; ++q) {
        double v = a[q];
        for (int off = 32; off > 0; off >>= 1)
            v += __shfl_down(v, off, 64);
        if ((threadIdx.x & 63) == 0)
            s_a[threadIdx.x >> 6][q] = v;
    }
    __syncthreads();
    if (threadIdx.x < 4)
        part[(blockIdx.y * gridDim.x + blockIdx.x) * 4 + threadIdx.x] =
            (s_a[0][threadIdx.x] + s_a[1][threadIdx.x]) + (s_a[2][threadIdx.x] + s_a[3][threadIdx.x]);
}
__global__ void __launch_bounds__(256) k_disk_on_body_final(const double *part, int nblocks, double *out)
{
    __shared__ double s_a[4][4];
#pragma unroll
    for (int q = 0; q < 4; ++q) {
        double v = 0.0;
        for (int n = threadIdx.x; n < nblocks; n += blockDim.x)
            v += part[n * 4 + q];
        for (int off = 32; off > 0; off >>= 1)
            v += __shfl_down(v, off, 64);
        if ((threadIdx.x & 63) == 0)
            s_a[threadIdx.x >> 6][q] = v;
    }
    __syncthreads();
    if (threadIdx.x < 4)
        out[threadIdx.x] = (s_a[0][threadIdx.x] + s_a[1][threadIdx.x]) + (s_a[2][threadIdx.x] + s_a[3][threadIdx.x]);
}

// ---------------------------------------------------------------------------
// boundary_conditions/{zero_gradient,reference,reflecting,outflow,keplerian_*,zero_shear}.cpp
// called in the order of boundary_conditions.cpp:65-114; one thread per phi column.
// All loads first, then all stores: the ghost values only depend on active rings (or on the
// reference fields), so the ~10 memory round trips of the sequential form collapse into one.
struct BcScalar {
    bool on;
    double v;
};
__device__ __forceinline__ BcScalar bc_scalar_load(const Dev &P, const double *x, const double *x0, int type, int outer,
                                                   int j)
{
    const int Irad = P.nr - 1;
    BcScalar r = {false, 0.0};
    if ((!outer && !P.is_first) || (outer && !P.is_last))
        return r;
    if (type == FCPT_BC_ZEROGRADIENT) {
        r.on = true;
        r.v = x[IDX(outer ? Irad - 1 : 1, j)];
    } else if (type == FCPT_BC_REFERENCE) {
        r.on = true;
        r.v = x0[IDX(outer ? Irad : 0, j)];
    }
    return r;
}
__global__ void k_boundary(const Dev P)
{
    const int j = blockIdx.x * blockDim.x + threadIdx.x;
    if (j >= P.nphi)
        return;
    const int Irad = P.nr - 1;
    const int Iv = P.nr; // max_radial of the vector grid
    // ---- loads ---------------------------------------------------------------------------
    BcScalar sg[2], en[2];
    for (int outer = 0; outer < 2; ++outer) {
        sg[outer] = bc_scalar_load(P, P.sigma, P.sigma0, P.bc_sigma[outer], outer, j);
        en[outer] = bc_scalar_load(P, P.energy, P.energy0, P.bc_energy[outer], outer, j);
    }
    bool vr_on[2] = {false, false}, va_on[2] = {false, false};
    double vr_g0[2] = {0.0, 0.0}, vr_g1[2] = {0.0, 0.0}, va_g[2] = {0.0, 0.0};
    for (int outer = 0; outer < 2; ++outer) {
        const int type = P.bc_vrad[outer];
        const int g0 = outer ? Iv : 0, g1 = outer ? Iv - 1 : 1, a = outer ? Iv - 2 : 2;
        if (type == FCPT_BC_REFLECTING) { // no rank guard in the reference (reflecting.cpp:15-40)
            vr_on[outer] = true;
            vr_g0[outer] = -P.vrad[IDX(a, j)];
            vr_g1[outer] = 0.0;
            continue;
        }
        if ((!outer && !P.is_first) || (outer && !P.is_last))
            continue;
        switch (type) {
        case FCPT_BC_ZEROGRADIENT:
            vr_on[outer] = true;
            vr_g0[outer] = vr_g1[outer] = P.vrad[IDX(a, j)];
            break;
        case FCPT_BC_REFERENCE:
            vr_on[outer] = true;
            vr_g0[outer] = P.vrad0[IDX(g0, j)];
            vr_g1[outer] = P.vrad0[IDX(g1, j)];
            break;
        case FCPT_BC_OUTFLOW: {
            const double va = P.vrad[IDX(a, j)];
            const bool inflow = outer ? (va < 0.0) : (va > 0.0);
            vr_on[outer] = true;
            vr_g0[outer] = vr_g1[outer] = inflow ? 0.0 : va;
            break;
        }
        case FCPT_BC_KEPLERIAN:
            vr_on[outer] = true;
            vr_g0[outer] = P.kep_vrad[outer] * sqrt(P.G * P.Mc / P.Rmed[g0]);
            vr_g1[outer] = P.kep_vrad[outer] * sqrt(P.G * P.Mc / P.Rmed[g1]);
            break;
        default:
            break;
        }
    }
    for (int outer = 0; outer < 2; ++outer) {
        const int type = P.bc_vaz[outer];
        if ((!outer && !P.is_first) || (outer && !P.is_last))
            continue;
        const int row = outer ? Irad : 0, act = outer ? Irad - 1 : 1;
        const double r = P.Rmed[row];
        switch (type) {
        case FCPT_BC_ZEROGRADIENT:
            va_on[outer] = true;
            va_g[outer] = P.vazi[IDX(act, j)];
            break;
        case FCPT_BC_REFERENCE:
            va_on[outer] = true;
            va_g[outer] = P.vazi0[IDX(row, j)];
            break;
        case FCPT_BC_KEPLERIAN:
            va_on[outer] = true;
            va_g[outer] = P.kep_vaz[outer] * sqrt(P.G * P.Mc / r) - r * P.omega_frame;
            break;
        case FCPT_BC_ZEROSHEAR:
            va_on[outer] = true;
            va_g[outer] = r * (P.vazi[IDX(act, j)] / P.Rmed[act]);
            break;
        default:
            break;
        }
    }
    // ---- stores --------------------------------------------------------------------------
    for (int outer = 0; outer < 2; ++outer) {
        const int row = outer ? Irad : 0;
        if (sg[outer].on)
            P.sigma[IDX(row, j)] = sg[outer].v;
        if (en[outer].on)
            P.energy[IDX(row, j)] = en[outer].v;
        if (vr_on[outer]) {
            P.vrad[IDX(outer ? Iv : 0, j)] = vr_g0[outer];
            P.vrad[IDX(outer ? Iv - 1 : 1, j)] = vr_g1[outer];
        }
        if (va_on[outer])
            P.vazi[IDX(row, j)] = va_g[outer];
    }
}

// boundary_conditions/damping.cpp:311-427 (reference), :429-557 (zero), :559-700 (mean):
// one block per damped ring.
__global__ void k_damping(const Dev P, double *q, double *q0, const double *radius, int lo, int type,
                          double rlim, double redge, double tau, int is_density)
{
    const int i = lo + blockIdx.x;
    const double dt = P.clk->dt;
    __shared__ double s_part[256];
    __shared__ double s_mean;
    if (type == FCPT_DAMP_MEAN) {
        double acc = 0.0;
        for (int j = threadIdx.x; j < P.nphi; j += blockDim.x)
            acc += q[IDX(i, j)];
        s_part[threadIdx.x] = acc;
        __syncthreads();
        for (int s = blockDim.x / 2; s > 0; s >>= 1) {
            if ((int)threadIdx.x < s)
                s_part[threadIdx.x] += s_part[threadIdx.x + s];
            __syncthreads();
        }
        if (threadIdx.x == 0) {
            s_mean = s_part[0] / P.nphi;
            q0[IDX(i, 0)] = s_mean;
        }
        __syncthreads();
    }
    const double t = (radius[i] - rlim) / (redge - rlim);
    const double factor = t * t;
    const double exp_factor = exp(-dt * factor / tau);
    for (int j = threadIdx.x; j < P.nphi; j += blockDim.x) {
        const double X = q[IDX(i, j)];
        double X0;
        if (type == FCPT_DAMP_REFERENCE)
            X0 = q0[IDX(i, j)];
        else if (type == FCPT_DAMP_MEAN)
            X0 = s_mean;
        else
            X0 = is_density ? P.sigma_floor_abs : 0.0;
        q[IDX(i, j)] = (X - X0) * exp_factor + X0;
    }
}

// ---------------------------------------------------------------------------
// Transport (TransportEuler.cpp).

// TransportEuler.cpp:306-337 flux_limiter
__device__ __forceinline__ double limiter(int type, double a, double b)
{
    if (type == FCPT_LIMITER_MC) {
        // minmod(0.5*(a+b), 2*minmod(a,b))
        double m = 0.0;
        if (a * b > 0.0)
            m = fabs(a) < fabs(b) ? a : b;
        const double c = 0.5 * (a + b), d = 2.0 * m;
        if (c * d > 0.0)
            return fabs(c) < fabs(d) ? c : d;
        return 0.0;
    }
    const double ab = a * b;
    return ab > 0.0 ? 2.0 * ab * fast_rcp1(a + b) : 0.0;
}

// Upwind "star" state at radial interface k (between rings k-1 and k),
// compute_star_radial (TransportEuler.cpp:349-406).  wm2..wp1 = Q at rings k-2..k+1.
// Per-interface geometry of compute_star_radial, loaded once with the (wavefront-uniform)
// interface index so it lives in scalar registers; the upwind choice then only selects
// between preloaded values instead of issuing lane-divergent loads.
struct StarGeo {
    double idr_m, idr_0, idr_p; // InvDiffRmed[k-1], [k], [k+1]
    double dr_lo, dr_hi;        // Rmed[k]-Rmed[k-1], Rmed[k+1]-Rmed[k]
    bool lim_lo, lim_hi;        // slope of ring k-1 / ring k is limited (not a closed boundary ring)
    bool open;                  // interface carries a flux (0 < k < Nr)
};
__device__ __forceinline__ StarGeo star_geo(const Dev &P, int k)
{
    StarGeo g;
    g.open = k > 0 && k < P.nr;
    const int kk = g.open ? k : 1;
    g.idr_m = P.InvDiffRmed[kk - 1];
    g.idr_0 = P.InvDiffRmed[kk];
    g.idr_p = P.InvDiffRmed[kk + 1];
    g.dr_lo = P.Rmed[kk] - P.Rmed[kk - 1];
    g.dr_hi = P.Rmed[kk + 1] - P.Rmed[kk];
    g.lim_lo = (kk - 1 != 0) && (kk - 1 != P.nr - 1);
    g.lim_hi = (kk != 0) && (kk != P.nr - 1);
    return g;
}
__device__ __forceinline__ double star_radial(const Dev &P, const StarGeo &g, double v, double dt,
                                              double wm2, double wm1, double w0, double wp1)
{
    if (!g.open)
        return 0.0; // row 0 is zeroed on every call, row Nr is never written
    // upwind cell c = k-1 (v > 0) or k; one limiter evaluation on the selected stencil
    const bool up = v > 0.0;
    const double x0 = up ? wm2 : wm1, x1 = up ? wm1 : w0, x2 = up ? w0 : wp1;
    const double ihi = up ? g.idr_0 : g.idr_p, ilo = up ? g.idr_m : g.idr_0;
    const bool lim = up ? g.lim_lo : g.lim_hi;
    const double dq = lim ? limiter(P.limiter, (x2 - x1) * ihi, (x1 - x0) * ilo) : 0.0;
    const double dist = up ? (g.dr_lo - v * dt) : -(g.dr_hi + v * dt);
    return x1 + dist * 0.5 * dq;
}

// compute_momenta_from_velocities (:471-493) + OneWindRad (:138-167) with all
// VanLeerRadial calls (:545-620) in one pass.  Reads Sigma, v_r, v_phi(, e) and
// writes the transported momenta / density / energy to set B, so the in-place
// ordering constraint of the reference ("Sigma MUST be last") is met by
// construction: every quantity sees the pre-transport density.
//
// One thread owns a phi column and marches RADIAL_ROWS rings outward keeping the
// 4-ring stencil of every specific quantity in registers, so each interface flux
// is evaluated once and each ring is loaded once per chunk (+4 halo rings).
// The specific momenta Work = (Sigma v)/Sigma are formed as v directly (equal to
// the reference's quotient to within 1 ulp).
#define RADIAL_ROWS 16

struct RadialRow { // specific quantities of one ring at this column (er: the energy itself)
    double s, rmp, rmm, lp, lm, e, er;
};
__device__ __forceinline__ RadialRow radial_load(const Dev &P, int k, int j, int jn, double vr_k,
                                                 double vr_k1)
{
    RadialRow w;
    if (k >= 0 && k < P.nr) {
        const double r = P.Rmed[k];
        w.s = P.sigma[IDX(k, j)];
        w.rmp = vr_k1;
        w.rmm = vr_k;
        w.lp = (P.vazi[IDX(k, jn)] + r * P.omega_frame) * r;
        w.lm = (P.vazi[IDX(k, j)] + r * P.omega_frame) * r;
        w.er = P.adiabatic ? P.energy[IDX(k, j)] : 0.0;
        w.e = P.adiabatic ? w.er / w.s : 0.0;
    } else {
        w.s = w.rmp = w.rmm = w.lp = w.lm = w.e = w.er = 0.0;
    }
    return w;
}
struct RadialFlux {
    double s, rmp, rmm, lp, lm, e;
};
// fluxes through interface k given rings k-2..k+1 (a,b,c,d) and v_r(k)
__device__ __forceinline__ RadialFlux radial_flux(const Dev &P, int k, double v, double dt,
                                                  const RadialRow &a, const RadialRow &b,
                                                  const RadialRow &c, const RadialRow &d)
{
    RadialFlux f;
    const StarGeo geo = star_geo(P, k);
    if (!geo.open) { // closed: QRStar/DensityStar row 0 zeroed, row Nr never written
        f.s = f.rmp = f.rmm = f.lp = f.lm = f.e = 0.0;
        return f;
    }
    const double rho = star_radial(P, geo, v, dt, a.s, b.s, c.s, d.s);
    const double g = dt * P.dphi * P.Rinf[k];
    f.s = g * 1.0 * rho * v;
    f.rmp = g * star_radial(P, geo, v, dt, a.rmp, b.rmp, c.rmp, d.rmp) * rho * v;
    f.rmm = g * star_radial(P, geo, v, dt, a.rmm, b.rmm, c.rmm, d.rmm) * rho * v;
    f.lp = g * star_radial(P, geo, v, dt, a.lp, b.lp, c.lp, d.lp) * rho * v;
    f.lm = g * star_radial(P, geo, v, dt, a.lm, b.lm, c.lm, d.lm) * rho * v;
    f.e = P.adiabatic ? g * star_radial(P, geo, v, dt, a.e, b.e, c.e, d.e) * rho * v : 0.0;
    return f;
}
template <bool ROWU> __device__ __forceinline__ void transport_radial_block(const Dev &P, int vb, int gx, int nvb)
{
    const int lb = xcd_block(vb, nvb);
    const int j = (lb % gx) * blockDim.x + threadIdx.x;
    const int r0_ = ((lb / gx) * blockDim.y + threadIdx.y) * RADIAL_ROWS;
    if (j >= P.nphi || r0_ >= P.nr)
        return;
    const int r0 = ROWU ? __builtin_amdgcn_readfirstlane(r0_) : r0_;
    const double dt = P.clk->dt;
    const int jn = JNEXT;
    const int nr = P.nr;
    const int r1 = r0 + RADIAL_ROWS < nr ? r0 + RADIAL_ROWS : nr;
    auto vr_at = [&](int k) { return (k >= 0 && k <= nr) ? P.vrad[IDX(k, j)] : 0.0; };
    // rings r0-2 .. r0+2 (the last one is the software-prefetched ring of the next iteration:
    // the marching loop is latency-bound unless each ring's loads are issued one iteration
    // before their first use)
    double v0 = vr_at(r0 - 2), v1 = vr_at(r0 - 1), v2 = vr_at(r0), v3 = vr_at(r0 + 1), v4 = vr_at(r0 + 2),
           v5 = vr_at(r0 + 3);
    RadialRow a = radial_load(P, r0 - 2, j, jn, v0, v1);
    RadialRow b = radial_load(P, r0 - 1, j, jn, v1, v2);
    RadialRow c = radial_load(P, r0, j, jn, v2, v3);
    RadialRow d = radial_load(P, r0 + 1, j, jn, v3, v4);
    RadialRow e = radial_load(P, r0 + 2, j, jn, v4, v5);
    RadialFlux fin = radial_flux(P, r0, v2, dt, a, b, c, d);
    for (int i = r0; i < r1; ++i) {
        // prefetch ring i+3 (used by the next iteration)
        const double v6 = vr_at(i + 4);
        const RadialRow f = radial_load(P, i + 3, j, jn, v5, v6);
        // rings i-1..i+2 around interface i+1
        const RadialFlux fout = radial_flux(P, i + 1, v3, dt, b, c, d, e);
        const double invsurf = P.InvSurf[i];
        const double s0 = c.s;
        // momenta of T1 for ring i (TransportEuler.cpp:484-490); c.lp/c.lm = (v_phi + r Omega) r
        P.rmpB[IDX(i, j)] = s0 * v3 + (fin.rmp - fout.rmp) * invsurf;
        P.rmmB[IDX(i, j)] = s0 * v2 + (fin.rmm - fout.rmm) * invsurf;
        P.lpB[IDX(i, j)] = s0 * c.lp + (fin.lp - fout.lp) * invsurf;
        P.lmB[IDX(i, j)] = s0 * c.lm + (fin.lm - fout.lm) * invsurf;
        if (P.adiabatic)
            P.eB[IDX(i, j)] = c.er + (fin.e - fout.e) * invsurf;
        P.sigB[IDX(i, j)] = s0 + (fin.s - fout.s) * invsurf;
        a = b; b = c; c = d; d = e; e = f;
        v2 = v3; v3 = v4; v4 = v5; v5 = v6;
        fin = fout;
    }
}

// The kernel proper walks gx * gy virtual blocks with a grid stride: launched with one block per
// virtual block in normal use, and with a small grid as the in-stream fallback of
// k_transport_fused (only_if: runs only when that kernel gave up; an idle fallback then costs a
// few hundred blocks that return at once, not thousands).
template <bool ROWU> __global__ void __launch_bounds__(256) k_transport_radial(const Dev P, const int *only_if, int gx, int gy)
{
    if (only_if && !*only_if)
        return;
    for (int vb = blockIdx.x; vb < gx * gy; vb += gridDim.x)
        transport_radial_block<ROWU>(P, vb, gx, gx * gy);
}

// compute_average_azimuthal_velocity (:174-189) + ComputeConstantResidual (:207-236):
// one wavefront per ring (4 rings per block), 16-byte loads with 16 in flight per lane, a
// butterfly for the ring sum; the per-ring scalars of the epilogue are fetched up front so the
// last lane-0 instructions do not queue behind three dependent memory round trips.
__global__ void __launch_bounds__(256) k_ring_mean(const Dev P, int with_shift, const double *part, int nparts, int pstride)
{
    // part != nullptr: the producer kernel left nparts partial sums per ring (fixed order, so the
    // result is deterministic); rings rewritten afterwards by a boundary condition are re-summed
    // from the grid.
    const int lane = threadIdx.x & 63;
    const int i = __builtin_amdgcn_readfirstlane(blockIdx.x * (blockDim.x >> 6) + (threadIdx.x >> 6));
    if (i >= P.nr)
        return;
    const double dt = with_shift ? P.clk->dt : 1.0;
    const double invr = P.InvRmed[i], rmed = P.Rmed[i];
    const bool ghost = (i == 0 && P.is_first && P.bc_vaz[0] != FCPT_BC_NONE) ||
                       (i == P.nr - 1 && P.is_last && P.bc_vaz[1] != FCPT_BC_NONE) ||
                       (!P.is_first && i < FCPT_OVERLAP) || (!P.is_last && i >= P.nr - FCPT_OVERLAP);
    double acc = 0.0;
    if (part && !ghost) {
        for (int n = lane; n < nparts; n += 64)
            acc += part[i * pstride + n];
    } else {
        const double *row = P.vazi + (size_t)i * P.nphi;
        const int npair = P.nphi >> 1;
        double acc2 = 0.0;
        int n = lane;
        for (; n + 15 * 64 < npair; n += 16 * 64) {
            D2 v[16];
#pragma unroll
            for (int u = 0; u < 16; ++u)
                v[u] = *(const D2 *)(row + 2 * (n + u * 64));
#pragma unroll
            for (int u = 0; u < 16; ++u) {
                acc += v[u].x;
                acc2 += v[u].y;
            }
        }
        for (; n < npair; n += 64) {
            const D2 v = *(const D2 *)(row + 2 * n);
            acc += v.x;
            acc2 += v.y;
        }
        if ((P.nphi & 1) && lane == 0)
            acc += row[P.nphi - 1];
        acc += acc2;
    }
    for (int off = 32; off > 0; off >>= 1)
        acc += __shfl_down(acc, off, 64);
    if (lane == 0) {
        const double mean = acc / (double)P.nphi;
        P.vmean[i] = mean;
        if (with_shift && i == 0)
            *P.shift_jump = 0;
        if (with_shift) {
            const double invdt = 1.0 / dt;
            const double Ntilde = mean * invr * dt * P.invdphi;
            const double Nround = floor(Ntilde + 0.5);
            P.nshift[i] = (int)Nround;
            const double vc = (Ntilde - Nround) * rmed * invdt * P.dphi;
            P.vconst[i] = vc;
            ShiftRow sr;
            sr.mean = mean, sr.vconst = vc, sr.nshift = (int)Nround, sr.pad0 = 0, sr.pad1[0] = sr.pad1[1] = 0.0;
            const DampRow dr = P.damp_tab[i]; // damping.cpp:311-427: X <- (X - X0) exp(-dt f / tau) + X0
            sr.es = exp(-dt * dr.fs / dr.ts);
            sr.ev = exp(-dt * dr.fv / dr.tv);
            sr.ev_top = 1.0;
            if (i == P.nr - 1) {
                const DampRow dn = P.damp_tab[P.nr];
                sr.ev_top = exp(-dt * dn.fv / dn.tv);
            }
            P.shift_tab[i] = sr;
        }
    }
}

// Upwind star state at azimuthal interface k (between cells k-1 and k),
// ComputeStarTheta (:416-466); wm2..wp1 = Q at cells k-2..k+1.
__device__ __forceinline__ double star_theta(const Dev &P, double v, double dt, double dxtheta,
                                             double invdxtheta, double wm2, double wm1, double w0,
                                             double wp1)
{
    const double ksi = v * dt;
    const bool up = ksi > 0.0;
    const double x0 = up ? wm2 : wm1, x1 = up ? wm1 : w0, x2 = up ? w0 : wp1;
    const double dq = 0.5 * limiter(P.limiter, (x2 - x1), (x1 - x0)) * invdxtheta;
    const double dist = up ? (dxtheta - ksi) : -(dxtheta + ksi);
    return x1 + dist * dq;
}

struct ThetaSet {
    const double *rmp, *rmm, *lp, *lm, *sig, *e;
};
struct ThetaOut {
    double *rmp, *rmm, *lp, *lm, *sig, *e;
};

// QuantitiesAdvection (:292-304) with all VanLeerTheta calls (:630-664) in one pass,
// out of place.  PASS 1: residual velocity v_phi - <v_phi> (+ constant residual when the
// FARGO split is off).  PASS 2: uniform residual, and the integer shift AdvectSHIFT
// (:238-268) is applied by the store (cell j lands in j + Nshift).
template <int PASS, bool ROWU> __global__ void k_transport_theta(const Dev P, ThetaSet in, ThetaOut out)
{
    CELL(0, P.nr);
    const double dt = P.clk->dt;
    const int nphi = P.nphi;
    int jj[5];
    jj[2] = j;
    jj[1] = j == 0 ? nphi - 1 : j - 1;
    jj[0] = jj[1] == 0 ? nphi - 1 : jj[1] - 1;
    jj[3] = j == nphi - 1 ? 0 : j + 1;
    jj[4] = jj[3] == nphi - 1 ? 0 : jj[3] + 1;

    int jout = j;
    if (PASS == 2) {
        int s = j + P.nshift[i];
        s %= nphi;
        if (s < 0)
            s += nphi;
        jout = s;
    }
    const bool skip = (PASS == 2) && !P.fast_transport; // NoSplitAdvection rows (:646)
    if (skip) {
        out.rmp[IDX(i, jout)] = in.rmp[IDX(i, j)];
        out.rmm[IDX(i, jout)] = in.rmm[IDX(i, j)];
        out.lp[IDX(i, jout)] = in.lp[IDX(i, j)];
        out.lm[IDX(i, jout)] = in.lm[IDX(i, j)];
        out.sig[IDX(i, jout)] = in.sig[IDX(i, j)];
        if (P.adiabatic)
            out.e[IDX(i, jout)] = in.e[IDX(i, j)];
        return;
    }
    double v0, v1; // velocity at interfaces j and j+1
    if (PASS == 1) {
        const double m = P.vmean[i];
        v0 = P.vazi[IDX(i, j)] - m;
        v1 = P.vazi[IDX(i, jj[3])] - m;
        if (!P.fast_transport) {
            v0 = P.vconst[i] + v0;
            v1 = P.vconst[i] + v1;
        }
    } else {
        v0 = v1 = P.vconst[i];
    }
    const double dxtheta = P.dphi * P.Rmed[i];
    const double invdxtheta = 1.0 / dxtheta;
    const double dxrad = (P.Rsup[i] - P.Rinf[i]) * dt;
    const double invsurf = P.InvSurf[i];
    double S[5], rS[5]; // Work = Q * (1/Sigma): within 1 ulp of the reference's Q / Sigma
#pragma unroll
    for (int a = 0; a < 5; ++a) {
        S[a] = in.sig[IDX(i, jj[a])];
        rS[a] = fast_rcp(S[a]);
    }
    const double rho0 = star_theta(P, v0, dt, dxtheta, invdxtheta, S[0], S[1], S[2], S[3]);
    const double rho1 = star_theta(P, v1, dt, dxtheta, invdxtheta, S[1], S[2], S[3], S[4]);
#define THETA_UPDATE(IN, OUT)                                                                    \
    {                                                                                            \
        double W[5];                                                                             \
        _Pragma("unroll") for (int a = 0; a < 5; ++a) W[a] = IN[IDX(i, jj[a])] * rS[a];         \
        const double q0 = star_theta(P, v0, dt, dxtheta, invdxtheta, W[0], W[1], W[2], W[3]);    \
        const double q1 = star_theta(P, v1, dt, dxtheta, invdxtheta, W[1], W[2], W[3], W[4]);    \
        double varq = dxrad * q0 * rho0 * v0;                                                    \
        varq -= dxrad * q1 * rho1 * v1;                                                          \
        OUT[IDX(i, jout)] = IN[IDX(i, j)] + varq * invsurf;                                      \
    }
    THETA_UPDATE(in.rmp, out.rmp);
    THETA_UPDATE(in.rmm, out.rmm);
    THETA_UPDATE(in.lp, out.lp);
    THETA_UPDATE(in.lm, out.lm);
    if (P.adiabatic)
        THETA_UPDATE(in.e, out.e);
    {
        // density: Work = 1 => star state 1
        double varq = dxrad * 1.0 * rho0 * v0;
        varq -= dxrad * 1.0 * rho1 * v1;
        out.sig[IDX(i, jout)] = S[2] + varq * invsurf;
    }
#undef THETA_UPDATE
}


// ---------------------------------------------------------------------------
// OneWindTheta (:270-288) in ONE kernel: residual pass, uniform pass and the integer
// shift.  A wavefront owns a segment of one ring; every lane keeps C contiguous cells of
// all transported quantities in registers.  Nothing is computed twice: a lane evaluates the
// limited slope of its own cells and the star state / flux at the lower face of its own
// cells; the left neighbour's edge value and slope and the right neighbour's first flux
// arrive by wavefront shuffle (no LDS, no barriers).
//   * "periodic" mode (Nphi <= 64 C, Nphi % C == 0): one wavefront holds the whole ring and
//     the shuffles wrap around.
//   * tiled mode: each pass invalidates two cells at either end of a segment (their stencil
//     leaves the segment), so segments of 64 C cells advance by 64 C - 8 and only the inner
//     cells are stored.
// Reads set B (+ v_phi, <v_phi>, Nshift), writes set A: 11 (13) doubles per cell.
#define THETA_HALO 4
template <int C, bool ADI>
__global__ void __launch_bounds__(256) k_transport_theta_fused(const Dev P, ThetaSet in, ThetaOut out,
                                                              int tiles, int periodic)
{
    const int lane = threadIdx.x & 63;
    const int wave = __builtin_amdgcn_readfirstlane(blockIdx.x * (blockDim.x >> 6) + (threadIdx.x >> 6));
    const int i = wave / tiles;
    if (i >= P.nr)
        return;
    const int tile = wave - i * tiles;
    const int nphi = P.nphi;
    const int nl = periodic ? nphi / C : 64;                    // lanes that own cells
    const int stride = periodic ? nphi : 64 * C - 2 * THETA_HALO;
    const int a = periodic ? 0 : tile * stride - THETA_HALO;    // first cell of the segment
    const bool act = lane < nl;
    const int ln = act ? lane : 0;
    int lsrc_l = ln - 1, lsrc_r = ln + 1;
    if (periodic) {
        lsrc_l = lsrc_l < 0 ? nl - 1 : lsrc_l;
        lsrc_r = lsrc_r >= nl ? 0 : lsrc_r;
    } else {
        lsrc_l = lsrc_l < 0 ? 0 : lsrc_l;
        lsrc_r = lsrc_r > 63 ? 63 : lsrc_r;
    }
    const double dt = P.clk->dt;
    const int row = i * nphi;
    // all cell indices of a segment lie in (-nphi, 2 nphi): one conditional fold replaces '%'
    auto wrap = [nphi](int j) { return j < 0 ? j + nphi : (j >= nphi ? j - nphi : j); };

    int idx[C];
    double S[C], Q[4][C], E[C], V[C];
    const double mean = P.vmean_c[i];
    const double vconst = P.vconst_c[i];
    const double vadd = P.fast_transport ? 0.0 : vconst; // ComputeConstantResidual, non-FARGO branch
#pragma unroll
    for (int c = 0; c < C; ++c) {
        idx[c] = wrap(a + ln * C + c);
        const int g = row + idx[c];
        S[c] = in.sig[g];
        Q[0][c] = in.rmp[g];
        Q[1][c] = in.rmm[g];
        Q[2][c] = in.lp[g];
        Q[3][c] = in.lm[g];
        E[c] = ADI ? in.e[g] : 0.0;
        V[c] = vadd + (P.vazi[g] - mean); // residual velocity at the lower face of cell c
    }
    const double dxtheta = P.dphi * P.Rmed[i];
    const double invdxtheta = 1.0 / dxtheta;
    const double dxrad = (P.Rsup[i] - P.Rinf[i]) * dt;
    const double invsurf = P.InvSurf[i];

    // ComputeStarTheta (:416-466) + the flux of VanLeerTheta (:655-658) at the lower faces of
    // the lane's cells for the array W[] (own cells); fac[c] = dxrad * rho*(c) * v(c) (or dxrad
    // v(c) for the density itself).  fl[C] is the right neighbour's first flux.
#define THETA_FLUX(fl, st, W, fac, HAVE_ST)                                                        \
    {                                                                                               \
        const double wl = __shfl(W[C - 1], lsrc_l, 64); /* cell -1 */                               \
        const double wr = __shfl(W[0], lsrc_r, 64);     /* cell C  */                               \
        double dq[C];                                                                               \
        _Pragma("unroll") for (int c = 0; c < C; ++c)                                               \
        {                                                                                           \
            const double wm = c == 0 ? wl : W[c == 0 ? 0 : c - 1];                                 \
            const double wp = c == C - 1 ? wr : W[c == C - 1 ? C - 1 : c + 1];                      \
            dq[c] = 0.5 * limiter(P.limiter, wp - W[c], W[c] - wm) * invdxtheta;                    \
        }                                                                                           \
        const double dql = __shfl(dq[C - 1], lsrc_l, 64); /* slope of cell -1 */                    \
        _Pragma("unroll") for (int c = 0; c < C; ++c)                                               \
        {                                                                                           \
            const double xa = up[c] ? (c == 0 ? wl : W[c == 0 ? 0 : c - 1]) : W[c];                 \
            const double sl = up[c] ? (c == 0 ? dql : dq[c == 0 ? 0 : c - 1]) : dq[c];              \
            const double star = xa + dist[c] * sl;                                                  \
            if (HAVE_ST)                                                                            \
                st[c] = star;                                                                       \
            fl[c] = (HAVE_ST) ? 0.0 : fac[c] * star;                                                \
        }                                                                                           \
    }

    for (int pass = 1; pass <= 2; ++pass) {
        if (pass == 2) {
            if (!P.fast_transport)
                break; // NoSplitAdvection: the uniform pass is skipped (:646)
#pragma unroll
            for (int c = 0; c < C; ++c)
                V[c] = vconst;
        }
        // per-face upwind data shared by all quantities
        bool up[C];
        double dist[C];
#pragma unroll
        for (int c = 0; c < C; ++c) {
            const double ksi = V[c] * dt;
            up[c] = ksi > 0.0;
            dist[c] = up[c] ? (dxtheta - ksi) : -(dxtheta + ksi);
        }
        double rho[C], rS[C], dummy[C];
        THETA_FLUX(dummy, rho, S, dummy, true); // DensityStar at the lower faces
#pragma unroll
        for (int c = 0; c < C; ++c)
            rS[c] = fast_rcp(S[c]);
        // varq = dxrad * Q* * rho* * v  (:655-658), evaluated as ((dxrad Q*) rho*) v
#define THETA_Q(X)                                                                                  \
        {                                                                                           \
            double W[C], fl[C + 1], qs[C];                                                          \
            _Pragma("unroll") for (int c = 0; c < C; ++c) W[c] = X[c] * rS[c];                      \
            THETA_FLUX(fl, qs, W, dummy, true);                                                     \
            _Pragma("unroll") for (int c = 0; c < C; ++c) fl[c] = dxrad * qs[c] * rho[c] * V[c];    \
            fl[C] = __shfl(fl[0], lsrc_r, 64);                                                      \
            _Pragma("unroll") for (int c = 0; c < C; ++c)                                           \
            {                                                                                       \
                double varq = fl[c];                                                                \
                varq -= fl[c + 1];                                                                  \
                X[c] += varq * invsurf;                                                             \
            }                                                                                       \
        }
        THETA_Q(Q[0]);
        THETA_Q(Q[1]);
        THETA_Q(Q[2]);
        THETA_Q(Q[3]);
        if (ADI)
            THETA_Q(E);
        {
            double fl[C + 1];
#pragma unroll
            for (int c = 0; c < C; ++c)
                fl[c] = dxrad * 1.0 * rho[c] * V[c];
            fl[C] = __shfl(fl[0], lsrc_r, 64);
#pragma unroll
            for (int c = 0; c < C; ++c) {
                double varq = fl[c];
                varq -= fl[c + 1];
                S[c] += varq * invsurf;
            }
        }
#undef THETA_Q
    }
#undef THETA_FLUX
    // AdvectSHIFT (:238-268): cell j lands in j + Nshift (Nshift folded into [0, nphi) once)
    int nshift = P.nshift_c[i] % nphi;
    nshift = nshift < 0 ? nshift + nphi : nshift;
#pragma unroll
    for (int c = 0; c < C; ++c) {
        const int pos = lane * C + c;
        const bool valid = act && (periodic || (pos >= THETA_HALO && pos < 64 * C - THETA_HALO));
        if (valid) {
            const int g = row + wrap(idx[c] + nshift);
            out.sig[g] = S[c];
            out.rmp[g] = Q[0][c];
            out.rmm[g] = Q[1][c];
            out.lp[g] = Q[2][c];
            out.lm[g] = Q[3][c];
            if (ADI)
                out.e[g] = E[c];
        }
    }
}

// compute_velocities_from_momenta (:498-535) + assure_minimum_value and the
// temperature floor/ceiling of Transport (:121-131); reads set B, writes the state.
// Wave damping of one value (damping.cpp:311-557): X <- (X - X0) exp(-dt f / tau) + X0 on rings
// whose per-ring type is non-zero (1: reference field, 2: zero / density floor).
__device__ __forceinline__ double damp_value(const Dev &P, double X, int type, double fac, double tau,
                                             double dt, const double *ref, int cell, double zero_target)
{
    if (type == 0)
        return X;
    const double exp_factor = exp(-dt * fac / tau);
    const double X0 = type == 1 ? ref[cell] : zero_target;
    return (X - X0) * exp_factor + X0;
}
// compute_velocities_from_momenta (:498-535) + assure_minimum_value and the temperature
// floor/ceiling of Transport (:121-131); reads a momenta set, writes the state.  With
// DAMP the reference/zero wave damping of the final boundary call (damping.cpp:754-774) is
// applied to the fresh values in the same pass (the per-cell operations commute with the
// ghost exchange that sits between them in the reference).
template <bool DAMP, bool ROWU> __global__ void k_velocities(const Dev P, ThetaSet in, const double *vr_src)
{
    CELL(0, P.nr);
    const double dt = P.clk->dt;
    if (i_ == P.nr - 1) { // v_r row Nr is not transported: it keeps its post-boundary value
        double v = vr_src[IDX(P.nr, j)];
        if (DAMP)
            v = damp_value(P, v, P.dtype_vr[P.nr], P.dfac_v[P.nr], P.dtau_v[P.nr], dt, P.vrad0, IDX(P.nr, j), 0.0);
        P.vrad[IDX(P.nr, j)] = v;
    }
    const int jp = JPREV;
    const double s = in.sig[IDX(i, j)];
    double vr = 0.0;
    if (i != 0)
        vr = (in.rmp[IDX(i - 1, j)] + in.rmm[IDX(i, j)]) / (in.sig[IDX(i - 1, j)] + s);
    double va = (in.lp[IDX(i, jp)] + in.lm[IDX(i, j)]) / (in.sig[IDX(i, jp)] + s) * P.InvRmed[i] -
                P.Rmed[i] * P.omega_frame;
    double sf = s < P.sigma_floor_abs ? P.sigma_floor_abs : s;
    double e = 0.0;
    if (P.adiabatic)
        e = clamp_energy(P, in.e[IDX(i, j)], sf);
    if (DAMP) {
        const int c = IDX(i, j);
        vr = damp_value(P, vr, P.dtype_vr[i], P.dfac_v[i], P.dtau_v[i], dt, P.vrad0, c, 0.0);
        const double fs = P.dfac_s[i], ts = P.dtau_s[i];
        va = damp_value(P, va, P.dtype_va[i], fs, ts, dt, P.vazi0, c, 0.0);
        sf = damp_value(P, sf, P.dtype_sig[i], fs, ts, dt, P.sigma0, c, P.sigma_floor_abs);
        if (P.adiabatic)
            e = damp_value(P, e, P.dtype_e[i], fs, ts, dt, P.energy0, c, 0.0);
    }
    P.vrad[IDX(i, j)] = vr;
    P.vazi[IDX(i, j)] = va;
    P.sigma[IDX(i, j)] = sf;
    if (P.adiabatic)
        P.energy[IDX(i, j)] = e;
}

// ---------------------------------------------------------------------------
// Azimuthal transport + velocities + floors + wave damping in ONE kernel, marching over rings.
// As k_transport_theta_fused, but a wavefront owns a phi segment in POST-shift coordinates and
// walks THETA_ROWS rings outward: for ring i it reads the cells that the integer shift maps
// onto its segment (input index = output index - Nshift[i]), runs both passes in registers,
// and -- because ring i-1 was processed by the same lanes one iteration earlier -- forms
//   v_r(i)   = (rm+(i-1) + rm-(i)) / (Sigma(i-1) + Sigma(i))                (:515-523)
//   v_phi(i) = (L+(j-1) + L-(j)) / (Sigma(j-1) + Sigma(j)) / r - r Omega    (:526-532)
// applies the density floor / temperature range (:121-131) and the reference/zero wave damping
// of the final boundary call, and stores the new state.  The transported momenta never go to
// memory: the sweep reads 6 (7) grids and writes 3 (4) instead of 11 + 8 (13 + 10) doubles per
// cell for k_transport_theta_fused + k_velocities.
// Validity: 4 cells at either end of a segment are lost to the two passes, one more on the
// left to the L+(j-1) neighbour.
#define THETA_ROWS 8
#define THETA_LO 6 /* even, so that a lane's two cells are both final or both halo */
#define THETA_HI 4

// 0.5 * flux_limiter(a, b) (TransportEuler.cpp:306-337): the factor 2 of van Leer's 2ab/(a+b)
// and the 0.5 of the half-cell slope cancel exactly.
__device__ __forceinline__ double half_limiter(int type, double a, double b)
{
    if (type == FCPT_LIMITER_MC)
        return 0.5 * limiter(type, a, b);
    const double ab = a * b;
    return ab > 0.0 ? ab * fast_rcp1(a + b) : 0.0;
}

// Upwind star states of one quantity on the C cells of a lane (compute_star_theta,
// TransportEuler.cpp:408-441): st = q_upwind + (dist / dxtheta) * half_limited_difference_upwind.
// MODE 0: per-cell upwind direction up[c] and distance factor d2[c]; MODE 1 / 2: the whole ring
// moves with one velocity > 0 / <= 0 (second FARGO pass), so the upwind choice is made at compile
// time and the selects disappear.
template <int C, bool PER, int MODE>
__device__ __forceinline__ void theta_star(int lim, int lsrc_l, int lsrc_r, const double (&W)[C], const bool (&up)[C],
                                           const double (&d2)[C], double d2u, double (&st)[C])
{
#define SH_PREV(x) (PER ? __shfl((x), lsrc_l, 64) : lane_prev(x))
#define SH_NEXT(x) (PER ? __shfl((x), lsrc_r, 64) : lane_next(x))
    const double wl = SH_PREV(W[C - 1]); // cell -1
    const double wr = SH_NEXT(W[0]);     // cell C
    double h[C];
#pragma unroll
    for (int c = 0; c < C; ++c) {
        const double wm = c == 0 ? wl : W[c == 0 ? 0 : c - 1];
        const double wp = c == C - 1 ? wr : W[c == C - 1 ? C - 1 : c + 1];
        h[c] = half_limiter(lim, wp - W[c], W[c] - wm);
    }
    if (MODE == 2) {
#pragma unroll
        for (int c = 0; c < C; ++c)
            st[c] = W[c] + d2u * h[c];
        return;
    }
    const double hl = SH_PREV(h[C - 1]); // slope of cell -1
#pragma unroll
    for (int c = 0; c < C; ++c) {
        const double wm = c == 0 ? wl : W[c == 0 ? 0 : c - 1];
        const double hm = c == 0 ? hl : h[c == 0 ? 0 : c - 1];
        if (MODE == 1)
            st[c] = wm + d2u * hm;
        else
            st[c] = (up[c] ? wm : W[c]) + d2[c] * (up[c] ? hm : h[c]);
    }
}

// One azimuthal pass (OneWindTheta's VanLeerTheta calls, TransportEuler.cpp:443-496,583-628) on
// the cells of a lane.  geo_dt = (Rsup-Rinf) * InvSurf * dt; V the per-cell velocity (MODE 0) or vu
// the ring velocity (MODE 1/2).  The interface mass flux F = geo_dt * v * rho* is formed once and
// shared by all quantities: Q += q*(c) F(c) - q*(c+1) F(c+1).
template <int C, bool ADI, bool PER, int MODE>
__device__ __forceinline__ void theta_pass(int lim, int lsrc_l, int lsrc_r, double geo_dt, double dxtheta, double invdx,
                                           double dt, const double (&V)[C], double vu, double (&S)[C], double (&Q)[4][C],
                                           double (&E)[C])
{
    bool up[C];
    double d2[C], d2u = 0.0;
    if (MODE == 0) {
#pragma unroll
        for (int c = 0; c < C; ++c) {
            const double ksi = V[c] * dt;
            up[c] = ksi > 0.0;
            d2[c] = (up[c] ? (dxtheta - ksi) : -(dxtheta + ksi)) * invdx;
        }
    } else {
        const double ksi = vu * dt;
#pragma unroll
        for (int c = 0; c < C; ++c) {
            up[c] = MODE == 1;
            d2[c] = 0.0;
        }
        d2u = (MODE == 1 ? (dxtheta - ksi) : -(dxtheta + ksi)) * invdx;
    }
    double rho[C], F[C + 1], rS[C];
    theta_star<C, PER, MODE>(lim, lsrc_l, lsrc_r, S, up, d2, d2u, rho);
    const double gvu = geo_dt * vu;
#pragma unroll
    for (int c = 0; c < C; ++c) {
        F[c] = (MODE == 0 ? geo_dt * V[c] : gvu) * rho[c];
        rS[c] = fast_rcp(S[c]);
    }
    F[C] = SH_NEXT(F[0]);
    auto advect = [&](double (&X)[C]) {
        double Wq[C], qs[C], fl[C + 1];
#pragma unroll
        for (int c = 0; c < C; ++c)
            Wq[c] = X[c] * rS[c];
        theta_star<C, PER, MODE>(lim, lsrc_l, lsrc_r, Wq, up, d2, d2u, qs);
#pragma unroll
        for (int c = 0; c < C; ++c)
            fl[c] = qs[c] * F[c];
        fl[C] = SH_NEXT(fl[0]);
#pragma unroll
        for (int c = 0; c < C; ++c)
            X[c] += fl[c] - fl[c + 1];
    };
    advect(Q[0]);
    advect(Q[1]);
    advect(Q[2]);
    advect(Q[3]);
    if (ADI)
        advect(E);
#pragma unroll
    for (int c = 0; c < C; ++c)
        S[c] += F[c] - F[c + 1];
}

template <int C, bool ADI, bool DAMP, bool PER>
__device__ __forceinline__ void transport_theta_march_block(const Dev &P, const double *va_pre, const double *vr_pre, const ThetaSet &in,
                                                            int tiles, int rows, int advance_clock, int vb, int nvb)
{
    constexpr int periodic = PER ? 1 : 0;
    // va_pre / vr_pre: the pre-transport (post-source, post-boundary) velocities; the new state goes to P's grids
    const int lane = threadIdx.x & 63;
    const int wave = __builtin_amdgcn_readfirstlane(xcd_block(vb, nvb) * (blockDim.x >> 6) + (threadIdx.x >> 6));
    const int chunk = wave / tiles;
    const int r0 = chunk * rows;
    const int nr = P.nr;
    if (r0 >= nr)
        return;
    const int r1 = r0 + rows < nr ? r0 + rows : nr;
    const int tile = wave - chunk * tiles;
    const int nphi = P.nphi;
    const int nl = periodic ? nphi / C : 64;
    const int stride = periodic ? nphi : 64 * C - (THETA_LO + THETA_HI);
    const int a = periodic ? 0 : tile * stride - THETA_LO; // first (output) cell of the segment
    const bool act = lane < nl;
    const int ln = act ? lane : 0;
    int lsrc_l = ln - 1, lsrc_r = ln + 1;
    if (periodic) {
        lsrc_l = lsrc_l < 0 ? nl - 1 : lsrc_l;
        lsrc_r = lsrc_r >= nl ? 0 : lsrc_r;
    } else {
        lsrc_l = lsrc_l < 0 ? 0 : lsrc_l;
        lsrc_r = lsrc_r > 63 ? 63 : lsrc_r;
    }
    const double dt = P.clk->dt;
    const int lim = P.limiter;
    auto wrap = [nphi](int j) { return j < 0 ? j + nphi : (j >= nphi ? j - nphi : j); };

    int jout[C];
    bool valid[C];
#pragma unroll
    for (int c = 0; c < C; ++c) {
        const int pos = lane * C + c;
        jout[c] = wrap(a + ln * C + c);
        valid[c] = act && (periodic || (pos >= THETA_LO && pos < 64 * C - THETA_HI && a + pos < nphi));
    }
    // 16-byte stores when, for every lane, the two cells are adjacent in memory and final together
    const bool pair_out =
        C == 2 && !PER && __builtin_amdgcn_ballot_w64(jout[C - 1] != jout[0] + 1 || valid[0] != valid[C - 1]) == 0;
    double rmp_prev[C], S_prev[C]; // transported rm+ and Sigma of ring i-1 at the same output cells
#pragma unroll
    for (int c = 0; c < C; ++c)
        rmp_prev[c] = S_prev[c] = 0.0;

    const int i0 = r0 > 0 ? r0 - 1 : 0;
    for (int i = i0; i < r1; ++i) {
        const int row = i * nphi;
        const double mean = P.vmean_c[i];
        const double vconst = P.vconst_c[i];
        const double vadd = P.fast_transport ? 0.0 : vconst;
        double S[C], Q[4][C], E[C], V[C];
        {
            int ns = P.nshift_c[i] % nphi;
            ns = ns < 0 ? ns + nphi : ns;
            int gin[C];
#pragma unroll
            for (int c = 0; c < C; ++c)
                gin[c] = row + wrap(jout[c] - ns); // the cell that AdvectSHIFT moves onto jout
            // 2 cells per lane: one 16-byte load per grid unless the ring seam falls inside a lane's pair
            const bool pairs = C == 2 && !PER && __builtin_amdgcn_ballot_w64(gin[C - 1] != gin[0] + 1) == 0;
            if (pairs) {
                const D2 s2 = LD2(in.sig + gin[0]), a2 = LD2(in.rmp + gin[0]), b2 = LD2(in.rmm + gin[0]);
                const D2 c2 = LD2(in.lp + gin[0]), d2 = LD2(in.lm + gin[0]), v2 = LD2(va_pre + gin[0]);
                D2 e2 = {0.0, 0.0};
                if (ADI)
                    e2 = LD2(in.e + gin[0]);
                S[0] = s2.x, S[C - 1] = s2.y;
                Q[0][0] = a2.x, Q[0][C - 1] = a2.y;
                Q[1][0] = b2.x, Q[1][C - 1] = b2.y;
                Q[2][0] = c2.x, Q[2][C - 1] = c2.y;
                Q[3][0] = d2.x, Q[3][C - 1] = d2.y;
                E[0] = e2.x, E[C - 1] = e2.y;
                V[0] = vadd + (v2.x - mean), V[C - 1] = vadd + (v2.y - mean);
            } else {
#pragma unroll
                for (int c = 0; c < C; ++c) {
                    const int g = gin[c];
                    S[c] = in.sig[g];
                    Q[0][c] = in.rmp[g];
                    Q[1][c] = in.rmm[g];
                    Q[2][c] = in.lp[g];
                    Q[3][c] = in.lm[g];
                    E[c] = ADI ? in.e[g] : 0.0;
                    V[c] = vadd + (va_pre[g] - mean);
                }
            }
        }
        const double dxtheta = P.g_dxtheta[i];
        const double invdx = P.g_inv_dxtheta[i];
        const double geo_dt = P.g_dr_invsurf[i] * dt;
#ifndef EXP_THETA_NOCOMP
        theta_pass<C, ADI, PER, 0>(lim, lsrc_l, lsrc_r, geo_dt, dxtheta, invdx, dt, V, 0.0, S, Q, E);
#else
#pragma unroll
        for (int c = 0; c < C; ++c) { Q[1][c] += Q[0][c] * 1e-9; Q[3][c] += (Q[2][c] + V[c]) * 1e-9; }
#endif
#ifndef EXP_THETA_NOCOMP
        if (P.fast_transport) {
            if (vconst * dt > 0.0)
                theta_pass<C, ADI, PER, 1>(lim, lsrc_l, lsrc_r, geo_dt, dxtheta, invdx, dt, V, vconst, S, Q, E);
            else
                theta_pass<C, ADI, PER, 2>(lim, lsrc_l, lsrc_r, geo_dt, dxtheta, invdx, dt, V, vconst, S, Q, E);
        }
#endif
        // compute_velocities_from_momenta + floors + damping for ring i (rings < r0 only prime rmp/S)
        if (i >= r0) {
            const double lp_l = SH_PREV(Q[2][C - 1]); // L+ and Sigma of cell j-1
            const double s_l = SH_PREV(S[C - 1]);
            const double fs = DAMP ? P.dfac_s[i] : 0.0, ts = DAMP ? P.dtau_s[i] : 1.0;
            const int tvr = DAMP ? P.dtype_vr[i] : 0, tva = DAMP ? P.dtype_va[i] : 0;
            const int tsg = DAMP ? P.dtype_sig[i] : 0, ten = DAMP ? P.dtype_e[i] : 0;
            const double fv = DAMP ? P.dfac_v[i] : 0.0, tv = DAMP ? P.dtau_v[i] : 1.0;
            const double invr = P.InvRmed[i], romega = P.g_r_omega[i];
            double o_vr[C], o_va[C], o_s[C], o_e[C];
#pragma unroll
            for (int c = 0; c < C; ++c) {
                const double lpm = c == 0 ? lp_l : Q[2][c == 0 ? 0 : c - 1];
                const double sm = c == 0 ? s_l : S[c == 0 ? 0 : c - 1];
                double vr = 0.0;
                if (i != 0)
                    vr = (rmp_prev[c] + Q[1][c]) * fast_rcp(S_prev[c] + S[c]);
                double va = (lpm + Q[3][c]) * fast_rcp(sm + S[c]) * invr - romega;
                double sf = S[c] < P.sigma_floor_abs ? P.sigma_floor_abs : S[c];
                double e = ADI ? clamp_energy(P, E[c], sf) : 0.0;
                const int g = row + jout[c];
                if (DAMP) {
                    vr = damp_value(P, vr, tvr, fv, tv, dt, P.vrad0, g, 0.0);
                    va = damp_value(P, va, tva, fs, ts, dt, P.vazi0, g, 0.0);
                    sf = damp_value(P, sf, tsg, fs, ts, dt, P.sigma0, g, P.sigma_floor_abs);
                    if (ADI)
                        e = damp_value(P, e, ten, fs, ts, dt, P.energy0, g, 0.0);
                }
                o_vr[c] = vr, o_va[c] = va, o_s[c] = sf, o_e[c] = e;
            }
            if (pair_out) { // both cells of the lane are final and adjacent in memory
                if (valid[0]) {
                    const int g = row + jout[0];
                    ST2(P.vrad + g, (D2{o_vr[0], o_vr[C - 1]}));
                    ST2(P.vazi + g, (D2{o_va[0], o_va[C - 1]}));
                    ST2(P.sigma + g, (D2{o_s[0], o_s[C - 1]}));
                    if (ADI)
                        ST2(P.energy + g, (D2{o_e[0], o_e[C - 1]}));
                }
            } else {
#pragma unroll
                for (int c = 0; c < C; ++c)
                    if (valid[c]) {
                        const int g = row + jout[c];
                        P.vrad[g] = o_vr[c];
                        P.vazi[g] = o_va[c];
                        P.sigma[g] = o_s[c];
                        if (ADI)
                            P.energy[g] = o_e[c];
                    }
            }
            if (i == nr - 1) { // v_r row Nr is not transported: it keeps its post-boundary value
#pragma unroll
                for (int c = 0; c < C; ++c)
                    if (valid[c]) {
                        double v = vr_pre[nr * nphi + jout[c]];
                        if (DAMP)
                            v = damp_value(P, v, P.dtype_vr[nr], P.dfac_v[nr], P.dtau_v[nr], dt, P.vrad0,
                                           nr * nphi + jout[c], 0.0);
                        P.vrad[nr * nphi + jout[c]] = v;
                    }
            }
        }
#pragma unroll
        for (int c = 0; c < C; ++c) {
            rmp_prev[c] = Q[0][c];
            S_prev[c] = S[c];
        }
    }
    if (wave == 0 && lane == 0 && advance_clock) { // sim::time += dt; N_hydro_iter++ (simulation.cpp:226-227)
        P.clk->time += dt;
        P.clk->n_hydro_iter += 1;
    }
#undef SH_PREV
#undef SH_NEXT
}

// grid-stride wrapper, as k_transport_radial
template <int C, bool ADI, bool DAMP, bool PER>
__global__ void __launch_bounds__(256) k_transport_theta_march(const Dev P, const double *va_pre, const double *vr_pre, ThetaSet in,
                                                              int tiles, int rows, int advance_clock, const int *only_if, int nvb)
{
    if (only_if && !*only_if)
        return;
    for (int vb = blockIdx.x; vb < nvb; vb += gridDim.x)
        transport_theta_march_block<C, ADI, DAMP, PER>(P, va_pre, vr_pre, in, tiles, rows, advance_clock, vb, nvb);
}

// ===========================================================================
// The whole Transport() (TransportEuler.cpp:112-136) in ONE pass over memory.
//
// A wavefront owns 64*C consecutive phi columns in PRE-shift coordinates and marches outward
// ring by ring.  Per step it loads one ring of Sigma, v_r, v_phi(, e) (the only HBM reads), and
//   R  radial sweep: specific momenta w(m), limited half slopes of ring m-1, the upwind fluxes
//      through interface m-1 (each evaluated once, shared mass flux), update of ring m-2
//      (compute_momenta_from_velocities + OneWindRad, :138-167,471-493,545-620) -- all column-local,
//      a rolling register window of three rings;
//   T  both azimuthal passes on ring m-2 (theta_pass, as k_transport_theta_march) with phi
//      neighbours by DPP lane shifts;
//   V  velocities from momenta, floors, wave damping (:498-535,121-131) and the store of the new
//      state at the POST-shift address (column + Nshift[i], AdvectSHIFT :238-268 is free).
// v_r(i) couples rings i-1 and i at one post-shift column, i.e. at lanes that differ by
// Nshift[i] - Nshift[i-1].  The FARGO shear limit of the CFL condition (cfl.cpp:207-220) keeps
// that difference in {-1, 0, 1} for every admissible dt, so one lane shift of the previous ring
// is enough; k_ring_mean raises P.shift_jump otherwise and the unfused kernels run instead.
// Nothing intermediate reaches memory: 3 (4) grids read + 3 (4) written instead of 8 + 9
// (10 + 11) doubles per cell for k_transport_radial + k_transport_theta_march.
// Validity in cells of a 64*C segment: right 1 (L+ needs v_phi(j+1)), 4 at either end for the
// two passes, left 1 for L+(j-1), 1 at either end for the v_r lane shift.
#define TF_ROWS 24
template <int C> struct TfHalo {
    static constexpr int lo = C == 2 ? 6 : 5; // even for C = 2: a lane's two cells are final together
    static constexpr int hi = 6;
};

// wave damping with the ring's precomputed exp(-dt f / tau) (k_ring_mean): types as damp_value
__device__ __forceinline__ double damp_apply(double X, int type, double ef, const double *ref, int cell, double zero_target)
{
    if (type == 0)
        return X;
    const double X0 = type == 1 ? ref[cell] : zero_target;
    return (X - X0) * ef + X0;
}

template <int C, bool ADI, bool DAMP, int LIM>
__global__ void __launch_bounds__(256) k_transport_fused(const Dev P, const Dev W, int tiles, int rows, int has_fallback)
{
    // P: view whose vrad/vazi are the velocities to transport; W: view that receives the new state
    constexpr int LO = TfHalo<C>::lo, HI = TfHalo<C>::hi;
    constexpr int NQ = ADI ? 6 : 5; // s, rmp, rmm, lp, lm(, e)
    constexpr int lim = LIM;
    const int lane = threadIdx.x & 63;
    const int wave = __builtin_amdgcn_readfirstlane(xcd_block(blockIdx.x, gridDim.x) * (blockDim.x >> 6) + (threadIdx.x >> 6));
    const int chunk = wave / tiles;
    const int r0 = chunk * rows;
    const int nr = P.nr, nphi = P.nphi;
    if (r0 >= nr)
        return;
    const int r1 = r0 + rows < nr ? r0 + rows : nr;
    if (wave == 0 && lane == 0) { // sim::time += dt; N_hydro_iter++ (simulation.cpp:226-227)
        W.clk->time += P.clk->dt;
        W.clk->n_hydro_iter += 1;
    }
    { // the lane shift of the previous ring covers |Nshift[i] - Nshift[i-1]| <= 1 only
        bool jump = false;
        int prev = P.nshift_c[r0 > 0 ? r0 - 1 : 0] % nphi;
        for (int i = r0; i < r1; ++i) {
            const int cur = P.nshift_c[i] % nphi;
            int dd = cur - prev;
            dd = dd < 0 ? -dd : dd;
            dd = dd > nphi / 2 ? nphi - dd : dd;
            jump = jump || dd > 1;
            prev = cur;
        }
        if (jump) {
            if (lane == 0) {
                *P.shift_jump = 1;
                if (!has_fallback) // nothing behind this kernel will redo the step: report it
                    W.clk->shear_error = 1;
            }
            return;
        }
    }
    const int tile = wave - chunk * tiles;
    const int stride = 64 * C - (LO + HI);
    const int a = tile * stride - LO; // first pre-shift column of the segment
    const double dt = P.clk->dt;
    auto wrap = [nphi](int j) { return j < 0 ? j + nphi : (j >= nphi ? j - nphi : j); };

    int jin[C];
    bool valid[C];
#pragma unroll
    for (int c = 0; c < C; ++c) {
        const int pos = lane * C + c;
        jin[c] = wrap(a + pos);
        valid[c] = pos >= LO && pos < 64 * C - HI && a + pos < nphi;
    }
    const bool pair_in = C == 2 && __builtin_amdgcn_ballot_w64(jin[C - 1] != jin[0] + 1) == 0;
    const bool pair_valid = C == 2 && __builtin_amdgcn_ballot_w64(valid[0] != valid[C - 1]) == 0;

    // rolling window: index 0 = ring m (newest), 1 = m-1, 2 = m-2
    double w[3][NQ][C];  // specific quantities: Sigma, v_r(ring+1), v_r(ring), (v_phi(j+1) + r Omega) r, (v_phi + r Omega) r(, e / Sigma)
    double er[3][C];     // the energy itself
    double vp[3][C];     // v_phi as loaded
    double d1[NQ][C];    // (w(m-1) - w(m-2)) InvDiffRmed[m-1]
    double hs1[NQ][C];   // limited half slope of ring m-2
    double F1[NQ][C];    // flux through interface m-2
    double rmp_prev[C], S_prev[C]; // transported rm+ and Sigma of the previous ring
#pragma unroll
    for (int c = 0; c < C; ++c) {
#pragma unroll
        for (int q = 0; q < NQ; ++q)
            w[0][q][c] = w[1][q][c] = w[2][q][c] = d1[q][c] = hs1[q][c] = F1[q][c] = 0.0;
        er[0][c] = er[1][c] = er[2][c] = vp[0][c] = vp[1][c] = vp[2][c] = 0.0;
        rmp_prev[c] = S_prev[c] = 0.0;
    }
    // raw loads of one ring: Sigma(k), v_phi(k)(, e(k)) and v_r(k+1); zeros outside the grid
    struct RingRaw {
        double sg[C], va[C], en[C], vr[C];
    };
    auto fetch = [&](int k, RingRaw &o) {
        const bool in_k = k >= 0 && k < nr;
        const bool in_v = k + 1 >= 0 && k + 1 <= nr;
#pragma unroll
        for (int c = 0; c < C; ++c)
            o.sg[c] = o.va[c] = o.en[c] = o.vr[c] = 0.0;
        const size_t row = (size_t)(in_k ? k : 0) * nphi, rowv = (size_t)(in_v ? k + 1 : 0) * nphi;
        if (pair_in) {
            if (in_k) {
                const D2 s2 = LD2(P.sigma + row + jin[0]), v2 = LD2(P.vazi + row + jin[0]);
                o.sg[0] = s2.x, o.sg[C - 1] = s2.y, o.va[0] = v2.x, o.va[C - 1] = v2.y;
                if (ADI) {
                    const D2 e2 = LD2(P.energy + row + jin[0]);
                    o.en[0] = e2.x, o.en[C - 1] = e2.y;
                }
            }
            if (in_v) {
                const D2 r2 = LD2(P.vrad + rowv + jin[0]);
                o.vr[0] = r2.x, o.vr[C - 1] = r2.y;
            }
        } else {
#pragma unroll
            for (int c = 0; c < C; ++c) {
                if (in_k) {
                    o.sg[c] = P.sigma[row + jin[c]];
                    o.va[c] = P.vazi[row + jin[c]];
                    if (ADI)
                        o.en[c] = P.energy[row + jin[c]];
                }
                if (in_v)
                    o.vr[c] = P.vrad[rowv + jin[c]];
            }
        }
    };
    // ring k (raw) -> newest window slot; vr_k = v_r(k) from the previous ring's fetch
    double vr_last[C];
    auto convert = [&](int k, const RingRaw &o) {
        const bool in_k = k >= 0 && k < nr;
        const ThetaRow tk = crow_load(P.theta_tab, in_k ? k : 0);
        const double r = tk.rmed, romega = tk.r_omega;
        const double va_n = lane_next(o.va[0]); // v_phi of cell j+1 of the last cell of the lane
#pragma unroll
        for (int c = 0; c < C; ++c) {
            const double van = c == C - 1 ? va_n : o.va[c == C - 1 ? c : c + 1];
            w[0][0][c] = o.sg[c];
            w[0][1][c] = in_k ? o.vr[c] : 0.0;                        // rm+ / Sigma = v_r(k+1)   (:484-485)
            w[0][2][c] = in_k ? vr_last[c] : 0.0;                     // rm- / Sigma = v_r(k)
            w[0][3][c] = in_k ? (van + romega) * r : 0.0;             // L+ / Sigma = (v_phi(j+1) + r Omega) r
            w[0][4][c] = in_k ? (o.va[c] + romega) * r : 0.0;         // L- / Sigma
            if (ADI) {
                w[0][NQ - 1][c] = in_k ? o.en[c] * fast_rcp(o.sg[c]) : 0.0;
                er[0][c] = o.en[c];
            }
            vp[0][c] = o.va[c];
            vr_last[c] = o.vr[c];
        }
    };
    // Software pipeline of the memory traffic: ring m+1 is in flight while ring m-2 is computed;
    // at the bottom of an iteration the arrived ring is converted, the loads of ring m+2 are
    // issued, and only then the iteration's stores.  The one s_waitcnt vmcnt(0) per iteration then
    // meets operations that are a whole compute phase old (vmcnt counts stores too; waiting right
    // behind them costs a round trip per ring at 2-3 waves per SIMD).
    RingRaw nxt;
    fetch(r0 - 4, nxt);
#pragma unroll
    for (int c = 0; c < C; ++c)
        vr_last[c] = nxt.vr[c]; // v_r(r0-3)
    fetch(r0 - 3, nxt);
    convert(r0 - 3, nxt);
    fetch(r0 - 2, nxt);
    int ns_prev = 0;

    for (int m = r0 - 3; m <= r1 + 1; ++m) {
        // ---- per-ring scalars of this iteration in one batch ----------------------------------
        const int k = m - 1, i = m - 2;
        const bool do_i = i >= r0 - 1 && i >= 0 && i < r1;
        const RadRow rk = crow_load(P.rad_tab, (k < -1 ? -1 : k) + 1);
        const ThetaRow ti = crow_load(P.theta_tab, do_i ? i : 0);
        const ShiftRow si = crow_load((const ShiftRow *)P.shift_tab, do_i ? i : 0);
        DampRow di;
        if (DAMP)
            di = crow_load(W.damp_tab, do_i ? i : 0);
        // ---- R: slopes of ring m-1, fluxes through interface k = m-1 --------------------------
        double F0[NQ][C];
        {
            const double idr_m = rk.idr_up;          // 1 / (Rmed[m] - Rmed[m-1]) when both rings exist
            const bool lim_ok = k > 0 && k < nr - 1; // boundary rings carry no slope (:360-372)
            const bool open = k > 0 && k < nr;       // interface carries a flux
            const double g = dt * rk.gphi;
            bool up[C];
            double dist[C];
#pragma unroll
            for (int c = 0; c < C; ++c) {
                const double v = w[1][2][c]; // v_r(m-1)
                up[c] = v > 0.0;
                dist[c] = up[c] ? (rk.dr_lo - v * dt) : -(rk.dr_hi + v * dt);
            }
            double Fc[C];
#pragma unroll
            for (int q = 0; q < NQ; ++q) {
#pragma unroll
                for (int c = 0; c < C; ++c) {
                    const double d0 = (w[0][q][c] - w[1][q][c]) * idr_m;
                    const double hs0 = lim_ok ? half_limiter(lim, d0, d1[q][c]) : 0.0; // ring m-1
                    const double st = (up[c] ? w[2][q][c] : w[1][q][c]) + dist[c] * (up[c] ? hs1[q][c] : hs0);
                    if (q == 0) {
                        Fc[c] = open ? g * st * w[1][2][c] : 0.0; // mass flux g rho* v
                        F0[q][c] = Fc[c];
                    } else {
                        F0[q][c] = st * Fc[c];
                    }
                    d1[q][c] = d0;
                    hs1[q][c] = hs0;
                }
            }
        }
        // ---- update of ring i = m-2, azimuthal passes, velocities -----------------------------
        bool out_on = false, out_pair = false;
        int out_g[C];
        double o_vr[C], o_va[C], o_s[C], o_e[C];
#pragma unroll
        for (int c = 0; c < C; ++c)
            out_g[c] = 0, o_vr[c] = o_va[c] = o_s[c] = o_e[c] = 0.0;
        if (do_i) {
            const double invsurf = ti.invsurf;
            double S[C], Q[4][C], E[C], V[C];
            const double mean = si.mean;
            const double vconst = si.vconst;
            const double vadd = P.fast_transport ? 0.0 : vconst;
#pragma unroll
            for (int c = 0; c < C; ++c) {
                const double s0 = w[2][0][c];
                S[c] = s0 + (F1[0][c] - F0[0][c]) * invsurf;
#pragma unroll
                for (int q = 0; q < 4; ++q)
                    Q[q][c] = s0 * w[2][q + 1][c] + (F1[q + 1][c] - F0[q + 1][c]) * invsurf;
                E[c] = ADI ? er[2][c] + (F1[NQ - 1][c] - F0[NQ - 1][c]) * invsurf : 0.0;
                V[c] = vadd + (vp[2][c] - mean);
            }
            const double dxtheta = ti.dxtheta;
            const double invdx = ti.inv_dxtheta;
            const double geo_dt = ti.dr_invsurf * dt;
            theta_pass<C, ADI, false, 0>(lim, 0, 0, geo_dt, dxtheta, invdx, dt, V, 0.0, S, Q, E);
            if (P.fast_transport) {
                if (vconst * dt > 0.0)
                    theta_pass<C, ADI, false, 1>(lim, 0, 0, geo_dt, dxtheta, invdx, dt, V, vconst, S, Q, E);
                else
                    theta_pass<C, ADI, false, 2>(lim, 0, 0, geo_dt, dxtheta, invdx, dt, V, vconst, S, Q, E);
            }
            int ns = si.nshift % nphi;
            ns = ns < 0 ? ns + nphi : ns;
            if (i >= r0) {
                // the previous ring sits Nshift[i] - Nshift[i-1] lanes further right
                int dsh = ns - ns_prev;
                dsh = dsh > nphi / 2 ? dsh - nphi : (dsh < -(nphi / 2) ? dsh + nphi : dsh);
                double rp[C], sp[C];
                if (dsh == 0) {
#pragma unroll
                    for (int c = 0; c < C; ++c)
                        rp[c] = rmp_prev[c], sp[c] = S_prev[c];
                } else if (dsh > 0) {
                    const double rn = lane_next(rmp_prev[0]), sn = lane_next(S_prev[0]);
#pragma unroll
                    for (int c = 0; c < C; ++c) {
                        rp[c] = c == C - 1 ? rn : rmp_prev[c == C - 1 ? c : c + 1];
                        sp[c] = c == C - 1 ? sn : S_prev[c == C - 1 ? c : c + 1];
                    }
                } else {
                    const double rl = lane_prev(rmp_prev[C - 1]), sl = lane_prev(S_prev[C - 1]);
#pragma unroll
                    for (int c = 0; c < C; ++c) {
                        rp[c] = c == 0 ? rl : rmp_prev[c == 0 ? 0 : c - 1];
                        sp[c] = c == 0 ? sl : S_prev[c == 0 ? 0 : c - 1];
                    }
                }
                const double lp_l = lane_prev(Q[2][C - 1]); // L+ and Sigma of cell j-1
                const double s_l = lane_prev(S[C - 1]);
                const double invr = ti.invr, romega = ti.r_omega;
                const int row = i * nphi;
                int jout[C];
#pragma unroll
                for (int c = 0; c < C; ++c) {
                    int jo = jin[c] + ns;
                    jout[c] = jo >= nphi ? jo - nphi : jo;
                    const double lpm = c == 0 ? lp_l : Q[2][c == 0 ? 0 : c - 1];
                    const double sm = c == 0 ? s_l : S[c == 0 ? 0 : c - 1];
                    double vr = 0.0;
                    if (i != 0)
                        vr = (rp[c] + Q[1][c]) * fast_rcp(sp[c] + S[c]);
                    double va = (lpm + Q[3][c]) * fast_rcp(sm + S[c]) * invr - romega;
                    double sf = S[c] < P.sigma_floor_abs ? P.sigma_floor_abs : S[c];
                    double e = ADI ? clamp_energy_fast(P, E[c], sf) : 0.0;
                    const int g = row + jout[c];
                    if (DAMP) {
                        vr = damp_apply(vr, di.tvr, si.ev, W.vrad0, g, 0.0);
                        va = damp_apply(va, di.tva, si.es, W.vazi0, g, 0.0);
                        sf = damp_apply(sf, di.tsg, si.es, W.sigma0, g, W.sigma_floor_abs);
                        if (ADI)
                            e = damp_apply(e, di.ten, si.es, W.energy0, g, 0.0);
                    }
                    o_vr[c] = vr, o_va[c] = va, o_s[c] = sf, o_e[c] = e;
                    out_g[c] = g;
                }
                out_on = true;
                out_pair = pair_valid && __builtin_amdgcn_ballot_w64(jout[C - 1] != jout[0] + 1) == 0;
            }
            ns_prev = ns;
#pragma unroll
            for (int c = 0; c < C; ++c) {
                rmp_prev[c] = Q[0][c];
                S_prev[c] = S[c];
            }
        }
        // ---- bottom: rotate, take ring m+1, start ring m+2, then this iteration's stores ------
#pragma unroll
        for (int c = 0; c < C; ++c) {
#pragma unroll
            for (int q = 0; q < NQ; ++q) {
                w[2][q][c] = w[1][q][c];
                w[1][q][c] = w[0][q][c];
                F1[q][c] = F0[q][c];
            }
            er[2][c] = er[1][c], er[1][c] = er[0][c];
            vp[2][c] = vp[1][c], vp[1][c] = vp[0][c];
        }
        if (m < r1 + 1) {
            convert(m + 1, nxt);
            if (m < r1)
                fetch(m + 2, nxt);
        }
        if (out_on) {
            if (out_pair) {
                if (valid[0]) {
                    ST2(W.vrad + out_g[0], (D2{o_vr[0], o_vr[C - 1]}));
                    ST2(W.vazi + out_g[0], (D2{o_va[0], o_va[C - 1]}));
                    ST2(W.sigma + out_g[0], (D2{o_s[0], o_s[C - 1]}));
                    if (ADI)
                        ST2(W.energy + out_g[0], (D2{o_e[0], o_e[C - 1]}));
                }
            } else {
#pragma unroll
                for (int c = 0; c < C; ++c)
                    if (valid[c]) {
                        W.vrad[out_g[c]] = o_vr[c];
                        W.vazi[out_g[c]] = o_va[c];
                        W.sigma[out_g[c]] = o_s[c];
                        if (ADI)
                            W.energy[out_g[c]] = o_e[c];
                    }
            }
            if (i == nr - 1) { // v_r row Nr is neither transported nor shifted: copied column by column
#pragma unroll
                for (int c = 0; c < C; ++c)
                    if (valid[c]) {
                        double v = P.vrad[nr * nphi + jin[c]];
                        if (DAMP) {
                            const DampRow dn = crow_load(W.damp_tab, nr);
                            v = damp_apply(v, dn.tvr, si.ev_top, W.vrad0, nr * nphi + jin[c], 0.0);
                        }
                        W.vrad[nr * nphi + jin[c]] = v;
                    }
            }
        }
    }
}

// ---------------------------------------------------------------------------
// cfl.cpp:185-376 condition_cfl.  k_ring_mean gives <v_phi>.  dt_cell = CFL / sqrt(sum of
// the squared inverse limits) and both sqrt and the quotient are monotone, so
// min_cells dt_cell == CFL / sqrt(max_cells sum): k_cfl_cells reduces the per-cell sums to one
// maximum per block (no atomics); k_cfl_final folds the block maxima, applies sqrt and the
// quotient once, and adds the per-ring FARGO shear limit (:207-220).
#define CFL_ROWS 8
// One thread owns a phi column and walks CFL_ROWS rings (v_r(i+1) of one ring is v_r(i) of the
// next, so every value is loaded once); per-block maxima, no atomics.
template <bool ROWU> __global__ void k_cfl_cells(const Dev P, double *part)
{
    const int j = blockIdx.x * blockDim.x + threadIdx.x;
    const int r0_ = P.first_active + (blockIdx.y * blockDim.y + threadIdx.y) * CFL_ROWS;
    const int r0 = ROWU ? __builtin_amdgcn_readfirstlane(r0_) : r0_;
    double s = 0.0;
    if (j < P.nphi && r0 < P.active_size) {
        const int jn = JNEXT;
        const int r1 = r0 + CFL_ROWS < P.active_size ? r0 + CFL_ROWS : P.active_size;
        const double lf = P.leapfrog ? 0.6 : 1.0;
        const double C2 = P.art_visc_factor * P.art_visc_factor;
        const double gg1 = P.gamma * (P.gamma - 1.0), inv_sqrt_gamma = 1.0 / sqrt(P.gamma);
        const double inv_limit = 1.0 / P.heating_cooling_cfl_limit;
        double vr0 = P.vrad[IDX(r0, j)];
        for (int i = r0; i < r1; ++i) {
            const double vr1 = P.vrad[IDX(i + 1, j)];
            const double inv_dxr = P.InvDiffRsup[i];         // 1 / (Rsup - Rinf)
            const double inv_dxa = P.InvRmed[i] * P.invdphi; // 1 / (Rmed dphi)
            const double inv_cell = dmax(inv_dxr, inv_dxa);  // 1 / min(dxRadial, dxAzimuthal)
            const double va = P.vazi[IDX(i, j)];
            const double van = P.vazi[IDX(i, jn)];
            const double vres = P.fast_transport ? va - P.vmean_c[i] : va;
            // isothermal: c_s and the alpha viscosity are per-ring constants (set once at init)
            double cs, nu;
            if (P.adiabatic && P.lazy_derived) { // k_adi_cs_h + k_viscosity in registers
                cs = sqrt(gg1 * P.energy[IDX(i, j)] * fast_rcp(P.sigma[IDX(i, j)]));
                const double H = cs * inv_sqrt_gamma * P.g_inv_omk[i];
                nu = P.alpha_viscosity ? P.alpha * H * cs : P.nu_const;
            } else {
                cs = P.adiabatic ? P.soundspeed[IDX(i, j)] : P.cs_ring[i];
                nu = P.adiabatic ? P.viscosity[IDX(i, j)] : (P.alpha_viscosity ? P.nu_ring[i] : P.nu_const);
            }
            const double invdt1 = cs * inv_cell;
            const double invdt2 = vr0 * inv_dxr;
            const double invdt3 = vres * inv_dxa;
            double invdt4;
            if (P.art_visc == FCPT_ARTVISC_SN) {
                double dvRadial = vr1 - vr0;
                double dvAzimuthal = van - va;
                dvRadial = dvRadial > 0.0 ? 0.0 : -dvRadial;
                dvAzimuthal = dvAzimuthal > 0.0 ? 0.0 : -dvAzimuthal;
                invdt4 = 4.0 * C2 * dmax(dvRadial * inv_dxr, dvAzimuthal * inv_dxa) * lf;
            } else { // the TW formula is also used for ArtificialViscosity: None (cfl.cpp:292)
                const double eps_rr = (vr1 - vr0) * P.InvDiffRsup[i];
                const double eps_pp = P.InvRmed[i] * ((van - va) * P.invdphi + 0.5 * (vr1 + vr0));
                const double mdiv_V = -dmin(eps_rr + eps_pp, 0.0);
                invdt4 = 4.0 * C2 * mdiv_V * lf;
            }
            const double invdt5 = 4.0 * nu * (inv_cell * inv_cell) * lf;
            double invdt6 = 0.0;
            if (P.adiabatic) {
                if (P.lazy_derived)
                    invdt6 = inv_limit * fabs((P.qplus[IDX(i, j)] - P.qminus[IDX(i, j)]) * fast_rcp(P.energy[IDX(i, j)])) * lf;
                else
                    invdt6 = inv_limit * fabs((P.qplus[IDX(i, j)] - P.qminus[IDX(i, j)]) / P.energy[IDX(i, j)]) * lf;
            }
            s = dmax(s, invdt1 * invdt1 + invdt2 * invdt2 + invdt3 * invdt3 + invdt4 * invdt4 + invdt5 * invdt5 +
                            invdt6 * invdt6);
            vr0 = vr1;
        }
    }
    for (int off = 32; off > 0; off >>= 1)
        s = dmax(s, __shfl_down(s, off, 64));
    __shared__ double s_w[4];
    const int tid = threadIdx.y * blockDim.x + threadIdx.x;
    if ((tid & 63) == 0)
        s_w[tid >> 6] = s;
    __syncthreads();
    if (tid == 0)
        part[blockIdx.y * gridDim.x + blockIdx.x] = dmax(dmax(s_w[0], s_w[1]), dmax(s_w[2], s_w[3]));
}
// Ring mean and per-cell limits in one pass: a block owns a ring, keeps its v_phi in registers
// (CFL_MAXP pairs per thread), sums them (<v_phi>, cfl.cpp:196-205), then evaluates the cells of
// the ring against that mean (:222-330) -- v_phi is read once instead of once by k_ring_mean and
// once by k_cfl_cells.  One partial maximum per ring.
#define CFL_MAXP 8
template <bool ADI> __global__ void __launch_bounds__(256) k_cfl_rings(const Dev P, double *part)
{
    const int i = xcd_block(blockIdx.x, gridDim.x); // neighbouring rings share the v_r row between them: same L2
    const int nphi = P.nphi, npair = nphi >> 1;
    const int t = threadIdx.x;
    const size_t row = (size_t)i * nphi;
    D2 va[CFL_MAXP];
    double acc = 0.0, acc2 = 0.0;
#pragma unroll
    for (int n = 0; n < CFL_MAXP; ++n) {
        const int p = t + n * 256;
        va[n] = D2{0.0, 0.0};
        if (p < npair)
            va[n] = *(const D2 *)(P.vazi + row + 2 * p);
    }
#pragma unroll
    for (int n = 0; n < CFL_MAXP; ++n) {
        acc += va[n].x;
        acc2 += va[n].y;
    }
    acc += acc2;
    for (int off = 32; off > 0; off >>= 1)
        acc += __shfl_down(acc, off, 64);
    __shared__ double s_w[4], s_m[4];
    if ((t & 63) == 0)
        s_w[t >> 6] = acc;
    __syncthreads();
    const double mean = ((s_w[0] + s_w[1]) + (s_w[2] + s_w[3])) / (double)nphi;
    if (t == 0)
        P.vmean[i] = mean;
    double s = 0.0;
    if (i >= P.first_active && i < P.active_size) {
        const double lf = P.leapfrog ? 0.6 : 1.0;
        const double C2 = P.art_visc_factor * P.art_visc_factor;
        const double inv_dxr = P.InvDiffRsup[i];         // 1 / (Rsup - Rinf)
        const double inv_rmed = P.InvRmed[i];
        const double inv_dxa = inv_rmed * P.invdphi;     // 1 / (Rmed dphi)
        const double inv_cell = dmax(inv_dxr, inv_dxa);  // 1 / min(dxRadial, dxAzimuthal)
        const double gg1 = P.gamma * (P.gamma - 1.0), inv_sqrt_gamma = 1.0 / sqrt(P.gamma);
        const double inv_limit = 1.0 / P.heating_cooling_cfl_limit;
        const double inv_omk = ADI ? P.g_inv_omk[i] : 0.0;
        const double cs_iso = ADI ? 0.0 : P.cs_ring[i];
        const double nu_iso = ADI ? 0.0 : (P.alpha_viscosity ? P.nu_ring[i] : P.nu_const);
        const double sub = P.fast_transport ? mean : 0.0;
#pragma unroll
        for (int n = 0; n < CFL_MAXP; ++n) {
            const int p = t + n * 256;
            if (p < npair) {
                const int j = 2 * p;
                const D2 r0 = *(const D2 *)(P.vrad + row + j), r1 = *(const D2 *)(P.vrad + row + nphi + j);
                const double van1 = P.vazi[row + (j + 2 >= nphi ? 0 : j + 2)]; // v_phi of cell j+2
                D2 e2 = {0.0, 0.0}, s2 = {1.0, 1.0}, qp = {0.0, 0.0}, qm = {0.0, 0.0};
                if (ADI) {
                    e2 = *(const D2 *)(P.energy + row + j);
                    s2 = *(const D2 *)(P.sigma + row + j);
                    qp = *(const D2 *)(P.qplus + row + j);
                    qm = *(const D2 *)(P.qminus + row + j);
                }
#pragma unroll
                for (int c = 0; c < 2; ++c) {
                    const double vr0 = c ? r0.y : r0.x, vr1 = c ? r1.y : r1.x;
                    const double v = c ? va[n].y : va[n].x, van = c ? van1 : va[n].y;
                    double cs = cs_iso, nu = nu_iso;
                    if (ADI) { // k_adi_cs_h + k_viscosity in registers
                        const double e = c ? e2.y : e2.x, sg = c ? s2.y : s2.x;
                        cs = sqrt(gg1 * e * fast_rcp(sg));
                        const double H = cs * inv_sqrt_gamma * inv_omk;
                        nu = P.alpha_viscosity ? P.alpha * H * cs : P.nu_const;
                    }
                    const double invdt1 = cs * inv_cell;
                    const double invdt2 = vr0 * inv_dxr;
                    const double invdt3 = (v - sub) * inv_dxa;
                    double invdt4;
                    if (P.art_visc == FCPT_ARTVISC_SN) {
                        double dvRadial = vr1 - vr0;
                        double dvAzimuthal = van - v;
                        dvRadial = dvRadial > 0.0 ? 0.0 : -dvRadial;
                        dvAzimuthal = dvAzimuthal > 0.0 ? 0.0 : -dvAzimuthal;
                        invdt4 = 4.0 * C2 * dmax(dvRadial * inv_dxr, dvAzimuthal * inv_dxa) * lf;
                    } else { // the TW formula is also used for ArtificialViscosity: None (cfl.cpp:292)
                        const double eps_rr = (vr1 - vr0) * inv_dxr;
                        const double eps_pp = inv_rmed * ((van - v) * P.invdphi + 0.5 * (vr1 + vr0));
                        const double mdiv_V = -dmin(eps_rr + eps_pp, 0.0);
                        invdt4 = 4.0 * C2 * mdiv_V * lf;
                    }
                    const double invdt5 = 4.0 * nu * (inv_cell * inv_cell) * lf;
                    double invdt6 = 0.0;
                    if (ADI) {
                        const double e = c ? e2.y : e2.x;
                        invdt6 = inv_limit * fabs(((c ? qp.y : qp.x) - (c ? qm.y : qm.x)) * fast_rcp(e)) * lf;
                    }
                    s = dmax(s, invdt1 * invdt1 + invdt2 * invdt2 + invdt3 * invdt3 + invdt4 * invdt4 +
                                    invdt5 * invdt5 + invdt6 * invdt6);
                }
            }
        }
    }
    for (int off = 32; off > 0; off >>= 1)
        s = dmax(s, __shfl_down(s, off, 64));
    if ((t & 63) == 0)
        s_m[t >> 6] = s;
    __syncthreads();
    if (t == 0)
        part[i] = dmax(dmax(s_m[0], s_m[1]), dmax(s_m[2], s_m[3]));
}
__global__ void __launch_bounds__(1024) k_cfl_final(const Dev P, const double *part, int nparts, int apply_policy)
{
    double smax = 0.0;
    for (int n = threadIdx.x; n < nparts; n += blockDim.x)
        smax = dmax(smax, part[n]);
    // FARGO shear limit, rings 0|1 (:207-208) and the active rings (:213-220)
    double dt = 1.0e300;
    for (int n = threadIdx.x; n < P.active_size; n += blockDim.x) {
        if (n == 0 || n >= P.first_active) {
            const double denom = fabs(P.vmean[n] * P.InvRmed[n] - P.vmean[n + 1] * P.InvRmed[n + 1]) + 1.0e-100;
            dt = dmin(dt, P.cfl * P.dphi / denom);
        }
    }
    for (int off = 32; off > 0; off >>= 1) {
        smax = dmax(smax, __shfl_down(smax, off, 64));
        dt = dmin(dt, __shfl_down(dt, off, 64));
    }
    __shared__ double s_s[16], s_d[16];
    const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
    if (lane == 0) {
        s_s[wave] = smax;
        s_d[wave] = dt;
    }
    __syncthreads();
    if (threadIdx.x == 0) {
        for (int w = 1; w < (int)(blockDim.x >> 6); ++w) {
            smax = dmax(smax, s_s[w]);
            dt = dmin(dt, s_d[w]);
        }
        if (nparts > 0)
            dt = dmin(dt, P.cfl / sqrt(smax));
        P.clk->cfl_bits = (unsigned long long)__double_as_longlong(dt);
        if (apply_policy) { // sim::CalculateTimeStep (simulation.cpp:100-118) for single-slab device loops
            const double a = P.cfl_max_var * P.clk->last_dt;
            const double rv = dt < a ? dt : a;
            P.clk->cfl_dt = rv;
            P.clk->last_dt = rv;
            P.clk->dt = rv;
        }
    }
}

// ---------------------------------------------------------------------------
// clock kernels (single thread)
__global__ void k_clock_set_dt(DevClock *clk, double dt) { clk->dt = dt; }
// leapfrog sub-steps: mode 0: step <- dt (host value), mode 1: step <- clk->dt (device value),
// mode 2: step <- the saved one; then clk->dt = factor * step.  The full step is parked in cfl_dt.
__global__ void k_clock_scale_dt(DevClock *clk, int mode, double dt, double factor)
{
    if (mode == 0)
        clk->cfl_dt = dt;
    else if (mode == 1)
        clk->cfl_dt = clk->dt;
    clk->dt = mode == 2 && factor == 1.0 ? clk->cfl_dt : clk->cfl_dt * factor;
}
__global__ void k_clock_advance(DevClock *clk)
{
    clk->time += clk->dt;
    clk->n_hydro_iter += 1;
}
__global__ void k_clock_export_cfl(const DevClock *clk, double *out)
{
    *out = __longlong_as_double((long long)clk->cfl_bits);
}
__global__ void k_clock_policy_ptr(DevClock *clk, double cfl_max_var, const double *cfl_global)
{
    const double cfl_dt = *cfl_global;
    const double a = cfl_max_var * clk->last_dt;
    const double rv = cfl_dt < a ? cfl_dt : a;
    clk->cfl_dt = rv;
    clk->last_dt = rv;
    clk->dt = rv;
}
// sim::CalculateTimeStep (simulation.cpp:100-118): rv = min(CFLmaxVar*last_dt, cfl_dt)
__global__ void k_clock_policy(DevClock *clk, double cfl_max_var, int use_device_cfl, double cfl_global)
{
    const double cfl_dt = use_device_cfl ? __longlong_as_double((long long)clk->cfl_bits) : cfl_global;
    const double a = cfl_max_var * clk->last_dt;
    const double rv = cfl_dt < a ? cfl_dt : a;
    clk->cfl_dt = rv;
    clk->last_dt = rv;
    clk->dt = rv;
}


// ---------------------------------------------------------------------------
// per-kernel HIP-event timing (fcpt_profile_start/stop)
const char *const kKernelNames[KID_COUNT] = {
    "k_potential", "k_source_vr", "k_source_va", "k_compression_heating", "k_tw_q", "k_tw_va", "k_tw_vr",
    "k_sn_q", "k_sn_e", "k_sn_vr", "k_sn_va", "k_temperature_range", "k_adi_cs_h", "k_iso_cs_h",
    "k_viscosity", "k_pressure", "k_temperature", "k_stress_diag", "k_stress_rphi", "k_visc_va",
    "k_visc_vr", "k_qplus_qminus", "k_substep3", "k_boundary", "k_damping", "k_transport_radial",
    "k_ring_mean", "k_transport_theta1", "k_transport_theta2", "k_velocities", "k_cfl_final",
    "k_cfl_cells", "k_clock", "k_src_fused", "k_av_fused", "k_visc_fused", "k_source_march",
    "k_transport_theta_fused", "k_transport_theta_march", "k_transport_fused"};

thread_local Profiler *g_prof = nullptr;

void Profiler::begin(int id, hipStream_t st)
{
    if (!((mask >> id) & 1ull) || used + 2 > (int)events.size())
        return;
    (void)hipEventRecord(events[used], st);
    open_id = id;
}
void Profiler::end(int id, hipStream_t st)
{
    if (open_id != id)
        return;
    (void)hipEventRecord(events[used + 1], st);
    ids.push_back(id);
    used += 2;
    open_id = -1;
}

#define KLAUNCH(id, kernel, grid, block, ...)                              \
    do {                                                                   \
        if (g_prof)                                                        \
            g_prof->begin((id), st);                                       \
        hipLaunchKernelGGL(kernel, (grid), (block), 0, st, __VA_ARGS__);   \
        if (g_prof)                                                        \
            g_prof->end((id), st);                                         \
    } while (0)

// ---------------------------------------------------------------------------
// launchers
#define LAUNCH2D(id, kernel, nrows, ...)                                             \
    do {                                                                             \
        if ((nrows) > 0) {                                                           \
            const Launch2D l = launch2d((nrows), P.nphi);                            \
            if (l.block.x >= 64)                                                     \
                KLAUNCH(id, (kernel<true>), l.grid, l.block, __VA_ARGS__);           \
            else                                                                     \
                KLAUNCH(id, (kernel<false>), l.grid, l.block, __VA_ARGS__);          \
        }                                                                            \
    } while (0)
#define LAUNCH2D_T(id, kernel, targ, nrows, ...)                                     \
    do {                                                                             \
        if ((nrows) > 0) {                                                           \
            const Launch2D l = launch2d((nrows), P.nphi);                            \
            if (l.block.x >= 64)                                                     \
                KLAUNCH(id, (kernel<targ, true>), l.grid, l.block, __VA_ARGS__);     \
            else                                                                     \
                KLAUNCH(id, (kernel<targ, false>), l.grid, l.block, __VA_ARGS__);    \
        }                                                                            \
    } while (0)

void launch_potential(const Dev &P, hipStream_t st) { LAUNCH2D(KID_POTENTIAL, k_potential, P.nr, P); }

void launch_source(const Dev &P, hipStream_t st)
{
    // update_with_sourceterms, SourceEuler.cpp:435-452
    LAUNCH2D(KID_SOURCE_VR, k_source_vr, P.maxmo_no_ghost_vr - P.one_no_ghost_vr, P);
    LAUNCH2D(KID_SOURCE_VA, k_source_va, P.max_no_ghost - P.zero_no_ghost, P);
    if (P.adiabatic)
        LAUNCH2D(KID_COMPRESSION, k_compression_heating, P.nr - 1, P);
}

void launch_artificial_viscosity(const Dev &P, hipStream_t st)
{
    // art_visc::update_with_artificial_viscosity, artificial_viscosity.cpp:11-26
    if (P.art_visc == FCPT_ARTVISC_TW) {
        LAUNCH2D(KID_TW_Q, k_tw_q, P.nr, P);
        LAUNCH2D(KID_TW_VA, k_tw_va, P.nr - 2, P);
        LAUNCH2D(KID_TW_VR, k_tw_vr, P.maxmo_no_ghost_vr - P.one_no_ghost_vr, P);
    } else if (P.art_visc == FCPT_ARTVISC_SN) {
        LAUNCH2D(KID_SN_Q, k_sn_q, P.nr, P);
        if (P.adiabatic && P.art_visc_dissipation)
            LAUNCH2D(KID_SN_E, k_sn_e, P.max_no_ghost - P.zero_no_ghost, P);
        LAUNCH2D(KID_SN_VR, k_sn_vr, P.maxmo_no_ghost_vr - P.one_no_ghost_vr, P);
        LAUNCH2D(KID_SN_VA, k_sn_va, P.max_no_ghost - P.zero_no_ghost, P);
    }
    if (P.adiabatic && P.art_visc_dissipation)
        LAUNCH2D(KID_TRANGE, k_temperature_range, P.nr, P);
}

void launch_recalculate_viscosity(const Dev &P, hipStream_t st)
{
    // recalculate_viscosity, SourceEuler.cpp:205-223 (AspectRatioMode 0)
    if (P.adiabatic)
        LAUNCH2D(KID_ADI_CS_H, k_adi_cs_h, P.nr, P);
    if (P.alpha_viscosity && P.adiabatic)
        LAUNCH2D(KID_VISCOSITY, k_viscosity, P.nr, P); // isothermal alpha-nu never changes after init
}

void launch_viscosity_field(const Dev &P, hipStream_t st) { LAUNCH2D(KID_VISCOSITY, k_viscosity, P.nr, P); }

void launch_iso_cs_h(const Dev &P, const double *cs_ring, hipStream_t st)
{
    LAUNCH2D(KID_ISO_CS_H, k_iso_cs_h, P.nr, P, cs_ring);
}

void launch_source_fused(const Dev &P, hipStream_t st)
{
    LAUNCH2D(KID_SRC_FUSED, k_src_fused, P.nr + 1, P);
    LAUNCH2D(KID_AV_FUSED, k_av_fused, P.nr + 1, P);
}
// whole source step in one marching pass (isothermal, Nphi >= 128); returns false if not applicable
int launch_source_march(const Dev &P, hipStream_t st)
{
    if (P.nphi < 128)
        return 0;
    if (P.adiabatic) {
        if (const char *e = getenv("FCPT_MARCH_SOURCE_ADI"))
            if (e[0] == '0')
                return 0;
        int rows = 24;
        if (const char *e = getenv("FCPT_SOURCE_ROWS")) // tuning knob
            rows = atoi(e) > 0 ? atoi(e) : rows;
        const int segs = (P.nphi + MARCH_VALID - 1) / MARCH_VALID;
        const int chunks = (P.nr + 1 + rows - 1) / rows;
        const dim3 grid((segs * chunks + 3) / 4), block(256);
        const bool cool = P.cooling_surface != 0 || P.cooling_beta != 0;
#define ADIK(AV_)                                                                                 \
    if (cool)                                                                                     \
        KLAUNCH(KID_SOURCE_MARCH, (k_source_march_adi<AV_, true>), grid, block, P, segs, rows);  \
    else                                                                                          \
        KLAUNCH(KID_SOURCE_MARCH, (k_source_march_adi<AV_, false>), grid, block, P, segs, rows)
        if (P.art_visc == FCPT_ARTVISC_TW) {
            ADIK(1);
        } else if (P.art_visc == FCPT_ARTVISC_SN) {
            ADIK(2);
        } else {
            ADIK(0);
        }
#undef ADIK
        return -segs; // marched, no ring sums
    }
    int rows = 24; // measured at 2048x4096: 16 / 24 / 32 / 48 / 64 rings -> 0.133 / 0.132 / 0.141 / 0.152 / 0.188 ms
    if (const char *e = getenv("FCPT_SOURCE_ROWS")) // tuning knob
        rows = atoi(e) > 0 ? atoi(e) : rows;
    const int segs = (P.nphi + MARCH_VALID - 1) / MARCH_VALID;
    // per-segment ring sums of v_phi, so that the transport's k_ring_mean reads 70 partials per ring
    // instead of the ring itself
    int ring_sums = segs <= P.ring_pstride;
    if (const char *e = getenv("FCPT_SOURCE_RING_PARTS"))
        ring_sums = ring_sums && e[0] != '0';
    const int chunks = (P.nr + 1 + rows - 1) / rows;
    const int waves = segs * chunks;
    const dim3 grid((waves + 3) / 4), block(256);
    if (P.art_visc == FCPT_ARTVISC_TW)
        KLAUNCH(KID_SOURCE_MARCH, k_source_march<1>, grid, block, P, segs, rows, ring_sums);
    else if (P.art_visc == FCPT_ARTVISC_SN)
        KLAUNCH(KID_SOURCE_MARCH, k_source_march<2>, grid, block, P, segs, rows, ring_sums);
    else
        KLAUNCH(KID_SOURCE_MARCH, k_source_march<0>, grid, block, P, segs, rows, ring_sums);
    return ring_sums ? segs : -segs; // < 0: marched, but no ring sums
}
void launch_viscous_fused(const Dev &P, hipStream_t st) { LAUNCH2D(KID_VISC_FUSED, k_visc_fused, P.nr + 1, P); }
void launch_substep3_after_fused(const Dev &P, hipStream_t st)
{
    // SubStep3 (SourceEuler.cpp:956-1051) with Q+ already evaluated by k_visc_fused
    LAUNCH2D(KID_TEMPERATURE, k_temperature, P.nr, P);
    LAUNCH2D(KID_SUBSTEP3, k_substep3, P.nr - 2, P, 1);
    LAUNCH2D(KID_TRANGE, k_temperature_range, P.nr, P);
}

void launch_stress(const Dev &P, hipStream_t st)
{
    LAUNCH2D(KID_STRESS_DIAG, k_stress_diag, P.nr, P);
    LAUNCH2D(KID_STRESS_RPHI, k_stress_rphi, P.nr - 1, P);
}

void launch_viscous_update(const Dev &P, hipStream_t st)
{
    LAUNCH2D(KID_VISC_VA, k_visc_va, P.nr - 2, P);
    LAUNCH2D(KID_VISC_VR, k_visc_vr, P.maxmo_no_ghost_vr - P.one_no_ghost_vr, P);
}

void launch_substep3_cooling_only(const Dev &P, hipStream_t st)
{
    // compute_heating_cooling_for_CFL at init (SourceEuler.cpp:1507-1547): Q+ = 0 (gas at rest), Q- / alpha
    LAUNCH2D(KID_SUBSTEP3, k_substep3, P.nr - 2, P, 0);
}

void launch_substep3(const Dev &P, int update_energy, hipStream_t st)
{
    // SubStep3, SourceEuler.cpp:956-1051 (update_energy = 1) or the Q+/Q- part of
    // compute_heating_cooling_for_CFL, :1507-1547 (update_energy = 0)
    if (update_energy)
        LAUNCH2D(KID_TEMPERATURE, k_temperature, P.nr, P);
    LAUNCH2D(KID_QPLUS, k_qplus_qminus, P.nr, P);
    LAUNCH2D(KID_SUBSTEP3, k_substep3, P.nr - 2, P, update_energy);
    if (update_energy)
        LAUNCH2D(KID_TRANGE, k_temperature_range, P.nr, P);
}

void launch_boundary(const Dev &P, hipStream_t st)
{
    const int bs = 256;
    KLAUNCH(KID_BOUNDARY, k_boundary, dim3((P.nphi + bs - 1) / bs), dim3(bs), P);
}

void launch_damping(const Dev &P, double *q, double *q0, const double *radius, const DampRange &r,
                    int is_density, hipStream_t st)
{
    if (r.type == FCPT_DAMP_NONE || r.lo > r.hi)
        return;
    KLAUNCH(KID_DAMPING, k_damping, dim3(r.hi - r.lo + 1), dim3(256), P, q, q0, radius, r.lo, r.type, r.rlim,
            r.redge, r.tau, is_density);
}

#define FALLBACK_BLOCKS 256 /* grid of the idle in-stream fallback kernels */
// one radial sweep + ring means (T1-T4); only_if: see k_transport_radial
static void launch_radial(const Dev &P, const int *only_if, hipStream_t st)
{
    const Launch2D l = launch2d((P.nr + RADIAL_ROWS - 1) / RADIAL_ROWS, P.nphi);
    const int gx = (int)l.grid.x, gy = (int)l.grid.y;
    const dim3 grid(only_if && gx * gy > FALLBACK_BLOCKS ? FALLBACK_BLOCKS : gx * gy);
    if (l.block.x >= 64)
        KLAUNCH(KID_TRANSPORT_RADIAL, k_transport_radial<true>, grid, l.block, P, only_if, gx, gy);
    else
        KLAUNCH(KID_TRANSPORT_RADIAL, k_transport_radial<false>, grid, l.block, P, only_if, gx, gy);
}
static void launch_shift_means(const Dev &P, hipStream_t st)
{
    KLAUNCH(KID_RING_MEAN, k_ring_mean, dim3((P.nr + 3) / 4), dim3(256), P, 1,
            P.src_ring_nparts ? (const double *)P.ring_part : (const double *)nullptr, P.src_ring_nparts, P.ring_pstride);
}
#define MARCHK(CC, PP, AA, DD)                                                                                      \
    KLAUNCH(KID_THETA_MARCH, (k_transport_theta_march<CC, AA, DD, PP>), grid, block, Wm, (const double *)P.vazi,   \
            (const double *)P.vrad, inB, tiles, rows, advance, only_if, nvb)
#define MARCHC(CC, PP)                   \
    if (P.adiabatic) {                   \
        if (Wm.damp_in_step)             \
            MARCHK(CC, PP, true, true);  \
        else                             \
            MARCHK(CC, PP, true, false); \
    } else {                             \
        if (Wm.damp_in_step)             \
            MARCHK(CC, PP, false, true); \
        else                             \
            MARCHK(CC, PP, false, false);\
    }
// azimuthal marching kernel on set B -> state grids of Wm; returns the tile count
static int launch_theta_march(const Dev &P, const Dev &Wm, int C, int periodic, int advance, const int *only_if,
                              hipStream_t st)
{
    ThetaSet inB = {P.rmpB, P.rmmB, P.lpB, P.lmB, P.sigB, P.eB};
    const int tstride = 64 * C - (THETA_LO + THETA_HI);
    const int tiles = periodic ? 1 : (P.nphi + tstride - 1) / tstride;
    int rows = THETA_ROWS;
    if (const char *e = getenv("FCPT_THETA_ROWS"))
        rows = atoi(e) > 0 ? atoi(e) : rows;
    const int chunks = (P.nr + rows - 1) / rows;
    const int waves = chunks * tiles;
    const int nvb = (waves + 3) / 4;
    const dim3 grid(only_if && nvb > FALLBACK_BLOCKS ? FALLBACK_BLOCKS : nvb), block(256);
    if (!periodic) { // tiled: 2 cells per lane (1, 4 and 6 were measured slower), DPP lane shifts
        MARCHC(2, false)
    } else if (C == 1) {
        MARCHC(1, true)
    } else if (C == 2) {
        MARCHC(2, true)
    } else {
        MARCHC(4, true)
    }
    return tiles;
}
#undef MARCHC
#undef MARCHK

TransportResult launch_transport(const Dev &P, const Dev &W, bool shear_safe, hipStream_t st)
{
    // P: view whose vrad/vazi are the velocities to transport; W: view that receives the new state
    // Transport, TransportEuler.cpp:112-136
    TransportResult res = {0, W.sigma, W.energy, W.vrad, W.vazi};
    // ---- everything in one kernel (tiled rings only) ------------------------------------------
    int CF = P.nphi >= 256 ? 1 : 0; // 1 cell per lane: 3 waves per SIMD (2 cells: 284 VGPRs, 1 wave)
    if (const char *e = getenv("FCPT_TRANSPORT_FUSED")) { // 0: off, 1 / 2: cells per lane
        const int v = atoi(e);
        CF = v == 0 ? 0 : ((v == 1 || v == 2) && P.nphi >= 128 * v ? v : CF);
    }
    if (CF) {
        Dev Wm = W; // the marching kernels cannot work in place
        Wm.sigma = W.sigA;
        Wm.energy = W.eA;
        Wm.vrad = P.vrad == W.vrad ? W.vrad_b : W.vrad;
        Wm.vazi = P.vazi == W.vazi ? W.vazi_b : W.vazi;
        launch_shift_means(P, st);
        int rows = TF_ROWS;
        if (const char *e = getenv("FCPT_TRANSPORT_ROWS"))
            rows = atoi(e) > 0 ? atoi(e) : rows;
        const int tstride = 64 * CF - (CF == 2 ? TfHalo<2>::lo + TfHalo<2>::hi : TfHalo<1>::lo + TfHalo<1>::hi);
        const int tiles = (P.nphi + tstride - 1) / tstride;
        const int chunks = (P.nr + rows - 1) / rows;
        const dim3 grid((chunks * tiles + 3) / 4), block(256);
        // shear_safe: dt comes from the CFL policy with CFL <= 0.8, so |Nshift[i] - Nshift[i-1]| <= 1 is
        // guaranteed (cfl.cpp:207-220) and the two idle fallback launches (5 us) are not queued; a
        // violation would still be detected and reported as FCPT_ESHEAR
        int fallback = shear_safe ? 0 : 1;
        if (const char *e = getenv("FCPT_TRANSPORT_FALLBACK"))
            fallback = e[0] != '0';
#define TFK(CC, AA, DD)                                                                                             \
    if (P.limiter == FCPT_LIMITER_MC)                                                                                \
        KLAUNCH(KID_TRANSPORT_FUSED, (k_transport_fused<CC, AA, DD, FCPT_LIMITER_MC>), grid, block, P, Wm, tiles, rows, fallback); \
    else                                                                                                             \
        KLAUNCH(KID_TRANSPORT_FUSED, (k_transport_fused<CC, AA, DD, FCPT_LIMITER_VANLEER>), grid, block, P, Wm, tiles, rows, fallback)
#define TFC(CC)               \
    if (P.adiabatic) {        \
        if (W.damp_in_step)   \
            TFK(CC, true, true);  \
        else                  \
            TFK(CC, true, false); \
    } else {                  \
        if (W.damp_in_step)   \
            TFK(CC, false, true); \
        else                  \
            TFK(CC, false, false);\
    }
        if (CF == 2) {
            TFC(2)
        } else {
            TFC(1)
        }
#undef TFC
#undef TFK
        // behind it, the two-kernel form: its blocks return at once unless the fused kernel met
        // |Nshift[i] - Nshift[i-1]| > 1 (a time step beyond the FARGO shear limit)
        if (fallback) {
            launch_radial(P, P.shift_jump, st);
            launch_theta_march(P, Wm, 2, 0, 0, P.shift_jump, st);
        }
        res.marched = tiles;
        res.sigma = Wm.sigma, res.energy = Wm.energy, res.vrad = Wm.vrad, res.vazi = Wm.vazi;
        return res;
    }
    launch_radial(P, nullptr, st);
    launch_shift_means(P, st);
    ThetaSet inB = {P.rmpB, P.rmmB, P.lpB, P.lmB, P.sigB, P.eB};
    ThetaOut outA = {P.rmpA, P.rmmA, P.lpA, P.lmA, P.sigA, P.eA};
    ThetaSet inA = {P.rmpA, P.rmmA, P.lpA, P.lmA, P.sigA, P.eA};
    ThetaOut outB = {P.rmpB, P.rmmB, P.lpB, P.lmB, P.sigB, P.eB};
    // fused azimuthal sweep when a lane-chunk size fits the ring, else the two-pass kernels
    int C = 0, periodic = 0;
    for (int c : {1, 2, 4})
        if (!C && P.nphi % c == 0 && P.nphi <= 64 * c && (c == 1 || P.nphi / c >= 1)) {
            C = c;
            periodic = 1;
        }
    if (!C && P.nphi > 64 * 2)
        C = 2;
    if (const char *e = getenv("FCPT_THETA_FUSED"))
        if (e[0] == '0')
            C = 0;
    bool march = C != 0;
    if (const char *e = getenv("FCPT_THETA_MARCH"))
        march = march && e[0] != '0';
    if (march) {
        res.marched = launch_theta_march(P, W, C, periodic, 1, nullptr, st);
    } else if (C) {
        const int tstride = 64 * C - 2 * THETA_HALO;
        const int tiles = periodic ? 1 : (P.nphi + tstride - 1) / tstride;
        const int waves = P.nr * tiles;
        const dim3 grid((waves + 3) / 4), block(256);
#define FUSED(CC)                                                                                      \
    if (P.adiabatic)                                                                                   \
        KLAUNCH(KID_THETA_FUSED, (k_transport_theta_fused<CC, true>), grid, block, P, inB, outA, tiles, periodic); \
    else                                                                                               \
        KLAUNCH(KID_THETA_FUSED, (k_transport_theta_fused<CC, false>), grid, block, P, inB, outA, tiles, periodic);
        if (C == 1) {
            FUSED(1)
        } else if (C == 2) {
            FUSED(2)
        } else {
            FUSED(4)
        }
#undef FUSED
        if (W.damp_in_step)
            LAUNCH2D_T(KID_VELOCITIES, k_velocities, true, P.nr, W, inA, (const double *)P.vrad);
        else
            LAUNCH2D_T(KID_VELOCITIES, k_velocities, false, P.nr, W, inA, (const double *)P.vrad);
    } else {
        LAUNCH2D_T(KID_THETA1, k_transport_theta, 1, P.nr, P, inB, outA);
        LAUNCH2D_T(KID_THETA2, k_transport_theta, 2, P.nr, P, inA, outB);
        if (W.damp_in_step)
            LAUNCH2D_T(KID_VELOCITIES, k_velocities, true, P.nr, W, inB, (const double *)P.vrad);
        else
            LAUNCH2D_T(KID_VELOCITIES, k_velocities, false, P.nr, W, inB, (const double *)P.vrad);
    }
    return res;
}

void launch_derived(const Dev &P, hipStream_t st)
{
    // recalculate_derived_disk_quantities, SourceEuler.cpp:225-249 (AspectRatioMode 0)
    if (P.adiabatic) {
        LAUNCH2D(KID_TEMPERATURE, k_temperature, P.nr, P);
        LAUNCH2D(KID_ADI_CS_H, k_adi_cs_h, P.nr, P);
        LAUNCH2D(KID_PRESSURE, k_pressure, P.nr, P);
        if (P.alpha_viscosity)
            LAUNCH2D(KID_VISCOSITY, k_viscosity, P.nr, P);
    } else {
        LAUNCH2D(KID_PRESSURE, k_pressure, P.nr, P);
    }
}

void launch_pressure(const Dev &P, hipStream_t st) { LAUNCH2D(KID_PRESSURE, k_pressure, P.nr, P); }
void launch_temperature(const Dev &P, hipStream_t st) { LAUNCH2D(KID_TEMPERATURE, k_temperature, P.nr, P); }

void launch_disk_on_body(const Dev &P, double x, double y, double r_object, double smoothing_fixed, double r_sm, double *out,
                         hipStream_t st)
{
    const int nrows = P.active_size - P.first_active;
    const dim3 grid((P.nphi + 255) / 256, nrows > 0 ? (nrows + DOB_ROWS - 1) / DOB_ROWS : 1), block(256);
    KLAUNCH(KID_POTENTIAL, k_disk_on_body, grid, block, P, x, y, r_object, smoothing_fixed, r_sm, P.cfl_part);
    KLAUNCH(KID_POTENTIAL, k_disk_on_body_final, dim3(1), dim3(256), (const double *)P.cfl_part, (int)(grid.x * grid.y), out);
}

void launch_cfl(const Dev &P, int apply_policy, hipStream_t st)
{
    // one block per ring: mean and cells in one pass (even Nphi up to 512 * CFL_MAXP; the isothermal
    // viscosity and sound speed per ring, or the lazily derived ones of the ideal EOS)
    bool rings = (P.nphi & 1) == 0 && P.nphi >= 128 && P.nphi <= 512 * CFL_MAXP && (!P.adiabatic || P.lazy_derived);
    if (const char *e = getenv("FCPT_CFL_RINGS"))
        rings = rings && e[0] != '0';
    if (rings) {
        if (P.adiabatic)
            KLAUNCH(KID_CFL_CELLS, k_cfl_rings<true>, dim3(P.nr), dim3(256), P, P.cfl_part);
        else
            KLAUNCH(KID_CFL_CELLS, k_cfl_rings<false>, dim3(P.nr), dim3(256), P, P.cfl_part);
        KLAUNCH(KID_CFL_INIT, k_cfl_final, dim3(1), dim3(1024), P, (const double *)P.cfl_part, P.nr, apply_policy);
        return;
    }
    KLAUNCH(KID_RING_MEAN, k_ring_mean, dim3((P.nr + 3) / 4), dim3(256), P, 0, (const double *)nullptr, 0, P.ring_pstride);
    const int nrows = P.active_size - P.first_active;
    int nparts = 0;
    if (nrows > 0) {
        const Launch2D l = launch2d((nrows + CFL_ROWS - 1) / CFL_ROWS, P.nphi);
        nparts = (int)(l.grid.x * l.grid.y);
        if (l.block.x >= 64)
            KLAUNCH(KID_CFL_CELLS, k_cfl_cells<true>, l.grid, l.block, P, P.cfl_part);
        else
            KLAUNCH(KID_CFL_CELLS, k_cfl_cells<false>, l.grid, l.block, P, P.cfl_part);
    }
    KLAUNCH(KID_CFL_INIT, k_cfl_final, dim3(1), dim3(1024), P, (const double *)P.cfl_part, nparts, apply_policy);
}

void launch_clock_export_cfl(DevClock *clk, double *out, hipStream_t st)
{
    KLAUNCH(KID_CLOCK, k_clock_export_cfl, dim3(1), dim3(1), (const DevClock *)clk, out);
}
void launch_clock_policy_ptr(DevClock *clk, double cfl_max_var, const double *cfl_global, hipStream_t st)
{
    KLAUNCH(KID_CLOCK, k_clock_policy_ptr, dim3(1), dim3(1), clk, cfl_max_var, cfl_global);
}
void launch_clock_scale_dt(DevClock *clk, int mode, double dt, double factor, hipStream_t st)
{
    KLAUNCH(KID_CLOCK, k_clock_scale_dt, dim3(1), dim3(1), clk, mode, dt, factor);
}
void launch_clock_set_dt(DevClock *clk, double dt, hipStream_t st)
{
    KLAUNCH(KID_CLOCK, k_clock_set_dt, dim3(1), dim3(1), clk, dt);
}
void launch_clock_advance(DevClock *clk, hipStream_t st)
{
    KLAUNCH(KID_CLOCK, k_clock_advance, dim3(1), dim3(1), clk);
}
void launch_clock_policy(DevClock *clk, double cfl_max_var, int use_device_cfl, double cfl_global,
                         hipStream_t st)
{
    KLAUNCH(KID_CLOCK, k_clock_policy, dim3(1), dim3(1), clk, cfl_max_var, use_device_cfl, cfl_global);
}

} // namespace fcpt
