// HIP kernels of the gas update for gfx950 (CDNA4).  FP64 throughout, phi is the
// contiguous (coalesced) axis of every grid, no MFMA (there is no contraction).
//
// Each kernel cites the reference loop nest it restates (paths relative to the
// reference's src/).  Operand order follows the reference so results agree with
// the CPU path to rounding.
#include <hip/hip_runtime.h>

#include "fcpt_kernels.h"

namespace fcpt {

#define IDX(i, j) ((size_t)(i) * (size_t)P.nphi + (size_t)(j))

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
#define CELL(row0, nrows)                                        \
    const int j = blockIdx.x * blockDim.x + threadIdx.x;         \
    const int i = (row0) + blockIdx.y * blockDim.y + threadIdx.y; \
    if (j >= P.nphi || i >= (row0) + (nrows))                    \
        return;
#define JNEXT (j == P.nphi - 1 ? 0 : j + 1)
#define JPREV (j == 0 ? P.nphi - 1 : j - 1)

__device__ __forceinline__ double dmin(double a, double b) { return b < a ? b : a; } // std::min
__device__ __forceinline__ double dmax(double a, double b) { return a < b ? b : a; } // std::max

// ---------------------------------------------------------------------------
// Pframeforce.cpp:21-94 CalculateNbodyPotential (+ Force.cpp:124-159 smoothing)
__global__ void k_potential(const Dev P)
{
    CELL(0, P.nr);
    const double x = P.Rmed[i] * P.cosphi[j];
    const double y = P.Rmed[i] * P.sinphi[j];
    const double smooth = P.thickness_smoothing * P.scale_height[IDX(i, j)];
    double pot = 0.0;
    for (int k = 0; k < P.nbodies; ++k) {
        const double dx = x - P.bx[k];
        const double dy = y - P.by[k];
        const double dist_2 = dx * dx + dy * dy;
        const double d_smoothed = sqrt(dist_2 + smooth * smooth);
        double klahr = 1.0;
        const double r_sm = P.brsm[k];
        if (r_sm > 0.0 && d_smoothed < r_sm) {
            const double q = d_smoothed / r_sm;
            klahr = ((q * q) * (q * q) - 2.0 * (q * q * q) + 2.0 * d_smoothed / r_sm);
        }
        pot += -P.G * P.bm[k] / d_smoothed * klahr;
    }
    pot += -P.indirect_x * x - P.indirect_y * y;
    P.potential[IDX(i, j)] = pot;
}

// SourceEuler.cpp:325-372 momentum_update_radial
__global__ void k_source_vr(const Dev P)
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
__global__ void k_source_va(const Dev P)
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
__global__ void k_compression_heating(const Dev P)
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
__global__ void k_tw_q(const Dev P)
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
__global__ void k_tw_va(const Dev P)
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
__global__ void k_tw_vr(const Dev P)
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
__global__ void k_sn_q(const Dev P)
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
__global__ void k_sn_e(const Dev P)
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
__global__ void k_sn_vr(const Dev P)
{
    CELL(P.one_no_ghost_vr, P.maxmo_no_ghost_vr - P.one_no_ghost_vr);
    const double dt = P.clk->dt;
    P.vrad[IDX(i, j)] = P.vrad[IDX(i, j)] - dt * 2.0 / (P.sigma[IDX(i, j)] + P.sigma[IDX(i - 1, j)]) *
                                                (P.qr[IDX(i, j)] - P.qr[IDX(i - 1, j)]) * P.InvDiffRmed[i];
}
// viscosity/artificial_viscosity.cpp:232-248 SN: v_phi
__global__ void k_sn_va(const Dev P)
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
__global__ void k_temperature_range(const Dev P)
{
    CELL(0, P.nr);
    P.energy[IDX(i, j)] = clamp_energy(P, P.energy[IDX(i, j)], P.sigma[IDX(i, j)]);
}

// SourceEuler.cpp:1054-1092 compute_sound_speed_normal + :1218-1251 compute_scale_height_old
// (adiabatic branch; the isothermal values are set once by k_iso_cs_h)
__global__ void k_adi_cs_h(const Dev P)
{
    CELL(0, P.nr);
    const double cs = sqrt(P.gamma * (P.gamma - 1.0) * P.energy[IDX(i, j)] / P.sigma[IDX(i, j)]);
    P.soundspeed[IDX(i, j)] = cs;
    const double r = P.Rmed[i];
    const double inv_omega_kepler = 1.0 / sqrt(P.G * P.Mc / (r * r * r));
    P.scale_height[IDX(i, j)] = cs / (sqrt(P.gamma)) * inv_omega_kepler;
}
__global__ void k_iso_cs_h(const Dev P, const double *cs_ring)
{
    CELL(0, P.nr);
    const double cs = cs_ring[i]; // h0 r^beta sqrt(GM/r), evaluated on the host (libm pow)
    P.soundspeed[IDX(i, j)] = cs;
    const double r = P.Rmed[i];
    const double inv_omega_kepler = 1.0 / sqrt(P.G * P.Mc / (r * r * r));
    P.scale_height[IDX(i, j)] = cs * inv_omega_kepler;
}
// viscosity/viscosity.cpp:98-137 update_viscosity
__global__ void k_viscosity(const Dev P)
{
    CELL(0, P.nr);
    if (P.alpha_viscosity)
        P.viscosity[IDX(i, j)] = P.alpha * P.scale_height[IDX(i, j)] * P.soundspeed[IDX(i, j)];
    else
        P.viscosity[IDX(i, j)] = P.nu_const;
}
// SourceEuler.cpp:1442-1473 compute_pressure
__global__ void k_pressure(const Dev P)
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
__global__ void k_temperature(const Dev P)
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
__global__ void k_stress_diag(const Dev P)
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
__global__ void k_stress_rphi(const Dev P)
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
__global__ void k_visc_va(const Dev P)
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
__global__ void k_visc_vr(const Dev P)
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
__global__ void k_qplus_qminus(const Dev P)
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
// SourceEuler.cpp:1000-1048: energy update of SubStep3 (update_energy != 0) or only the
// alpha rescaling of compute_heating_cooling_for_CFL (:1520-1545)
__global__ void k_substep3(const Dev P, int update_energy)
{
    CELL(1, P.nr - 2);
    const double dt = P.clk->dt;
    const double H = P.scale_height[IDX(i, j)];
    const double sigma = P.sigma[IDX(i, j)];
    const double energy = P.energy[IDX(i, j)];
    const double alpha = substep3_alpha(P, H, sigma, energy);
    const double Qplus = P.qplus[IDX(i, j)] / alpha;
    double Qminus = P.qminus[IDX(i, j)] / alpha;
    if (update_energy) {
        double energy_new = energy + dt * (Qplus - Qminus);
        const double SigmaFloor = 10.0 * P.sigma0_val * P.sigma_floor_rel;
        if (sigma < SigmaFloor) {
            // tau_eff is 0 without cooling => equilibrium energy 0 (raised to the floor below)
            energy_new = 0.0;
            Qminus = Qplus;
        }
        P.energy[IDX(i, j)] = energy_new;
    }
    P.qplus[IDX(i, j)] = Qplus;
    P.qminus[IDX(i, j)] = Qminus;
}

// ---------------------------------------------------------------------------
// boundary_conditions/{zero_gradient,reference,reflecting,outflow,keplerian_*,zero_shear}.cpp
// called in the order of boundary_conditions.cpp:65-114; one thread per phi column.
__device__ __forceinline__ void bc_scalar(const Dev &P, double *x, const double *x0, int type,
                                          int outer, int j)
{
    const int Irad = P.nr - 1;
    if (!outer) {
        if (!P.is_first)
            return;
        if (type == FCPT_BC_ZEROGRADIENT)
            x[IDX(0, j)] = x[IDX(1, j)];
        else if (type == FCPT_BC_REFERENCE)
            x[IDX(0, j)] = x0[IDX(0, j)];
    } else {
        if (!P.is_last)
            return;
        if (type == FCPT_BC_ZEROGRADIENT)
            x[IDX(Irad, j)] = x[IDX(Irad - 1, j)];
        else if (type == FCPT_BC_REFERENCE)
            x[IDX(Irad, j)] = x0[IDX(Irad, j)];
    }
}
__global__ void k_boundary(const Dev P)
{
    const int j = blockIdx.x * blockDim.x + threadIdx.x;
    if (j >= P.nphi)
        return;
    bc_scalar(P, P.sigma, P.sigma0, P.bc_sigma[0], 0, j);
    bc_scalar(P, P.sigma, P.sigma0, P.bc_sigma[1], 1, j);
    bc_scalar(P, P.energy, P.energy0, P.bc_energy[0], 0, j);
    bc_scalar(P, P.energy, P.energy0, P.bc_energy[1], 1, j);
    double *vr = P.vrad;
    const double *v0 = P.vrad0;
    const int Iv = P.nr; // max_radial of the vector grid
    for (int outer = 0; outer < 2; ++outer) {
        const int type = P.bc_vrad[outer];
        if (type == FCPT_BC_REFLECTING) { // no rank guard in the reference (reflecting.cpp:15-40)
            if (!outer) {
                vr[IDX(0, j)] = -vr[IDX(2, j)];
                vr[IDX(1, j)] = 0;
            } else {
                vr[IDX(Iv, j)] = -vr[IDX(Iv - 2, j)];
                vr[IDX(Iv - 1, j)] = 0;
            }
            continue;
        }
        if ((!outer && !P.is_first) || (outer && !P.is_last))
            continue;
        const int g0 = outer ? Iv : 0, g1 = outer ? Iv - 1 : 1, a = outer ? Iv - 2 : 2;
        switch (type) {
        case FCPT_BC_ZEROGRADIENT:
            vr[IDX(g0, j)] = vr[IDX(a, j)];
            vr[IDX(g1, j)] = vr[IDX(a, j)];
            break;
        case FCPT_BC_REFERENCE:
            vr[IDX(g0, j)] = v0[IDX(g0, j)];
            vr[IDX(g1, j)] = v0[IDX(g1, j)];
            break;
        case FCPT_BC_OUTFLOW: {
            const double va = vr[IDX(a, j)];
            const bool inflow = outer ? (va < 0.0) : (va > 0.0);
            vr[IDX(g1, j)] = inflow ? 0.0 : va;
            vr[IDX(g0, j)] = inflow ? 0.0 : va;
            break;
        }
        case FCPT_BC_KEPLERIAN:
            vr[IDX(g0, j)] = P.kep_vrad[outer] * sqrt(P.G * P.Mc / P.Rmed[g0]);
            vr[IDX(g1, j)] = P.kep_vrad[outer] * sqrt(P.G * P.Mc / P.Rmed[g1]);
            break;
        default:
            break;
        }
    }
    for (int outer = 0; outer < 2; ++outer) {
        const int type = P.bc_vaz[outer];
        if ((!outer && !P.is_first) || (outer && !P.is_last))
            continue;
        const int row = outer ? P.nr - 1 : 0, act = outer ? P.nr - 2 : 1;
        const double r = P.Rmed[row];
        switch (type) {
        case FCPT_BC_ZEROGRADIENT:
            P.vazi[IDX(row, j)] = P.vazi[IDX(act, j)];
            break;
        case FCPT_BC_REFERENCE:
            P.vazi[IDX(row, j)] = P.vazi0[IDX(row, j)];
            break;
        case FCPT_BC_KEPLERIAN:
            P.vazi[IDX(row, j)] = P.kep_vaz[outer] * sqrt(P.G * P.Mc / r) - r * P.omega_frame;
            break;
        case FCPT_BC_ZEROSHEAR:
            P.vazi[IDX(row, j)] = r * (P.vazi[IDX(act, j)] / P.Rmed[act]);
            break;
        default:
            break;
        }
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
    if (a * b > 0.0)
        return 2.0 * a * b / (a + b);
    return 0.0;
}

// Upwind "star" state at radial interface k (between rings k-1 and k),
// compute_star_radial (TransportEuler.cpp:349-406).  wm2..wp1 = Q at rings k-2..k+1.
__device__ __forceinline__ double star_radial(const Dev &P, int k, double v, double dt, double wm2,
                                              double wm1, double w0, double wp1)
{
    if (k <= 0 || k >= P.nr)
        return 0.0; // row 0 is zeroed on every call, row Nr is never written
    if (v > 0.0) {
        double dq = 0.0;
        if (k - 1 != 0 && k - 1 != P.nr - 1)
            dq = limiter(P.limiter, (w0 - wm1) * P.InvDiffRmed[k], (wm1 - wm2) * P.InvDiffRmed[k - 1]);
        return wm1 + (P.Rmed[k] - P.Rmed[k - 1] - v * dt) * 0.5 * dq;
    }
    double dq = 0.0;
    if (k != 0 && k != P.nr - 1)
        dq = limiter(P.limiter, (wp1 - w0) * P.InvDiffRmed[k + 1], (w0 - wm1) * P.InvDiffRmed[k]);
    return w0 - (P.Rmed[k + 1] - P.Rmed[k] + v * dt) * 0.5 * dq;
}

// compute_momenta_from_velocities (:471-493) + OneWindRad (:138-167) with all
// VanLeerRadial calls (:545-620) in one pass.  Reads Sigma, v_r, v_phi(, e) and
// writes the transported momenta / density / energy to set B, so the in-place
// ordering constraint of the reference ("Sigma MUST be last") is met by
// construction: every quantity sees the pre-transport density.
__global__ void k_transport_radial(const Dev P)
{
    CELL(0, P.nr);
    const double dt = P.clk->dt;
    const int jn = JNEXT;
    const int nr = P.nr;
    // rings i-2 .. i+2 of this column
    double S[5], VR[6], W_rmp[5], W_rmm[5], W_lp[5], W_lm[5], W_e[5];
#pragma unroll
    for (int a = 0; a < 6; ++a) {
        const int k = i - 2 + a;
        VR[a] = (k >= 0 && k <= nr) ? P.vrad[IDX(k, j)] : 0.0;
    }
#pragma unroll
    for (int a = 0; a < 5; ++a) {
        const int k = i - 2 + a;
        if (k >= 0 && k < nr) {
            const double s = P.sigma[IDX(k, j)];
            const double r = P.Rmed[k];
            const double va = P.vazi[IDX(k, j)], van = P.vazi[IDX(k, jn)];
            S[a] = s;
            // Work = Q / DENSITY_INT with Q the momentum of T1
            W_rmp[a] = (s * VR[a + 1]) / s;
            W_rmm[a] = (s * VR[a]) / s;
            W_lp[a] = (s * (van + r * P.omega_frame) * r) / s;
            W_lm[a] = (s * (va + r * P.omega_frame) * r) / s;
            W_e[a] = P.adiabatic ? P.energy[IDX(k, j)] / s : 0.0;
        } else {
            S[a] = W_rmp[a] = W_rmm[a] = W_lp[a] = W_lm[a] = W_e[a] = 0.0;
        }
    }
    // interfaces i (inf) and i+1 (sup)
    const double v_inf = VR[2], v_sup = VR[3];
    const double rho_inf = star_radial(P, i, v_inf, dt, S[0], S[1], S[2], S[3]);
    const double rho_sup = star_radial(P, i + 1, v_sup, dt, S[1], S[2], S[3], S[4]);
    const double f_inf = dt * P.dphi * P.Rinf[i];
    const double f_sup = dt * P.dphi * P.Rsup[i];
    const double invsurf = P.InvSurf[i];
    const double s0 = S[2];
#define RADIAL_UPDATE(W, Q0, OUT)                                                      \
    {                                                                                  \
        const double q_inf = star_radial(P, i, v_inf, dt, W[0], W[1], W[2], W[3]);     \
        const double q_sup = star_radial(P, i + 1, v_sup, dt, W[1], W[2], W[3], W[4]); \
        const double varq_inf = f_inf * q_inf * rho_inf * v_inf;                       \
        const double varq_sup = f_sup * q_sup * rho_sup * v_sup;                       \
        OUT[IDX(i, j)] = (Q0) + (varq_inf - varq_sup) * invsurf;                       \
    }
    RADIAL_UPDATE(W_rmp, s0 * VR[3], P.rmpB);
    RADIAL_UPDATE(W_rmm, s0 * VR[2], P.rmmB);
    {
        const double r = P.Rmed[i];
        const double va = P.vazi[IDX(i, j)], van = P.vazi[IDX(i, jn)];
        RADIAL_UPDATE(W_lp, s0 * (van + r * P.omega_frame) * r, P.lpB);
        RADIAL_UPDATE(W_lm, s0 * (va + r * P.omega_frame) * r, P.lmB);
    }
    if (P.adiabatic)
        RADIAL_UPDATE(W_e, P.energy[IDX(i, j)], P.eB);
    {
        // density: Work = Sigma / DENSITY_INT = 1 exactly => star state 1 (0 on the closed rows)
        const double q_inf = (i <= 0) ? 0.0 : 1.0;
        const double q_sup = (i + 1 >= nr) ? 0.0 : 1.0;
        const double varq_inf = f_inf * q_inf * rho_inf * v_inf;
        const double varq_sup = f_sup * q_sup * rho_sup * v_sup;
        P.sigB[IDX(i, j)] = s0 + (varq_inf - varq_sup) * invsurf;
    }
#undef RADIAL_UPDATE
}

// compute_average_azimuthal_velocity (:174-189) + ComputeConstantResidual (:207-236):
// one block per ring; wavefront shuffles + LDS for the ring sum.
__global__ void k_ring_mean(const Dev P, int with_shift)
{
    const int i = blockIdx.x;
    double acc = 0.0;
    for (int j = threadIdx.x; j < P.nphi; j += blockDim.x)
        acc += P.vazi[IDX(i, j)];
    for (int off = 32; off > 0; off >>= 1)
        acc += __shfl_down(acc, off, 64);
    __shared__ double s_w[4];
    const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
    if (lane == 0)
        s_w[wave] = acc;
    __syncthreads();
    if (threadIdx.x == 0) {
        double sum = s_w[0];
        for (int w = 1; w < (int)(blockDim.x >> 6); ++w)
            sum += s_w[w];
        const double mean = sum / (double)P.nphi;
        P.vmean[i] = mean;
        if (with_shift) {
            const double dt = P.clk->dt;
            const double invdt = 1.0 / dt;
            const double Ntilde = mean * P.InvRmed[i] * dt * P.invdphi;
            const double Nround = floor(Ntilde + 0.5);
            P.nshift[i] = (int)Nround;
            P.vconst[i] = (Ntilde - Nround) * P.Rmed[i] * invdt * P.dphi;
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
    if (ksi > 0.0) {
        const double dq = 0.5 * limiter(P.limiter, (w0 - wm1), (wm1 - wm2)) * invdxtheta;
        return wm1 + (dxtheta - ksi) * dq;
    }
    const double dq = 0.5 * limiter(P.limiter, (wp1 - w0), (w0 - wm1)) * invdxtheta;
    return w0 - (dxtheta + ksi) * dq;
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
template <int PASS> __global__ void k_transport_theta(const Dev P, ThetaSet in, ThetaOut out)
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
    double S[5];
#pragma unroll
    for (int a = 0; a < 5; ++a)
        S[a] = in.sig[IDX(i, jj[a])];
    const double rho0 = star_theta(P, v0, dt, dxtheta, invdxtheta, S[0], S[1], S[2], S[3]);
    const double rho1 = star_theta(P, v1, dt, dxtheta, invdxtheta, S[1], S[2], S[3], S[4]);
#define THETA_UPDATE(IN, OUT)                                                                    \
    {                                                                                            \
        double W[5];                                                                             \
        _Pragma("unroll") for (int a = 0; a < 5; ++a) W[a] = IN[IDX(i, jj[a])] / S[a];           \
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

// compute_velocities_from_momenta (:498-535) + assure_minimum_value and the
// temperature floor/ceiling of Transport (:121-131); reads set B, writes the state.
__global__ void k_velocities(const Dev P)
{
    CELL(0, P.nr);
    const int jp = JPREV;
    const double s = P.sigB[IDX(i, j)];
    if (i == 0)
        P.vrad[IDX(i, j)] = 0.0;
    else
        P.vrad[IDX(i, j)] = (P.rmpB[IDX(i - 1, j)] + P.rmmB[IDX(i, j)]) / (P.sigB[IDX(i - 1, j)] + s);
    P.vazi[IDX(i, j)] =
        (P.lpB[IDX(i, jp)] + P.lmB[IDX(i, j)]) / (P.sigB[IDX(i, jp)] + s) * P.InvRmed[i] -
        P.Rmed[i] * P.omega_frame;
    const double sf = s < P.sigma_floor_abs ? P.sigma_floor_abs : s;
    P.sigma[IDX(i, j)] = sf;
    if (P.adiabatic)
        P.energy[IDX(i, j)] = clamp_energy(P, P.eB[IDX(i, j)], sf);
}

// ---------------------------------------------------------------------------
// cfl.cpp:185-376 condition_cfl.  k_ring_mean gives <v_phi>; k_cfl_init seeds the
// running minimum with the shear criterion of rings 0|1 (:207-208); k_cfl_cells adds
// the per-ring shear limit (:213-220) and the six per-cell limits (:243-328).
__global__ void k_cfl_init(const Dev P)
{
    const double denom = fabs(P.vmean[0] * P.InvRmed[0] - P.vmean[1] * P.InvRmed[1]) + 1.0e-100;
    const double dt_core = P.cfl * P.dphi / denom;
    P.clk->cfl_bits = (unsigned long long)__double_as_longlong(dt_core);
}
__global__ void k_cfl_cells(const Dev P)
{
    const int j = blockIdx.x * blockDim.x + threadIdx.x;
    const int i = P.first_active + blockIdx.y * blockDim.y + threadIdx.y;
    double dt_cell = 1.0e300;
    if (j < P.nphi && i < P.active_size) {
        const int jn = JNEXT;
        const double dxRadial = P.Rsup[i] - P.Rinf[i];
        const double dxAzimuthal = P.Rmed[i] * P.dphi;
        const double cell_size = dmin(dxRadial, dxAzimuthal);
        const double lf = P.leapfrog ? 0.6 : 1.0;
        const double va = P.vazi[IDX(i, j)];
        const double vres = P.fast_transport ? va - P.vmean[i] : va;
        const double vr0 = P.vrad[IDX(i, j)], vr1 = P.vrad[IDX(i + 1, j)];
        const double invdt1 = P.soundspeed[IDX(i, j)] / cell_size;
        const double invdt2 = vr0 / dxRadial;
        const double invdt3 = vres / dxAzimuthal;
        const double C2 = P.art_visc_factor * P.art_visc_factor;
        double invdt4;
        if (P.art_visc == FCPT_ARTVISC_SN) {
            double dvRadial = vr1 - vr0;
            double dvAzimuthal = P.vazi[IDX(i, jn)] - va;
            dvRadial = dvRadial > 0.0 ? 0.0 : -dvRadial;
            dvAzimuthal = dvAzimuthal > 0.0 ? 0.0 : -dvAzimuthal;
            invdt4 = 4.0 * C2 * dmax(dvRadial / dxRadial, dvAzimuthal / dxAzimuthal) * lf;
        } else { // the TW formula is also used for ArtificialViscosity: None (cfl.cpp:292)
            const double eps_rr = (vr1 - vr0) * P.InvDiffRsup[i];
            const double eps_pp =
                P.InvRmed[i] * ((P.vazi[IDX(i, jn)] - va) * P.invdphi + 0.5 * (vr1 + vr0));
            const double mdiv_V = -dmin(eps_rr + eps_pp, 0.0);
            invdt4 = 4.0 * C2 * mdiv_V * lf;
        }
        const double invdt5 = 4.0 * P.viscosity[IDX(i, j)] / (cell_size * cell_size) * lf;
        double invdt6 = 0.0;
        if (P.adiabatic) {
            const double inv_limit = 1.0 / P.heating_cooling_cfl_limit;
            invdt6 = inv_limit * fabs((P.qplus[IDX(i, j)] - P.qminus[IDX(i, j)]) / P.energy[IDX(i, j)]) * lf;
        }
        dt_cell = P.cfl / sqrt(invdt1 * invdt1 + invdt2 * invdt2 + invdt3 * invdt3 + invdt4 * invdt4 +
                               invdt5 * invdt5 + invdt6 * invdt6);
        if (j == 0) {
            const double denom = fabs(P.vmean[i] * P.InvRmed[i] - P.vmean[i + 1] * P.InvRmed[i + 1]) + 1.0e-100;
            dt_cell = dmin(dt_cell, P.cfl * P.dphi / denom);
        }
    }
    // min over the block: wavefront shuffles, then LDS across the 4 waves
    for (int off = 32; off > 0; off >>= 1)
        dt_cell = dmin(dt_cell, __shfl_down(dt_cell, off, 64));
    __shared__ double s_w[4];
    const int tid = threadIdx.y * blockDim.x + threadIdx.x;
    if ((tid & 63) == 0)
        s_w[tid >> 6] = dt_cell;
    __syncthreads();
    if (tid == 0) {
        const double m = dmin(dmin(s_w[0], s_w[1]), dmin(s_w[2], s_w[3]));
        // positive doubles order like their bit patterns
        atomicMin(&P.clk->cfl_bits, (unsigned long long)__double_as_longlong(m));
    }
}

// ---------------------------------------------------------------------------
// clock kernels (single thread)
__global__ void k_clock_set_dt(DevClock *clk, double dt) { clk->dt = dt; }
__global__ void k_clock_advance(DevClock *clk)
{
    clk->time += clk->dt;
    clk->n_hydro_iter += 1;
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
    "k_ring_mean", "k_transport_theta1", "k_transport_theta2", "k_velocities", "k_cfl_init",
    "k_cfl_cells", "k_clock"};

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
            KLAUNCH(id, kernel, l.grid, l.block, __VA_ARGS__);                       \
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

void launch_transport(const Dev &P, hipStream_t st)
{
    // Transport, TransportEuler.cpp:112-136
    LAUNCH2D(KID_TRANSPORT_RADIAL, k_transport_radial, P.nr, P);
    KLAUNCH(KID_RING_MEAN, k_ring_mean, dim3(P.nr), dim3(256), P, 1);
    ThetaSet inB = {P.rmpB, P.rmmB, P.lpB, P.lmB, P.sigB, P.eB};
    ThetaOut outA = {P.rmpA, P.rmmA, P.lpA, P.lmA, P.sigA, P.eA};
    ThetaSet inA = {P.rmpA, P.rmmA, P.lpA, P.lmA, P.sigA, P.eA};
    ThetaOut outB = {P.rmpB, P.rmmB, P.lpB, P.lmB, P.sigB, P.eB};
    LAUNCH2D(KID_THETA1, k_transport_theta<1>, P.nr, P, inB, outA);
    LAUNCH2D(KID_THETA2, k_transport_theta<2>, P.nr, P, inA, outB);
    LAUNCH2D(KID_VELOCITIES, k_velocities, P.nr, P);
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

void launch_cfl(const Dev &P, hipStream_t st)
{
    KLAUNCH(KID_RING_MEAN, k_ring_mean, dim3(P.nr), dim3(256), P, 0);
    KLAUNCH(KID_CFL_INIT, k_cfl_init, dim3(1), dim3(1), P);
    const int nrows = P.active_size - P.first_active;
    if (nrows > 0) {
        const Launch2D l = launch2d(nrows, P.nphi);
        KLAUNCH(KID_CFL_CELLS, k_cfl_cells, l.grid, l.block, P);
    }
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
