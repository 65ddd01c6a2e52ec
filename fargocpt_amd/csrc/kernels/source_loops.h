// Part of fcpt_kernels.hip (one translation unit, namespace fcpt): potential; one kernel per reference loop nest of the source step; EOS; SubStep3 and its cooling terms.
// Not a stand-alone header: included once, in the order given there.

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

// Pframeforce.cpp:96-189 CalculateAccelOnGas (BodyForceFromPotential: no): rows 1 .. Nr-1 of ACCEL_RADIAL / ACCEL_AZIMUTHAL,
// the operations in the reference's order (the one source kernel without fast reciprocals: sqrt and an IEEE division)
template <bool ROWU> __global__ void k_accel_on_gas(const Dev P)
{
    CELL(1, P.nr - 1);
    const double r = P.Rmed[i];
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
    double ax = P.indirect_x, ay = P.indirect_y;
    for (int k = 0; k < P.nbodies; ++k) {
        const double dx = x - P.bx[k];
        const double dy = y - P.by[k];
        const double dist_2 = dx * dx + dy * dy;
        const double dist_2_sm = dist_2 + smooth * smooth;
        const double dist_sm = sqrt(dist_2_sm);
        const double dist_3_sm = dist_sm * dist_2_sm;
        const double inv_dist_3_sm = 1.0 / dist_3_sm;
        double klahr = 1.0;
        const double r_sm = P.brsm[k];
        if (r_sm > 0.0 && dist_sm < r_sm) {
            const double q = dist_sm / r_sm;
            klahr = -(3.0 * ((q * q) * (q * q)) - 4.0 * (q * q * q));
        }
        ax -= dx * P.G * P.bm[k] * inv_dist_3_sm * klahr;
        ay -= dy * P.G * P.bm[k] * inv_dist_3_sm * klahr;
    }
    P.accel_r[IDX(i, j)] = (x * ax + y * ay) / r;
    P.accel_az[IDX(i, j)] = (x * ay - y * ax) / r;
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
    const double gradphi = P.accel_force ? -(P.accel_r[IDX(i, j)] + P.accel_r[IDX(i - 1, j)]) * 0.5 // :348-353
                                         : (P.potential[IDX(i, j)] - P.potential[IDX(i - 1, j)]) * P.InvDiffRmed[i];
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
    const double gradphi = P.accel_force ? -(P.accel_az[IDX(i, j)] + P.accel_az[IDX(i, jp)]) * 0.5 // :406-411
                                         : (P.potential[IDX(i, j)] - P.potential[IDX(i, jp)]) * invdxtheta;
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

// recalculate_derived_disk_quantities (SourceEuler.cpp:225-249) / recalculate_viscosity (:205-223) of the ideal EOS in
// ONE launch: temperature, sound speed, scale height, pressure, viscosity are pointwise functions of Sigma and e, with
// the expressions of k_temperature, k_adi_cs_h, k_pressure, k_viscosity.  what: bit 0 temperature, bit 1 pressure
// (c_s, H and nu always).  Narrow grids run these between the per-loop kernels: four launches of 5 us each were a
// fifth of their step.
template <bool ROWU> __global__ void k_adi_derived(const Dev P, int what)
{
    CELL(0, P.nr);
    const double e = P.energy[IDX(i, j)], sg = P.sigma[IDX(i, j)];
    if (what & 1) {
        const double c_v_inv = P.mu / P.Rgas * (P.gamma - 1.0);
        P.temperature[IDX(i, j)] = c_v_inv * e / sg;
    }
    const double cs = sqrt(P.gamma * (P.gamma - 1.0) * e / sg);
    P.soundspeed[IDX(i, j)] = cs;
    const double r = P.Rmed[i];
    const double inv_omega_kepler = 1.0 / sqrt(P.G * P.Mc / (r * r * r));
    const double H = cs / (sqrt(P.gamma)) * inv_omega_kepler;
    P.scale_height[IDX(i, j)] = H;
    if (what & 2)
        P.pressure[IDX(i, j)] = (P.gamma - 1.0) * e;
    if (P.alpha_viscosity)
        P.viscosity[IDX(i, j)] = P.alpha * H * cs;
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
// VISCOSITY_SIGMA_RP (viscosity.cpp:228-252): nu Sigma averaged over the four cells around the corner; rows 0 and Nr hold 0
__device__ __forceinline__ double nusig_rp(const Dev &P, int i, int j, int jp)
{
    if (i < 1 || i >= P.nr)
        return 0.0;
    const double nu = 0.25 * (P.viscosity[IDX(i, j)] + P.viscosity[IDX(i - 1, j)] + P.viscosity[IDX(i, jp)] +
                              P.viscosity[IDX(i - 1, jp)]);
    const double sigma =
        0.25 * (P.sigma[IDX(i, j)] + P.sigma[IDX(i - 1, j)] + P.sigma[IDX(i, jp)] + P.sigma[IDX(i - 1, jp)]);
    return nu * sigma;
}
// viscosity/viscosity.cpp:256-348: correction factors of the pseudo-implicit viscosity (StabilizeViscosity 1|2),
// rows 1..Nr-1.  Not on the marching path: three small kernels extra per kick.
template <bool ROWU> __global__ void k_visc_factors(const Dev P)
{
    CELL(1, P.nr - 1);
    const int jp = JPREV, jn = JNEXT;
    const double NuSig_rp = nusig_rp(P, i, j, jp);
    const double NuSig_rp_ip = nusig_rp(P, i + 1, j, jp);
    const double NuSig_rp_jp = nusig_rp(P, i, jn, j);
    const double sg = P.sigma[IDX(i, j)], sg_jm = P.sigma[IDX(i, jp)], sg_im = P.sigma[IDX(i - 1, j)];
    const double NuSigma = P.viscosity[IDX(i, j)] * sg;
    const double NuSigma_jm = P.viscosity[IDX(i, jp)] * sg_jm;
    const double NuSigma_im = P.viscosity[IDX(i - 1, j)] * sg_im;
    const double Ra = P.Rinf[i], rs = P.Rsup[i];
    const double TwoDiffRaSq = 2.0 / (rs * rs - Ra * Ra);
    const double FourThirdInvRbInvdphiSq = 4.0 / 3.0 / P.Rmed[i] * P.invdphi * P.invdphi;
    const double a0 = NuSig_rp * P.g_ra3[i] * P.InvDiffRmed[i];
    const double a1 = NuSig_rp_ip * P.g_ra3[i + 1] * P.InvDiffRmed[i + 1];
    const double cphi_rp = -P.InvRmed[i] * TwoDiffRaSq * (a1 + a0);
    const double cphi_pp = -FourThirdInvRbInvdphiSq * (NuSigma + NuSigma_jm);
    const double sigma_avg_phi = 0.5 * (sg + sg_jm);
    P.cfac_phi[IDX(i, j)] = (cphi_rp + cphi_pp) / (sigma_avg_phi * P.Rmed[i]);
    const double sigma_avg_r = 0.5 * (sg + sg_im);
    const double cr_rp = -(NuSig_rp_jp + NuSig_rp) / (P.dphi * P.dphi * Ra);
    const double cr_pp_1 = 2.0 * NuSigma * (0.5 * P.InvRmed[i] + 1.0 / 3.0 * Ra * P.InvDiffRsupRb[i]);
    const double cr_pp_2 = 2.0 * NuSigma_im * (0.5 * P.InvRmed[i - 1] - 1.0 / 3.0 * Ra * P.InvDiffRsupRb[i - 1]);
    const double cr_rr_1 = P.Rmed[i] * 2.0 * NuSigma * (-P.InvDiffRsup[i] + 1.0 / 3.0 * Ra * P.InvDiffRsupRb[i]);
    const double cr_rr_2 =
        -1.0 * P.Rmed[i - 1] * 2.0 * NuSigma_im * (P.InvDiffRsup[i - 1] - 1.0 / 3.0 * Ra * P.InvDiffRsupRb[i - 1]);
    const double cr_pp = -0.5 * (cr_pp_1 + cr_pp_2);
    const double cr_rr = P.InvDiffRmed[i] * (cr_rr_1 + cr_rr_2);
    const double Rmed_mid = 0.5 * (P.Rmed[i] + P.Rmed[i - 1]);
    P.cfac_r[IDX(i, j)] = P.radial_viscosity_factor * (cr_rr + cr_rp + cr_pp) / (sigma_avg_r * Rmed_mid);
}
// viscosity/viscosity.cpp:386-391|413-417: corr = 1 / (max(1 + dt c, 0) - dt c)
__device__ __forceinline__ double visc_corr(double dt, double c) { return 1.0 / (dmax(1.0 + dt * c, 0.0) - dt * c); }
// viscosity/viscosity.cpp:368-394: v_phi update
template <bool ROWU> __global__ void k_visc_va(const Dev P)
{
    CELL(1, P.nr - 2);
    const double dt = P.clk->dt;
    const int jp = JPREV;
    const double sigma_avg = 0.5 * (P.sigma[IDX(i, j)] + P.sigma[IDX(i, jp)]);
    const double ra1 = P.Rinf[i + 1], ra0 = P.Rinf[i];
    double dVp = dt * P.InvRmed[i] / (sigma_avg) *
                 ((2.0 / (ra1 * ra1 - ra0 * ra0)) *
                            (ra1 * ra1 * P.trp[IDX(i + 1, j)] - ra0 * ra0 * P.trp[IDX(i, j)]) +
                        (P.tpp[IDX(i, j)] - P.tpp[IDX(i, jp)]) * P.invdphi);
    if (P.stabilize == 1)
        dVp *= visc_corr(dt, P.cfac_phi[IDX(i, j)]);
    P.vazi[IDX(i, j)] += dVp;
}
// viscosity/viscosity.cpp:396-421: v_r update
template <bool ROWU> __global__ void k_visc_vr(const Dev P)
{
    CELL(P.one_no_ghost_vr, P.maxmo_no_ghost_vr - P.one_no_ghost_vr);
    const double dt = P.clk->dt;
    const int jn = JNEXT;
    const double sigma_avg = 0.5 * (P.sigma[IDX(i, j)] + P.sigma[IDX(i - 1, j)]);
    double dVr = dt / (sigma_avg)*P.radial_viscosity_factor * 2.0 / (P.Rmed[i] + P.Rmed[i - 1]) *
                       ((P.Rmed[i] * P.trr[IDX(i, j)] - P.Rmed[i - 1] * P.trr[IDX(i - 1, j)]) * P.InvDiffRmed[i] +
                        (P.trp[IDX(i, jn)] - P.trp[IDX(i, j)]) * P.invdphi -
                        0.5 * (P.tpp[IDX(i, j)] + P.tpp[IDX(i - 1, j)]));
    if (P.stabilize == 1)
        dVr *= visc_corr(dt, P.cfac_r[IDX(i, j)]);
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
// opacity.cpp:170-297 bell(): Bell & Lin (1994), eight regions smoothed across their borders; cgs in / cgs out
__device__ double opacity_bell(double density, double temperature)
{
    const double power1 = 2.8369e-2, power2 = 1.1464e-2, power3 = 2.2667e-1;
    const double t234 = 1.46e3, t456 = 4.51e3, t678 = 2.37e6;
    const double ak1 = 2.e-4, ak2 = 2.e16, ak3 = 0.1e0;
    const double bk3 = 10., bk4 = 2.e-15, bk5 = 1e4, bk6 = 1e4, bk7 = 1.5e10, bk8 = 0.348;
    if (temperature < 1.0)
        temperature = 10.0;
    if (temperature > t234 * pow(density, power1)) {
        const double ts4 = 1.e-4 * temperature; // to avoid overflow
        const double density13 = pow(density, 1.0 / 3.0);
        const double density23 = density13 * density13;
        const double ts42 = ts4 * ts4;
        const double ts44 = ts42 * ts42;
        const double ts48 = ts44 * ts44;
        if (temperature > t456 * pow(density, power2)) {
            if ((temperature < t678 * pow(density, power3)) || ((density <= 1e10) && (temperature < 1e4))) {
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
            const double o3 = bk3 * sqrt(ts4);
            const double o4 = bk4 * density / (ts48 * ts48 * ts48);
            const double o5 = bk5 * density23 * ts42 * ts4;
            const double o4an = pow(o4, 4.0), o3an = pow(o3, 4.0);
            return pow((o4an * o3an / (o4an + o3an)) + pow(o5 / (1.0 + 6.561e-5 / ts48 * 1e2 * density23), 4.0), 0.25);
        }
    } else {
        const double t2 = temperature * temperature;
        const double t4 = t2 * t2;
        const double t8 = t4 * t4;
        const double t10 = t8 * t2;
        const double o1 = ak1 * t2;
        const double o2 = ak2 * temperature / t8;
        const double o3 = ak3 * sqrt(temperature);
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
    else if (P.opacity == FCPT_OPACITY_BELL)
        kappa = opacity_bell(rho * P.density_cgs, temperatureCGS) * (1.0 / P.opacity_cgs);
    else if (P.opacity == FCPT_OPACITY_CONST)
        kappa = P.kappa_const;
    else
        kappa = P.kappa_const * (temperatureCGS * temperatureCGS);
    kappa = P.kappa_factor * kappa;
    const double tau = P.tau_factor * (1.0 / P.density_factor) * kappa * sigma;
    if (P.opacity == FCPT_OPACITY_SIMPLE)
        return 3.0 / 8.0 * tau; // D'Angelo et al. 2003 eq. (28)
    if (P.heating_star) // irradiated disk, D'Angelo & Marzari 2012
        return 3.0 / 8.0 * tau + 0.5 + 1.0 / (4.0 * tau + P.tau_min);
    return 3.0 / 8.0 * tau + sqrt(3.0) / 4.0 + 1.0 / (4.0 * tau + P.tau_min);
}
// calculate_qminus (SourceEuler.cpp:931-950) at one cell of rows [1, Nr-1): beta cooling
// (thermal_relaxation :632-786, without the opacity-based Ziampras variants) and thermal surface
// cooling (:790-820).  tau_eff is returned for SubStep3's low-density branch (0 without surface cooling).
struct Cooling {
    double qminus, tau_eff, qplus_star; // qplus_star: irradiation by the bodies (SourceEuler.cpp:538-612)
};
__device__ __forceinline__ Cooling cooling_terms(const Dev &P, int i, int j, int cell, double sigma, double energy, double H)
{
    Cooling c = {0.0, 0.0, 0.0};
    const double tkick = P.clk->time - (P.kick_time_shift ? P.clk->dt : 0.0);
    if (P.cooling_beta && !(P.cooling_at_init && P.cooling_beta_reference == FCPT_BETAREF_REFERENCE)) {
        double beta_inv = 1 / P.cooling_beta_value;
        if (P.cooling_beta_ramp_up > 0.0) {
            const double x = 2 * tkick / P.cooling_beta_ramp_up;
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
    if (P.cooling_surface || P.heating_star) {
        const double T = P.mu / P.Rgas * (P.gamma - 1.0) * energy / sigma; // compute_temperature
        c.tau_eff = tau_eff_of(P, sigma, H, T);
        if (P.cooling_surface) {
            const double T2 = T * T, Tm2 = P.tmin * P.tmin;
            c.qminus += P.cooling_radiative_factor * 2 * P.sigma_sb * (T2 * T2 - Tm2 * Tm2) / c.tau_eff;
        }
    }
    if (P.heating_star) {
        const double xc = P.Rmed[i] * P.cosphi[j], yc = P.Rmed[i] * P.sinphi[j];
        const double HoverR = H * P.InvRmed[i]; // ASPECTRATIO
        for (int k = 0; k < P.nbodies; ++k) {
            if (!(P.btemp[k] > 0.0))
                continue;
            double ramping = 1.0;
            if (tkick < P.bramp[k]) {
                const double cs = cos(tkick * M_PI / 2.0 / P.bramp[k]);
                ramping = 1.0 - cs * cs;
            }
            const double x = P.bx[k], y = P.by[k];
            const double R_star = P.bradius[k], T_star = P.btemp[k];
            const double min_dist = (x * x + y * y > 1e-10) ? dmax(R_star, P.brsm[k]) : R_star;
            const double distance = dmax(sqrt((x - xc) * (x - xc) + (y - yc) * (y - yc)), min_dist);
            const double roverd = distance < R_star ? 1.0 : R_star / distance;
            const double W_G = 0.4 * roverd + HoverR * (9.0 / 7.0 - 1.0); // Chiang & Goldreich (1997)
            const double Ts2 = T_star * T_star;
            const double T_irrad_pow4 = (1.0 - 0.5) * (Ts2 * Ts2) * (roverd * roverd) * W_G;
            c.qplus_star += ramping * (2.0 * P.sigma_sb * T_irrad_pow4 / c.tau_eff);
        }
    }
    return c;
}
// SourceEuler.cpp:1000-1048: energy update of SubStep3 (update_energy != 0) or only the
// alpha rescaling of compute_heating_cooling_for_CFL (:1520-1545)
// update_energy = 2: followed by SetTemperatureFloorCeilValues on all rings (k_temperature_range) in the same launch
// one cell of it; qplus_in / qminus_in: the heating and cooling rates found so far
__device__ __forceinline__ void substep3_cell(const Dev &P, int i, int j, double qplus_in, double qminus_in, int update_energy)
{
    if (i < 1 || i >= P.nr - 1) { // SubStep3 itself runs on rings [1, Nr-1)
        if (update_energy == 2)
            P.energy[IDX(i, j)] = clamp_energy(P, P.energy[IDX(i, j)], P.sigma[IDX(i, j)]);
        return;
    }
    const double dt = P.clk->dt;
    const double H = P.scale_height[IDX(i, j)];
    const double sigma = P.sigma[IDX(i, j)];
    const double energy = P.energy[IDX(i, j)];
    const double alpha = substep3_alpha(P, H, sigma, energy);
    const Cooling cool = cooling_terms(P, i, j, IDX(i, j), sigma, energy, H);
    const double Qplus = (qplus_in + cool.qplus_star) / alpha;
    double Qminus = (qminus_in + cool.qminus) / alpha;
    if (update_energy) {
        double energy_new = energy + dt * (Qplus - Qminus);
        const double SigmaFloor = 10.0 * P.sigma0_val * P.sigma_floor_rel;
        if (sigma < SigmaFloor) {
            // the energy at which the current heating and cooling balance (0 without surface cooling: tau_eff = 0)
            const double e4 = Qplus * cool.tau_eff / (2.0 * P.sigma_sb);
            energy_new = sqrt(sqrt(e4)) * (P.Rgas / P.mu * sigma / (P.gamma - 1.0));
            Qminus = Qplus;
        }
        if (update_energy == 2)
            energy_new = clamp_energy(P, energy_new, sigma);
        P.energy[IDX(i, j)] = energy_new;
    }
    P.qplus[IDX(i, j)] = Qplus;
    P.qminus[IDX(i, j)] = Qminus;
}
template <bool ROWU> __global__ void k_substep3(const Dev P, int update_energy)
{
    CELL(0, P.nr);
    const bool inner = i >= 1 && i < P.nr - 1;
    substep3_cell(P, i, j, inner ? P.qplus[IDX(i, j)] : 0.0, inner ? P.qminus[IDX(i, j)] : 0.0, update_energy);
}
