// Part of fcpt_kernels.hip (one translation unit, namespace fcpt): the three out-of-place source kernels (narrow rings).
// Not a stand-alone header: included once, in the order given there.

// ===========================================================================
// Fused source step (default path).  The reference's source / artificial-viscosity /
// viscous-stress substeps are 13 loop nests that stream ~45 grids; here they are three
// out-of-place kernels that stream 17: intermediate tensors (Q_rr, Q_pp, div v, tau_*) are
// re-evaluated from the velocities in registers instead of being stored.
//   k_src_fused : (v_r, v_phi)   -> (v_r_b, v_phi_b)   S1 + S2
//   k_av_fused  : (v_r_b,v_phi_b)-> (v_r, v_phi) [,e]  S3 + artificial viscosity (+ T range)
//   k_visc_fused: (v_r, v_phi)   -> (v_r_b, v_phi_b)   stress tensor + viscous update [+ Q+, SubStep3]
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
        // SubStep3 on this cell (cell-local: Q+, Sigma, e, H of the cell; k_visc_fused itself never reads e)
        substep3_cell(P, i, j, qplus, 0.0, 2);
    }
}
