// Part of fcpt_kernels.hip (one translation unit, namespace fcpt): the wave-marching source kernels (isothermal and ideal EOS).
// Not a stand-alone header: included once, in the order given there.

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
// Reciprocals of the source step: every one of them scales a source term that enters the state times dt (pressure
// gradient / (Sigma + Sigma), viscous and artificial-viscous forces / Sigma, heating rates / alpha ...), so its error
// reaches the state damped by dt a / v ~ 1e-2 or less: one Newton step on v_rcp_f64 (<= 2e-15 relative, measured:
// profiles/tools/rcp_accuracy.hip) instead of two.  (The transport's reciprocals act on full-magnitude quantities and
// keep both steps.)
#ifndef FAST_RCP_SRC
#define FAST_RCP_SRC fast_rcp1
#endif
// (The same shortcut for the roots -- one Newton step on v_rsq_f64 for the potential's smoothed distance, c_s as
//  x rsqrt(x) -- was measured too: -0.5 % per step, and the 4x-CFL full-size case against the oracle at 1.005e-10 instead
//  of inside the 1e-10 bar.  Not taken.)
#ifndef FAST_RSQRT_SRC
#define FAST_RSQRT_SRC fast_rsqrt
#endif
// byte offset of cell (ring r_, this lane's column) in a grid: 32-bit (see ld_off)
#define MOFF(r_) (((unsigned)(r_) * (unsigned)nphi + (unsigned)j) * 8u)
#ifndef ADI_OFF32
#define ADI_OFF32 1 /* loads only: with the stores in this form too the 128-register kernel spills 12 bytes and loses 14 us */
#endif
#if ADI_OFF32 & 1
#define ADI_LD(g_, r_) ld_off(g_, MOFF(r_))
#else
#define ADI_LD(g_, r_) (g_)[IDX(r_, j)]
#endif
#if ADI_OFF32 & 2
#define ADI_ST(g_, r_, v_) st_off(g_, MOFF(r_), v_)
#else
#define ADI_ST(g_, r_, v_) (g_)[IDX(r_, j)] = (v_)
#endif
#define MARCH_VALID 59
#define MARCH_LO 3

// StabilizeViscosity (viscosity.cpp:256-348): the correction factors c1_phi, c1_r of ring k -- the diagonal of the
// Jacobian of the viscous acceleration -- from the nu Sigma products the stress stages of the march hold anyway:
// ns_k / ns_k1 / ns_k_jn = VISCOSITY_SIGMA_RP at the corners (k, j), (k+1, j), (k, j+1); nsg, nsg_jm, nsg_im =
// nu Sigma of the cells (k, j), (k, j-1), (k-1, j); sg, sg_jm, sg_im the densities.
struct ViscFactors {
    double cphi, cr;
};
__device__ __forceinline__ ViscFactors visc_factors_row(const Dev &P, int k, double ns_k, double ns_k1, double ns_k_jn,
                                                        double nsg, double nsg_jm, double nsg_im, double sg, double sg_jm,
                                                        double sg_im)
{
    const double Ra = P.Rinf[k], rs = P.Rsup[k];
    const double TwoDiffRaSq = 2.0 / (rs * rs - Ra * Ra);
    const double FourThirdInvRbInvdphiSq = 4.0 / 3.0 / P.Rmed[k] * P.invdphi * P.invdphi;
    const double a0 = ns_k * P.g_ra3[k] * P.InvDiffRmed[k];
    const double a1 = ns_k1 * P.g_ra3[k + 1] * P.InvDiffRmed[k + 1];
    const double cphi_rp = -P.InvRmed[k] * TwoDiffRaSq * (a1 + a0);
    const double cphi_pp = -FourThirdInvRbInvdphiSq * (nsg + nsg_jm);
    const double sigma_avg_phi = 0.5 * (sg + sg_jm);
    ViscFactors f;
    f.cphi = (cphi_rp + cphi_pp) / (sigma_avg_phi * P.Rmed[k]);
    const double sigma_avg_r = 0.5 * (sg + sg_im);
    const double cr_rp = -(ns_k_jn + ns_k) / (P.dphi * P.dphi * Ra);
    const double cr_pp_1 = 2.0 * nsg * (0.5 * P.InvRmed[k] + 1.0 / 3.0 * Ra * P.InvDiffRsupRb[k]);
    const double cr_pp_2 = 2.0 * nsg_im * (0.5 * P.InvRmed[k - 1] - 1.0 / 3.0 * Ra * P.InvDiffRsupRb[k - 1]);
    const double cr_rr_1 = P.Rmed[k] * 2.0 * nsg * (-P.InvDiffRsup[k] + 1.0 / 3.0 * Ra * P.InvDiffRsupRb[k]);
    const double cr_rr_2 = -1.0 * P.Rmed[k - 1] * 2.0 * nsg_im * (P.InvDiffRsup[k - 1] - 1.0 / 3.0 * Ra * P.InvDiffRsupRb[k - 1]);
    const double cr_pp = -0.5 * (cr_pp_1 + cr_pp_2);
    const double cr_rr = P.InvDiffRmed[k] * (cr_rr_1 + cr_rr_2);
    const double Rmed_mid = 0.5 * (P.Rmed[k] + P.Rmed[k - 1]);
    f.cr = P.radial_viscosity_factor * (cr_rr + cr_rp + cr_pp) / (sigma_avg_r * Rmed_mid);
    return f;
}
// viscosity.cpp:386-391|413-417: corr = 1 / (max(1 + dt c, 0) - dt c)
__device__ __forceinline__ double visc_corr_march(double dt, double c) { return 1.0 / (dmax(1.0 + dt * c, 0.0) - dt * c); }

// STAB: StabilizeViscosity 1 | 2 -- the correction factors are formed in stage E and stored (the CFL condition of
// mode 2 and fcpt_download read the grids), mode 1 also damps the viscous velocity update with them.
// ACC: BodyForceFromPotential: no -- the window carries ACCEL_RADIAL (in the potential's place) and ACCEL_AZIMUTHAL of
// CalculateAccelOnGas instead of the potential (SourceEuler.cpp:348-353, 406-411)
template <int AV, bool STAB, bool ACC = false> // AV 0: none, 1: TW, 2: SN
// (six wavefronts per SIMD = 80 VGPRs, asked for: the plain kernels settle there by themselves, but one more live value --
//  the trace's time stamp, the step length from the in-kernel CFL fold -- tips the allocator's own choice to 82 = five)
__global__ void __launch_bounds__(256) __attribute__((amdgpu_waves_per_eu(STAB ? 4 : (ACC ? 5 : 6), STAB ? 4 : (ACC ? 5 : 6))))
k_source_march(const Dev P, int segs, int rows_per_chunk, int ring_sums, int bc_fold)
{
    const int lane = threadIdx.x & 63;
    const int nr = P.nr, nphi = P.nphi;
    // bc_fold bit 1: this launch stands behind the ring kernel of the CFL reduction without k_cfl_final in between --
    // every workgroup folds the reduction and applies the time-step policy itself (all of its wavefronts: before any returns)
    double dt;
    if (bc_fold & 2) { // (into scalar registers, where the value loaded from the clock lives too: two VGPRs less for the whole kernel)
        const double v = cfl_fold_in_step(P);
        const unsigned lo = __builtin_amdgcn_readfirstlane((unsigned)__double2loint(v)), hi = __builtin_amdgcn_readfirstlane((unsigned)__double2hiint(v));
        dt = __hiloint2double((int)hi, (int)lo);
    } else {
        dt = P.clk->dt;
    }
    bc_fold &= 1;
    int wave, seg, k0, k1;
    if (P.sm_sched) { // (segment, first ring, one past the last) of every wavefront in the order of dispatch: source_schedule()
        wave = __builtin_amdgcn_readfirstlane(blockIdx.x * MARCH_WAVES + (threadIdx.x >> 6));
        const int __attribute__((address_space(4))) *e = (const int __attribute__((address_space(4))) *)P.sm_sched + 4 * wave;
        seg = e[0], k0 = e[1], k1 = e[2];
        if (k0 >= k1)
            return;
    } else {
        wave = __builtin_amdgcn_readfirstlane(xcd_block(blockIdx.x, gridDim.x) * MARCH_WAVES + (threadIdx.x >> 6));
        const int chunk = wave / segs;
        seg = wave - chunk * segs;
        k0 = chunk * rows_per_chunk;
        if (k0 > nr)
            return;
        k1 = (k0 + rows_per_chunk < nr + 1) ? k0 + rows_per_chunk : nr + 1; // v_r has rows 0..nr
    }
#ifdef SM_TRACE /* profiles/tools/wave_trace_source.py: start, end (10 ns ticks), first ring, one past the last of every wavefront */
    if (lane == 0) {
        P.temperature[4 * wave] = (double)wall_clock64(); // (a grid neither marching kernel touches)
        P.temperature[4 * wave + 2] = (double)k0;
        P.temperature[4 * wave + 3] = (double)k1;
    }
#endif
    const int jraw = seg * MARCH_VALID - MARCH_LO + lane;
    const int j = jraw < 0 ? jraw + nphi : (jraw >= nphi ? jraw - nphi : jraw);
    const bool store_lane = lane >= MARCH_LO && lane < MARCH_LO + MARCH_VALID && jraw < nphi;
    const double C2 = P.art_visc_factor * P.art_visc_factor;

#define NEXT(x) lane_next(x) /* value of cell j+1 */
#define PREV(x) lane_prev(x) /* value of cell j-1 */
    auto crow = [nr](int r) { return r < 0 ? 0 : (r > nr - 1 ? nr - 1 : r); };   // cell rows
    auto vrow = [nr](int r) { return r < 0 ? 0 : (r > nr ? nr : r); };           // v_r rows

    // rolling state (suffix _1.._3 = rings m-1..m-3)
    double S_m = 0, S_1 = 0, S_2 = 0, S_3 = 0, Sp_m = 0, Sp_1 = 0, Sp_2 = 0; // Sigma and Sigma(j-1)
    double F_m = 0, F_1 = 0;                                               // potential (ACC: radial acceleration)
    const double *const fgrid = ACC ? P.accel_r : P.potential;
    double G_m = 0;                                                        // ACC: azimuthal acceleration
    double va0_m = 0, va0_1 = 0, va0n_m = 0, va0n_1 = 0;                   // v_phi (input) and (j+1)
    double vr1_m = 0, vr1_1 = 0, va1_m = 0, va1_1 = 0;                      // after source terms
    double qr_1 = 0, qr_2 = 0, qp_1 = 0, qp_2 = 0;                          // Q_rr/Q_pp (TW) or q_r/q_phi (SN)
    double vr2_1 = 0, vr2_2 = 0, va2_1 = 0, va2_2 = 0;                      // after artificial viscosity
    double trr_2 = 0, trr_3 = 0, tpp_2 = 0, tpp_3 = 0, trp_1 = 0, trp_2 = 0;
    double nsrp_1 = 0, nsrp_2 = 0; // STAB: nu Sigma at the corners of rings m-1, m-2 (VISCOSITY_SIGMA_RP)
    double nu_d_prev = 0;          // STAB: viscosity of ring m-3

    // ring k0-3 is the "previous" ring of the first iteration
    {
        const int r = crow(k0 - 3);
        S_m = ld_off(P.sigma, MOFF(r));
        F_m = ld_off(fgrid, MOFF(r));
        va0_m = ld_off(P.vazi, MOFF(r));
    }
    // software prefetch of the next input ring (requested before the first ring is used: one round trip, not two)
    int rn = k0 - 2;
    double pS = ld_off(P.sigma, MOFF(crow(rn))), pF = ld_off(fgrid, MOFF(crow(rn)));
    double pG = ACC ? ld_off(P.accel_az, MOFF(crow(rn))) : 0.0;
    double pVa = ld_off(P.vazi, MOFF(crow(rn))), pVr = ld_off(P.vrad, MOFF(vrow(rn)));
    asm volatile("" : "+v"(S_m), "+v"(va0_m), "+v"(pVr)); // (keeps the lane shifts below, and their wait, behind the last request)
    Sp_m = PREV(S_m);
    va0n_m = NEXT(va0_m);

    for (int m = k0 - 2; m <= k1 + 1; ++m) {
        const SrcRow R = crow_load(P.src_tab, m + 2); // every per-ring factor of this iteration, one batch
        // ---- shift the window, take the prefetched ring m, prefetch ring m+1 ------------
        S_3 = S_2; S_2 = S_1; S_1 = S_m; Sp_2 = Sp_1; Sp_1 = Sp_m;
        F_1 = F_m;
        va0_1 = va0_m; va0n_1 = va0n_m;
        vr1_1 = vr1_m; va1_1 = va1_m;
        S_m = pS; F_m = pF; va0_m = pVa;
        if (ACC)
            G_m = pG;
        const double vr0_m = pVr;
        {
            const int r = m + 1;
            pS = ld_off(P.sigma, MOFF(crow(r)));
            pF = ld_off(fgrid, MOFF(crow(r)));
            if (ACC)
                pG = ld_off(P.accel_az, MOFF(crow(r)));
            pVa = ld_off(P.vazi, MOFF(crow(r)));
            pVr = ld_off(P.vrad, MOFF(vrow(r)));
        }
        Sp_m = PREV(S_m);
        va0n_m = NEXT(va0_m);
        const double Fp_m = ACC ? PREV(G_m) : PREV(F_m);

        // ---- A: source terms on ring m ---------------------------------------------------
        {
            const int r = m;
            const double P_m = S_m * R.cs2_m, P_1 = S_1 * R.cs2_m1, Pp_m = Sp_m * R.cs2_m;
            vr1_m = vr0_m;
            if (r >= P.one_no_ghost_vr && r < P.maxmo_no_ghost_vr) {
                double gradp = 2.0 * FAST_RCP_SRC(S_m + S_1);
                gradp *= (P_m - P_1);
                gradp *= R.idr_m;
                const double gradphi = ACC ? -(F_m + F_1) * 0.5 : (F_m - F_1) * R.idr_m;
                const double vsum = va0_m + va0n_m + va0_1 + va0n_1;
                const double vt = 0.25 * vsum + R.rinf_om_m;
                const double vt2 = vt * vt;
                vr1_m = vr0_m + dt * (-gradp - gradphi + vt2 * R.inv_rinf_m);
            }
            va1_m = va0_m;
            if (r >= P.zero_no_ghost && r < P.max_no_ghost) {
                const double invdxtheta = R.inv_dxt_m; // 2 / (dphi (Rsup + Rinf))
                const double gradp = 2.0 * FAST_RCP_SRC(S_m + Sp_m) * (P_m - Pp_m) * invdxtheta;
                const double gradphi = ACC ? -(G_m + Fp_m) * 0.5 : (F_m - Fp_m) * invdxtheta;
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
                    va2_1 = va1_1 + 2.0 * dt * (R.inv_rsum_c * FAST_RCP_SRC(sigma_phi_avg)) * (qp_1 - qpp_p) * P.invdphi;
                }
                if (upd_vr) {
                    const double sigma_r_avg = 0.5 * (S_1 + S_2);
                    const double rm = R.rmed_c, rmm = R.rmed_cm1;
                    vr2_1 = vr1_1 + P.radial_viscosity_factor * dt * FAST_RCP_SRC(sigma_r_avg) * 2.0 * R.inv_drmed2_c *
                                        ((qr_1 * rm - qr_2 * rmm) - 0.5 * (qp_1 + qp_2) * (rm - rmm));
                }
            } else if (AV == 2) {
                const double qphi_p = PREV(qp_1);
                if (upd_vr)
                    vr2_1 = vr1_1 - dt * 2.0 * FAST_RCP_SRC(S_1 + S_2) * (qr_1 - qr_2) * R.idr_c;
                if (r >= P.zero_no_ghost && r < P.max_no_ghost) {
                    const double invdxtheta = R.inv_dxtheta_c;
                    va2_1 = va1_1 - dt * 2.0 * FAST_RCP_SRC(S_1 + Sp_1) * (qp_1 - qphi_p) * invdxtheta;
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
            if (STAB) {
                nsrp_2 = nsrp_1;
                nsrp_1 = 0.0;
            }
            if (r >= 1 && r <= nr - 1) {
                const double dvazirdr = (va2_1 * R.inv_rmed_r - va2_2 * R.inv_rmed_rm1) * R.idr_r;
                const double dvrdphi = (vr2_1 - vr2p_1) * P.invdphi;
                const double drp = R.rinf_r * dvazirdr + dvrdphi * R.inv_rinf_r;
                const double nu = R.nu_avg_r;
                const double sigma = 0.25 * (S_1 + S_2 + Sp_1 + Sp_2);
                trp_1 = nu * sigma * drp;
                if (STAB)
                    nsrp_1 = nu * sigma;
            }
        }
        // ---- E: viscous update of ring k = m-2 and store ----------------------------------
        {
            const int k = m - 2;
            const double tpp_p = PREV(tpp_2);
            const double trp_n = NEXT(trp_2);
            const double nsrp_jn = STAB ? NEXT(nsrp_2) : 0.0; // corner (k, j+1)
            if (k >= k0 && k < k1) {
                double vr3 = vr2_2, va3 = va2_2;
                double corr_phi = 1.0, corr_r = 1.0;
                if (STAB && k >= 1 && k <= nr - 1) {
                    const ViscFactors f = visc_factors_row(P, k, nsrp_2, nsrp_1, nsrp_jn, R.nu_d * S_2, R.nu_d * Sp_2,
                                                           nu_d_prev * S_3, S_2, Sp_2, S_3);
                    if (store_lane) {
                        st_off(P.cfac_phi, MOFF(k), f.cphi);
                        st_off(P.cfac_r, MOFF(k), f.cr);
                    }
                    if (P.stabilize == 1) {
                        corr_phi = visc_corr_march(dt, f.cphi);
                        corr_r = visc_corr_march(dt, f.cr);
                    }
                }
                if (k >= 1 && k < nr - 1) {
                    const double sigma_avg = 0.5 * (S_2 + Sp_2);
                    const double dVp = dt * R.inv_rmed_k * FAST_RCP_SRC(sigma_avg) *
                                       (R.two_inv_dra2_k * (R.ra1sq_k * trp_1 - R.ra0sq_k * trp_2) +
                                        (tpp_2 - tpp_p) * P.invdphi);
                    va3 = va2_2 + (STAB ? dVp * corr_phi : dVp);
                }
                if (k >= P.one_no_ghost_vr && k < P.maxmo_no_ghost_vr) {
                    const double sigma_avg = 0.5 * (S_2 + S_3);
                    const double dVr = dt * FAST_RCP_SRC(sigma_avg) * P.radial_viscosity_factor * 2.0 * R.inv_rmsum_k *
                                       ((R.rmed_k * trr_2 - R.rmed_km1 * trr_3) * R.idr_k +
                                        (trp_n - trp_2) * P.invdphi - 0.5 * (tpp_2 + tpp_3));
                    vr3 = vr2_2 + (STAB ? dVr * corr_r : dVr);
                }
                if (store_lane) {
                    st_off(P.vrad_b, MOFF(k), vr3);
                    if (k < nr)
                        st_off(P.vazi_b, MOFF(k), va3);
                }
                if (ring_sums && k < nr) { // this segment's share of sum_j v_phi(k, j) for the transport's <v_phi>
                    const double part = wave_sum(store_lane ? va3 : 0.0);
                    if (lane == 63)
                        P.ring_part[k * P.ring_pstride + seg] = part;
                }
            }
        }
        if (STAB)
            nu_d_prev = R.nu_d;
    }
    // bc_fold: apply_boundary_condition(final = false) of step_Euler (simulation.cpp:208) on this wavefront's columns of
    // the post-kick view, right behind its own stores of the rings the conditions read (rows 1, 2 / nr-2 .. nr: the
    // same lane wrote them; the launcher folds only if the last chunk holds at least three rows)
    if (bc_fold && store_lane) {
        const int sides = (k0 == 0 ? 1 : 0) | (k1 == nr + 1 ? 2 : 0);
        if (sides) {
            Dev Q = P;
            Q.vrad = P.vrad_b;
            Q.vazi = P.vazi_b;
            boundary_column(Q, j, sides);
        }
    }
#ifdef SM_TRACE
    if (lane == 0)
        P.temperature[4 * wave + 1] = (double)wall_clock64();
#endif
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
// POT: the potential of ring m is evaluated here (CalculateNbodyPotential, Pframeforce.cpp:21-94, with the
// smoothing length ThicknessSmoothing * H of the cell, Force.cpp:124-159) instead of read from the grid
// k_potential would have to refresh every step, because H follows the energy.
template <int AV, bool COOL, bool POT, bool STAB, bool ACC = false> // AV 0: none, 1: TW, 2: SN; ACC: see k_source_march
__device__ __forceinline__ void source_march_adi_body(const Dev &P, int segs, int rows_per_chunk, int ring_sums, int bc_fold)
{
    const int lane = threadIdx.x & 63;
    const int nr = P.nr, nphi = P.nphi;
    // bc_fold bit 1: this launch stands behind the ring kernel of the CFL reduction without k_cfl_final in between --
    // every workgroup folds the reduction and applies the time-step policy itself (all of its wavefronts: before any returns)
    double dt;
    if (bc_fold & 2) { // (into scalar registers, where the value loaded from the clock lives too: two VGPRs less for the whole kernel)
        const double v = cfl_fold_in_step(P);
        const unsigned lo = __builtin_amdgcn_readfirstlane((unsigned)__double2loint(v)), hi = __builtin_amdgcn_readfirstlane((unsigned)__double2hiint(v));
        dt = __hiloint2double((int)hi, (int)lo);
    } else {
        dt = P.clk->dt;
    }
    bc_fold &= 1;
    int wave, seg, k0, k1;
    if (P.sm_sched) { // (segment, first ring, one past the last) of every wavefront in the order of dispatch: source_schedule()
        wave = __builtin_amdgcn_readfirstlane(blockIdx.x * MARCH_WAVES + (threadIdx.x >> 6));
        const int __attribute__((address_space(4))) *e = (const int __attribute__((address_space(4))) *)P.sm_sched + 4 * wave;
        seg = e[0], k0 = e[1], k1 = e[2];
        if (k0 >= k1)
            return;
    } else {
        wave = __builtin_amdgcn_readfirstlane(xcd_block(blockIdx.x, gridDim.x) * MARCH_WAVES + (threadIdx.x >> 6));
        const int chunk = wave / segs;
        seg = wave - chunk * segs;
        k0 = chunk * rows_per_chunk;
        if (k0 > nr)
            return;
        k1 = (k0 + rows_per_chunk < nr + 1) ? k0 + rows_per_chunk : nr + 1; // v_r has rows 0..nr
    }
#ifdef SM_TRACE /* profiles/tools/wave_trace_source.py: start, end (10 ns ticks), first ring, one past the last of every wavefront */
    if (lane == 0) {
        P.temperature[4 * wave] = (double)wall_clock64(); // (a grid neither marching kernel touches)
        P.temperature[4 * wave + 2] = (double)k0;
        P.temperature[4 * wave + 3] = (double)k1;
    }
#endif
    const int jraw = seg * MARCH_VALID - MARCH_LO + lane;
    const int j = jraw < 0 ? jraw + nphi : (jraw >= nphi ? jraw - nphi : jraw);
    const bool store_lane = lane >= MARCH_LO && lane < MARCH_LO + MARCH_VALID && jraw < nphi;
    const double C2 = P.art_visc_factor * P.art_visc_factor;
    const double gm1 = P.gamma - 1.0;
    const double inv_sqrt_gamma = 1.0 / sqrt(P.gamma);
    const bool dissipate = P.art_visc_dissipation != 0;
    constexpr bool cooling = COOL;

#define NEXT(x) lane_next_keep(x) /* value of cell j+1 (the read-modify-write form: see device_util.h) */
#define PREV(x) lane_prev_keep(x) /* value of cell j-1 */
    auto crow = [nr](int r) { return r < 0 ? 0 : (r > nr - 1 ? nr - 1 : r); };   // cell rows
    auto vrow = [nr](int r) { return r < 0 ? 0 : (r > nr ? nr : r); };           // v_r rows

    // rolling state (suffix _1.._3 = rings m-1..m-3)
    static_assert(!(ACC && POT), "the inline potential is the potential's path");
    double S_m = 0, S_1 = 0, S_2 = 0, S_3 = 0, Sp_m = 0; // Sigma and Sigma(j-1)
    double F_m = 0, F_1 = 0;                                               // potential (ACC: radial acceleration)
    const double *const fgrid = ACC ? P.accel_r : P.potential;
    double G_m = 0;                                                        // ACC: azimuthal acceleration
    // (register diet, 142 -> 128 VGPRs = 4 wavefronts per SIMD: the pressure (gamma - 1) e, Sigma(j-1) and nu(j-1)
    //  of ring m-2, v_phi(j+1) of ring m-1 and this lane's cos / sin are re-formed where they are used instead of
    //  carried in the rolling window -- the same values, bit for bit)
    double e0_m = 0, e0_1 = 0;                                             // energy as loaded
    double e2_1 = 0, e2_2 = 0;                                             // after S3 + dissipation + floor
    double nu_1 = 0, nu_2 = 0, nup_1 = 0, H_1 = 0, H_2 = 0;                // viscosity (and at j-1), scale height
    double va0_m = 0, va0_1 = 0, va0n_m = 0;                               // v_phi (input) and (j+1)
    double vr1_m = 0, vr1_1 = 0, va1_m = 0, va1_1 = 0;                      // after source terms
    double qr_1 = 0, qr_2 = 0, qp_1 = 0, qp_2 = 0;                          // Q_rr/Q_pp (TW) or q_r/q_phi (SN)
    double vr2_1 = 0, vr2_2 = 0, va2_1 = 0, va2_2 = 0;                      // after artificial viscosity
    double trr_2 = 0, trr_3 = 0, tpp_2 = 0, tpp_3 = 0, trp_1 = 0, trp_2 = 0;
    double nsrp_1 = 0, nsrp_2 = 0, nu_3 = 0; // STAB: nu Sigma at the corners of rings m-1, m-2; viscosity of ring m-3
    // 1 / (Sigma + Sigma(i-1)) and 1 / (Sigma + Sigma(j-1)) of a ring serve stages A, C and E of three consecutive
    // iterations: formed once and carried (a v_rcp_f64 with its two Newton steps costs eight FMAs; four of the 28 per
    // ring saved, -1.4 % per step).  1 / Sigma (potential, V0, SubStep3) does not fit the 128 registers as well.
    double rr_m = 0, rr_1 = 0, rp_m = 0, rp_1 = 0;

    // k_potential in registers: this lane's column (cos phi_j, sin phi_j) against the bodies
    const double gg1 = P.gamma * gm1;
    auto potential_of = [&](int r, double sg, double en) {
        const double rmed = P.Rmed[r];
        const double x = rmed * P.cosphi[j], y = rmed * P.sinphi[j]; // (cache hits after the first ring)
        // (ThicknessSmoothing H)^2 with H = c_s / (sqrt(gamma) Omega_K), c_s^2 = gamma (gamma - 1) e / Sigma: no root needed
        const double hk = P.thickness_smoothing * inv_sqrt_gamma * P.g_inv_omk[r];
        const double smooth2 = (gg1 * en * FAST_RCP_SRC(sg)) * (hk * hk);
        double pot = 0.0;
        for (int k = 0; k < P.nbodies; ++k) {
            const double dx = x - P.bx[k];
            const double dy = y - P.by[k];
            const double dist_2 = dx * dx + dy * dy;
            const double d2s = dist_2 + smooth2;
            const double inv_d = FAST_RSQRT_SRC(d2s); // 1 / d_smoothed
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
        return pot;
    };
    // ring k0-3 is the "previous" ring of the first iteration
    {
        const int r = crow(k0 - 3);
        S_m = ADI_LD(P.sigma, r);
        va0_m = ADI_LD(P.vazi, r);
        e0_m = ADI_LD(P.energy, r);
        if (!POT)
            F_m = ADI_LD(fgrid, r);
    }
    // software prefetch of the next input ring (requested before the first ring is used: one round trip, not two)
    int rn = k0 - 2;
    double pS = ADI_LD(P.sigma, crow(rn)), pF = POT ? 0.0 : ADI_LD(fgrid, crow(rn)), pE = ADI_LD(P.energy, crow(rn));
    double pG = ACC ? ADI_LD(P.accel_az, crow(rn)) : 0.0;
    double pVa = ADI_LD(P.vazi, crow(rn)), pVr = ADI_LD(P.vrad, vrow(rn));
    if (POT)
        F_m = potential_of(crow(k0 - 3), S_m, e0_m);
    Sp_m = PREV(S_m);
    va0n_m = NEXT(va0_m);

    for (int m = k0 - 2; m <= k1 + 1; ++m) {
        const SrcRow R = crow_load(P.src_tab, m + 2); // every per-ring factor of this iteration, one batch
        // ---- shift the window, take the prefetched ring m, prefetch ring m+1 ------------
        if (STAB)
            S_3 = S_2;
        S_2 = S_1; S_1 = S_m;
        const double Sp_2 = PREV(S_2), Sp_1 = PREV(S_1);
        const double rr_2 = rr_1, rp_2 = rp_1;
        rr_1 = rr_m; rp_1 = rp_m;
        F_1 = F_m; e0_1 = e0_m;
        va0_1 = va0_m;
        vr1_1 = vr1_m; va1_1 = va1_m;
        if (STAB)
            nu_3 = nu_2;
        e2_2 = e2_1; nu_2 = nu_1; H_2 = H_1;
        const double nup_2 = PREV(nu_2);
        S_m = pS; F_m = pF; va0_m = pVa; e0_m = pE;
        if (ACC)
            G_m = pG;
        const double vr0_m = pVr;
        {
            const int r = m + 1;
            pS = ADI_LD(P.sigma, crow(r));
            if (!POT)
                pF = ADI_LD(fgrid, crow(r));
            if (ACC)
                pG = ADI_LD(P.accel_az, crow(r));
            pE = ADI_LD(P.energy, crow(r));
            pVa = ADI_LD(P.vazi, crow(r));
            pVr = ADI_LD(P.vrad, vrow(r));
        }
        if (POT)
            F_m = potential_of(crow(m), S_m, e0_m);
        const double Pr_m = gm1 * e0_m, Pr_1 = gm1 * e0_1;
        Sp_m = PREV(S_m);
        rr_m = FAST_RCP_SRC(S_m + S_1);
        rp_m = FAST_RCP_SRC(S_m + Sp_m);
        va0n_m = NEXT(va0_m);
        const double va0n_1 = NEXT(va0_1);
        const double Fp_m = ACC ? PREV(G_m) : PREV(F_m);
        const double Prp_m = PREV(Pr_m);

        // ---- A: source terms on ring m ---------------------------------------------------
        {
            const int r = m;
            vr1_m = vr0_m;
            if (r >= P.one_no_ghost_vr && r < P.maxmo_no_ghost_vr) {
                double gradp = 2.0 * rr_m;
                gradp *= (Pr_m - Pr_1);
                gradp *= R.idr_m;
                const double gradphi = ACC ? -(F_m + F_1) * 0.5 : (F_m - F_1) * R.idr_m;
                const double vsum = va0_m + va0n_m + va0_1 + va0n_1;
                const double vt = 0.25 * vsum + R.rinf_om_m;
                const double vt2 = vt * vt;
                vr1_m = vr0_m + dt * (-gradp - gradphi + vt2 * R.inv_rinf_m);
            }
            va1_m = va0_m;
            if (r >= P.zero_no_ghost && r < P.max_no_ghost) {
                const double invdxtheta = R.inv_dxt_m; // 2 / (dphi (Rsup + Rinf))
                const double gradp = 2.0 * rp_m * (Pr_m - Prp_m) * invdxtheta;
                const double gradphi = ACC ? -(G_m + Fp_m) * 0.5 : (F_m - Fp_m) * invdxtheta;
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
                e = e * exp_small(-gm1 * dt * DIV_V);
            }
            if (AV == 1) {
                const double eps_rr = (vr1_m - vr1_1) * R.inv_drsup_b;
                const double eps_pp = R.inv_rmed_b * ((va1n_1 - va1_1) * P.invdphi + 0.5 * (vr1_m + vr1_1));
                const double div_V = dmin(eps_rr + eps_pp, 0.0);
                const double l_sq = R.lsq_b;
                qr_1 = l_sq * S_1 * -div_V * (eps_rr - 1.0 / 3.0 * div_V);
                qp_1 = l_sq * S_1 * -div_V * (eps_pp - 1.0 / 3.0 * div_V);
                if (dissipate && r > P.zero_no_ghost && r < P.max_no_ghost) {
                    const double Qplus = -l_sq * div_V * S_1 * (1.0 / 3.0) * // (a multiply: the reference's `* 1.0 / 3.0` is an IEEE division per cell)
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
            const double cs = sqrt(P.gamma * gm1 * e * FAST_RCP_SRC(S_1));
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
                    // 1 / sigma_phi_avg = 1 / (0.5 (S_1 + Sp_1)) = 2 rp_1, bit for bit
                    va2_1 = va1_1 + 2.0 * dt * (R.inv_rsum_c * (2.0 * rp_1)) * (qp_1 - qpp_p) * P.invdphi;
                }
                if (upd_vr) {
                    const double rm = R.rmed_c, rmm = R.rmed_cm1;
                    vr2_1 = vr1_1 + P.radial_viscosity_factor * dt * (2.0 * rr_1) * 2.0 * R.inv_drmed2_c *
                                        ((qr_1 * rm - qr_2 * rmm) - 0.5 * (qp_1 + qp_2) * (rm - rmm));
                }
            } else if (AV == 2) {
                const double qphi_p = PREV(qp_1);
                if (upd_vr)
                    vr2_1 = vr1_1 - dt * 2.0 * rr_1 * (qr_1 - qr_2) * R.idr_c;
                if (r >= P.zero_no_ghost && r < P.max_no_ghost)
                    va2_1 = va1_1 - dt * 2.0 * rp_1 * (qp_1 - qphi_p) * R.inv_dxtheta_b;
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
            if (STAB) {
                nsrp_2 = nsrp_1;
                nsrp_1 = 0.0;
            }
            if (r >= 1 && r <= nr - 1) {
                const double dvazirdr = (va2_1 * R.inv_rmed_r - va2_2 * R.inv_rmed_rm1) * R.idr_r;
                const double dvrdphi = (vr2_1 - vr2p_1) * P.invdphi;
                const double drp = R.rinf_r * dvazirdr + dvrdphi * R.inv_rinf_r;
                const double nu = 0.25 * (nu_1 + nu_2 + nup_1 + nup_2);
                const double sigma = 0.25 * (S_1 + S_2 + Sp_1 + Sp_2);
                trp_1 = nu * sigma * drp;
                if (STAB)
                    nsrp_1 = nu * sigma;
            }
        }
        // ---- E: viscous update, viscous heating and SubStep3 of ring k = m-2, store -------
        {
            const int k = m - 2;
            const double tpp_p = PREV(tpp_2);
            const double trp_n = NEXT(trp_2);
            const double trp_1n = NEXT(trp_1);
            const double nsrp_jn = STAB ? NEXT(nsrp_2) : 0.0; // corner (k, j+1)
            if (k >= k0 && k < k1) {
                double vr3 = vr2_2, va3 = va2_2;
                const bool row_va = k >= 1 && k < nr - 1;
                double corr_phi = 1.0, corr_r = 1.0;
                if (STAB && k >= 1 && k <= nr - 1) {
                    const ViscFactors f = visc_factors_row(P, k, nsrp_2, nsrp_1, nsrp_jn, nu_2 * S_2, nup_2 * Sp_2,
                                                           nu_3 * S_3, S_2, Sp_2, S_3);
                    if (store_lane) {
                        ADI_ST(P.cfac_phi, k, f.cphi);
                        ADI_ST(P.cfac_r, k, f.cr);
                    }
                    if (P.stabilize == 1) {
                        corr_phi = visc_corr_march(dt, f.cphi);
                        corr_r = visc_corr_march(dt, f.cr);
                    }
                }
                if (row_va) {
                    const double dVp = dt * R.inv_rmed_k * (2.0 * rp_2) *
                                       (R.two_inv_dra2_k * (R.ra1sq_k * trp_1 - R.ra0sq_k * trp_2) +
                                        (tpp_2 - tpp_p) * P.invdphi);
                    va3 = va2_2 + (STAB ? dVp * corr_phi : dVp);
                }
                if (k >= P.one_no_ghost_vr && k < P.maxmo_no_ghost_vr) {
                    const double dVr = dt * (2.0 * rr_2) * P.radial_viscosity_factor * 2.0 * R.inv_rmsum_k *
                                       ((R.rmed_k * trr_2 - R.rmed_km1 * trr_3) * R.idr_k +
                                        (trp_n - trp_2) * P.invdphi - 0.5 * (tpp_2 + tpp_3));
                    vr3 = vr2_2 + (STAB ? dVr * corr_r : dVr);
                }
                double qplus = 0.0, qminus = 0.0, e = e2_2;
                if (k < nr) {
                    if (P.heating_viscous && row_va && nu_2 != 0.0) { // viscous_heating
                        const double tau_r_phi = 0.25 * (trp_2 + trp_1 + trp_n + trp_1n);
                        double q = FAST_RCP_SRC(2.0 * nu_2 * S_2) * (trr_2 * trr_2 + 2 * (tau_r_phi * tau_r_phi) + tpp_2 * tpp_2);
                        q += (2.0 / 9.0) * nu_2 * S_2 * (divv_2 * divv_2);
                        q *= P.heating_viscous_factor;
                        qplus += q;
                    }
                    if (row_va) { // SubStep3, rows [1, Nr-1)
                        const double bb = P.b_fac * FAST_RCP_SRC(S_2), b2 = bb * bb; // substep3_alpha
                        const double alpha = 1.0 + H_2 * P.alpha_fac * (b2 * b2) * (e * e * e); // alpha_fac = 2 * 4 sigma_SB / c
                        const double ralpha = FAST_RCP_SRC(alpha);
                        double tau_eff = 0.0;
                        if (cooling) { // calculate_qminus
                            const Cooling cool = cooling_terms(P, k, j, IDX(k, j), S_2, e, H_2);
                            qminus = cool.qminus * ralpha;
                            tau_eff = cool.tau_eff;
                            qplus += cool.qplus_star;
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
                    ADI_ST(P.vrad_b, k, vr3);
                    if (k < nr) {
                        ADI_ST(P.vazi_b, k, va3);
                        ADI_ST(P.energy_b, k, e);
                        if (!P.q_skip) { // (the grids are outputs: fcpt_run_steps has them written by its last step only -- two of ten grids of traffic)
                            ADI_ST(P.qplus, k, qplus);
                            ADI_ST(P.qminus, k, qminus);
                        }
                        ADI_ST(P.qdiff, k, qplus - qminus); // what the CFL condition needs of the two (cfl.cpp:303-316)
                        // step_LeapFrog evaluates the mid-step potential with the scale height this kick's
                        // recalculate_viscosity left behind (simulation.cpp:340-378), not with that of the
                        // transported state: keep the grid for it
                        if (P.leapfrog)
                            ADI_ST(P.scale_height, k, H_2);
                    }
                }
                if (ring_sums && k < nr) { // this segment's share of sum_j v_phi(k, j) for the transport's <v_phi>
                    const double part = wave_sum(store_lane ? va3 : 0.0);
                    if (lane == 63)
                        P.ring_part[k * P.ring_pstride + seg] = part;
                }
            }
        }
    }
    if (bc_fold && store_lane) { // the pre-transport boundary call on this wavefront's edge columns (see k_source_march)
        const int sides = (k0 == 0 ? 1 : 0) | (k1 == nr + 1 ? 2 : 0);
        if (sides) {
            Dev Q = P;
            Q.vrad = P.vrad_b;
            Q.vazi = P.vazi_b;
            Q.energy = P.energy_b;
            boundary_column(Q, j, sides);
        }
    }
#ifdef SM_TRACE
    if (lane == 0)
        P.temperature[4 * wave + 1] = (double)wall_clock64();
#endif
#undef NEXT
#undef PREV
}

// The kernels proper.  Without the cooling terms the body fits 128 VGPRs when the register allocator is told to aim
// for 4 wavefronts per SIMD (127 VGPRs, no scratch; left alone it settles at 134 = 3 wavefronts); the instantiations
// with the opacity laws (220+ VGPRs) and with StabilizeViscosity keep their natural allocation.
template <int AV, bool POT>
__global__ void __launch_bounds__(256) __attribute__((amdgpu_waves_per_eu(4, 4)))
k_source_march_adi(const Dev P, int segs, int rows_per_chunk, int ring_sums, int bc_fold)
{
    source_march_adi_body<AV, false, POT, false>(P, segs, rows_per_chunk, ring_sums, bc_fold);
}
template <int AV, bool COOL, bool POT, bool STAB>
__global__ void __launch_bounds__(256) k_source_march_adi_wide(const Dev P, int segs, int rows_per_chunk, int ring_sums, int bc_fold)
{
    source_march_adi_body<AV, COOL, POT, STAB>(P, segs, rows_per_chunk, ring_sums, bc_fold);
}
// BodyForceFromPotential: no (never the default: one instantiation per artificial viscosity, cooling and
// StabilizeViscosity decided at run time inside would cost the registers of the widest case, so they stay template flags)
template <int AV, bool COOL, bool STAB>
__global__ void __launch_bounds__(256) k_source_march_adi_acc(const Dev P, int segs, int rows_per_chunk, int ring_sums, int bc_fold)
{
    source_march_adi_body<AV, COOL, false, STAB, true>(P, segs, rows_per_chunk, ring_sums, bc_fold);
}
