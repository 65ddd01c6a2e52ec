// Part of fcpt_kernels.hip (one translation unit, namespace fcpt): k_transport_fused: the whole Transport() in one marching kernel.
// Not a stand-alone header: included once, in the order given there.

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
// is enough; k_ring_mean raises P.shift_jump otherwise and this launch does the radial sweep of the
// two-kernel transport instead (see there).
// Nothing intermediate reaches memory: 3 (4) grids read + 3 (4) written instead of 8 + 9
// (10 + 11) doubles per cell for k_transport_radial + k_transport_theta_march.
// Validity in cells of a 64*C segment: right 1 (L+ needs v_phi(j+1)), 4 at either end for the
// two passes, left 1 for L+(j-1), 1 at either end for the v_r lane shift.
// The chunks of one launch: `count` of them, the first `lead` are chunks 0..lead-1 of the grid, the others follow
// `skip` chunks further up.  All chunks at once: {n, n, 0, 1}.  Slabs with neighbours march the chunks that hold
// the rings the neighbours are waiting for first (fcpt_step_device_begin): {1 + tail, 1, gap, 1} then {gap, 0, 1, 0}.
#define TF_XCD_CHUNKS 16 /* launches of at least this many chunks deal whole chunks to the XCDs */
// sched != null: wavefront w of the launch (blockIdx.x * 4 + its index in the workgroup) marches tile sched[4w] over rings
// [sched[4w+1], sched[4w+2]) -- transport_schedule() in launch.h; count = entries, lead / skip do not apply.
struct TfChunks {
    int count, lead, skip, advance_clock;
    const int *sched;
};
template <int C> struct TfHalo {
    static constexpr int lo = C == 2 ? 6 : 5; // even for C = 2: a lane's two cells are final together
    static constexpr int hi = 6;
};

// wave damping with the ring's precomputed exp(-dt f / tau) (k_ring_mean): types as damp_value.  The load of the
// reference value and its use sit in one basic block: a load whose use is behind another branch leaves the compiler's
// s_waitcnt pass with a "maybe pending" register at every later store of the loop, and the pinned prefetch (bottom
// of the loop) would be waited for on the spot again.
__device__ __forceinline__ double damp_apply(double X, int type, double ef, const double *ref, unsigned cell_off, double zero_target)
{
    if (type == 1) {
        const double X0 = ld_off(ref, cell_off);
        return (X - X0) * ef + X0;
    }
    if (type != 0)
        return (X - zero_target) * ef + zero_target;
    return X;
}

// THERM: the cell-local CFL terms of the new state are stored with it (ideal EOS; its own instantiation, because the
// extra live values cost the kernel five dwords of scratch at 128 VGPRs)
template <int C, bool ADI, bool DAMP, int LIM, bool THERM>
__device__ __forceinline__ void transport_fused_body(const Dev &P, const Dev &W, int tiles, int rows, int has_fallback,
                                                     const TfChunks &ch)
{
    // P: view whose vrad/vazi are the velocities to transport; W: view that receives the new state
    constexpr int LO = TfHalo<C>::lo, HI = TfHalo<C>::hi;
    constexpr int NQ = ADI ? 6 : 5; // s, rmp, rmm, lp, lm(, e)
    constexpr int lim = LIM;
    // Register diet of the ideal-EOS instantiation (156 -> 128 VGPRs = 4 instead of 3 wavefronts per SIMD): the raw
    // energy, the raw v_phi and the slope differences of the previous ring are re-derived from the specific
    // quantities of the rolling window instead of being carried along (ulp-level differences: e = (e / Sigma) Sigma,
    // v_phi = ((v_phi + r Omega) r) / r - r Omega)
    constexpr bool DIET = ADI;
    // PIN: the prefetch of ring m+2 is pinned behind convert() of ring m+1 (bottom of the loop).  -1.8 % per step for
    // the ideal EOS when it was introduced; the isothermal kernel gained nothing then (its time was set by the tail of
    // slow wavefronts) and 1.9 % once the chunks were dealt slow ones first (four A/B pairs each).
    // DVP: v_phi re-derived from (v_phi + r Omega) r instead of carried: a carried copy would share the prefetch registers.
    constexpr bool PIN = true, DVP = true;
    const int lane = threadIdx.x & 63;
    // Chunks are dealt to the 8 XCDs round-robin (workgroup b runs on XCD b % 8; all tiles of a chunk on one XCD, whose
    // L2 then serves their shared halo columns), in an order that starts at both ends of the slab and works inward:
    // the rings of the damping zones cost ~1.5x (three more loads per cell, waited for on the spot), and with the
    // chunks in radial order on contiguous XCD ranges the outer zone's wavefronts started last, on one XCD, and ran
    // on alone (2.98 of 4 wavefronts per SIMD on average; -5.5 % kernel time, -3 % / -4.6 % per step, three A/B pairs).
    // (launches of fewer than TF_XCD_CHUNKS chunks -- short slabs -- deal workgroups instead: every XCD gets work)
    if (ch.advance_clock && blockIdx.x == 0 && threadIdx.x == 0) { // sim::time += dt; N_hydro_iter++ (simulation.cpp:226-227)
        clock_advance(W.clk, P.clk->dt);
    }
    // k_ring_mean found a ring pair beyond the one-lane shift (a dt beyond the FARGO shear limit, or a source step that
    // changed v_phi violently): this launch runs the radial sweep of the two-kernel transport instead -- all of its
    // workgroups, with a grid stride -- and the gated azimuthal launch queued behind it finishes the step.
    if (shift_jump_raised(P.shift_jump)) {
        if (has_fallback) {
            const int gx = (P.nphi + 255) / 256, gy = (P.nr + RADIAL_ROWS - 1) / RADIAL_ROWS; // launch2d() of Nphi >= 256
            for (int vb = blockIdx.x; vb < gx * gy; vb += gridDim.x)
                transport_radial_block<true>(P, vb, gx, gx * gy, RADIAL_ROWS);
        } else if (blockIdx.x == 0 && threadIdx.x == 0) {
            W.clk->shear_error = 1; // nothing behind this kernel will redo the step: report it
        }
        return;
    }
    const int nr = P.nr, nphi = P.nphi;
    int tile, r0, r1, trace_slot;
    if (ch.sched) { // (tile, first ring, one past the last) of every wavefront in the order of dispatch: transport_schedule()
        const int wave = __builtin_amdgcn_readfirstlane(blockIdx.x * MARCH_WAVES + (threadIdx.x >> 6));
        if (wave >= ch.count)
            return;
        const int __attribute__((address_space(4))) *e = (const int __attribute__((address_space(4))) *)ch.sched + 4 * wave;
        tile = e[0], r0 = e[1], r1 = e[2];
        trace_slot = wave;
        if (r0 >= r1)
            return;
    } else {
        int chunk_l, wave;
        if (ch.count >= TF_XCD_CHUNKS) {
            const int xcd = blockIdx.x & 7, wq = (blockIdx.x >> 3) * MARCH_WAVES + (threadIdx.x >> 6);
            const int zq = __builtin_amdgcn_readfirstlane(wq / tiles);
            chunk_l = xcd + 8 * zq; // chunk within this launch
            wave = __builtin_amdgcn_readfirstlane(chunk_l * tiles + (wq - zq * tiles));
        } else {
            wave = __builtin_amdgcn_readfirstlane(blockIdx.x * MARCH_WAVES + (threadIdx.x >> 6));
            chunk_l = wave / tiles;
        }
        if (chunk_l >= ch.count)
            return;
        int chunk = chunk_l < ch.lead ? chunk_l : chunk_l + ch.skip;
        if (ch.lead == ch.count && ch.skip == 0) // all chunks in one launch: 0, n-1, 1, n-2, ...
            chunk = (chunk_l & 1) ? ch.count - 1 - (chunk_l >> 1) : (chunk_l >> 1);
        r0 = chunk * rows, r1 = r0 + rows < nr ? r0 + rows : nr;
        if (r0 >= nr)
            return;
        tile = wave - chunk_l * tiles;
        trace_slot = chunk * tiles + tile;
    }
    (void)trace_slot;
#ifdef TF_TRACE /* profiles/tools/wave_trace_transport.py: start and end of every wavefront, 10 ns ticks, in the temperature grid (which no marching kernel touches) */
    if (lane == 0) {
        W.temperature[4 * trace_slot] = (double)wall_clock64();
        W.temperature[4 * trace_slot + 2] = (double)r0;
        W.temperature[4 * trace_slot + 3] = (double)r1;
    }
#endif
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
    double er[DIET ? 1 : 3][C];     // the energy itself
    double vp[DVP ? 1 : 3][C];      // v_phi as loaded
    double d1[DIET ? 1 : NQ][C];    // (w(m-1) - w(m-2)) InvDiffRmed[m-1]
    double idr_prev = 0.0;          // DIET: InvDiffRmed[m-1], to re-form d1
    double hs1[NQ][C];   // limited half slope of ring m-2
    double F1[NQ][C];    // flux through interface m-2
    double rmp_prev[C], S_prev[C]; // transported rm+ and Sigma of the previous ring
#pragma unroll
    for (int c = 0; c < C; ++c) {
#pragma unroll
        for (int q = 0; q < NQ; ++q) {
            w[0][q][c] = w[1][q][c] = w[2][q][c] = hs1[q][c] = F1[q][c] = 0.0;
            if (!DIET)
                d1[q][c] = 0.0;
        }
        if (!DIET)
            er[0][c] = er[1][c] = er[2][c] = 0.0;
        if (!DVP)
            vp[0][c] = vp[1][c] = vp[2][c] = 0.0;
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
        const unsigned row = (unsigned)(in_k ? k : 0) * (unsigned)nphi, rowv = (unsigned)(in_v ? k + 1 : 0) * (unsigned)nphi;
        if (pair_in) {
            if (in_k) {
                const unsigned ob = (row + (unsigned)jin[0]) * 8u;
                const D2 s2 = ld2_off(P.sigma, ob), v2 = ld2_off(P.vazi, ob);
                o.sg[0] = s2.x, o.sg[C - 1] = s2.y, o.va[0] = v2.x, o.va[C - 1] = v2.y;
                if (ADI) {
                    const D2 e2 = ld2_off(P.energy, ob);
                    o.en[0] = e2.x, o.en[C - 1] = e2.y;
                }
            }
            if (in_v) {
                const D2 r2 = ld2_off(P.vrad, (rowv + (unsigned)jin[0]) * 8u);
                o.vr[0] = r2.x, o.vr[C - 1] = r2.y;
            }
        } else {
#pragma unroll
            for (int c = 0; c < C; ++c) {
                if (in_k) {
                    const unsigned ob = (row + (unsigned)jin[c]) * 8u;
                    o.sg[c] = ld_off(P.sigma, ob);
                    o.va[c] = ld_off(P.vazi, ob);
                    if (ADI)
                        o.en[c] = ld_off(P.energy, ob);
                }
                if (in_v)
                    o.vr[c] = ld_off(P.vrad, (rowv + (unsigned)jin[c]) * 8u);
            }
        }
    };
    // ring k (raw) -> newest window slot; vr_k = v_r(k) from the previous ring's fetch
    double vr_last[C];
    // (r, romega: Rmed[k] and Rmed[k] OmegaFrame, from the caller's batch of per-ring scalars)
    auto convert = [&](int k, const RingRaw &o, double r, double romega) {
        const bool in_k = k >= 0 && k < nr;
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
                w[0][NQ - 1][c] = in_k ? o.en[c] * FAST_RCP_TR(o.sg[c]) : 0.0;
                if (!DIET)
                    er[0][c] = o.en[c];
            }
            if (!DVP)
                vp[0][c] = o.va[c];
            vr_last[c] = o.vr[c];
        }
    };
    // Software pipeline of the memory traffic: ring m+1 is in flight while ring m-2 is computed;
    // at the bottom of an iteration the arrived ring is converted, the loads of ring m+2 are
    // issued, and only then the iteration's stores.  The one s_waitcnt vmcnt(0) per iteration then
    // meets operations that are a whole compute phase old (vmcnt counts stores too; waiting right
    // behind them costs a round trip per ring at 2-3 waves per SIMD).
    // (the three rings that start a chunk are requested together: one memory round trip before the loop, not two)
    RingRaw nxt;
    {
        RingRaw first, second;
        fetch(r0 - 4, first);
        fetch(r0 - 3, second);
        fetch(r0 - 2, nxt);
#pragma unroll
        for (int c = 0; c < C; ++c)
            vr_last[c] = first.vr[c]; // v_r(r0-3)
        const ThetaRow t3 = crow_load(P.theta_tab, r0 - 3 >= 0 ? r0 - 3 : 0);
        convert(r0 - 3, second, t3.rmed, t3.r_omega);
    }
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
        // ... and of ring m+1, which the bottom of this iteration turns into specific quantities
        typedef const double __attribute__((address_space(4))) *cdptr;
        const cdptr tn = (cdptr)(const void *)(P.theta_tab + (m + 1 >= 0 && m + 1 < nr ? m + 1 : 0));
        const double r_next = tn[offsetof(ThetaRow, rmed) / sizeof(double)], romega_next = tn[offsetof(ThetaRow, r_omega) / sizeof(double)];
        // ---- R: slopes of ring m-1, fluxes through interface k = m-1 --------------------------
        // (the first iterations of a chunk only fill the window: the first slope that reaches a result is that of ring
        //  r0-2 -- as hs1 of the flux through interface r0-1 -- formed at m = r0-1 from the differences of rings r0-3 ..
        //  r0-1; the kernels that carry the previous difference instead of re-forming it need it from m = r0-2)
        double F0[NQ][C];
#pragma unroll
        for (int q = 0; q < NQ; ++q)
#pragma unroll
            for (int c = 0; c < C; ++c)
                F0[q][c] = 0.0;
        if (m >= r0 - (DIET ? 1 : 2)) {
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
                    const double dprev = DIET ? (w[1][q][c] - w[2][q][c]) * idr_prev : d1[q][c];
                    const double hs0 = lim_ok ? half_limiter(lim, d0, dprev) : 0.0; // ring m-1
                    // (both candidates with the lane's own distance, then one select: see theta_star)
                    const double st_up = w[2][q][c] + dist[c] * hs1[q][c], st_dn = w[1][q][c] + dist[c] * hs0;
                    const double st = up[c] ? st_up : st_dn;
                    if (q == 0) {
                        Fc[c] = open ? g * st * w[1][2][c] : 0.0; // mass flux g rho* v
                        F0[q][c] = Fc[c];
                    } else {
                        F0[q][c] = st * Fc[c];
                    }
                    if (!DIET)
                        d1[q][c] = d0;
                    hs1[q][c] = hs0;
                }
            }
        }
        // ---- update of ring i = m-2, azimuthal passes, velocities -----------------------------
        bool out_on = false, out_pair = false;
        unsigned out_g[C]; // byte offsets of the cells this lane stores
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
                E[c] = ADI ? (DIET ? s0 * w[2][NQ - 1][c] : er[2][c]) + (F1[NQ - 1][c] - F0[NQ - 1][c]) * invsurf : 0.0;
                V[c] = vadd + ((DVP ? w[2][4][c] * ti.invr - ti.r_omega : vp[2][c]) - mean);
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
                const unsigned row = (unsigned)i * (unsigned)nphi;
                int jout[C];
#pragma unroll
                for (int c = 0; c < C; ++c) {
                    int jo = jin[c] + ns;
                    jout[c] = jo >= nphi ? jo - nphi : jo;
                    const double lpm = c == 0 ? lp_l : Q[2][c == 0 ? 0 : c - 1];
                    const double sm = c == 0 ? s_l : S[c == 0 ? 0 : c - 1];
                    double vr = 0.0;
                    if (i != 0)
                        vr = (rp[c] + Q[1][c]) * FAST_RCP_TR(sp[c] + S[c]);
                    double va = (lpm + Q[3][c]) * FAST_RCP_TR(sm + S[c]) * invr - romega;
                    double sf = S[c] < P.sigma_floor_abs ? P.sigma_floor_abs : S[c];
                    double e = ADI ? clamp_energy_fast(P, E[c], sf) : 0.0;
                    const unsigned g = (row + (unsigned)jout[c]) * 8u;
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
            if (!DIET)
                er[2][c] = er[1][c], er[1][c] = er[0][c];
            if (!DVP)
                vp[2][c] = vp[1][c], vp[1][c] = vp[0][c];
        }
        if (DIET)
            idr_prev = rk.idr_up;
        if (m < r1 + 1) {
            convert(m + 1, nxt, r_next, romega_next);
            // The loads of fetch() must go straight into the registers convert() has just read.  Left alone, the
            // compiler sinks convert() below them, loads into fresh registers and copies those back behind an
            // s_waitcnt vmcnt(0): the prefetch becomes a blocking load, once per ring.  The empty asm pins the
            // results of convert() above it and (memory clobber) the loads below it.
            if (PIN) {
#pragma unroll
                for (int c = 0; c < C; ++c) {
#pragma unroll
                    for (int q = 0; q < NQ; ++q)
                        asm volatile("" : "+v"(w[0][q][c]));
                    asm volatile("" : "+v"(vr_last[c]));
                }
                asm volatile("" ::: "memory");
            }
            if (m < r1)
                fetch(m + 2, nxt);
        }
        if (ADI && THERM && out_on) { // the cell-local CFL terms of the new state (cfl_thermal_term)
            const ThermalRing tr = thermal_ring(W, i);
#pragma unroll
            for (int c = 0; c < C; ++c)
                if (valid[c])
                    st_off(W.cfl_thermal, out_g[c], cfl_thermal_term(W, tr, o_s[c], o_e[c], ld_off(W.qplus, out_g[c]), ld_off(W.qminus, out_g[c])));
        }
        if (out_on) {
            if (out_pair) {
                if (valid[0]) {
                    st2_off(W.vrad, out_g[0], (D2{o_vr[0], o_vr[C - 1]}));
                    st2_off(W.vazi, out_g[0], (D2{o_va[0], o_va[C - 1]}));
                    st2_off(W.sigma, out_g[0], (D2{o_s[0], o_s[C - 1]}));
                    if (ADI)
                        st2_off(W.energy, out_g[0], (D2{o_e[0], o_e[C - 1]}));
                }
            } else {
#pragma unroll
                for (int c = 0; c < C; ++c)
                    if (valid[c]) {
                        st_off(W.vrad, out_g[c], o_vr[c]);
                        st_off(W.vazi, out_g[c], o_va[c]);
                        st_off(W.sigma, out_g[c], o_s[c]);
                        if (ADI)
                            st_off(W.energy, out_g[c], o_e[c]);
                    }
            }
            if (i == nr - 1) { // v_r row Nr is neither transported nor shifted: copied column by column
#pragma unroll
                for (int c = 0; c < C; ++c)
                    if (valid[c]) {
                        const unsigned gt = ((unsigned)nr * (unsigned)nphi + (unsigned)jin[c]) * 8u;
                        double v = ld_off(P.vrad, gt);
                        if (DAMP) {
                            const DampRow dn = crow_load(W.damp_tab, nr);
                            v = damp_apply(v, dn.tvr, si.ev_top, W.vrad0, gt, 0.0);
                        }
                        st_off(W.vrad, gt, v);
                    }
            }
        }
    }
#ifdef TF_TRACE
    if (lane == 0)
        W.temperature[4 * trace_slot + 1] = (double)wall_clock64();
#endif
}

// The kernels proper.  One cell per lane: the register allocator is told to aim for 4 wavefronts per SIMD (<= 128
// VGPRs, no scratch in any instantiation; left alone the ideal-EOS one with the CFL terms settles at 136 = 3
// wavefronts).  Two cells per lane (the tuning variant transport_fused = 2, 230-290 VGPRs) keeps its natural allocation.
template <int C, bool ADI, bool DAMP, int LIM>
__global__ void __launch_bounds__(256) __attribute__((amdgpu_waves_per_eu(4, 4)))
k_transport_fused(const Dev P, const Dev W, int tiles, int rows, int has_fallback, const TfChunks ch)
{
    static_assert(C == 1, "the 4-wavefront kernel is the one-cell-per-lane form");
    transport_fused_body<C, ADI, DAMP, LIM, false>(P, W, tiles, rows, has_fallback, ch);
}
template <int C, bool ADI, bool DAMP, int LIM>
__global__ void __launch_bounds__(256) __attribute__((amdgpu_waves_per_eu(4, 4)))
k_transport_fused_therm(const Dev P, const Dev W, int tiles, int rows, int has_fallback, const TfChunks ch)
{
    static_assert(C == 1 && ADI, "the cell-local CFL terms belong to the energy equation");
    transport_fused_body<C, ADI, DAMP, LIM, true>(P, W, tiles, rows, has_fallback, ch);
}
template <int C, bool ADI, bool DAMP, int LIM>
__global__ void __launch_bounds__(256) k_transport_fused_wide(const Dev P, const Dev W, int tiles, int rows, int has_fallback,
                                                              const TfChunks ch)
{
    transport_fused_body<C, ADI, DAMP, LIM, false>(P, W, tiles, rows, has_fallback, ch);
}
