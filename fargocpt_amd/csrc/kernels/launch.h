// Part of fcpt_kernels.hip (one translation unit, namespace fcpt): kernel names, per-kernel HIP-event profiler, launchers.
// Not a stand-alone header: included once, in the order given there.

// ---------------------------------------------------------------------------
// per-kernel HIP-event timing (fcpt_profile_start/stop)
const char *const kKernelNames[KID_COUNT] = {
    "k_potential", "k_source_vr", "k_source_va", "k_compression_heating", "k_tw_q", "k_tw_va", "k_tw_vr",
    "k_sn_q", "k_sn_e", "k_sn_vr", "k_sn_va", "k_temperature_range", "k_adi_derived", "k_iso_cs_h",
    "k_viscosity", "k_pressure", "k_temperature", "k_stress_diag", "k_stress_rphi", "k_visc_va",
    "k_visc_vr", "k_qplus_qminus", "k_substep3", "k_boundary", "k_damping", "k_transport_radial",
    "k_ring_mean", "k_transport_theta<1>", "k_transport_theta<2>", "k_velocities", "k_cfl_final",
    "k_cfl_cells", "k_clock", "k_src_fused", "k_av_fused", "k_visc_fused", "k_source_march",
    "k_transport_theta_march", "k_transport_fused", "k_massflow", "k_cfl_rings", "k_theta_march_gated_boundary",
    "k_exchange_copy", "k_disk_on_body", "k_visc_factors", "k_source_march_adi", "k_source_march_adi_wide",
    "k_transport_fused_therm", "k_transport_fused_wide", "k_step_coop", "k_accel_on_gas", "k_source_march_adi_acc",
    "k_transport_radial_means", "k_cfl_rings_bc"};

thread_local Profiler *g_prof = nullptr;

void Profiler::begin(int id, hipStream_t st)
{
    if (!((mask >> id) & 1ull) || used + 2 > (int)events.size())
        return;
    if (stride > 1 && (seen++ % stride) != 0)
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
void launch_accel_on_gas(const Dev &P, hipStream_t st) { LAUNCH2D(KID_ACCEL_ON_GAS, k_accel_on_gas, P.nr - 1, P); }
void launch_body_force(const Dev &P, hipStream_t st)
{
    if (P.accel_force)
        launch_accel_on_gas(P, st);
    else
        launch_potential(P, st);
}

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
    // recalculate_viscosity, SourceEuler.cpp:205-223 (AspectRatioMode 0): c_s, H (and nu when alpha) in one launch;
    // the isothermal alpha-nu never changes after init
    if (P.adiabatic)
        LAUNCH2D(KID_ADI_CS_H, k_adi_derived, P.nr, P, 0);
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
// Rings per marching chunk.  A marching wavefront is a serial chain of (rows + pre-roll) ring iterations, and the
// GPU holds a fixed number of them at a time (CUs x 4 SIMDs x the kernel's wavefronts per SIMD).  What a launch costs
// is the number of ROUNDS of resident wavefronts -- an integer -- times the length of the chain: the chunk length
// that minimises (rows + pre-roll) x rounds is taken.  Measured (round 2; 2048 rings unless noted):
//   source march, ideal EOS, Nphi = 4096: 36 rings (3 990 wavefronts, one round) 0.5277 ms per step, 24 rings (two
//     rounds, the second 47 % full) 0.5415, 35 rings (4 130 wavefronts: two rounds of long chains) 0.575; Nphi = 6144:
//     54 rings 0.774 against 0.789 at 24; 1024 x 3072: 14 rings 0.2345-0.2362 against 0.2424 at 7;
//   source march, isothermal, Nphi = 6144: 36 rings 0.5104-0.5148 against 0.5303-0.5322 at 24; Nphi = 4096: 24 rings
//     (one round) 0.368, 23 (two) 0.397;
//   transport: see transport_rows().
// Small grids have fewer wavefronts than slots at any length: the shortest chunks (4 rings) win there.
// compute units of the current device (one device per process)
static int g_cus_override = 0; // fcpt_selftest_chunk_tables: the host logic for a device of that many CUs
static int device_cus()
{
    if (g_cus_override > 0)
        return g_cus_override;
    static int n_cu = 0;
    if (!n_cu) {
        int dev = 0, v = 0;
        (void)hipGetDevice(&dev);
        if (hipDeviceGetAttribute(&v, hipDeviceAttributeMultiprocessorCount, dev) != hipSuccess || v <= 0)
            v = 64;
        n_cu = v;
    }
    return n_cu;
}
// Rings per thread (or wavefront) of the marching kernels of the per-loop path (k_transport_radial, k_cfl_cells,
// k_transport_theta_march): a thread that owns `rows` rings is a serial chain of rows (+ pre-roll) dependent
// iterations, which pays off only when there are more cells than the GPU has lanes to put them on.  On the small
// grids of the reference's tests the shortest chain wins (measured, round 3, shock tube 4096 x 4: k_transport_radial
// 31.5 us at 16 rings per thread, k_transport_theta_march 27.6 us at 8 rings per wavefront -- 57 % of a 104 us step of
// 15 launches; profiles/r03_narrow_kernels.txt).
static int march_len(const Dev &P, int rows_full)
{
    const long lanes = (long)device_cus() * 4 * 8 * 64; // every SIMD eight wavefronts deep
    const long cells = (long)P.nr * P.nphi;
    int rows = rows_full;
    while (rows > 1 && cells / rows < lanes)
        rows >>= 1;
    return rows;
}
// wavefronts per SIMD of the source-march instantiation that will run: 6 (isothermal), 4 (with StabilizeViscosity; ideal
// EOS), 2 (ideal EOS with cooling terms or StabilizeViscosity)
static int source_occupancy(const Dev &P)
{
    const bool wide_adi = P.adiabatic && (P.stabilize || P.cooling_surface || P.cooling_beta || P.heating_star || P.accel_force);
    return P.adiabatic ? (wide_adi ? 2 : 4) : (P.stabilize ? 4 : 6);
}
static int source_rows(const Dev &P)
{
    if (P.opt.source_rows > 0)
        return P.opt.source_rows;
    const int segs = (P.nphi + MARCH_VALID - 1) / MARCH_VALID;
    const int occ = source_occupancy(P);
    const long slots = (long)device_cus() * 4 * occ;
    int r = 4;
    long best_cost = 0;
    for (int rows = 4; rows <= 64; ++rows) {
        const long waves = (long)segs * ((P.nr + 1 + rows - 1) / rows);
        const long cost = (rows + 4) * ((waves + slots - 1) / slots);
        if (best_cost == 0 || cost < best_cost) {
            best_cost = cost;
            r = rows;
        }
    }
    // the boundary call folded into the kick needs the last chunk to hold rows nr-2 .. nr: a slightly longer chunk if
    // the division leaves fewer than three rows over
    for (int dr = 0; dr < 4; ++dr) {
        const int chunks = (P.nr + 1 + r + dr - 1) / (r + dr);
        if ((P.nr + 1) - (chunks - 1) * (r + dr) >= 3)
            return r + dr;
    }
    return r;
}
// Rank-matched chunks for the marching source kernels: one entry (segment, first ring, one past the last, 0) per
// wavefront, indexed by blockIdx.x * 4 + wavefront of the block, i.e. in the order of dispatch.
//
// Why: these kernels run as ONE round of wavefronts (the cheapest form: source_rows()), all starting together.  A SIMD
// issues for its oldest wavefront first, and the dispatcher hands every CU of an XCD one workgroup before any gets its
// second: the q-th wavefront an XCD receives sits at rank q / (4 x CUs) of its SIMD and advances at a rate that falls
// with the rank -- the trace of k_source_march_adi at 2048 x 4096 (profiles/r03_sm_wave_trace_ideal_uniform.txt) shows
// the four ranks ending at 125 / 135 / 152 / 172 us of 181, i.e. rates 1 : 0.92 : 0.79 : 0.65 while all four are resident,
// and the GPU a third empty for the last 58 us.  Chunk lengths in proportion to the rate of the rank that will march them
// ((rings + pre-roll) ~ 1 - g rank / (ranks - 1)) end all wavefronts together.
// Every XCD keeps a contiguous eighth of the rings (its L2 serves the shared halo cells); within it every segment
// (column of 59 cells) is cut into as many chunks as fit the XCD's slots once, and the chunks of the columns are dealt
// to the dispatch order in a snake (0 .. segs-1, segs-1 .. 0, ...) so that every column gets nearly the same mix of ranks.
// Empty where equal chunks stay: source_rows > 0, source_graded = 0, more than one round, chains beyond 64 rings.
std::vector<int> source_schedule(const Dev &P)
{
    std::vector<int> out;
    if (P.nphi < 128 || P.opt.source_rows > 0 || P.opt.source_graded == 0)
        return out;
    const int segs = (P.nphi + MARCH_VALID - 1) / MARCH_VALID;
    const int occ = source_occupancy(P);
    const int wpr = device_cus() / 8 * 4; // wavefronts of one rank in an XCD: one per SIMD
    const int rows = P.nr + 1;            // v_r has rows 0 .. nr
    const int PRE = 4;
    if (wpr < 4 || occ < 2 || rows < 64)
        return out;
    int cpc = wpr * occ / segs; // chunks per column in an XCD's eighth of the rings
    if (cpc < 2)
        return out;
    const int rx = rows / 8;
    if (cpc > rx / 6)
        return out; // short chunks: the grid does not fill the slots once (source_rows() picks the shortest equal chunks)
    if ((rx + 1 + cpc - 1) / cpc > 64)
        return out; // one round would be chains longer than any measured: several rounds of equal chunks
    const double g = (P.opt.source_graded > 0 && P.opt.source_graded < 100 ? P.opt.source_graded : (P.adiabatic ? 35 : 20)) * 0.01;
    std::vector<double> w(occ);
    for (int r = 0; r < occ; ++r)
        w[r] = 1.0 - g * r / (occ - 1);
    const int nblk = (cpc * segs + 3) / 4; // workgroups per XCD
    out.assign((size_t)nblk * 8 * 4 * 4, 0);
    for (int x = 0; x < 8; ++x) {
        const int A = (int)((long)x * rows / 8), B = (int)((long)(x + 1) * rows / 8), n = B - A;
        for (int c = 0; c < segs; ++c) {
            double sw = 0.0;
            for (int j = 0; j < cpc; ++j) {
                const int q = j * segs + ((j & 1) ? segs - 1 - c : c);
                sw += w[q / wpr < occ ? q / wpr : occ - 1];
            }
            const double scale = (n + (double)cpc * PRE) / sw;
            double edge = 0.0;
            int k0 = A;
            for (int j = 0; j < cpc; ++j) {
                const int q = j * segs + ((j & 1) ? segs - 1 - c : c);
                edge += scale * w[q / wpr < occ ? q / wpr : occ - 1] - PRE;
                int k1 = j == cpc - 1 ? B : A + (int)(edge + 0.5);
                if (k1 < k0 + 3 || k1 > B) { // (cannot happen with the bounds above; equal chunks rather than a wrong table)
                    out.clear();
                    return out;
                }
                const size_t t = ((size_t)(q / 4) * 8 + x) * 4 + (q & 3);
                out[4 * t] = c, out[4 * t + 1] = k0, out[4 * t + 2] = k1;
                k0 = k1;
            }
        }
    }
    return out;
}
// Chunks of graded length for k_transport_fused, in dispatch order: (first ring, one past the last) pairs.
//
// Why: the wavefront trace of the kernel (profiles/tools/wave_trace_transport.py, profiles/r03_tf_wave_trace_uniform.txt) shows
// equal chunks leaving a long tail.  At 2048 x 4096 the 8 034 wavefronts of 103 twenty-ring chunks take two rounds of
// the 4 096 slots; a SIMD issues for its OLDEST wavefront first, so the four wavefronts of a SIMD finish 58 ... 95 us
// after a common start, the second round starts staggered over 40 us and ends staggered over 58 us, during which the
// GPU holds 1 900 wavefronts on average: 188 us for 150 us of full-occupancy work.  Long chunks first and ever shorter
// ones behind them (guided self-scheduling) let the slots run dry together: the last wavefronts a slot receives are
// short, and their pre-roll (4 cheap + 1 full iteration per chunk) is paid on a small share of the rings only.
//
// Three lengths (see the body for the numbers).  Chunks are taken from both ends of the
// slab alternately (the damping zones -- costlier rings, `slow` = 1 -- sit at the ends and start first, and they
// count 1.4 rings each).  Returns an empty vector where equal chunks stay: tuning runs (transport_rows > 0,
// transport_graded = 0) and grids whose wavefronts fit the slots once (the shortest chunks win there: transport_rows()).
static int transport_rows(const Dev &P);
static std::vector<int> transport_chunk_list(const Dev &P, const std::vector<int> &slow, const std::vector<int> *lengths)
{
    std::vector<int> out;
    if (P.nphi < 256 || P.opt.transport_rows > 0 || P.opt.transport_graded == 0)
        return out;
    if (P.opt.transport_fused == 0 || P.opt.transport_fused == 2)
        return out;
    const int tstride = 64 - (TfHalo<1>::lo + TfHalo<1>::hi);
    const long tiles = (P.nphi + tstride - 1) / tstride;
    const long slots = (long)device_cus() * 4 * 4; // 4 wavefronts per SIMD (128 VGPRs)
    const double conc = (double)slots / (double)tiles; // chunks resident at once
    const bool explicit_spec = lengths && !lengths->empty(); // fcpt_set_transport_chunks / FCPT_TF_SCHEDULE: tuning runs and tests
    if (!explicit_spec) {
        const int rows_u = transport_rows(P);
        if ((long)((P.nr + rows_u - 1) / rows_u) * tiles <= slots || conc < 16.0)
            return out; // equal chunks need one round only / rings so long that a few chunks fill an XCD
    }
    const int COST = 10, COST_SLOW = 14; // tenths of a ring
    long total = 0;
    for (int i = 0; i < P.nr; ++i)
        total += (i < (int)slow.size() && slow[i]) ? COST_SLOW : COST;
    // lengths in dispatch order, in rings of cost
    std::vector<int> len;
    if (explicit_spec) {
        len = *lengths;
    } else {
        // level 0: the same number of chunks for every XCD (they are dealt round-robin), enough of them to fill the
        // XCD's slots once; 74 % of the cost there, then three chunks per XCD of 0.43 of that length, the rest at 0.21
        // (measured at 2048 x 4096, 78 tiles, 512 slots per XCD: 28 x 56, 12 x 24, 6 ...: profiles/r03_tf_schedule_sweep.txt)
        const long slots_xcd = slots / 8;
        const int k0 = (int)((slots_xcd + tiles - 1) / tiles);
        const int n0 = 8 * k0;
        int big = P.opt.transport_big > 0 ? P.opt.transport_big : (int)(0.74 * (double)total / COST / n0 + 0.5);
        big = big < 4 ? 4 : big;
        const double ladder = (P.opt.transport_ladder > 0 && P.opt.transport_ladder <= 100 ? P.opt.transport_ladder : 43) * 0.01;
        const int n1 = 8 * ((int)(0.43 * k0 + 0.5) < 1 ? 1 : (int)(0.43 * k0 + 0.5));
        const int l1 = (int)(big * ladder + 0.5) < 4 ? 4 : (int)(big * ladder + 0.5);
        const int l2 = (int)(big * ladder * 0.5 + 0.5) < 4 ? 4 : (int)(big * ladder * 0.5 + 0.5);
        for (int k = 0; k < n0; ++k)
            len.push_back(big);
        for (int k = 0; k < n1; ++k)
            len.push_back(l1);
        len.push_back(l2); // ... repeated to the end
    }
    int lo = 0, hi = P.nr;
    for (size_t k = 0; lo < hi; ++k) {
        const int lk = len[k < len.size() ? k : len.size() - 1];
        const long target = (long)(lk < 1 ? 1 : lk) * COST;
        long cost = 0;
        if ((k & 1) == 0) {
            const int r0 = lo;
            while (lo < hi && cost < target)
                cost += (lo < (int)slow.size() && slow[lo]) ? COST_SLOW : COST, ++lo;
            if (hi - lo < 3) // no crumbs
                lo = hi;
            out.push_back(r0), out.push_back(lo);
        } else {
            const int r1 = hi;
            while (lo < hi && cost < target)
                cost += (hi - 1 < (int)slow.size() && slow[hi - 1]) ? COST_SLOW : COST, --hi;
            if (hi - lo < 3)
                hi = lo;
            out.push_back(hi), out.push_back(r1);
        }
    }
    return out;
}
// One round of wavefronts (grids whose equal chunks fit the slots once): chunk lengths matched to the SIMD rank of the
// wavefront, exactly as source_schedule() does for the source marches -- the trace of the 1024 x 3072 transport
// (profiles/r03_tf_wave_trace_config3_uniform.txt) shows all 3 712 wavefronts resident for 50 us and then leaving over
// the next 43.  Entries (tile, first ring, one past the last, 0) indexed by blockIdx.x * 4 + wavefront of the workgroup.
static std::vector<int> transport_rank_table(const Dev &P, const std::vector<int> &slow)
{
    std::vector<int> out;
    if (P.opt.transport_rank_grade == 0)
        return out;
    const int tstride = 64 - (TfHalo<1>::lo + TfHalo<1>::hi);
    const int tiles = (P.nphi + tstride - 1) / tstride;
    const int occ = 4, PRE = 5;
    const int wpr = device_cus() / 8 * 4; // wavefronts of one rank in an XCD: one per SIMD
    const int rows = P.nr;
    if (wpr < 4 || rows < 128)
        return out;
    const int cpc = wpr * occ / tiles; // chunks per tile column in an XCD's eighth of the rings
    const int rx = rows / 8;
    if (cpc < 2 || rx / cpc < 10)
        return out; // short chunks (the grid does not fill the slots with chunks of ten rings): transport_rows()'s equal ones
    if ((rx + cpc) / cpc > 64)
        return out;
    const double g = (P.opt.transport_rank_grade > 0 && P.opt.transport_rank_grade < 100 ? P.opt.transport_rank_grade : (P.adiabatic ? 45 : 60)) * 0.01;
    double w[4];
    for (int r = 0; r < occ; ++r)
        w[r] = 1.0 - g * r / (occ - 1);
    // cost of the rings: a damping-zone ring (reference values loaded and waited for) counts 1.4 -- the XCDs get equal
    // cost, not equal numbers of rings (the zones sit in the first and the last XCD's range), and so do the chunks
    std::vector<double> cum(rows + 1, 0.0);
    for (int i = 0; i < rows; ++i)
        cum[i + 1] = cum[i] + ((i < (int)slow.size() && slow[i]) ? 1.4 : 1.0);
    auto ring_at = [&](double cost) { // first ring index whose cumulated cost reaches `cost`
        int lo = 0, hi = rows;
        while (lo < hi) {
            const int mid = (lo + hi) / 2;
            if (cum[mid] < cost)
                lo = mid + 1;
            else
                hi = mid;
        }
        return lo;
    };
    const int nblk = (cpc * tiles + 3) / 4;
    out.assign((size_t)nblk * 8 * 4 * 4, 0);
    for (int x = 0; x < 8; ++x) {
        const int A = x == 0 ? 0 : ring_at(cum[rows] * x / 8.0), B = x == 7 ? rows : ring_at(cum[rows] * (x + 1) / 8.0);
        const double n = cum[B] - cum[A];
        for (int c = 0; c < tiles; ++c) {
            double sw = 0.0;
            for (int j = 0; j < cpc; ++j) {
                const int q = j * tiles + ((j & 1) ? tiles - 1 - c : c);
                sw += w[q / wpr < occ ? q / wpr : occ - 1];
            }
            const double scale = (n + (double)cpc * PRE) / sw;
            double edge = 0.0;
            int k0 = A;
            for (int j = 0; j < cpc; ++j) {
                const int q = j * tiles + ((j & 1) ? tiles - 1 - c : c);
                edge += scale * w[q / wpr < occ ? q / wpr : occ - 1] - PRE;
                const int k1 = j == cpc - 1 ? B : ring_at(cum[A] + edge);
                if (k1 < k0 + 2 || k1 > B) {
                    out.clear();
                    return out;
                }
                const size_t t = ((size_t)(q / 4) * 8 + x) * 4 + (q & 3);
                out[4 * t] = c, out[4 * t + 1] = k0, out[4 * t + 2] = k1;
                k0 = k1;
            }
        }
    }
    return out;
}
// The table k_transport_fused runs from: per wavefront (tile, first ring, one past the last, 0) in the order of
// dispatch -- graded chunks (several rounds of wavefronts: transport_chunk_list, every chunk's tiles side by side on one
// XCD), rank-matched chunks (one round: transport_rank_table), or empty: equal chunks of transport_rows() rings.
std::vector<int> transport_schedule(const Dev &P, const std::vector<int> &slow, const std::vector<int> *lengths)
{
    std::vector<int> out;
    if (P.nphi < 256 || P.opt.transport_rows > 0 || P.opt.transport_fused == 0 || P.opt.transport_fused == 2)
        return out;
    const std::vector<int> chunks = transport_chunk_list(P, slow, lengths);
    if (chunks.empty()) {
        const bool explicit_spec = lengths && !lengths->empty();
        if (explicit_spec || P.opt.transport_graded == 0)
            return out;
        // one round of equal chunks?
        const int tstride = 64 - (TfHalo<1>::lo + TfHalo<1>::hi);
        const long tiles = (P.nphi + tstride - 1) / tstride;
        const int rows_u = transport_rows(P);
        if ((long)((P.nr + rows_u - 1) / rows_u) * tiles > (long)device_cus() * 4 * 4)
            return out;
        return transport_rank_table(P, slow);
    }
    const int tstride = 64 - (TfHalo<1>::lo + TfHalo<1>::hi);
    const int tiles = (P.nphi + tstride - 1) / tstride;
    const int count = (int)(chunks.size() / 2);
    // as the kernel deals equal chunks: workgroup b runs on XCD b % 8; chunk c on XCD c % 8, its tiles side by side
    const int nblk = 8 * ((((count + 7) / 8) * tiles + 3) / 4);
    out.assign((size_t)nblk * 4 * 4, 0);
    for (int b = 0; b < nblk; ++b)
        for (int wv = 0; wv < 4; ++wv) {
            const int xcd = b & 7, wq = (b >> 3) * 4 + wv, zq = wq / tiles, c = xcd + 8 * zq;
            if (c >= count)
                continue;
            const size_t t = (size_t)b * 4 + wv;
            out[4 * t] = wq - zq * tiles, out[4 * t + 1] = chunks[2 * c], out[4 * t + 2] = chunks[2 * c + 1];
        }
    return out;
}
// test hook (no GPU needed): the two tables for a grid, an EOS and a device of n_cu compute units; the first
// damp_inner and the last damp_outer rings load reference values in the transport (damping zones)
void selftest_chunk_tables(int nr, int nphi, int n_cu, int adiabatic, int damp_inner, int damp_outer, const Options &opt,
                           std::vector<int> &transport, std::vector<int> &source)
{
    Dev P;
    std::memset(&P, 0, sizeof(P));
    P.nr = nr, P.nphi = nphi, P.adiabatic = adiabatic, P.opt = opt;
    P.damp_in_step = (damp_inner > 0 || damp_outer > 0) ? 1 : 0;
    std::vector<int> slow(nr > 0 ? nr : 0, 0);
    for (int i = 0; i < nr; ++i)
        slow[i] = (i < damp_inner || i >= nr - damp_outer) ? 1 : 0;
    g_cus_override = n_cu;
    transport = transport_schedule(P, slow, nullptr);
    source = source_schedule(P);
    g_cus_override = 0;
}
// The transport deals whole chunks to the 8 XCDs (k_transport_fused), so the rounds are counted per XCD; and its
// chunks are not equal: the rings of the damping zones (folded into the kernel) cost ~1.5x and are started first, which
// adds half a round to the last one.  cost = (rows + 5) x (rounds - 1 + slow).  Measured: 2048 x 4096 (78 tiles): 20
// rings (13 chunks per XCD, 1 014 wavefronts for 512 slots: 2 rounds) 0.363 ms per step; 18 (15 chunks: 3 rounds) 0.377;
// 24 (2 rounds of longer chains) 0.367-0.371; 40 (7 chunks on some XCDs = 546 wavefronts: 2 rounds of 45) 0.41;
// 1024 x 3072 ideal (58 tiles): 16 rings (8 chunks per XCD, one round) 0.2374-0.2383 against 0.2434-0.2448 at 8
// and 0.250 at 14 (10 chunks per XCD: 580 wavefronts, two rounds).
static int transport_rows(const Dev &P)
{
    if (P.opt.transport_rows > 0)
        return P.opt.transport_rows;
    const int CF = 1; // cells per lane of the default kernel
    const int tstride = 64 * CF - (TfHalo<1>::lo + TfHalo<1>::hi);
    const long tiles = (P.nphi + tstride - 1) / tstride;
    const long slots_xcd = (long)device_cus() / 8 * 4 * 4; // 4 wavefronts per SIMD (128 VGPRs)
    const double slow = P.damp_in_step ? 1.5 : 1.0;
    int r = 4;
    double best_cost = 0.0;
    for (int rows = 4; rows <= 32; ++rows) { // (longer single-round chains are unmeasured)
        const long chunks = (P.nr + rows - 1) / rows;
        // (launches of fewer than TF_XCD_CHUNKS chunks deal workgroups, not chunks: all wavefronts over all slots)
        const long rounds = chunks >= TF_XCD_CHUNKS ? (((chunks + 7) / 8) * tiles + slots_xcd - 1) / slots_xcd
                                                    : (chunks * tiles + 8 * slots_xcd - 1) / (8 * slots_xcd);
        const double cost = (rows + 5) * (rounds - 1 + slow);
        if (best_cost == 0.0 || cost < best_cost * (1.0 - 1e-12)) {
            best_cost = cost;
            r = rows;
        }
    }
    return r;
}
// whole source step in one marching pass (Nphi >= 128); returns 0 if not applicable, else +-segments (> 0: ring sums
// of v_phi were left for the transport).  fold_bc: the caller's next call is apply_boundary_condition(final = false) on
// the kick's result -- *bc_folded reports whether the kernel applied it itself (boundary_column on its edge chunks)
// will launch_source_march() take the step?
bool source_march_applies(const Dev &P)
{
    if (P.nphi < 128)
        return false;
    if ((long long)(P.nr + 1) * P.nphi >= (1ll << 29))
        return false; // the kernels address cells by 32-bit byte offsets (ld_off): grids below 4 GiB
    return !P.adiabatic || P.opt.march_source_adi != 0;
}
// fold_cfl: the launch stands directly behind the ring kernel of the CFL reduction (launch_cfl / launch_cfl_bc with
// apply_policy = 2): its workgroups fold the reduction and apply the time-step policy themselves (cfl_fold_in_step)
int launch_source_march(const Dev &P, hipStream_t st, bool fold_bc, bool *bc_folded, bool fold_cfl)
{
    if (bc_folded)
        *bc_folded = false;
    if (!source_march_applies(P))
        return 0;
    int bc_fold = 0;
    const bool sched = P.sm_sched_n > 0 && P.opt.source_rows <= 0; // rank-matched chunks (every one of them >= 3 rings)
    {
        // the boundary conditions read rows 1, 2 and nr-2 .. nr of the kick's result: the wavefront that applies them
        // must have stored those rows itself
        const int rows = sched ? 3 : source_rows(P), chunks = (P.nr + 1 + rows - 1) / rows;
        const int last_rows = sched ? 3 : (P.nr + 1) - (chunks - 1) * rows;
        if (fold_bc && P.opt.bc_fold != 0 && rows >= 3 && last_rows >= 3 && P.nr >= 6 && (!P.adiabatic || P.opt.march_source_adi != 0)) {
            bc_fold = 1;
            if (bc_folded)
                *bc_folded = true;
        }
    }
    if (fold_cfl)
        bc_fold |= 2;
    if (P.adiabatic) {
        const int rows = source_rows(P);
        const int segs = (P.nphi + MARCH_VALID - 1) / MARCH_VALID;
        const int chunks = (P.nr + 1 + rows - 1) / rows;
        const dim3 grid(sched ? P.sm_sched_n / 4 : (segs * chunks + 3) / 4), block(256);
        const bool cool = P.cooling_surface != 0 || P.cooling_beta != 0 || P.heating_star != 0;
        const int ring_sums = segs <= P.ring_pstride && P.opt.source_ring_parts != 0;
#define ADIKS(AV_, COOL_, POT_)                                                                                               \
    if (P.stabilize)                                                                                                          \
        KLAUNCH(KID_SOURCE_MARCH_ADI_WIDE, (k_source_march_adi_wide<AV_, COOL_, POT_, true>), grid, block, P, segs, rows, ring_sums, bc_fold); \
    else                                                                                                                      \
        KLAUNCH(KID_SOURCE_MARCH_ADI_WIDE, (k_source_march_adi_wide<AV_, COOL_, POT_, false>), grid, block, P, segs, rows, ring_sums, bc_fold)
#define ADIKP(AV_, POT_)                                                                                    \
    if (cool) {                                                                                             \
        ADIKS(AV_, true, POT_);                                                                             \
    } else if (P.stabilize) {                                                                               \
        ADIKS(AV_, false, POT_);                                                                            \
    } else {                                                                                                \
        KLAUNCH(KID_SOURCE_MARCH_ADI, (k_source_march_adi<AV_, POT_>), grid, block, P, segs, rows, ring_sums, bc_fold); \
    }
#define ADIKA(AV_, COOL_)                                                                                                  \
    if (P.stabilize)                                                                                                       \
        KLAUNCH(KID_SOURCE_MARCH_ADI_ACC, (k_source_march_adi_acc<AV_, COOL_, true>), grid, block, P, segs, rows, ring_sums, bc_fold); \
    else                                                                                                                   \
        KLAUNCH(KID_SOURCE_MARCH_ADI_ACC, (k_source_march_adi_acc<AV_, COOL_, false>), grid, block, P, segs, rows, ring_sums, bc_fold)
#define ADIK(AV_)                 \
    if (P.accel_force) {          \
        if (cool) {               \
            ADIKA(AV_, true);     \
        } else {                  \
            ADIKA(AV_, false);    \
        }                         \
    } else if (P.inline_potential) { \
        ADIKP(AV_, true);         \
    } else {                      \
        ADIKP(AV_, false);        \
    }
        if (P.art_visc == FCPT_ARTVISC_TW) {
            ADIK(1);
        } else if (P.art_visc == FCPT_ARTVISC_SN) {
            ADIK(2);
        } else {
            ADIK(0);
        }
#undef ADIK
#undef ADIKA
#undef ADIKP
#undef ADIKS
        return ring_sums ? segs : -segs; // < 0: marched, but no ring sums
    }
    // measured at 2048x4096: 16 / 24 / 32 / 48 / 64 rings -> 0.133 / 0.132 / 0.141 / 0.152 / 0.188 ms
    const int rows = source_rows(P);
    const int segs = (P.nphi + MARCH_VALID - 1) / MARCH_VALID;
    // per-segment ring sums of v_phi, so that the transport's k_ring_mean reads 70 partials per ring
    // instead of the ring itself
    const int ring_sums = segs <= P.ring_pstride && P.opt.source_ring_parts != 0;
    const int chunks = (P.nr + 1 + rows - 1) / rows;
    const int waves = sched ? P.sm_sched_n : segs * chunks;
    const dim3 grid((waves + 3) / 4), block(256);
#define ISOKA(AV_, ACC_)                                                                                    \
    if (P.stabilize)                                                                                        \
        KLAUNCH(KID_SOURCE_MARCH, (k_source_march<AV_, true, ACC_>), grid, block, P, segs, rows, ring_sums, bc_fold); \
    else                                                                                                    \
        KLAUNCH(KID_SOURCE_MARCH, (k_source_march<AV_, false, ACC_>), grid, block, P, segs, rows, ring_sums, bc_fold)
#define ISOK(AV_)          \
    if (P.accel_force) {   \
        ISOKA(AV_, true);  \
    } else {               \
        ISOKA(AV_, false); \
    }
    if (P.art_visc == FCPT_ARTVISC_TW) {
        ISOK(1);
    } else if (P.art_visc == FCPT_ARTVISC_SN) {
        ISOK(2);
    } else {
        ISOK(0);
    }
#undef ISOK
#undef ISOKA
    return ring_sums ? segs : -segs; // < 0: marched, but no ring sums
}
// stress tensor + viscous update, and for the energy equation viscous heating and -- cell-local, on the Q+ just
// formed -- SubStep3 (SourceEuler.cpp:956-1051) with the temperature floor / ceiling behind it (the TEMPERATURE grid
// the reference refreshes there is read by nothing before recalculate_derived_disk_quantities rewrites it)
void launch_viscous_fused(const Dev &P, hipStream_t st) { LAUNCH2D(KID_VISC_FUSED, k_visc_fused, P.nr + 1, P); }

// viscosity.cpp:256-348: the correction factors depend on nu and Sigma only
void launch_visc_factors(const Dev &P, hipStream_t st)
{
    if (P.stabilize)
        LAUNCH2D(KID_VISC_FACTORS, k_visc_factors, P.nr - 1, P);
}
void launch_stress(const Dev &P, hipStream_t st)
{
    LAUNCH2D(KID_STRESS_DIAG, k_stress_diag, P.nr, P);
    LAUNCH2D(KID_STRESS_RPHI, k_stress_rphi, P.nr - 1, P);
    launch_visc_factors(P, st);
}

void launch_viscous_update(const Dev &P, hipStream_t st)
{
    LAUNCH2D(KID_VISC_VA, k_visc_va, P.nr - 2, P);
    LAUNCH2D(KID_VISC_VR, k_visc_vr, P.maxmo_no_ghost_vr - P.one_no_ghost_vr, P);
}

void launch_substep3_cooling_only(const Dev &P, hipStream_t st)
{
    // compute_heating_cooling_for_CFL at init (SourceEuler.cpp:1507-1547): Q+ = 0 (gas at rest), Q- / alpha
    LAUNCH2D(KID_SUBSTEP3, k_substep3, P.nr, P, 0);
}

void launch_substep3(const Dev &P, int update_energy, hipStream_t st)
{
    // SubStep3, SourceEuler.cpp:956-1051 (update_energy = 1) or the Q+/Q- part of
    // compute_heating_cooling_for_CFL, :1507-1547 (update_energy = 0)
    LAUNCH2D(KID_QPLUS, k_qplus_qminus, P.nr, P);
    LAUNCH2D(KID_SUBSTEP3, k_substep3, P.nr, P, update_energy ? 2 : 0);
}

// rows [7,14) and [nr-14,nr-7) -> buffers (unpack = 0), buffers -> rows [0,7) and [nr-7,nr) (unpack = 1)
void launch_exchange_copy(const Dev &P, double *inner, double *outer, int unpack, hipStream_t st)
{
    ExchangeArgs a;
    a.field[0] = P.sigma;
    a.field[1] = P.vrad;
    a.field[2] = P.vazi;
    a.field[3] = P.energy;
    a.buf[0] = inner;
    a.buf[1] = outer;
    a.row0[0] = unpack ? 0 : FCPT_OVERLAP;
    a.row0[1] = unpack ? P.nr - FCPT_OVERLAP : P.nr - 2 * FCPT_OVERLAP;
    a.nq = P.adiabatic ? 4 : 3;
    a.nphi = P.nphi;
    a.unpack = unpack;
    const size_t npair = ((size_t)FCPT_OVERLAP * P.nphi) >> 1;
    int bx = (int)((npair + 255) / 256);
    bx = bx < 1 ? 1 : (bx > 64 ? 64 : bx);
    KLAUNCH(KID_EXCHANGE_COPY, k_exchange_copy, dim3(bx, 2 * a.nq), dim3(256), a);
}

void launch_selftest_half_limiter(int type, long long n, const double *a, const double *b, double *out, hipStream_t st)
{
    hipLaunchKernelGGL(k_selftest_half_limiter, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, st, type, n, a, b, out);
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

#ifndef FALLBACK_BLOCKS
#define FALLBACK_BLOCKS 256 /* grid of the idle in-stream fallback kernels */
#endif
// one radial sweep + ring means (T1-T4); only_if: see k_transport_radial
static void launch_radial(const Dev &P, const int *only_if, hipStream_t st)
{
    const int rows = march_len(P, RADIAL_ROWS);
    const Launch2D l = launch2d((P.nr + rows - 1) / rows, P.nphi);
    const int gx = (int)l.grid.x, gy = (int)l.grid.y;
    const dim3 grid(only_if && gx * gy > FALLBACK_BLOCKS ? FALLBACK_BLOCKS : gx * gy);
    if (l.block.x >= 64)
        KLAUNCH(KID_TRANSPORT_RADIAL, k_transport_radial<true>, grid, l.block, P, only_if, gx, gy, rows);
    else
        KLAUNCH(KID_TRANSPORT_RADIAL, k_transport_radial<false>, grid, l.block, P, only_if, gx, gy, rows);
}
void launch_massflow(const Dev &P, hipStream_t st)
{
    if (P.massflow)
        LAUNCH2D(KID_MASSFLOW, k_massflow, P.nr - 1, P);
}
void launch_shift_means(const Dev &P, hipStream_t st)
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
    const int rows = P.opt.theta_rows > 0 ? P.opt.theta_rows : march_len(P, THETA_ROWS);
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
// the gated azimuthal launch launch_transport(defer_gated) left out -- alone, or with the final boundary call of the step
// on `boundary_view` (the state after the transport's pointer swap) in the same launch (k_theta_march_gated_boundary)
void launch_gated_theta(const GatedTheta &g, const Dev *boundary_view, hipStream_t st)
{
    const Dev &P = g.P, &Wm = g.Wm;
    if (!boundary_view) {
        launch_theta_march(P, Wm, 2, 0, 0, P.shift_jump, st);
        return;
    }
    ThetaSet inB = {P.rmpB, P.rmmB, P.lpB, P.lmB, P.sigB, P.eB};
    const int tstride = 64 * 2 - (THETA_LO + THETA_HI);
    const int tiles = (P.nphi + tstride - 1) / tstride;
    const int rows = P.opt.theta_rows > 0 ? P.opt.theta_rows : march_len(P, THETA_ROWS);
    const int chunks = (P.nr + rows - 1) / rows;
    const int nvb = (chunks * tiles + 3) / 4;
    const int ntheta = nvb > FALLBACK_BLOCKS ? FALLBACK_BLOCKS : nvb;
    const dim3 grid(ntheta + (boundary_view->nphi + 255) / 256), block(256);
#define GTB(AA, DD)                                                                                                       \
    KLAUNCH(KID_THETA_GATED_BOUNDARY, (k_theta_march_gated_boundary<2, AA, DD, false>), grid, block, Wm, (const double *)P.vazi, \
            (const double *)P.vrad, inB, tiles, rows, nvb, ntheta, *boundary_view)
    if (P.adiabatic) {
        if (Wm.damp_in_step) {
            GTB(true, true);
        } else {
            GTB(true, false);
        }
    } else {
        if (Wm.damp_in_step) {
            GTB(false, true);
        } else {
            GTB(false, false);
        }
    }
#undef GTB
}
#undef MARCHC
#undef MARCHK

// the fused kernel runs, nothing is queued behind it, and there are chunks between the two ends
bool transport_can_split(const Dev &P, bool shear_safe)
{
    if (P.nphi < 256 || !shear_safe)
        return false;
    if (P.opt.transport_fallback != 0)
        return false; // the fallback kernels behind the fused one need all of its chunks in one launch
    if (P.opt.transport_fused >= 0 || P.opt.transport_rows > 0)
        return false; // tuning runs keep the one-launch form
    if (P.opt.transport_split == 0)
        return false;
    const int rows = transport_rows(P);
    const int chunks = (P.nr + rows - 1) / rows, c_lo = (P.nr - 2 * FCPT_OVERLAP) / rows;
    const int lead = (2 * FCPT_OVERLAP + rows - 1) / rows; // chunks that hold rows [0, 14)
    return c_lo > lead && c_lo < chunks;
}
// part: TRANSPORT_ALL, or -- for slabs with neighbours, when transport_can_split() -- launch_shift_means, then
// TRANSPORT_INTERIOR on a side stream and TRANSPORT_EDGES (the chunks holding the rings a neighbour receives, rows
// [7,14) and [nr-14,nr-7)) on the caller's stream, so that the ghost exchange runs under the interior chunks.
TransportResult launch_transport(const Dev &P, const Dev &W, hipStream_t st, int part, GatedTheta *defer_gated)
{
    // P: view whose vrad/vazi are the velocities to transport; W: view that receives the new state
    // Transport, TransportEuler.cpp:112-136
    TransportResult res = {0, W.sigma, W.energy, W.vrad, W.vazi, 0, 0, 0};
    // ---- everything in one kernel (tiled rings only) ------------------------------------------
    int CF = P.nphi >= 256 ? 1 : 0; // 1 cell per lane: 3 waves per SIMD (2 cells: 284 VGPRs, 1 wave)
    if (P.opt.transport_fused >= 0) { // 0: off, 1 / 2: cells per lane
        const int v = P.opt.transport_fused;
        CF = v == 0 ? 0 : ((v == 1 || v == 2) && P.nphi >= 128 * v ? v : CF);
    }
    if ((long long)(P.nr + 1) * P.nphi >= (1ll << 29))
        CF = 0; // the fused kernel addresses its grids with 32-bit byte offsets (4 GiB each)
    if (CF) {
        Dev Wm = W; // the marching kernels cannot work in place
        Wm.sigma = W.sigA;
        Wm.energy = W.eA;
        Wm.vrad = P.vrad == W.vrad ? W.vrad_b : W.vrad;
        Wm.vazi = P.vazi == W.vazi ? W.vazi_b : W.vazi;
        if (part == TRANSPORT_ALL)
            launch_shift_means(P, st); // else: the caller queued it ahead of both parts
        const int rows = transport_rows(P);
        const int tstride = 64 * CF - (CF == 2 ? TfHalo<2>::lo + TfHalo<2>::hi : TfHalo<1>::lo + TfHalo<1>::hi);
        const int tiles = (P.nphi + tstride - 1) / tstride;
        const int chunks = (P.nr + rows - 1) / rows;
        // The azimuthal half of the two-kernel transport is always queued behind the fused kernel (one idle launch) and
        // runs only if a ring pair exceeds the one-lane shift.  The CFL condition's shear limit
        // (cfl.cpp:207-220) bounds |Nshift[i] - Nshift[i-1]| for the velocities it saw, but the source step that
        // follows can change v_phi enough to break it in violent flows (the fuzzer found one: an ideal-gas
        // spreading ring), and the reference shifts by any amount.  FCPT_TRANSPORT_FALLBACK=0 drops the launches
        // for flows known to be benign; a violation is then reported as FCPT_ESHEAR.
        const int fallback = P.opt.transport_fallback != 0;
        TfChunks ch = {chunks, chunks, 0, 1, nullptr};
        const bool sched = part == TRANSPORT_ALL && P.tf_sched_n > 0 && P.opt.transport_rows <= 0 && CF == 1;
        if (sched)
            ch = TfChunks{P.tf_sched_n, P.tf_sched_n, 0, 1, P.tf_sched};
        const int c_lo = (P.nr - 2 * FCPT_OVERLAP) / rows;    // first chunk of the outer tail (holds row nr - 14)
        const int lead = (2 * FCPT_OVERLAP + rows - 1) / rows; // chunks that hold rows [0, 14)
        if (part == TRANSPORT_EDGES)
            ch = TfChunks{lead + (chunks - c_lo), lead, c_lo - lead, 1, nullptr};
        else if (part == TRANSPORT_INTERIOR)
            ch = TfChunks{c_lo - lead, 0, lead, 0, nullptr};
        res.split = part != TRANSPORT_ALL;
        // (8 XCDs x the wavefronts of ceil(count / 8) chunks, four to a workgroup: see the chunk mapping in the kernel)
        const dim3 grid(sched ? (ch.count + 3) / 4
                              : (ch.count >= TF_XCD_CHUNKS ? 8 * ((((ch.count + 7) / 8) * tiles + 3) / 4) : (ch.count * tiles + 3) / 4)),
            block(256);
#define TFK2(ID, KK, CC, AA, DD)                                                                    \
    if (P.limiter == FCPT_LIMITER_MC)                                                                \
        KLAUNCH(ID, (KK<CC, AA, DD, FCPT_LIMITER_MC>), grid, block, P, Wm, tiles, rows, fallback, ch); \
    else                                                                                             \
        KLAUNCH(ID, (KK<CC, AA, DD, FCPT_LIMITER_VANLEER>), grid, block, P, Wm, tiles, rows, fallback, ch)
#define TFK(CC, AA, DD)                                \
    if (CC == 1 && AA && Wm.cfl_thermal) {             \
        TFK2(KID_TRANSPORT_FUSED_THERM, k_transport_fused_therm, 1, true, DD); \
    } else if (CC == 1) {                              \
        TFK2(KID_TRANSPORT_FUSED, k_transport_fused, 1, AA, DD); \
    } else {                                           \
        TFK2(KID_TRANSPORT_FUSED_WIDE, k_transport_fused_wide, 2, AA, DD); \
    }
#define TFC(CC)                    \
    if (P.adiabatic) {             \
        if (W.damp_in_step) {      \
            TFK(CC, true, true)    \
        } else {                   \
            TFK(CC, true, false)   \
        }                          \
    } else {                       \
        if (W.damp_in_step) {      \
            TFK(CC, false, true)   \
        } else {                   \
            TFK(CC, false, false)  \
        }                          \
    }
        if (CF == 2) {
            TFC(2)
        } else {
            TFC(1)
        }
#undef TFC
#undef TFK
#undef TFK2
        // behind it, the azimuthal march of the two-kernel form: its blocks return at once unless k_ring_mean met
        // |Nshift[i] - Nshift[i-1]| > 1 (a time step beyond the FARGO shear limit) -- the fused launch then ran the
        // radial sweep.  (Round 2 did both sweeps in this second launch with a hand-rolled grid barrier between them;
        // the flag now being known before the fused launch starts, no barrier is needed.)
        if (fallback && defer_gated && part == TRANSPORT_ALL) { // the caller queues it with the final boundary call (launch_gated_theta)
            defer_gated->P = P;
            defer_gated->Wm = Wm;
            res.gated_pending = 1;
        } else if (fallback) {
            launch_theta_march(P, Wm, 2, 0, 0, P.shift_jump, st);
        }
        res.marched = tiles;
        res.thermal = CF == 1 && P.adiabatic && Wm.cfl_thermal != nullptr;
        res.sigma = Wm.sigma, res.energy = Wm.energy, res.vrad = Wm.vrad, res.vazi = Wm.vazi;
        return res;
    }
    { // radial sweep + ring means (two independent kernels of the reference's sequence) as one launch
        const int rows = march_len(P, RADIAL_ROWS);
        const Launch2D l = launch2d((P.nr + rows - 1) / rows, P.nphi);
        const int gx = (int)l.grid.x, gy = (int)l.grid.y;
        const dim3 grid(gx * gy + (P.nr + 3) / 4);
        const double *part = P.src_ring_nparts ? (const double *)P.ring_part : (const double *)nullptr;
        if (l.block.x >= 64)
            KLAUNCH(KID_TRANSPORT_RADIAL_MEANS, k_transport_radial_means<true>, grid, l.block, P, gx, gy, rows, part, P.src_ring_nparts, P.ring_pstride);
        else
            KLAUNCH(KID_TRANSPORT_RADIAL_MEANS, k_transport_radial_means<false>, grid, l.block, P, gx, gy, rows, part, P.src_ring_nparts, P.ring_pstride);
    }
    ThetaSet inB = {P.rmpB, P.rmmB, P.lpB, P.lmB, P.sigB, P.eB};
    ThetaOut outA = {P.rmpA, P.rmmA, P.lpA, P.lmA, P.sigA, P.eA};
    ThetaSet inA = {P.rmpA, P.rmmA, P.lpA, P.lmA, P.sigA, P.eA};
    ThetaOut outB = {P.rmpB, P.rmmB, P.lpB, P.lmB, P.sigB, P.eB};
    // ring-marching azimuthal kernel when a lane-chunk size fits the ring, else (and with FCPT_THETA_MARCH=0 or
    // FCPT_THETA_FUSED=0) the per-pass kernels
    int C = 0, periodic = 0;
    for (int c : {1, 2, 4})
        if (!C && P.nphi % c == 0 && P.nphi <= 64 * c && (c == 1 || P.nphi / c >= 1)) {
            C = c;
            periodic = 1;
        }
    if (!C && P.nphi > 64 * 2)
        C = 2;
    if (P.opt.theta_fused == 0)
        C = 0;
    const bool march = C != 0 && P.opt.theta_march != 0;
    if (march) {
        // the kernel reads the pre-transport v_phi and v_r of a ring (halo columns included) while other
        // wavefronts already store the new ones: never in place (the per-loop source step leaves its result in
        // the state grids themselves, the marching one in the *_b twins)
        Dev Wm = W;
        Wm.vrad = P.vrad == W.vrad ? W.vrad_b : W.vrad;
        Wm.vazi = P.vazi == W.vazi ? W.vazi_b : W.vazi;
        res.marched = launch_theta_march(P, Wm, C, periodic, 1, nullptr, st);
        res.thermal = P.adiabatic && Wm.cfl_thermal != nullptr;
        res.vrad = Wm.vrad, res.vazi = Wm.vazi;
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
        LAUNCH2D(KID_ADI_CS_H, k_adi_derived, P.nr, P, 3); // T, c_s, H, P, nu
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
    KLAUNCH(KID_DISK_ON_BODY, k_disk_on_body, grid, block, P, x, y, r_object, smoothing_fixed, r_sm, P.cfl_part);
    KLAUNCH(KID_DISK_ON_BODY, k_disk_on_body_final, dim3(1), dim3(256), (const double *)P.cfl_part, (int)(grid.x * grid.y), out);
}

// rings whose CFL terms read nothing the ghost exchange or the boundary kernels write: ring i reads rows i
// (and i+1 of v_r); fcpt_exchange_unpack writes rows [0,7) and [nr-7,nr)
#define CFL_EDGE_LO (FCPT_OVERLAP + 1)
#define CFL_EDGE_HI (FCPT_OVERLAP + 2)
bool cfl_by_rings(const Dev &P)
{
    // one block per ring: mean and cells in one pass (even Nphi up to 1024 * CFL_MAXP = 8192; the isothermal
    // viscosity and sound speed per ring, or the lazily derived ones of the ideal EOS)
    return (P.nphi & 1) == 0 && P.nphi >= 128 && P.nphi <= 1024 * CFL_MAXP && (!P.adiabatic || P.lazy_derived) &&
           P.stabilize != 2 && P.opt.cfl_rings != 0;
}
// rings of 2049 .. 4096 cells: 1024 threads with two cell pairs each and ALL their loads ahead of the ring sum, instead
// of 256 with eight.  Isothermal (two grids): 37 -> 33 us at 2048 x 4096 (profiles/r03_ab_cfl_threads.txt).  Ideal EOS
// (five grids, 100 VGPRs = one 1024-thread workgroup per CU): 65 against 52 us in the bench's units, 0.505-0.508
// against 0.491-0.496 ms per step (profiles/r03_ab_cfl_hoist.txt) -- it keeps the 256-thread form.
// (round 3, later: 512 threads with four pairs each read 22.7-23.6 us where 1024 with two read 23.8-25.6, the step
//  0.3187 against 0.3197 ms, three A/B pairs, profiles/r03_ab_cfl_512.txt: the isothermal built-in; ideal EOS 62-65 us
//  against 56 for its 256-thread form)
static int cfl_block_form(const Dev &P) // 0: 256 threads per ring, 1: 1024, 2: 512 (the wide forms load everything ahead of the ring sum)
{
    return P.opt.cfl_wide_blocks < 0 ? (P.adiabatic ? 0 : 2) : P.opt.cfl_wide_blocks;
}
static bool cfl_wide_blocks(const Dev &P) { return cfl_block_form(P) != 0; }
static void launch_cfl_rings(const Dev &P, int r1, int n1, int r2, int n2, int finalize, hipStream_t st)
{
    if (n1 + n2 <= 0)
        return;
    const bool wide = P.nphi > 512 * CFL_MAXP;
#ifdef FCPT_CFL_NT /* tuning builds: NT threads per ring, CFL_MAXP * 256 / NT pairs each (Nphi <= 4096) */
    if (!wide) {
        if (P.adiabatic)
            KLAUNCH(KID_CFL_RINGS, (k_cfl_rings<true, CFL_MAXP * 256 / FCPT_CFL_NT, FCPT_CFL_NT>), dim3(n1 + n2), dim3(FCPT_CFL_NT), P, P.cfl_part, r1, n1, r2, finalize);
        else
            KLAUNCH(KID_CFL_RINGS, (k_cfl_rings<false, CFL_MAXP * 256 / FCPT_CFL_NT, FCPT_CFL_NT>), dim3(n1 + n2), dim3(FCPT_CFL_NT), P, P.cfl_part, r1, n1, r2, finalize);
        return;
    }
#endif
    if (!wide && P.nphi > 2048 && cfl_wide_blocks(P)) {
        if (cfl_block_form(P) == 2) { // 512 threads with four cell pairs each
            if (P.adiabatic)
                KLAUNCH(KID_CFL_RINGS, (k_cfl_rings<true, CFL_MAXP / 2, 512>), dim3(n1 + n2), dim3(512), P, P.cfl_part, r1, n1, r2, finalize);
            else
                KLAUNCH(KID_CFL_RINGS, (k_cfl_rings<false, CFL_MAXP / 2, 512>), dim3(n1 + n2), dim3(512), P, P.cfl_part, r1, n1, r2, finalize);
        } else if (P.adiabatic)
            KLAUNCH(KID_CFL_RINGS, (k_cfl_rings<true, CFL_MAXP / 4, 1024>), dim3(n1 + n2), dim3(1024), P, P.cfl_part, r1, n1, r2, finalize);
        else
            KLAUNCH(KID_CFL_RINGS, (k_cfl_rings<false, CFL_MAXP / 4, 1024>), dim3(n1 + n2), dim3(1024), P, P.cfl_part, r1, n1, r2, finalize);
        return;
    }
    if (P.adiabatic && wide)
        KLAUNCH(KID_CFL_RINGS, (k_cfl_rings<true, 2 * CFL_MAXP>), dim3(n1 + n2), dim3(256), P, P.cfl_part, r1, n1, r2, finalize);
    else if (P.adiabatic)
        KLAUNCH(KID_CFL_RINGS, (k_cfl_rings<true, CFL_MAXP>), dim3(n1 + n2), dim3(256), P, P.cfl_part, r1, n1, r2, finalize);
    else if (wide)
        KLAUNCH(KID_CFL_RINGS, (k_cfl_rings<false, 2 * CFL_MAXP>), dim3(n1 + n2), dim3(256), P, P.cfl_part, r1, n1, r2, finalize);
    else
        KLAUNCH(KID_CFL_RINGS, (k_cfl_rings<false, CFL_MAXP>), dim3(n1 + n2), dim3(256), P, P.cfl_part, r1, n1, r2, finalize);
}
// launch_cfl with the final boundary call of the previous step inside the ring launch (see k_cfl_rings_bc); the caller
// has checked cfl_bc_mergeable()
// (worth it only where the ring launch is several rounds of workgroups long -- measured, three / two A/B pairs each,
//  profiles/r03_ab_bc_in_cfl.txt: 2048 x 4096 isothermal 0.3311 against 0.3343 ms per step, ideal EOS 0.511 against
//  0.514; on a grid whose workgroups are all resident at once the four waiting ones start with the rest and spin:
//  512 x 1536 0.0755 against 0.0725, 1024 x 3072 ideal 0.223 against 0.222)
bool cfl_bc_mergeable(const Dev &P)
{
    return cfl_by_rings(P) && P.nr >= 8 && ((long long)P.nr * P.nphi >= (1ll << 22) || P.opt.bc_in_cfl == 2); // (2: tests)
}
void launch_cfl_bc(const Dev &P, int apply_policy, hipStream_t st)
{
    const bool wide = P.nphi > 512 * CFL_MAXP;
#define CFLBC(ADI_, MAXP_, NT_)                                                                                              \
    KLAUNCH(KID_CFL_RINGS_BC, (k_cfl_rings_bc<ADI_, MAXP_, NT_>), dim3((P.nphi + NT_ - 1) / NT_ + P.nr), dim3(NT_), P, P.cfl_part, \
            (P.nphi + NT_ - 1) / NT_)
    if (!wide && P.nphi > 2048 && cfl_wide_blocks(P)) {
        if (cfl_block_form(P) == 2) {
            if (P.adiabatic) {
                CFLBC(true, CFL_MAXP / 2, 512);
            } else {
                CFLBC(false, CFL_MAXP / 2, 512);
            }
        } else if (P.adiabatic) {
            CFLBC(true, CFL_MAXP / 4, 1024);
        } else {
            CFLBC(false, CFL_MAXP / 4, 1024);
        }
    } else if (P.adiabatic && wide) {
        CFLBC(true, 2 * CFL_MAXP, 256);
    } else if (P.adiabatic) {
        CFLBC(true, CFL_MAXP, 256);
    } else if (wide) {
        CFLBC(false, 2 * CFL_MAXP, 256);
    } else {
        CFLBC(false, CFL_MAXP, 256);
    }
#undef CFLBC
    if (apply_policy != 2)
        KLAUNCH(KID_CFL_INIT, k_cfl_final, dim3(1), dim3(1024), P, (const double *)P.cfl_part, P.nr, apply_policy);
}
// the fold a launch_cfl / launch_cfl_bc with apply_policy = 2 left out, for a caller whose marching source kernel did not run after all
void launch_cfl_final(const Dev &P, int apply_policy, hipStream_t st)
{
    KLAUNCH(KID_CFL_INIT, k_cfl_final, dim3(1), dim3(1024), P, (const double *)P.cfl_part, P.nr, apply_policy);
}
// phase 1 of a split CFL: the interior rings only (returns false when the one-block-per-ring kernel does not apply)
bool launch_cfl_interior(const Dev &P, hipStream_t st)
{
    if (!cfl_by_rings(P) || P.nr <= CFL_EDGE_LO + CFL_EDGE_HI)
        return false;
    launch_cfl_rings(P, CFL_EDGE_LO, P.nr - CFL_EDGE_LO - CFL_EDGE_HI, 0, 0, 0, st);
    return true;
}
void launch_cfl(const Dev &P, int apply_policy, hipStream_t st, bool interior_done)
{
    if (cfl_by_rings(P)) {
        // (finalize = 0: the final fold as its own small launch.  Letting the last workgroup of k_cfl_rings do it --
        // cfl_last_workgroup, one agent-scope release per workgroup -- was measured at 110 instead of 36 + 6 us: on
        // this GPU a device-scope release writes the XCD's L2 back, 2048 times per launch.)
        if (interior_done)
            launch_cfl_rings(P, 0, CFL_EDGE_LO, P.nr - CFL_EDGE_HI, CFL_EDGE_HI, 0, st);
        else
            launch_cfl_rings(P, 0, P.nr, 0, 0, 0, st);
        if (apply_policy != 2) // (2: the marching source kernel queued next folds for itself)
            KLAUNCH(KID_CFL_INIT, k_cfl_final, dim3(1), dim3(1024), P, (const double *)P.cfl_part, P.nr, apply_policy);
        return;
    }
    KLAUNCH(KID_RING_MEAN, k_ring_mean, dim3((P.nr + 3) / 4), dim3(256), P, 0, (const double *)nullptr, 0, P.ring_pstride);
    const int nrows = P.active_size - P.first_active;
    int nparts = 0;
    if (nrows > 0) {
        const int rows = march_len(P, CFL_ROWS);
        const Launch2D l = launch2d((nrows + rows - 1) / rows, P.nphi);
        nparts = (int)(l.grid.x * l.grid.y);
        if (l.block.x >= 64)
            KLAUNCH(KID_CFL_CELLS, k_cfl_cells<true>, l.grid, l.block, P, P.cfl_part, rows);
        else
            KLAUNCH(KID_CFL_CELLS, k_cfl_cells<false>, l.grid, l.block, P, P.cfl_part, rows);
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
