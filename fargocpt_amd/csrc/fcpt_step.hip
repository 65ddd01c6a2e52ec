// The hot path of the C ABI (include/fargocpt_hip.h): CFL reduction, the gas step, the final boundary call and the
// run loop, with its hipGraph replay for launch-bound grids.
#include "fcpt_ctx.h"

using namespace fcpt;

namespace fcpt {

// boundary_conditions.cpp:65-114
void apply_boundary_view(fcpt_ctx *c, const Dev &P, bool final, bool damping_done)
{
    if (final && c->d.damping && !damping_done) {
        // damping.cpp:754-774, order of damping_vector: vrad, vaz, sigma, energy
        for (int o = 0; o < 2; ++o)
            launch_damping(P, P.vrad, P.vrad0, P.Rinf.p, c->damp[0][o], 0, c->stream);
        for (int o = 0; o < 2; ++o)
            launch_damping(P, P.vazi, P.vazi0, P.Rmed.p, c->damp[1][o], 0, c->stream);
        for (int o = 0; o < 2; ++o)
            launch_damping(P, P.sigma, P.sigma0, P.Rmed.p, c->damp[2][o], 1, c->stream);
        if (P.adiabatic)
            for (int o = 0; o < 2; ++o)
                launch_damping(P, P.energy, P.energy0, P.Rmed.p, c->damp[3][o], 0, c->stream);
    }
    launch_boundary(P, c->stream);
}
void apply_boundary(fcpt_ctx *c, bool final) { apply_boundary_view(c, c->P, final); }

// isothermal pressure is Sigma c_s^2 with c_s fixed per ring: the marching source kernel forms
// it in registers, so the grid is only materialised for callers that ask for it
// (ideal EOS with the marching source step: the same holds for T, c_s, H and nu)
void ensure_pressure(fcpt_ctx *c)
{
    if (!c->pressure_valid) {
        if (c->P.adiabatic)
            launch_derived(c->P, c->stream);
        else
            launch_pressure(c->P, c->stream);
        c->pressure_valid = true;
    }
}

} // namespace fcpt

namespace {

// one gas "kick" (source terms, artificial viscosity, viscosity, SubStep3) with the step length
// currently in the device clock.  Returns true if the result is in (vrad_b, vazi_b).
bool enqueue_kick(fcpt_ctx *c, bool fold_bc = false)
{
    const Dev &P = c->P;
    hipStream_t st = c->stream;
    c->kick_bc_folded = false;
    if (c->fold_pending && !c->fused_source) {
        launch_cfl_final(P, 1, st);
        c->fold_pending = false;
    }
    if (c->fused_source) {
        // one pass: (v[, e]) -> (v_b[, e_b]); fold_bc: and the boundary call that follows the first kick of a step
        const bool fold_cfl = c->fold_pending && c->march_source && source_march_applies(P);
        if (c->fold_pending && !fold_cfl)
            launch_cfl_final(P, 1, st);
        c->fold_pending = false;
        int segs = 0;
        if (c->march_source && c->skip_q_store && P.adiabatic && cfl_by_rings(P) && !P.cfl_thermal) {
            Dev Q = P; // (Q+ and Q- themselves are read by nothing on the device in this configuration: the ring kernel of the CFL reduction takes their difference)
            Q.q_skip = 1;
            segs = launch_source_march(Q, st, fold_bc, &c->kick_bc_folded, fold_cfl);
        } else if (c->march_source) {
            segs = launch_source_march(P, st, fold_bc, &c->kick_bc_folded, fold_cfl);
        }
        c->skip_q_store = false;
        c->src_parts = segs > 0 ? segs : 0;
        c->kick_energy_b = segs != 0 && P.adiabatic;
        c->qdiff_valid = segs != 0 && P.adiabatic; // the march wrote Q+ - Q- beside Q+ and Q-; the loop kernels do not
        if (!segs) {
            ensure_pressure(c);
            launch_source_fused(P, st);          // (v) -> (v_b) -> (v)
            launch_recalculate_viscosity(P, st);
            launch_viscous_fused(P, st);         // (v) -> (v_b); ideal EOS: + viscous heating + SubStep3
        }
        return true;
    }
    c->qdiff_valid = false;
    ensure_pressure(c);
    launch_source(P, st);
    launch_artificial_viscosity(P, st);
    launch_recalculate_viscosity(P, st);
    launch_stress(P, st);
    launch_viscous_update(P, st);
    if (P.adiabatic)
        launch_substep3(P, 1, st);
    return false;
}

void enqueue_potential(fcpt_ctx *c, bool midstep)
{
    Dev &P = c->P;
    if (midstep && P.adiabatic && P.lazy_derived) {
        // the mid-step potential of step_LeapFrog sees the scale height of the first kick (left in the
        // grid by k_source_march_adi), not one derived from the transported state
        Dev M = P;
        M.lazy_derived = 0;
        if (c->has_mid)
            for (int k = 0; k < P.nbodies; ++k) {
                M.bx[k] = c->mx[k];
                M.by[k] = c->my[k];
                M.bm[k] = c->mm[k];
                M.brsm[k] = c->mrsm[k];
            }
        launch_body_force(M, c->stream);
        c->potential_valid = false;
        return;
    }
    if (midstep && c->has_mid) {
        Dev M = P;
        for (int k = 0; k < P.nbodies; ++k) {
            M.bx[k] = c->mx[k];
            M.by[k] = c->my[k];
            M.bm[k] = c->mm[k];
            M.brsm[k] = c->mrsm[k];
        }
        launch_body_force(M, c->stream);
        c->potential_valid = false; // the grid now holds the mid-step potential
        return;
    }
    if (P.inline_potential) { // k_source_march_adi evaluates it ring by ring; the grid is only filled on request
        c->potential_valid = false;
        return;
    }
    if (P.adiabatic || !c->potential_valid) {
        launch_body_force(P, c->stream); // CalculateNbodyPotential | CalculateAccelOnGas; static when H and the bodies are
        c->potential_valid = true;
    }
}

// the gas part of step_Euler up to Transport (simulation.cpp:167-217), or of step_LeapFrog
// (simulation.cpp:316-393): kick 1/2 (dt/2), drift (dt), kick 2/2 (dt/2).  `dt_dev`: the step

} // namespace

namespace fcpt {

void enqueue_step(fcpt_ctx *c, bool dt_dev, double dt, bool shear_safe, bool split)
{
    join_side(c);
    c->cfl_interior = false;
    const Dev &P = c->P;
    hipStream_t st = c->stream;
    const bool frog = c->d.integrator == FCPT_INTEGRATOR_LEAPFROG;
    if (frog)
        launch_clock_scale_dt(P.clk, dt_dev ? 1 : 0, dt, 0.5, st); // dt <- step/2, keeps step in cfl_dt
    else if (!dt_dev)
        launch_clock_set_dt(P.clk, dt, st);
    enqueue_potential(c, false);
    // (after an upload of a state grid the ghost rings may not satisfy the boundary conditions yet: the folded form
    //  rewrites Sigma's ghost ring while neighbouring wavefronts may still read it -- harmless only when the values
    //  are the ones already there, so that one step takes the separate launch)
    //  likewise a second fcpt_step without fcpt_post in between: the ghost rings are the transport's, not a boundary call's)
    const bool in_b = enqueue_kick(c, !c->ghosts_unknown && !c->stepped);
    c->ghosts_unknown = false;
    Dev Q = P; // view with the post-kick velocities
    if (in_b) {
        Q.vrad = P.vrad_b;
        Q.vazi = P.vazi_b;
    }
    if (c->kick_energy_b)
        Q.energy = P.energy_b;
    c->kick_energy_b = false;
    Q.src_ring_nparts = in_b ? c->src_parts : 0; // ring sums of v_phi left by k_source_march
    c->src_parts = 0;
    if (!c->kick_bc_folded) // (else the source march applied it on its edge chunks)
        apply_boundary_view(c, Q, false);
    c->kick_bc_folded = false;
    if (frog)
        launch_clock_scale_dt(P.clk, 2, 0.0, 1.0, st); // dt <- step (saved in cfl_dt)
    launch_massflow(Q, st); // WriteMassFlow: what this Transport() carries through the interfaces
    TransportResult tr;
    if (split && !frog && transport_can_split(Q, shear_safe) && c->side) {
        launch_shift_means(Q, st);
        (void)hipEventRecord(c->e_fork, st);
        (void)hipStreamWaitEvent(c->side, c->e_fork, 0);
        (void)launch_transport(Q, P, c->side, TRANSPORT_INTERIOR);
        (void)hipEventRecord(c->e_join, c->side);
        tr = launch_transport(Q, P, st, TRANSPORT_EDGES);
        c->join_pending = true;
    } else {
        tr = launch_transport(Q, P, st, TRANSPORT_ALL, c->want_gated_deferred && !frog ? &c->gated : nullptr);
        c->gated_pending = tr.gated_pending != 0;
    }
    c->want_gated_deferred = false;
    if (!tr.marched)
        launch_clock_advance(P.clk, st);
    // a marching transport kernel stored the cell-local CFL terms with the new Sigma and e; they stay those of the
    // final state if nothing but boundary rings and ghost rows changes before the next CFL reduction (the wave
    // damping folded into that kernel, or no damping zone on this slab)
    c->thermal_valid = tr.thermal != 0 && tr.marched > 0 && !frog && (!c->damp_any || P.damp_in_step != 0);
    // the marching transport is out of place: the new state may sit in the scratch twins
    if (tr.sigma != c->P.sigma)
        std::swap(c->P.sigma, c->P.sigA);
    if (tr.energy != c->P.energy)
        std::swap(c->P.energy, c->P.eA);
    if (tr.vrad != c->P.vrad)
        std::swap(c->P.vrad, c->P.vrad_b);
    if (tr.vazi != c->P.vazi)
        std::swap(c->P.vazi, c->P.vazi_b);
    c->grid[FCPT_F_SIGMA] = c->P.sigma;
    c->grid[FCPT_F_VRAD] = c->P.vrad;
    c->grid[FCPT_F_VAZI] = c->P.vazi;
    c->grid[FCPT_F_ENERGY] = c->P.energy;
    if (frog) {
        launch_clock_scale_dt(P.clk, 2, 0.0, 0.5, st); // dt <- step/2
        enqueue_potential(c, true);
        c->pressure_valid = false; // compute_pressure(data), simulation.cpp:378
        if (P.adiabatic && !P.lazy_derived)
            ensure_pressure(c);
        c->P.kick_time_shift = 1; // SubStep3 of the second kick runs at midstep_time (simulation.cpp:388)
        const bool home = enqueue_kick(c);
        c->P.kick_time_shift = 0;
        if (home) { // result in the *_b buffers: bring it home
            const size_t ns = (size_t)P.nr * P.nphi * sizeof(double), nv = (size_t)(P.nr + 1) * P.nphi * sizeof(double);
            (void)hipMemcpyAsync(P.vrad, P.vrad_b, nv, hipMemcpyDeviceToDevice, st);
            (void)hipMemcpyAsync(P.vazi, P.vazi_b, ns, hipMemcpyDeviceToDevice, st);
            if (c->kick_energy_b)
                (void)hipMemcpyAsync(P.energy, P.energy_b, ns, hipMemcpyDeviceToDevice, st);
            c->kick_energy_b = false;
        }
        launch_clock_scale_dt(P.clk, 2, 0.0, 1.0, st); // dt <- step, for the damping of the final boundary call
    }
    c->stepped = true;
}

void flush_deferred_boundary(fcpt_ctx *c)
{
    if (c->bc_deferred) {
        c->bc_deferred = false;
        launch_boundary(c->P, c->stream);
    }
}

void enqueue_post(fcpt_ctx *c, bool may_defer_boundary)
{
    join_side(c);
    // the damping of the final boundary call was applied by k_velocities when damp_in_step
    const bool damping_done = c->P.damp_in_step != 0 && c->stepped;
    // fcpt_run_steps: when the call is nothing but the ghost-ring kernel (no separate damping launches, no derived grids
    // to refresh behind it) and the next launch of the stream is the one-block-per-ring CFL kernel, that launch carries it
    const bool defer = may_defer_boundary && (damping_done || !c->damp_any) && !(c->P.adiabatic && !c->P.lazy_derived) && cfl_bc_mergeable(c->P);
    if (c->gated_pending) { // (enqueue_device_step asked the transport to leave its gated fallback launch to this call)
        c->gated_pending = false;
        if (!defer && (damping_done || !c->d.damping)) { // the call is nothing but the ghost-ring kernel: one launch for both
            launch_gated_theta(c->gated, &c->P, c->stream);
            c->stepped = false;
            if (c->P.adiabatic && !c->P.lazy_derived) {
                launch_derived(c->P, c->stream);
                c->pressure_valid = true;
            } else {
                c->pressure_valid = false;
            }
            return;
        }
        launch_gated_theta(c->gated, nullptr, c->stream);
    }
    if (defer)
        c->bc_deferred = true;
    else
        apply_boundary_view(c, c->P, true, damping_done);
    c->stepped = false;
    if (c->P.adiabatic && !c->P.lazy_derived) {
        launch_derived(c->P, c->stream);
        c->pressure_valid = true;
    } else {
        c->pressure_valid = false; // recalculate_derived_disk_quantities: P only, evaluated lazily
    }
}

void enqueue_cfl(fcpt_ctx *c, int apply_policy)
{
    join_side(c);
    c->P.cfl_thermal_on = c->thermal_valid ? 1 : 0;
    c->P.qdiff_on = c->qdiff_valid ? 1 : 0;
    if (c->bc_deferred && !c->cfl_interior && cfl_bc_mergeable(c->P)) {
        c->bc_deferred = false;
        launch_cfl_bc(c->P, apply_policy, c->stream); // + the final boundary call of the step before
    } else {
        flush_deferred_boundary(c);
        launch_cfl(c->P, apply_policy, c->stream, c->cfl_interior);
    }
    c->cfl_interior = false;
}

} // namespace fcpt

extern "C" {

// condition_cfl for the rings that neither the ghost exchange nor the boundary kernels touch, to be queued between
// fcpt_exchange_pack and the wait for the neighbours' rings: it runs while they are on the wire.  The next
// fcpt_cfl / fcpt_cfl_device evaluates the remaining rings and reduces.  A no-op (the whole CFL runs later)
// whenever that split would not see the final state: damping outside the step kernels, narrow rings, ...
int fcpt_cfl_begin(fcpt_ctx *c)
{
    if (!c)
        return FCPT_EINVAL;
    join_side(c);
    c->cfl_interior = false;
    const bool state_final = c->stepped && (!c->damp_any || c->P.damp_in_step != 0); // fcpt_post will not damp
    if (!state_final)
        return FCPT_OK;
    if (c->P.opt.cfl_split == 0)
        return FCPT_OK;
    ProfScope prof_scope(c);
    c->P.cfl_thermal_on = c->thermal_valid ? 1 : 0;
    c->P.qdiff_on = c->qdiff_valid ? 1 : 0;
    c->cfl_interior = launch_cfl_interior(c->P, c->stream);
    HIPCHK(hipGetLastError());
    return FCPT_OK;
}

int fcpt_cfl(fcpt_ctx *c, double *dt_local)
{
    if (!c || !dt_local)
        return FCPT_EINVAL;
    ProfScope prof_scope(c);
    enqueue_cfl(c, 0);
    HIPCHK(hipGetLastError());
    DevClock k;
    if (int rc = read_clock(c, &k))
        return rc;
    double v;
    std::memcpy(&v, &k.cfl_bits, sizeof(v));
    *dt_local = v;
    return FCPT_OK;
}

int fcpt_cfl_device(fcpt_ctx *c, double *d_dt_local)
{
    if (!c || !d_dt_local)
        return FCPT_EINVAL;
    ProfScope prof_scope(c);
    enqueue_cfl(c, 0);
    launch_clock_export_cfl(c->P.clk, d_dt_local, c->stream);
    HIPCHK(hipGetLastError());
    return FCPT_OK;
}

int fcpt_calculate_timestep_device(fcpt_ctx *c, const double *d_cfl_global)
{
    if (c && !d_cfl_global)
        d_cfl_global = c->d_cfl; // what fcpt_cfl_allreduce left
    if (!c || !d_cfl_global)
        return FCPT_EINVAL;
    ProfScope prof_scope(c);
    launch_clock_policy_ptr(c->P.clk, c->d.cfl_max_var, d_cfl_global, c->stream);
    c->policy_dt_dev = true;
    HIPCHK(hipGetLastError());
    return FCPT_OK;
}

int fcpt_step_device(fcpt_ctx *c)
{
    if (!c)
        return FCPT_EINVAL;
    ProfScope prof_scope(c);
    enqueue_step(c, true, 0.0, c->policy_dt_dev && c->d.cfl <= 0.8);
    c->policy_dt_dev = false;
    HIPCHK(hipGetLastError());
    return FCPT_OK;
}

// fcpt_step_device for slabs with neighbours: the chunks of the transport that hold the rings the neighbours are
// waiting for (rows [7,14), [nr-14,nr-7)) are marched on the caller's stream, all others on an internal stream,
// so that fcpt_exchange_pack and the transfers queued next run under the interior chunks.  Until
// fcpt_step_device_end only fcpt_exchange_pack may be called.
int fcpt_step_device_begin(fcpt_ctx *c)
{
    if (!c)
        return FCPT_EINVAL;
    if (!c->side) {
        HIPCHK(hipStreamCreateWithFlags(&c->side, hipStreamNonBlocking));
        HIPCHK(hipEventCreateWithFlags(&c->e_fork, hipEventDisableTiming));
        HIPCHK(hipEventCreateWithFlags(&c->e_join, hipEventDisableTiming));
    }
    ProfScope prof_scope(c);
    enqueue_step(c, true, 0.0, c->policy_dt_dev && c->d.cfl <= 0.8, true);
    c->policy_dt_dev = false;
    HIPCHK(hipGetLastError());
    return FCPT_OK;
}

int fcpt_step_device_end(fcpt_ctx *c)
{
    if (!c)
        return FCPT_EINVAL;
    join_side(c);
    return FCPT_OK;
}

int fcpt_post_device(fcpt_ctx *c)
{
    if (!c)
        return FCPT_EINVAL;
    join_side(c);
    ProfScope prof_scope(c);
    enqueue_post(c);
    HIPCHK(hipGetLastError());
    return FCPT_OK;
}

int fcpt_calculate_timestep(fcpt_ctx *c, double cfl_dt_global, double *dt)
{
    if (!c || !dt)
        return FCPT_EINVAL;
    launch_clock_policy(c->P.clk, c->d.cfl_max_var, 0, cfl_dt_global, c->stream);
    DevClock k;
    if (int rc = read_clock(c, &k))
        return rc;
    *dt = k.last_dt;
    c->policy_dt_host = k.last_dt;
    return FCPT_OK;
}

int fcpt_snap_to_monitor(const fcpt_ctx *cc, double cfl_dt, double *step_dt)
{
    fcpt_ctx *c = const_cast<fcpt_ctx *>(cc);
    if (!c || !step_dt)
        return FCPT_EINVAL;
    DevClock k;
    if (int rc = read_clock(c, &k))
        return rc;
    // simulation.cpp:528-540
    const double time_next_monitor = (k.n_monitor + 1) * c->d.monitor_timestep;
    const double time_left_till_write = time_next_monitor - k.time;
    const bool overshoot = cfl_dt > time_left_till_write;
    const double dt_stretch_factor = 0.05;
    const bool almost_there = time_left_till_write < cfl_dt * (1 + dt_stretch_factor);
    *step_dt = (overshoot || almost_there) ? time_left_till_write : cfl_dt;
    return FCPT_OK;
}

int fcpt_step(fcpt_ctx *c, double dt)
{
    if (!c)
        return FCPT_EINVAL;
    ProfScope prof_scope(c);
    enqueue_step(c, false, dt, c->policy_dt_host > 0.0 && dt <= c->policy_dt_host * (1.0 + 1e-12) && c->d.cfl <= 0.8);
    c->policy_dt_host = -1.0;
    HIPCHK(hipGetLastError());
    return FCPT_OK;
}

int fcpt_post(fcpt_ctx *c, double dt)
{
    if (!c)
        return FCPT_EINVAL;
    ProfScope prof_scope(c);
    launch_clock_set_dt(c->P.clk, dt, c->stream);
    enqueue_post(c);
    HIPCHK(hipGetLastError());
    return FCPT_OK;
}

int fcpt_apply_boundary(fcpt_ctx *c, double dt, int32_t final)
{
    if (!c)
        return FCPT_EINVAL;
    join_side(c);
    ProfScope prof_scope(c);
    launch_clock_set_dt(c->P.clk, dt, c->stream);
    if (final && c->damp_any)
        c->thermal_valid = false; // the wave damping changes Sigma and e of the damping zones
    apply_boundary(c, final != 0);
    HIPCHK(hipGetLastError());
    return FCPT_OK;
}

} // extern "C"
namespace {
unsigned launch_flags(const fcpt_ctx *c)
{
    return (c->potential_valid ? 1u : 0u) | (c->pressure_valid ? 2u : 0u) | (c->stepped ? 4u : 0u) |
           (c->cfl_interior ? 8u : 0u) | (c->kick_energy_b ? 16u : 0u) | (c->fused_source ? 32u : 0u) |
           (c->march_source ? 64u : 0u) | (c->has_mid ? 128u : 0u) | (c->join_pending ? 256u : 0u) |
           (c->thermal_valid ? 512u : 0u) | (c->ghosts_unknown ? 1024u : 0u) | (c->qdiff_valid ? 2048u : 0u) |
           (c->bc_deferred ? 4096u : 0u) | ((unsigned)c->src_parts << 13);
}
bool graph_wanted(const fcpt_ctx *c)
{
    if (c->profiling || c->graph_failed || c->comm)
        return false;
    if (c->P.opt.graph_steps >= 0)
        return c->P.opt.graph_steps != 0;
    // launch-bound grids: the kernels of a step are shorter than the host's launch calls
    return (long long)c->P.nr * c->P.nphi <= 131072;
}
// one iteration of the device-resident loop: CFL reduction -> policy kernel -> step -> post
void enqueue_device_step(fcpt_ctx *c)
{
    // (built-in: grids below 4 M cells -- measured, profiles/r03_ab_cfl_fold_in_source.txt: 512 x 1536 -2.4 %, 1024 x 3072
    //  ideal EOS -1.2 %; at 2048 x 4096 the 1 400 workgroups folding 48 KB each cost the kernel what the launch saved)
    // (the fold of the CFL reduction inside the marching source kernel: Euler steps whose first launch behind the CFL
    //  reduction -- but for k_potential, which reads no step length -- is that kernel; ring kernel of the reduction in use)
    const bool frog = c->d.integrator == FCPT_INTEGRATOR_LEAPFROG;
    const int want = c->P.opt.cfl_fold_in_source;
    const bool fold = (want < 0 ? (long long)c->P.nr * c->P.nphi < (1ll << 22) : want != 0) && !frog && c->fused_source &&
                      c->march_source && source_march_applies(c->P) && cfl_by_rings(c->P);
    enqueue_cfl(c, fold ? 2 : 1);
    c->fold_pending = fold;
    {
        // the gated launch of the fallback transport together with the final boundary call, where that call will be
        // nothing but the ghost-ring kernel and does not ride in the next CFL launch (grids below 4 M cells)
        const bool pure_bc = (c->P.damp_in_step != 0 || !c->d.damping) && !(c->P.adiabatic && !c->P.lazy_derived);
        const bool rides_in_cfl = c->P.opt.bc_in_cfl != 0 && (c->P.damp_in_step != 0 || !c->damp_any) &&
                                  !(c->P.adiabatic && !c->P.lazy_derived) && cfl_bc_mergeable(c->P);
        c->want_gated_deferred = c->P.opt.gate_in_boundary != 0 && !frog && pure_bc && !rides_in_cfl;
    }
    enqueue_step(c, true, 0.0, c->d.cfl <= 0.8);
    enqueue_post(c, c->P.opt.bc_in_cfl != 0); // (the boundary call may ride in the next iteration's CFL launch)
}
// capture `cycle` steps; true if the host-side state is back where it started (the graph can be replayed)
bool capture_graph(fcpt_ctx *c, int cycle)
{
    if (!c->capture_stream && hipStreamCreateWithFlags(&c->capture_stream, hipStreamNonBlocking) != hipSuccess)
        return false;
    const Dev P0 = c->P;
    const unsigned f0 = launch_flags(c);
    hipStream_t user = c->stream;
    c->stream = c->capture_stream;
    bool ok = hipStreamBeginCapture(c->capture_stream, hipStreamCaptureModeThreadLocal) == hipSuccess;
    if (ok) {
        for (int n = 0; n < cycle; ++n)
            enqueue_device_step(c);
        hipGraph_t g = nullptr;
        ok = hipStreamEndCapture(c->capture_stream, &g) == hipSuccess && g != nullptr;
        c->graph = g;
    }
    c->stream = user;
    (void)hipGetLastError();
    const bool periodic = std::memcmp(&P0, &c->P, sizeof(Dev)) == 0 && f0 == launch_flags(c);
    if (ok && periodic)
        ok = hipGraphInstantiate(&c->graph_exec, c->graph, nullptr, nullptr, 0) == hipSuccess;
    if (!ok || !periodic) {
        // nothing was executed during the capture: put the host-side view back and step without a graph
        c->P = P0;
        c->grid[FCPT_F_SIGMA] = c->P.sigma;
        c->grid[FCPT_F_VRAD] = c->P.vrad;
        c->grid[FCPT_F_VAZI] = c->P.vazi;
        c->grid[FCPT_F_ENERGY] = c->P.energy;
        c->potential_valid = f0 & 1u;
        c->pressure_valid = f0 & 2u;
        c->stepped = f0 & 4u;
        c->cfl_interior = f0 & 8u;
        c->kick_energy_b = f0 & 16u;
        c->thermal_valid = f0 & 512u;
        c->ghosts_unknown = f0 & 1024u;
        c->qdiff_valid = f0 & 2048u;
        c->bc_deferred = f0 & 4096u;
        c->src_parts = (int)(f0 >> 13);
        drop_graph(c);
        return false;
    }
    c->graph_cycle = cycle;
    c->graph_P = c->P;
    c->graph_flags = f0;
    return true;
}
} // namespace
extern "C" {

// sim::run's loop (simulation.cpp:515-553) for a single slab.
int fcpt_run_steps(fcpt_ctx *c, int64_t nsteps, int32_t snap, int64_t *done)
{
    if (!c)
        return FCPT_EINVAL;
    ProfScope prof_scope(c);
    int64_t n = 0;
    const bool slabs = c->comm && (c->peer_inner >= 0 || c->peer_outer >= 0); // this slab has neighbours
    if (slabs && !snap) {
        // several slabs, dt never leaves the device: CFL -> MIN over the slabs (cfl.cpp:379) -> policy kernel -> step
        // -> ghost exchange (simulation.cpp:236) -> post, all on one stream
        for (; n < nsteps; ++n) {
            if (int rc = enqueue_cfl_allreduce(c))
                return rc;
            launch_clock_policy_ptr(c->P.clk, c->d.cfl_max_var, c->d_cfl, c->stream);
            c->skip_q_store = c->d.integrator != FCPT_INTEGRATOR_LEAPFROG && n + 1 < nsteps; // Q+ / Q- (outputs) by the last step only
            enqueue_step(c, true, 0.0, c->d.cfl <= 0.8);
            if (int rc = enqueue_exchange(c))
                return rc;
            enqueue_post(c);
        }
        HIPCHK(hipGetLastError());
    } else if (slabs) {
        // monitor-time snapping: the reduced dt comes to the host, as in sim::run
        const double t_final = (double)c->d.nsnapshots * c->d.nmonitor * c->d.monitor_timestep;
        for (; n < nsteps; ++n) {
            DevClock k;
            if (int rc = read_clock(c, &k))
                return rc;
            if (t_final > 0 && !(k.time < t_final))
                break;
            double cfl_dt, dt, step_dt;
            if (int rc = fcpt_cfl_allreduce(c, &cfl_dt))
                return rc;
            if (int rc = fcpt_calculate_timestep(c, cfl_dt, &dt))
                return rc;
            if (int rc = fcpt_snap_to_monitor(c, dt, &step_dt))
                return rc;
            const double time_next_monitor = (k.n_monitor + 1) * c->d.monitor_timestep;
            if (int rc = fcpt_step(c, step_dt))
                return rc;
            if (int rc = fcpt_exchange(c))
                return rc;
            if (int rc = fcpt_post(c, step_dt))
                return rc;
            if (int rc = read_clock(c, &k))
                return rc;
            if (std::fabs(time_next_monitor - k.time) < 1e-6 * dt) {
                fcpt_clock hc = {k.time, k.last_dt, k.n_hydro_iter, k.n_monitor + 1, 0};
                hc.n_snapshot = hc.n_monitor / (uint32_t)(c->d.nmonitor > 0 ? c->d.nmonitor : 1);
                if (int rc = fcpt_set_clock(c, &hc))
                    return rc;
            }
        }
    } else if (!snap) {
        // dt never leaves the device: CFL reduction -> policy kernel -> step -> post
        if (graph_wanted(c) && nsteps >= 8) {
            join_side(c);
            // A captured cycle is replayed while the host-side state that decided its launches (the whole Dev view, the
            // lazy-evaluation flags) is what it was at capture: bodies, options, uploads change it for good; the parity
            // of the transport's ping-pong and the boundary call that rides in the next CFL launch (deferred inside
            // this function only) are put back by one or two plain steps -- the same two that let the lazily evaluated
            // grids settle before the first capture.
            auto valid = [&] {
                return c->graph_exec && std::memcmp(&c->graph_P, &c->P, sizeof(Dev)) == 0 && c->graph_flags == launch_flags(c);
            };
            for (int lead = 0; lead < 2 && !valid(); ++lead, ++n)
                enqueue_device_step(c);
            if (!valid()) {
                drop_graph(c);
                if (!capture_graph(c, 2) && !capture_graph(c, 4))
                    c->graph_failed = true;
            }
            if (c->graph_exec) {
                for (; n + c->graph_cycle <= nsteps; n += c->graph_cycle) {
                    HIPCHK(hipGraphLaunch(c->graph_exec, c->stream));
                    ++c->graph_replays;
                }
            }
        }
        const bool frog_loop = c->d.integrator == FCPT_INTEGRATOR_LEAPFROG;
        for (; n < nsteps; ++n) {
            c->skip_q_store = !frog_loop && n + 1 < nsteps; // Q+ / Q- (outputs) by the last step only
            enqueue_device_step(c);
        }
        flush_deferred_boundary(c); // the last step's boundary call has no CFL launch to ride in
        HIPCHK(hipGetLastError());
    } else {
        const double t_final = (double)c->d.nsnapshots * c->d.nmonitor * c->d.monitor_timestep;
        for (; n < nsteps; ++n) {
            DevClock k;
            if (int rc = read_clock(c, &k))
                return rc;
            if (t_final > 0 && !(k.time < t_final))
                break;
            double cfl_dt, dt, step_dt;
            if (int rc = fcpt_cfl(c, &cfl_dt))
                return rc;
            if (int rc = fcpt_calculate_timestep(c, cfl_dt, &dt))
                return rc;
            if (int rc = fcpt_snap_to_monitor(c, dt, &step_dt))
                return rc;
            const double time_next_monitor = (k.n_monitor + 1) * c->d.monitor_timestep;
            if (int rc = fcpt_step(c, step_dt))
                return rc;
            if (int rc = fcpt_post(c, step_dt))
                return rc;
            if (int rc = read_clock(c, &k))
                return rc;
            const bool towrite = std::fabs(time_next_monitor - k.time) < 1e-6 * dt;
            if (towrite) {
                fcpt_clock hc = {k.time, k.last_dt, k.n_hydro_iter, k.n_monitor + 1, 0};
                hc.n_snapshot = hc.n_monitor / (uint32_t)(c->d.nmonitor > 0 ? c->d.nmonitor : 1);
                if (int rc = fcpt_set_clock(c, &hc))
                    return rc;
            }
        }
    }
    if (done)
        *done = n;
    return FCPT_OK;
}

} // extern "C"
