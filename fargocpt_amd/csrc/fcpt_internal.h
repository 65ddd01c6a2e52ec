// Internal declarations shared by the host and HIP translation units.
#ifndef FCPT_INTERNAL_H
#define FCPT_INTERNAL_H

#include "../../include/fargocpt_hip.h"

#include <cstddef>
#include <cstdint>
#include <vector>

#ifndef M_PI
#define M_PI 3.14159265358979323846
#endif

namespace fcpt {

void set_error(const char *fmt, ...);

// Radial 1-D arrays of one slab (src/global.h:62-79), nr + FCPT_GEOM_PAD + 1 entries.
struct HostGeometry {
    std::vector<double> Rmed, Rinf, Rsup, Surf, InvRmed, InvRinf, InvSurf, InvDiffRmed, InvDiffRsup,
        InvDiffRsupRb;
    double dphi = 0, invdphi = 0;
    double cf_growth = 0, cf_inv_log_growth = 0, cf_opt_const = 0; // src/find_cell_id.cpp
};

int build_radii(const fcpt_desc &d, double *radii);
int split_domain(const fcpt_desc &d, fcpt_split &out);
void build_geometry(const fcpt_desc &d, const fcpt_split &s, const double *radii, HostGeometry &g);
int rmed_id(const fcpt_desc &d, const fcpt_split &s, const HostGeometry &g, double r);
int rinf_id(const fcpt_desc &d, const fcpt_split &s, const HostGeometry &g, double r);

// Simulation clock as it lives in device memory; kernels read dt from here so a
// step never needs a host round trip.
struct DevClock {
    double time;
    double last_dt;
    double dt;     // the step length currently in force
    double cfl_dt; // result of the last CFL policy evaluation
    unsigned long long cfl_bits; // running min of the CFL reduction (bit pattern of a positive double)
    unsigned long long n_hydro_iter;
    unsigned int n_monitor;
    unsigned int n_snapshot;
    unsigned int shear_error; // sticky: k_transport_fused met |dNshift| > 1 in a step launched without fallback
    unsigned int last_dt_pending; // the CFL fold inside the source march left dt but not last_dt (other workgroups were still reading it): clock_advance completes it
    // hydro_dt_logger (hydro_dt_logger.h:13-34): smallest / largest step since the last fcpt_dt_statistics(reset)
    double dt_min, dt_max;
};
#ifdef __HIPCC__
// sim::time += dt; N_hydro_iter++ (simulation.cpp:226-227) and the step-size log, by one thread per step
__device__ __forceinline__ void clock_advance(DevClock *clk, double dt)
{
    clk->time += dt;
    clk->n_hydro_iter += 1;
    if (clk->last_dt_pending) { // sim::CalculateTimeStep's last_dt = dt (simulation.cpp:100-118), deferred by cfl_fold_in_step
        clk->last_dt = dt;
        clk->last_dt_pending = 0;
    }
    clk->dt_min = dt < clk->dt_min ? dt : clk->dt_min;
    clk->dt_max = dt > clk->dt_max ? dt : clk->dt_max;
}
#endif

// Per-row damping description, built on the host (damping.cpp:311-427).
struct DampRange {
    int lo, hi;      // inclusive row range, lo > hi = disabled
    int type;        // FCPT_DAMP_*
    double rlim, redge, tau;
};

// Read-only per-ring array.  Device reads go through the constant address space, so a
// wavefront-uniform index becomes a scalar-cache load (s_load) even inside kernels that also
// store to global memory; plain `const double*` members would be re-read with 64-lane vector
// loads at full memory latency because the compiler cannot rule out aliasing with the stores.
// These arrays are written only by hipMemcpy / an earlier kernel, never by the reading kernel.
struct CArr {
    const double *p;
#ifdef __HIPCC__
    __device__ __forceinline__ double operator[](int i) const
    {
        return ((const double __attribute__((address_space(4))) *)p)[i];
    }
#endif
};
struct CArrI {
    const int *p;
#ifdef __HIPCC__
    __device__ __forceinline__ int operator[](int i) const
    {
        return ((const int __attribute__((address_space(4))) *)p)[i];
    }
#endif
};

// Packed per-ring rows for the marching kernels: one wide scalar load (s_load_dwordx8/x16) per row
// instead of one s_load + s_waitcnt per array, the dominant stall of a 2-3 waves/SIMD kernel.
struct alignas(32) RadRow { // radial interface k, stored at index k + 1 (k = -1 .. nr)
    double dr_lo;  // Rmed[k] - Rmed[k-1]   (k clamped to an open interface)
    double dr_hi;  // Rmed[k+1] - Rmed[k]
    double gphi;   // dphi * Rinf[k]
    double idr_up; // InvDiffRmed[k+1] if rings k and k+1 both exist, else 0
};
// everything iteration m of k_source_march needs (rings m, m-1, m-2), stored at index m + 2;
// fields of rings outside the grid are 0 and guarded by the kernel's row-range tests
struct alignas(64) SrcRow {
    // stage A, ring m
    double cs2_m, cs2_m1, idr_m, rinf_om_m, inv_rinf_m, inv_dxt_m;
    // stage B, ring crow(m-1)
    double inv_drsup_b, inv_rmed_b, lsq_b;
    // stage C, ring m-1
    double inv_rsum_c, rmed_c, rmed_cm1, inv_drmed2_c, idr_c, inv_dxtheta_c;
    // stage D, diagonal on ring crow(m-2), r-phi on ring m-1
    double rinf_d1, rinf_d0, inv_drsuprb_d, inv_rmed_d, inv_drsup_d, nu_d;
    double inv_rmed_r, inv_rmed_rm1, idr_r, rinf_r, inv_rinf_r, nu_avg_r;
    // stage E, ring k = m-2
    double inv_rmed_k, two_inv_dra2_k, ra1sq_k, ra0sq_k, inv_rmsum_k, rmed_k, rmed_km1, idr_k;
    // energy equation, ring crow(m-1): Rinf[r+1], Rinf[r], InvDiffRsupRb[r], 1/Omega_K(Rmed[r]), 1/(dphi Rmed[r])
    double rinf_b1, rinf_b0, inv_drsuprb_b, inv_omk_b, inv_dxtheta_b;
};
static_assert(sizeof(SrcRow) == 320, "SrcRow is five 64-byte scalar loads");
struct alignas(64) ThetaRow { // ring i
    double invsurf, dxtheta, inv_dxtheta, dr_invsurf, invr, r_omega, rmed, pad;
};
struct alignas(64) ShiftRow { // ring i, written by k_ring_mean (transport call)
    double mean, vconst;
    double es, ev, ev_top; // exp(-dt f / tau) of the wave damping: cell-centred, v_r, v_r row i+1 (last ring only)
    int nshift, pad0;
    double pad1[2];
};
struct alignas(64) DampRow { // ring i (nr + 1 rows: v_r has row nr)
    double fs, ts, fv, tv;
    int tvr, tva, tsg, ten;
    double pad[2];
};

// Kernel-selection switches of one context (fcpt_set_option / fcpt_get_option).  The FCPT_* environment
// variables of the same names only seed the defaults in fcpt_create; nothing on the launch path reads the
// environment.  A value of -1 means "not set: the launcher's built-in choice".
struct Options {
    int transport_fused;    // 0: off, 1 | 2: k_transport_fused with 1 | 2 cells per lane (rings of >= 256 cells)
    int transport_rows;     // rings per marching chunk of k_transport_fused (> 0: equal chunks of that many rings)
    int transport_graded;   // 1: chunks of graded length, long ones first (transport_schedule()); 0: equal chunks
    int transport_big;      // graded: rings per chunk of the first round (-1: from the grid and the wavefront slots)
    int transport_rank_grade; // one round of wavefronts: per cent by which the chunks of the last rank of a SIMD are shorter than those of the first (transport_rank_table(); 0: equal chunks, -1: built-in)
    int transport_ladder;   // graded: per cent of the previous length each further round of chunks gets (-1: built-in)
    int source_rows;        // ... of k_source_march(_adi) (> 0: equal chunks of that many rings)
    int source_graded;      // per cent by which the wavefronts a SIMD receives last get shorter chunks than those it receives first (source_schedule(); 0: equal chunks, -1: built-in)
    int theta_rows;         // ... of k_transport_theta_march
    int transport_fallback; // 1: the two-kernel transport is queued behind every k_transport_fused
    int transport_split;    // fcpt_step_device_begin may split the transport around the ghost exchange
    int fused_source;       // 0: per-loop source kernels
    int march_source;       // 0: the three fused source kernels instead of the marching one
    int march_source_adi;   // 0: ... for the ideal EOS only
    int theta_march;        // 0: per-pass azimuthal kernels
    int theta_fused;        // 0: same (older name, kept)
    int cfl_rings;          // 0: k_ring_mean + k_cfl_cells instead of k_cfl_rings
    int cfl_wide_blocks;    // rings of 2049 .. 4096 cells: 0: k_cfl_rings with 256 threads per ring, 1 | 2: 1024 | 512 threads and all loads ahead of the ring sum, -1: built-in (isothermal 2, ideal EOS 0)
    int cfl_fold_in_source; // fcpt_run_steps on one slab: 0: k_cfl_final as its own launch, 1: every workgroup of the marching source kernel folds the partial maxima itself (-1: built-in)
    int gate_in_boundary;   // fcpt_run_steps on one slab: 1: the idle gated launch of the fallback transport and the final boundary call of the step are ONE launch where the boundary call is its own kernel (grids below 4 M cells)
    int cfl_split;          // fcpt_cfl_begin evaluates the interior rings ahead of the ghost exchange
    int source_ring_parts;  // 0: the transport's ring mean re-reads v_phi
    int fused_damping;      // 0: the wave damping as separate kernels in the final boundary call
    int inline_potential;   // 0: ideal EOS: k_potential every step instead of the evaluation inside k_source_march_adi
    int cfl_thermal;        // 1: ideal EOS: the transport stores the cell-local CFL terms, the CFL kernel reads 3 grids, not 6
    int bc_fold;            // 0: the pre-transport boundary call as its own launch instead of inside the source march
    int bc_in_cfl;          // 0: fcpt_run_steps launches the final boundary call of a step by itself instead of inside the next CFL launch (1: on grids of >= 4M cells, 2: on any grid)
    int comm_overlap;       // fcpt_exchange: transfers on the library's communication stream under the interior CFL
    int comm_loopback;      // rehearsal on one GPU: both "neighbours" of the slab are the slab itself
    int graph_steps;        // fcpt_run_steps: replay a captured hipGraph of one step (launch-bound narrow grids)
    int profile_stride;     // fcpt_profile_start: every n-th launch of the selected kernels is timed (an event pair costs ~3 us of stream time)
};
#define FCPT_OPTION_NAMES                                                                                        \
    X(transport_fused) X(transport_rows) X(transport_graded) X(transport_big) X(transport_ladder) X(transport_rank_grade) X(source_rows) X(source_graded) X(theta_rows) X(transport_fallback) X(transport_split)    \
    X(fused_source) X(march_source) X(march_source_adi) X(theta_march) X(theta_fused) X(cfl_rings) X(cfl_wide_blocks) X(cfl_fold_in_source) X(gate_in_boundary) X(cfl_split)   \
    X(source_ring_parts) X(fused_damping) X(inline_potential) X(cfl_thermal) X(bc_fold) X(bc_in_cfl) X(comm_overlap) X(comm_loopback) X(graph_steps) X(profile_stride)

// Everything a kernel needs: geometry, grids, parameters.  Passed by value.
struct Dev {
    int nr, nphi;
    double dphi, invdphi;
    // geometry
    CArr Rmed, Rinf, Rsup, Surf, InvRmed, InvRinf, InvSurf, InvDiffRmed, InvDiffRsup, InvDiffRsupRb;
    const double *cosphi, *sinphi; // cos/sin(dphi * j), j < nphi (SideEuler.cpp:60-63)
    CArr cs_ring, nu_ring; // isothermal sound speed and alpha-viscosity per ring
    // per-ring reciprocals used by the marching source kernel (host-evaluated, IEEE):
    //   g_inv_dxt_src = 2/(dphi (Rsup+Rinf)), g_inv_rsum = 1/(Rsup+Rinf), g_inv_drmed2 = 1/(Rmed[i]^2-Rmed[i-1]^2),
    //   g_inv_dra2 = 1/(Rinf[i+1]^2-Rinf[i]^2), g_inv_rmsum = 1/(Rmed[i]+Rmed[i-1])
    CArr g_inv_dxt_src, g_inv_rsum, g_inv_drmed2, g_inv_dra2, g_inv_rmsum;
    //   g_dxtheta = dphi Rmed, g_inv_dxtheta = 1/(dphi Rmed), g_dr_invsurf = (Rsup-Rinf) InvSurf, g_r_omega = Rmed OmegaFrame
    CArr g_dxtheta, g_inv_dxtheta, g_dr_invsurf, g_r_omega;
    CArr g_inv_omk;  // 1 / Omega_K(Rmed[i])
    int q_skip;       // 1: k_source_march_adi leaves Q+ and Q- unwritten (only their difference, which the CFL kernel reads): set for the steps of fcpt_run_steps that are not its last
    int lazy_derived; // ideal EOS + marching source step: c_s, H, nu, T, P are formed in registers where needed,
                      // the grids are only materialised for callers that ask for them
    int inline_potential; // ... and the potential of the Euler step inside k_source_march_adi
    // state
    double *sigma, *vrad, *vazi, *energy;
    double *vrad_b, *vazi_b; // intermediate velocities of the fused source step
    double *energy_b;        // energy after the marching source step (ideal EOS)
    // derived
    double *pressure, *soundspeed, *scale_height, *viscosity, *temperature, *potential;
    // reference (t = 0) copies
    double *sigma0, *vrad0, *vazi0, *energy0;
    // source-step scratch
    double *qr, *qphi, *divv, *trr, *tpp, *trp /* (nr+1) rows */, *qplus, *qminus;
    // StabilizeViscosity (viscosity.cpp:256-348): correction factors c1_phi, c1_r; null when it is 0
    double *cfac_phi, *cfac_r;
    double *massflow; // MASSFLOW (data.h:76), (nr+1) rows; null without WriteMassFlow
    // BodyForceFromPotential: no -- ACCEL_RADIAL / ACCEL_AZIMUTHAL of CalculateAccelOnGas (Pframeforce.cpp:96-189),
    // (nr+1) rows of which 1 .. nr-1 are written; null otherwise
    double *accel_r, *accel_az;
    int accel_force; // 1: the source step takes the bodies' pull from these grids, not from the potential's gradient
    CArr g_ra3;    // pow(Rinf[i], 3)
    int stabilize; // 0: off, 1: damp the viscous velocity update, 2: limit the time step
    // transport: momenta / density / energy, two sets (A: after pass 1, B: radial + final)
    double *rmpA, *rmmA, *lpA, *lmA, *sigA, *eA;
    double *rmpB, *rmmB, *lpB, *lmB, *sigB, *eB;
    double *vmean;  // per ring <v_phi>
    double *vconst; // per ring constant residual velocity
    int *nshift;    // per ring integer shift
    CArr vmean_c, vconst_c; // the same arrays for kernels that only read them
    const SrcRow *src_tab;
    const RadRow *rad_tab;
    const ThetaRow *theta_tab;
    ShiftRow *shift_tab;
    const DampRow *damp_tab;
    const int *sm_sched;    // wavefronts of the marching source kernels in dispatch order: (segment, first ring, one past the last, 0)
    int sm_sched_n;         // ... how many, a multiple of 4 (0: equal chunks of source_rows() rings)
    const int *tf_sched;    // wavefronts of k_transport_fused in dispatch order: (tile, first ring, one past the last, 0)
    int tf_sched_n;         // ... how many (0: equal chunks of transport_rows() rings)
    int *shift_jump;        // set by k_transport_fused when |Nshift[i]-Nshift[i-1]| > 1 somewhere: the unfused kernels take over
    // wave damping folded into the end of the transport step: per-ring factor f = ((r-r_lim)/(r_edge-r_lim))^2
    // and time scale tau for scalar (dfac_s/dtau_s, nr) and vector (dfac_v/dtau_v, nr+1) grids, and the
    // per-ring damping type of each field (0 none, 1 reference, 2 zero)
    CArr dfac_s, dtau_s, dfac_v, dtau_v;
    CArrI dtype_vr, dtype_va, dtype_sig, dtype_e;
    int damp_in_step;
    CArrI nshift_c;
    double *cfl_part; // per-block maxima of the CFL reduction
    double *cfl_thermal; // ideal EOS: invdt1^2 + invdt5^2 + invdt6^2 per cell, left by the marching transport (null: off)
    int cfl_thermal_on;  // ... and valid for the current state: k_cfl_rings reads it instead of Sigma, e, Q+, Q-
    double *qdiff;       // ideal EOS: Q+ - Q- of the last kick, written by k_source_march_adi beside Q+ and Q-
    int qdiff_on;        // ... and current: the CFL kernel reads it instead of the two grids
    double *cfl_export;  // non-null: the final fold also leaves the slab's CFL step here (the MIN all-reduce's operand)
    int *cfl_tickets; // 1 + CFL_TICKET_LANES counters of the "last workgroup folds" scheme (zero between launches)
    // per-ring partial sums of v_phi left by k_source_march for the transport's ring mean
    // (pstride entries per ring, src_ring_nparts of them valid, 0 = not available)
    double *ring_part;
    int ring_pstride, src_ring_nparts;
    DevClock *clk;
    // split
    int zero_no_ghost, one_no_ghost_vr, max_no_ghost, maxmo_no_ghost_vr, first_active, active_size;
    int is_first, is_last;
    // parameters
    int adiabatic, art_visc, art_visc_dissipation, heating_viscous, fast_transport, limiter, leapfrog;
    int alpha_viscosity;
    double gamma, mu, Rgas, G, Mc, sigma_sb, c_light, aspect_ratio, flaring_index;
    double tmin, tmax, sigma_floor_abs, sigma_floor_rel, sigma0_val;
    // cooling terms of SubStep3 (calculate_qminus)
    int cooling_surface, opacity, cooling_beta, cooling_beta_reference, cooling_at_init;
    int heating_star;        // some body irradiates (parameters::heating_star_enabled)
    double btemp[FCPT_MAX_BODIES], bradius[FCPT_MAX_BODIES], bramp[FCPT_MAX_BODIES];
    int kick_time_shift; // 1: second leapfrog kick -- its SubStep3 runs at clk.time - clk.dt (midstep_time)
    double cooling_radiative_factor, kappa_const, kappa_factor, tau_factor, tau_min, density_factor;
    double cooling_beta_value, cooling_beta_ramp_up, temperature_cgs, density_cgs, opacity_cgs;
    CArr g_omk; // Omega_K(Rmed[i])
    double emin_fac, emax_fac; // T_min|max / mu * R / (gamma - 1): energy floor / ceiling per unit Sigma
    double b_fac;              // mu (gamma - 1) / R of SubStep3's alpha
    double alpha_fac;          // 2 * 4 sigma_SB / c of the same (one multiply per cell instead of an IEEE division)
    double alpha, nu_const, radial_viscosity_factor, art_visc_factor, heating_viscous_factor;
    double omega_frame, thickness_smoothing, cfl, cfl_max_var, heating_cooling_cfl_limit;
    double monitor_timestep;
    int bc_sigma[2], bc_energy[2], bc_vrad[2], bc_vaz[2];
    double kep_vaz[2], kep_vrad[2];
    // bodies
    int nbodies;
    double bx[FCPT_MAX_BODIES], by[FCPT_MAX_BODIES], bm[FCPT_MAX_BODIES], brsm[FCPT_MAX_BODIES];
    double indirect_x, indirect_y;
    Options opt; // host side only: which kernels the launchers pick
};

} // namespace fcpt

#endif
