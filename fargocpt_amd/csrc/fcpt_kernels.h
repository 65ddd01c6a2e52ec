// Launch functions of fcpt_kernels.hip (all asynchronous on the given stream).
#ifndef FCPT_KERNELS_H
#define FCPT_KERNELS_H

#include <vector>
#include <hip/hip_runtime.h>

#include "fcpt_internal.h"

namespace fcpt {

// One id per __global__ function family; kKernelNames[id] is the function's name as rocprofv3 prints it (without
// namespace and template arguments), so that fcpt_profile_stop's table can be matched to profiles/*.csv by name.
// KID_CLOCK stands for the single-thread k_clock_* kernels (set_dt, scale_dt, advance, policy, policy_ptr, export_cfl).
enum KernelId {
    KID_POTENTIAL, KID_SOURCE_VR, KID_SOURCE_VA, KID_COMPRESSION, KID_TW_Q, KID_TW_VA, KID_TW_VR,
    KID_SN_Q, KID_SN_E, KID_SN_VR, KID_SN_VA, KID_TRANGE, KID_ADI_CS_H, KID_ISO_CS_H,
    KID_VISCOSITY, KID_PRESSURE, KID_TEMPERATURE, KID_STRESS_DIAG, KID_STRESS_RPHI, KID_VISC_VA,
    KID_VISC_VR, KID_QPLUS, KID_SUBSTEP3, KID_BOUNDARY, KID_DAMPING, KID_TRANSPORT_RADIAL,
    KID_RING_MEAN, KID_THETA1, KID_THETA2, KID_VELOCITIES, KID_CFL_INIT, KID_CFL_CELLS, KID_CLOCK,
    KID_SRC_FUSED, KID_AV_FUSED, KID_VISC_FUSED, KID_SOURCE_MARCH, KID_THETA_MARCH,
    KID_TRANSPORT_FUSED, KID_MASSFLOW, KID_CFL_RINGS, KID_THETA_GATED_BOUNDARY, KID_EXCHANGE_COPY,
    KID_DISK_ON_BODY, KID_VISC_FACTORS, KID_SOURCE_MARCH_ADI, KID_SOURCE_MARCH_ADI_WIDE,
    KID_TRANSPORT_FUSED_THERM, KID_TRANSPORT_FUSED_WIDE, KID_STEP_COOP, KID_ACCEL_ON_GAS,
    KID_SOURCE_MARCH_ADI_ACC, KID_TRANSPORT_RADIAL_MEANS, KID_CFL_RINGS_BC, KID_COUNT
};
static_assert(KID_COUNT <= 64, "fcpt_profile_start selects kernels with a 64-bit mask");
extern const char *const kKernelNames[KID_COUNT];

// HIP-event stopwatch around selected kernel launches (on the launch stream).
struct Profiler {
    unsigned long long mask = 0;
    std::vector<hipEvent_t> events;
    std::vector<int> ids;
    int used = 0;
    int open_id = -1;
    int stride = 1, seen = 0; // every stride-th launch of the selected kernels carries the event pair (option profile_stride)
    void begin(int id, hipStream_t st);
    void end(int id, hipStream_t st);
};
extern thread_local Profiler *g_prof;

void launch_potential(const Dev &P, hipStream_t st);
void launch_accel_on_gas(const Dev &P, hipStream_t st);
void launch_body_force(const Dev &P, hipStream_t st); // the potential, or the accelerations (simulation.cpp:167-175)
void launch_source(const Dev &P, hipStream_t st);
void launch_artificial_viscosity(const Dev &P, hipStream_t st);
void launch_recalculate_viscosity(const Dev &P, hipStream_t st);
void launch_viscosity_field(const Dev &P, hipStream_t st);
void launch_iso_cs_h(const Dev &P, const double *cs_ring, hipStream_t st);
void launch_stress(const Dev &P, hipStream_t st);
void launch_visc_factors(const Dev &P, hipStream_t st);
void launch_viscous_update(const Dev &P, hipStream_t st);
void launch_substep3(const Dev &P, int update_energy, hipStream_t st);
void launch_boundary(const Dev &P, hipStream_t st);
void launch_selftest_half_limiter(int type, long long n, const double *a, const double *b, double *out, hipStream_t st);
void launch_exchange_copy(const Dev &P, double *inner, double *outer, int unpack, hipStream_t st);
void launch_damping(const Dev &P, double *q, double *q0, const double *radius, const DampRange &r,
                    int is_density, hipStream_t st);
// Where Transport() left the new state: the marching kernels cannot work in place (neighbouring
// wavefronts still read the old rings), so they write Sigma / e to the A scratch grids and the
// velocities to whichever of vrad|vrad_b, vazi|vazi_b was not the input; the caller swaps the
// context's pointers accordingly.
struct TransportResult {
    int marched;  // > 0: a marching kernel ran (new state complete, clock advanced)
    double *sigma, *energy, *vrad, *vazi;
    int split;    // only a part of the chunks was marched
    int thermal;  // the kernel left the cell-local CFL terms of the new state in Dev::cfl_thermal
    int gated_pending; // launch_transport(defer_gated): the gated azimuthal launch of the fallback is still to be queued (launch_gated_theta)
};
// the arguments of that launch, kept by the caller until the final boundary call of the step
struct GatedTheta {
    Dev P, Wm;
};
void launch_gated_theta(const GatedTheta &g, const Dev *boundary_view, hipStream_t st); // boundary_view != null: + the boundary call, one launch
enum { TRANSPORT_ALL = 0, TRANSPORT_EDGES = 1, TRANSPORT_INTERIOR = 2 };
TransportResult launch_transport(const Dev &P, const Dev &W, hipStream_t st, int part = TRANSPORT_ALL, GatedTheta *defer_gated = nullptr);
bool transport_can_split(const Dev &P, bool shear_safe);
std::vector<int> source_schedule(const Dev &P);
void selftest_chunk_tables(int nr, int nphi, int n_cu, int adiabatic, int damp_inner, int damp_outer, const Options &opt,
                           std::vector<int> &transport, std::vector<int> &source);
std::vector<int> transport_schedule(const Dev &P, const std::vector<int> &slow_rings, const std::vector<int> *lengths);
void launch_shift_means(const Dev &P, hipStream_t st);
void launch_massflow(const Dev &P, hipStream_t st);
void launch_substep3_cooling_only(const Dev &P, hipStream_t st);
void launch_disk_on_body(const Dev &P, double x, double y, double r_object, double smoothing_fixed, double r_sm, double *out,
                         hipStream_t st);
void launch_source_fused(const Dev &P, hipStream_t st);
int launch_source_march(const Dev &P, hipStream_t st, bool fold_bc, bool *bc_folded, bool fold_cfl = false);
bool source_march_applies(const Dev &P);
void launch_cfl_final(const Dev &P, int apply_policy, hipStream_t st);
bool cfl_by_rings(const Dev &P);
void launch_viscous_fused(const Dev &P, hipStream_t st);
void launch_derived(const Dev &P, hipStream_t st);
void launch_pressure(const Dev &P, hipStream_t st);
void launch_temperature(const Dev &P, hipStream_t st);
void launch_cfl(const Dev &P, int apply_policy, hipStream_t st, bool interior_done = false);
bool launch_cfl_interior(const Dev &P, hipStream_t st);
bool cfl_bc_mergeable(const Dev &P);
void launch_cfl_bc(const Dev &P, int apply_policy, hipStream_t st);
void launch_clock_set_dt(DevClock *clk, double dt, hipStream_t st);
void launch_clock_scale_dt(DevClock *clk, int mode, double dt, double factor, hipStream_t st);
void launch_clock_export_cfl(DevClock *clk, double *out, hipStream_t st);
void launch_clock_policy_ptr(DevClock *clk, double cfl_max_var, const double *cfl_global, hipStream_t st);
void launch_clock_advance(DevClock *clk, hipStream_t st);
void launch_clock_policy(DevClock *clk, double cfl_max_var, int use_device_cfl, double cfl_global,
                         hipStream_t st);

} // namespace fcpt
#endif
