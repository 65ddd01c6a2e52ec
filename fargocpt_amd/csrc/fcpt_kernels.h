// Launch functions of fcpt_kernels.hip (all asynchronous on the given stream).
#ifndef FCPT_KERNELS_H
#define FCPT_KERNELS_H

#include <hip/hip_runtime.h>

#include "fcpt_internal.h"

namespace fcpt {

void launch_potential(const Dev &P, hipStream_t st);
void launch_source(const Dev &P, hipStream_t st);
void launch_artificial_viscosity(const Dev &P, hipStream_t st);
void launch_recalculate_viscosity(const Dev &P, hipStream_t st);
void launch_viscosity_field(const Dev &P, hipStream_t st);
void launch_iso_cs_h(const Dev &P, const double *cs_ring, hipStream_t st);
void launch_stress(const Dev &P, hipStream_t st);
void launch_viscous_update(const Dev &P, hipStream_t st);
void launch_substep3(const Dev &P, int update_energy, hipStream_t st);
void launch_boundary(const Dev &P, hipStream_t st);
void launch_damping(const Dev &P, double *q, double *q0, const double *radius, const DampRange &r,
                    int is_density, hipStream_t st);
void launch_transport(const Dev &P, hipStream_t st);
void launch_derived(const Dev &P, hipStream_t st);
void launch_pressure(const Dev &P, hipStream_t st);
void launch_temperature(const Dev &P, hipStream_t st);
void launch_cfl(const Dev &P, hipStream_t st);
void launch_clock_set_dt(DevClock *clk, double dt, hipStream_t st);
void launch_clock_advance(DevClock *clk, hipStream_t st);
void launch_clock_policy(DevClock *clk, double cfl_max_var, int use_device_cfl, double cfl_global,
                         hipStream_t st);

} // namespace fcpt
#endif
