/*
 * fargo_oracle.h -- TEST INFRASTRUCTURE ONLY.
 *
 * CPU restatement (plain C99) of the FargoCPT gas update, used as the parity
 * oracle for the HIP path.  Only tests/, __graft_entry__.smoke() and bench.py's
 * cpu_baseline leg may load this library; the product (fargocpt_amd/) never
 * does.  It shares only the descriptor / split / clock *types* with the product
 * header include/fargocpt_hip.h, so the same configuration drives both sides.
 *
 * Every function in fargo_oracle.c cites the reference file:line it follows.
 */
#ifndef FARGO_ORACLE_H
#define FARGO_ORACLE_H

#include "../include/fargocpt_hip.h"

#ifdef __cplusplus
extern "C" {
#endif

typedef struct orc_ctx orc_ctx;

int orc_split_domain(const fcpt_desc *d, fcpt_split *out);
int orc_radii(const fcpt_desc *d, double *radii);
int orc_initial_fields(fcpt_desc *d, const double *radii, double *sigma, double *vrad,
                       double *vazi, double *energy);

int orc_create(const fcpt_desc *d, const double *radii, orc_ctx **out);
int orc_destroy(orc_ctx *c);
int orc_get_split(const orc_ctx *c, fcpt_split *out);
int orc_get_clock(const orc_ctx *c, fcpt_clock *out);
int orc_set_clock(orc_ctx *c, const fcpt_clock *in);
int orc_upload(orc_ctx *c, int32_t field, const double *host);
int orc_download(orc_ctx *c, int32_t field, double *host);
int orc_set_bodies(orc_ctx *c, int32_t n, const double *x, const double *y, const double *mass,
                   const double *rsm, double indirect_x, double indirect_y);
int orc_set_bodies_midstep(orc_ctx *c, int32_t n, const double *x, const double *y, const double *mass,
                           const double *rsm);
int orc_init_physics(orc_ctx *c);
int orc_set_body_irradiation(orc_ctx *c, int32_t n, const double *temperature, const double *radius,
                             const double *rampup_time);
int orc_disk_on_body_accel(orc_ctx *c, double x, double y, double r_object, double smoothing_fixed,
                           double cubic_smoothing_radius, double out[4]);
int orc_cfl(orc_ctx *c, double *dt_local);
int orc_calculate_timestep(orc_ctx *c, double cfl_dt_global, double *dt);
int orc_snap_to_monitor(const orc_ctx *c, double cfl_dt, double *step_dt);
int orc_step(orc_ctx *c, double dt);
int orc_exchange_count(const orc_ctx *c, uint64_t *count);
int orc_exchange_pack(orc_ctx *c, double *send_inner, double *send_outer);
int orc_exchange_unpack(orc_ctx *c, const double *recv_inner, const double *recv_outer);
int orc_post(orc_ctx *c, double dt);
int orc_apply_boundary(orc_ctx *c, double dt, int32_t final);
int orc_recalculate_derived(orc_ctx *c);
int orc_run_steps(orc_ctx *c, int64_t nsteps, int32_t snap, int64_t *nsteps_done);

/* geometry arrays for tests: which = 0 Rmed, 1 Rinf, 2 Rsup, 3 Surf, 4 InvDiffRmed,
 * 5 InvDiffRsup; out must hold nr + FCPT_GEOM_PAD + 1 doubles */
int orc_geometry(const orc_ctx *c, int32_t which, double *out);
/* per-ring integer shifts of the last transport call (TransportEuler.cpp:207-236) */
int orc_last_nshift(const orc_ctx *c, int32_t *out);

#ifdef __cplusplus
}
#endif
#endif
