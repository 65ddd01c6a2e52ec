/*
 * fargo_oracle.c -- TEST INFRASTRUCTURE ONLY (see fargo_oracle.h).
 *
 * Plain-C restatement of the per-timestep gas update of kimweiskopf/fargocpt
 * (reference snapshot 2025-03-21).  Loop structure, loop bounds, operand order
 * and in-place update order follow the reference so that rounding matches a
 * `-O2 -ffp-contract=off` build of it; each function cites the file:line it
 * restates (paths relative to the reference's src/).
 *
 * PARITY PINNING: the reference binary cannot be built in this image (it needs
 * GSL, which is absent, and links the whole program: yaml-cpp, LLNL units,
 * REBOUND, MPI).  The oracle is therefore pinned by the reference's own
 * known-answer tests -- test/shockTube (analytic_shock.dat + thresholds of
 * check_results.py:18-23), test/spreading_ring (calc_deviation.py:43-66),
 * test/TemperatureTest and test/irradiation (their check_results.py) --
 * see tests/test_oracle_known_answers.py.  PARITY UNPINNED for one switch:
 * StabilizeViscosity 1|2 is never on in the reference's tests; its correction
 * factors are checked against the Jacobian diagonal of the viscous force
 * (same file), where they are applied is a restatement only.
 *
 * Out of scope (not restated): N-body integration, self-gravity, FLD, dust,
 * S-curve cooling and the Bell opacity, variable-gamma EOS,
 * BodyForceFromPotential=no, composite BCs (custom / centerofmass), the MassDelta
 * boundary bookkeeping (the MASSFLOW grid of WriteMassFlow is restated).
 */
#include "fargo_oracle.h"

#include <float.h>
#include <math.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>

#ifndef M_PI
#define M_PI 3.14159265358979323846
#endif

struct orc_ctx {
    fcpt_desc d;
    fcpt_split s;
    int nr, nphi;
    double dphi, invdphi;
    double *radii; /* global interfaces, nr_global + PAD + 1 */
    /* local geometry, nr + PAD + 1 entries each (global.h:62-79) */
    double *Rmed, *Rinf, *Rsup, *Surf, *InvRmed, *InvRinf, *InvSurf, *InvDiffRmed, *InvDiffRsup,
        *InvDiffRsupRb;
    /* cell finder (find_cell_id.cpp:47-106) */
    double cf_growth, cf_inv_log_growth, cf_opt_const, cf_first_cell;

    /* state grids (data.h) */
    double *sigma, *vrad, *vazi, *energy;
    double *pressure, *soundspeed, *scale_height, *viscosity, *temperature, *potential;
    double *accel_r, *accel_az; /* ACCEL_RADIAL, ACCEL_AZIMUTHAL (data.cpp:49-58), (Nr+1) x Nphi */
    double *sigma0, *vrad0, *vazi0, *energy0;
    double *qr, *qphi, *divv, *trr, *tpp, *trp /* vector */, *qplus, *qminus, *density_int;
    double *nusig, *nusig_rp /* vector */, *cfac_phi, *cfac_r; /* StabilizeViscosity (viscosity.cpp:256-348) */
    double *tau_eff; /* kappa_eff (compute.cpp:41-87); 0 without surface cooling */
    double *massflow; /* MASSFLOW (data.h:76), vector grid; NULL without WriteMassFlow */
    double btemp[FCPT_MAX_BODIES], bradius[FCPT_MAX_BODIES], bramp[FCPT_MAX_BODIES]; /* irradiating bodies */
    int heating_star; /* parameters::heating_star_enabled */
    /* transport scratch (TransportEuler.cpp:32-50) */
    double *rmp, *rmm, *lp, *lm, *vres, *vmean, *work, *qrstar /* vector */, *densstar /* vector */,
        *tempshift, *dq;
    int *nshift;
    int *nosplit;
    /* cfl scratch (cfl.cpp:11-12) */
    double *cfl_vmean, *cfl_vres;
    int viscosity_calculated; /* static bool of viscosity::update_viscosity */
    int big;                  /* grid large enough for OpenMP teams to pay off */

    /* bodies */
    int nbodies;
    double bx[FCPT_MAX_BODIES], by[FCPT_MAX_BODIES], bm[FCPT_MAX_BODIES], brsm[FCPT_MAX_BODIES];
    double indirect_x, indirect_y;
    /* bodies at the mid-step time of a leapfrog step (simulation.cpp:363-366) */
    int has_mid;
    double mx[FCPT_MAX_BODIES], my[FCPT_MAX_BODIES], mm[FCPT_MAX_BODIES], mrsm[FCPT_MAX_BODIES];

    fcpt_clock clk;
};

#define IDX(c, i, j) ((size_t)(i) * (size_t)(c)->nphi + (size_t)(j))

/* ------------------------------------------------------------------------ */
/* split.cpp:34-88 SplitDomain, standard branch                              */
int orc_split_domain(const fcpt_desc *d, fcpt_split *o)
{
    if (!d || !o || d->nranks < 1 || d->rank < 0 || d->rank >= d->nranks)
        return FCPT_EINVAL;
    const int NR = d->nr_global, P = d->nranks, rank = d->rank;
    const int size_low = NR / P, size_high = size_low + 1, remainder = NR % P;
    if (P > 1 && size_low < 2 * FCPT_OVERLAP)
        return FCPT_ESPLIT;
    int imin, imax;
    if (rank < remainder) {
        imin = size_high * rank;
        imax = imin + size_high - 1;
    } else {
        imin = size_high * remainder + (rank - remainder) * size_low;
        imax = imin + size_low - 1;
    }
    if (rank > 0)
        imin -= FCPT_OVERLAP;
    if (rank < P - 1)
        imax += FCPT_OVERLAP;
    const int nr = imax - imin + 1;
    const int first = rank == 0, last = rank == P - 1;
    o->nr = nr;
    o->imin = imin;
    o->imax = imax;
    o->zero_no_ghost = first ? 1 : 0;
    o->one_no_ghost_vr = first ? 2 : 1;
    o->max_no_ghost = nr - (last ? 1 : 0);
    o->maxmo_no_ghost_vr = nr + 1 - (last ? 2 : 1);
    o->zero_or_active = first ? 0 : FCPT_OVERLAP;
    o->radial_first_active = first ? FCPT_GHOSTCELLS_B : FCPT_OVERLAP;
    o->max_or_active = nr - (last ? 0 : FCPT_OVERLAP);
    o->radial_active_size = nr - (last ? FCPT_GHOSTCELLS_B : FCPT_OVERLAP);
    o->is_first = first;
    o->is_last = last;
    return FCPT_OK;
}

/* init.cpp:92-145 init_radialarrays, grid construction */
int orc_radii(const fcpt_desc *d, double *Radii)
{
    if (!d || !Radii || d->nr_global < 3)
        return FCPT_EINVAL;
    const int N = d->nr_global;
    const double RMIN = d->rmin, RMAX = d->rmax;
    switch (d->radial_spacing) {
    case FCPT_SPACING_LOGARITHMIC: {
        const double f = pow(RMAX / RMIN, 1.0 / ((double)N - 2.0));
        for (int n = 0; n <= N + FCPT_GEOM_PAD; ++n)
            Radii[n] = RMIN * pow(f, (double)n - 1.0);
        break;
    }
    case FCPT_SPACING_ARITHMETIC: {
        const double interval = (RMAX - RMIN) / (double)(N - 2.0);
        for (int n = 0; n <= N + FCPT_GEOM_PAD; ++n)
            Radii[n] = RMIN + interval * (double)(n - 1.0);
        break;
    }
    case FCPT_SPACING_EXPONENTIAL: {
        const double cgf = pow(RMAX / RMIN, 1.0 / ((double)N - 2.0));
        const double first = RMIN * (cgf - 1.0) * d->exponential_cell_size_factor;
        const double f = (RMAX - RMIN) / first;
        double g = 1.02;
        const double Nr = (double)N - 2.0;
        for (int i = 0; i < 500000; ++i)
            g = g - ((pow(g, Nr) - g * f + f - 1)) / (Nr * pow(g, Nr - 1.0) - f);
        for (int n = 0; n <= N + FCPT_GEOM_PAD; ++n)
            Radii[n] = RMIN + first * (pow(g, (double)n - 1.0) - 1.0) / (g - 1.0);
        break;
    }
    default:
        return FCPT_EINVAL;
    }
    return FCPT_OK;
}

/* init.cpp:169-225 local geometry arrays */
static void init_geometry(orc_ctx *c)
{
    const int n1 = c->nr + FCPT_GEOM_PAD; /* NRadial + search_buffer */
    const int imin = c->s.imin;
    const double *Radii = c->radii;
    for (int i = 0; i < n1; ++i) {
        /* local index i + imin may run past the global pad for interior slabs:
         * clamp reads to the last available interface (values past the slab are
         * never used by the update) */
        int gi = i + imin;
        const int gmax = c->d.nr_global + FCPT_GEOM_PAD;
        if (gi + 1 > gmax)
            gi = gmax - 1;
        c->Rinf[i] = Radii[gi];
        c->Rsup[i] = Radii[gi + 1];
        const double rs = c->Rsup[i], ri = c->Rinf[i];
        c->Rmed[i] = 2.0 / 3.0 * (rs * rs * rs - ri * ri * ri);
        c->Rmed[i] = c->Rmed[i] / (rs * rs - ri * ri);
        c->Surf[i] = M_PI * (rs * rs - ri * ri) / (double)c->nphi;
        c->InvRmed[i] = 1.0 / c->Rmed[i];
        c->InvSurf[i] = 1.0 / c->Surf[i];
        c->InvDiffRsup[i] = 1.0 / (rs - ri);
        c->InvDiffRsupRb[i] = 1.0 / ((rs - ri) * c->Rmed[i]);
        c->InvRinf[i] = 1.0 / ri;
    }
    c->InvDiffRmed[0] = 0.0; /* never written by the reference; t_radialarray is zero-filled */
    for (int i = 1; i < c->nr + 1; ++i)
        c->InvDiffRmed[i] = 1.0 / (c->Rmed[i] - c->Rmed[i - 1]);

    /* find_cell_id.cpp:47-94 */
    const fcpt_desc *d = &c->d;
    const double N = (double)d->nr_global;
    c->cf_growth = c->cf_inv_log_growth = c->cf_opt_const = c->cf_first_cell = 0.0;
    if (d->radial_spacing == FCPT_SPACING_LOGARITHMIC) {
        const double g = pow(d->rmax / d->rmin, 1.0 / (N - 2.0));
        c->cf_growth = g;
        c->cf_opt_const = 3.0 / 2.0 / d->rmin * (1 - pow(g, 2.0)) / (1 - pow(g, 3.0));
        c->cf_inv_log_growth = 1.0 / log(g);
    } else if (d->radial_spacing == FCPT_SPACING_ARITHMETIC) {
        c->cf_growth = (N - 2.0) / (d->rmax - d->rmin);
    } else {
        const double cgf = pow(d->rmax / d->rmin, 1.0 / (N - 2.0));
        const double first = d->rmin * (cgf - 1.0) * d->exponential_cell_size_factor;
        const double f = (d->rmax - d->rmin) / first;
        double g = 1.02;
        const double Nr = N - 2.0;
        for (int i = 0; i < 500000; ++i)
            g = g - ((pow(g, Nr) - g * f + f - 1)) / (Nr * pow(g, Nr - 1.0) - f);
        c->cf_growth = g;
        c->cf_first_cell = first;
        c->cf_inv_log_growth = 1.0 / log(g);
        c->cf_opt_const = (g - 1.0) / first;
    }
}

/* find_cell_id.cpp:217-253,360-390 get_rmed_id (+ NDEBUG-style clamp of :207-214) */
static int get_rmed_id(const orc_ctx *c, double r)
{
    int id;
    const int imin = c->s.imin;
    if (c->d.radial_spacing == FCPT_SPACING_LOGARITHMIC) {
        id = (int)floor(log(r * c->cf_opt_const) * c->cf_inv_log_growth) - imin + 1;
    } else if (c->d.radial_spacing == FCPT_SPACING_ARITHMETIC) {
        id = (int)floor((r - c->d.rmin) * c->cf_growth) - imin + 1;
        if (id >= 0 && id < c->nr + FCPT_GEOM_PAD && c->Rmed[id] > r)
            id--;
    } else {
        const double tmp = (r - c->d.rmin) * c->cf_opt_const + 1.0;
        id = (int)floor(log(tmp) * c->cf_inv_log_growth) - imin + 1;
        if (id >= 0 && id < c->nr + FCPT_GEOM_PAD && c->Rmed[id] > r)
            id--;
    }
    return id;
}

/* find_cell_id.cpp:265-285 get_rinf_id */
static int get_rinf_id(const orc_ctx *c, double r)
{
    const int imin = c->s.imin;
    if (c->d.radial_spacing == FCPT_SPACING_LOGARITHMIC)
        return (int)floor(log(r / c->d.rmin) * c->cf_inv_log_growth) - imin + 1;
    if (c->d.radial_spacing == FCPT_SPACING_ARITHMETIC)
        return (int)floor((r - c->d.rmin) * c->cf_growth) - imin + 1;
    const double tmp = (r - c->d.rmin) * c->cf_opt_const + 1.0;
    return (int)floor(log(tmp) * c->cf_inv_log_growth) - imin + 1;
}

/* find_cell_id.cpp:15-41 clamp_r_id_to_{rmed,radii}_grid */
static unsigned clamp_r_id(const orc_ctx *c, int id, int is_vector)
{
    const int mx = c->nr - (is_vector ? 0 : 1);
    if (id < 0)
        id = 0;
    else if (id > mx)
        id = mx;
    return (unsigned)id;
}

/* Theo.cpp:246-249 */
static double omega_kepler(const orc_ctx *c, double r)
{
    return sqrt(c->d.G * c->d.hydro_center_mass / (r * r * r));
}

/* ------------------------------------------------------------------------ */
/* initial conditions                                                        */

/* modified Bessel function I_nu(x), nu > 0, x >= 0: ascending series
 * sum_k (x/2)^(2k+nu) / (k! Gamma(k+nu+1)); all terms positive => stable.
 * Stands in for gsl_sf_bessel_Inu (init.cpp:381,398). */
static double bessel_inu(double nu, double x)
{
    if (x == 0.0)
        return 0.0;
    const double hx = 0.5 * x;
    const double lhx = log(hx);
    /* largest term is near k ~ x/2; work in logs relative to it to avoid overflow */
    double sum = 0.0;
    const double q = hx * hx;
    /* start from k=0 term in log form, accumulate with running ratio */
    double logterm0 = nu * lhx - lgamma(nu + 1.0);
    /* find the max log term for scaling */
    double kmax = floor(0.5 * (sqrt(nu * nu + 4.0 * q) - nu));
    if (kmax < 0)
        kmax = 0;
    double logmax = (2.0 * kmax + nu) * lhx - lgamma(kmax + 1.0) - lgamma(kmax + nu + 1.0);
    double term = exp(logterm0 - logmax);
    for (int k = 0; k < 100000; ++k) {
        sum += term;
        const double ratio = q / ((double)(k + 1) * ((double)(k + 1) + nu));
        term *= ratio;
        if (k > kmax && term < sum * 1e-18)
            break;
    }
    return sum * exp(logmax);
}

/* Theo.cpp:128-155,166-180 */
static double support_azi_pressure(const fcpt_desc *d, double R)
{
    const double h = d->aspect_ratio * pow(R, d->flaring_index);
    return (2.0 * d->flaring_index - 1.0 - d->sigma_slope) * (h * h);
}
static double support_azi_smoothing_derivative(const fcpt_desc *d, double R)
{
    const double F = d->flaring_index;
    const double h = d->aspect_ratio * pow(R, F);
    const double eps = d->thickness_smoothing;
    const double he2 = (h * eps) * (h * eps);
    const double s = sqrt(1 + he2);
    return (1.0 + (F + 1.0) * he2) / (s * s * s);
}
static double smoothed_v_az(const fcpt_desc *d, double R, double M)
{
    const double support = support_azi_smoothing_derivative(d, R) + support_azi_pressure(d, R);
    const double vk_2 = d->G * M / R;
    return sqrt(vk_2 * support);
}
/* Theo.cpp:86-99 */
static double initial_energy(const fcpt_desc *d, double R, double M)
{
    const double h0 = d->aspect_ratio, F = d->flaring_index, S = d->sigma_slope;
    return 1.0 / (d->adiabatic_index - 1.0) * d->sigma0 * (h0 * h0) * pow(R, -S - 1.0 + 2.0 * F) *
           d->G * M;
}
/* Theo.cpp:215-243 */
static double initial_viscous_radial_speed(const fcpt_desc *d, double R, double M)
{
    if (d->viscous_alpha > 0) {
        const double sqrt_gamma = d->eos == FCPT_EOS_IDEAL ? sqrt(d->adiabatic_index) : 1.0;
        const double v_k = sqrt(d->G * M / R);
        const double h = d->aspect_ratio * pow(R, d->flaring_index);
        const double cs = sqrt_gamma * h * v_k;
        const double H = h * R;
        const double nu = d->viscous_alpha * cs * H;
        return -3.0 * nu / R * (-d->sigma_slope + 2.0 * d->flaring_index + 1.0);
    }
    const double nu = d->constant_viscosity;
    return -3.0 * nu / R * (-d->sigma_slope + .5);
}

/* viscosity/viscous_radial_speed.cpp:39-206 (profile cut-offs are out of scope) */
static double vs_get_sigma(const fcpt_desc *d, double R)
{
    double density = d->sigma0 * pow(R, -d->sigma_slope);
    const double density_floor = d->sigma_floor * d->sigma0;
    return density > density_floor ? density : density_floor;
}
static double vs_get_nu2(const fcpt_desc *d, double R, double M, double Sigma)
{
    const double v_k = sqrt(d->G * M / R);
    const double h = d->aspect_ratio * pow(R, d->flaring_index);
    double cs_adb, H;
    if (d->eos == FCPT_EOS_IDEAL) {
        const double gamma = d->adiabatic_index;
        double energy = 1.0 * 1.0 / (gamma - 1.0) * Sigma * ((h * v_k) * (h * v_k));
        const double e_floor = d->minimum_temperature * Sigma / d->mu * d->Rgas / (gamma - 1.0);
        const double e_ceil = d->maximum_temperature * Sigma / d->mu * d->Rgas / (gamma - 1.0);
        energy = energy > e_floor ? energy : e_floor;
        energy = energy < e_ceil ? energy : e_ceil;
        cs_adb = sqrt(gamma * (gamma - 1.0) * energy / Sigma);
        const double cs_iso = sqrt((gamma - 1.0) * energy / Sigma);
        const double omega_k = v_k / R;
        H = cs_iso / omega_k;
    } else {
        cs_adb = h * v_k;
        H = h * R;
    }
    return d->viscous_alpha * cs_adb * H;
}
typedef double (*vs_fn)(const fcpt_desc *, double, double);
static double vs_derive(const fcpt_desc *d, double r, double mass, vs_fn f)
{
    const double x = r;
    const double h = 8.0e-4 * x;
    const double f1 = -1.0 * f(d, x + 2.0 * h, mass);
    const double f2 = 8.0 * f(d, x + h, mass);
    const double f3 = -8.0 * f(d, x - h, mass);
    const double f4 = 1.0 * f(d, x - 2.0 * h, mass);
    return (f1 + f2 + f3 + f4) / (12.0 * h);
}
static double vs_get_w(const fcpt_desc *d, double r, double mass)
{
    return smoothed_v_az(d, r, mass) / r;
}
static double vs_get_r2_w(const fcpt_desc *d, double r, double mass)
{
    return (r * r) * vs_get_w(d, r, mass);
}
static double vs_get_nu_S_r3_dwdr(const fcpt_desc *d, double r, double mass)
{
    const double dw_dr = vs_derive(d, r, mass, vs_get_w);
    const double Sigma = vs_get_sigma(d, r);
    const double nu = vs_get_nu2(d, r, mass, Sigma);
    return nu * Sigma * (r * r * r) * dw_dr;
}
static double vs_get_vr(const fcpt_desc *d, double r, double mass)
{
    const double num = 1.0 / r * vs_derive(d, r, mass, vs_get_nu_S_r3_dwdr);
    const double Sigma = vs_get_sigma(d, r);
    const double den = Sigma * vs_derive(d, r, mass, vs_get_r2_w);
    return num / den;
}

/* init.cpp:255-343 init_physics up to (not including) init_euler/BCs:
 * init_gas_density (:937-1004), spreading ring (:358-413), shock tube (:423-441),
 * renormalize_sigma_and_report (:1150-1185), init_gas_energy (:1257-1300),
 * init_gas_velocities (:1616-1631,:1725-1772). */
int orc_initial_fields(fcpt_desc *d, const double *Radii, double *sigma, double *vrad,
                       double *vazi, double *energy)
{
    fcpt_split s;
    int rc = orc_split_domain(d, &s);
    if (rc)
        return rc;
    if (!Radii || !sigma || !vrad || !vazi)
        return FCPT_EINVAL;
    const int nr = s.nr, nphi = d->nphi, imin = s.imin;
    const int adi = d->eos == FCPT_EOS_IDEAL;
    if (adi && !energy)
        return FCPT_EINVAL;
    double *Rmed = (double *)malloc(sizeof(double) * (size_t)(nr + 1));
    double *Rinf = (double *)malloc(sizeof(double) * (size_t)(nr + 1));
    double *Rsup = (double *)malloc(sizeof(double) * (size_t)(nr + 1));
    double *Surf = (double *)malloc(sizeof(double) * (size_t)(nr + 1));
    if (!Rmed || !Rinf || !Rsup || !Surf)
        return FCPT_ENOMEM;
    for (int i = 0; i < nr; ++i) {
        Rinf[i] = Radii[i + imin];
        Rsup[i] = Radii[i + imin + 1];
        const double rs = Rsup[i], ri = Rinf[i];
        Rmed[i] = 2.0 / 3.0 * (rs * rs * rs - ri * ri * ri);
        Rmed[i] = Rmed[i] / (rs * rs - ri * ri);
        Surf[i] = M_PI * (rs * rs - ri * ri) / (double)nphi;
    }
    const double M = d->hydro_center_mass;
    double sigma0 = d->sigma0;

    if (d->ic == FCPT_IC_SHOCKTUBE) {
        /* init.cpp:423-441 */
        const double rs0 = Radii[1], ri0 = Radii[0];
        double g0 = 2.0 / 3.0 * (rs0 * rs0 * rs0 - ri0 * ri0 * ri0);
        g0 = g0 / (rs0 * rs0 - ri0 * ri0); /* GlobalRmed[0] */
        for (int i = 0; i < nr; ++i)
            for (int j = 0; j < nphi; ++j) {
                double density = 1.0, e = 2.5;
                if (Rmed[i] - g0 > 0.5) {
                    density = 0.125;
                    e = 2.0 * 0.125;
                }
                sigma[(size_t)i * nphi + j] = density;
                if (energy)
                    energy[(size_t)i * nphi + j] = e;
            }
    } else {
        /* init.cpp:948-960 */
        for (int i = 0; i < nr; ++i)
            for (int j = 0; j < nphi; ++j) {
                const double density = sigma0 * pow(Rmed[i], -d->sigma_slope);
                const double density_floor = d->sigma_floor * sigma0;
                sigma[(size_t)i * nphi + j] = density > density_floor ? density : density_floor;
            }
        if (d->ic == FCPT_IC_SPREADING_RING) {
            /* init.cpp:358-413 */
            const double R0 = 1.0;
            int R0_id = 0;
            for (int i = 0; i < nr; ++i)
                if (Rsup[i] > R0 && R0 > Rinf[i])
                    R0_id = i;
            const double Disk_Mass = d->disk_mass;
            const double tau0 = 0.016;
            const double x0 = Rmed[R0_id] / R0;
            const double I0 = bessel_inu(0.25, 2.0 * x0 / tau0);
            const double Sigma0 = Disk_Mass / (M_PI * R0 * R0) * 1.0 / (tau0 * pow(x0, 0.25)) * I0 *
                                  exp(-(1.0 + x0 * x0) / tau0);
            for (int i = 0; i < nr; ++i)
                for (int j = 0; j < nphi; ++j) {
                    const double density_floor = Sigma0 * d->sigma_floor;
                    const double x = Rmed[i] / R0;
                    const double I = bessel_inu(0.25, 2.0 * x / tau0);
                    double density = Disk_Mass / (M_PI * R0 * R0) * 1.0 / (tau0 * pow(x, 0.25)) * I *
                                     exp(-(1.0 + x * x) / tau0);
                    density = density > density_floor ? density : density_floor;
                    sigma[(size_t)i * nphi + j] = density;
                    if (energy)
                        energy[(size_t)i * nphi + j] = 0.0;
                }
        }
        /* init.cpp:1063-1104 profile cutoff at the outer boundary, util.cpp:69-81 cutoff_outer */
        if (d->profile_cutoff_outer)
            for (int n_radial = 0; n_radial < nr; ++n_radial)
                for (int n_azimuthal = 0; n_azimuthal < nphi; ++n_azimuthal) {
                    const double r = Rmed[n_radial];
                    const double density_damped =
                        sigma[(size_t)n_radial * nphi + n_azimuthal] *
                        (1.0 / (1.0 + exp((r - d->profile_cutoff_point_outer) / d->profile_cutoff_width_outer)));
                    const double density_floor = d->sigma_floor * d->sigma0;
                    sigma[(size_t)n_radial * nphi + n_azimuthal] = fmax(density_damped, density_floor);
                }
        /* init.cpp:1106-1146 profile cutoff at the inner boundary, util.cpp:90-93 cutoff_inner */
        if (d->profile_cutoff_inner)
            for (int n_radial = 0; n_radial < nr; ++n_radial)
                for (int n_azimuthal = 0; n_azimuthal < nphi; ++n_azimuthal) {
                    const double r = Rmed[n_radial];
                    const double density_damped =
                        sigma[(size_t)n_radial * nphi + n_azimuthal] *
                        (1.0 / (1.0 + exp((d->profile_cutoff_point_inner - r) / d->profile_cutoff_width_inner)));
                    const double density_floor = d->sigma_floor * d->sigma0;
                    sigma[(size_t)n_radial * nphi + n_azimuthal] = fmax(density_damped, density_floor);
                }
        /* init.cpp:1257-1300 (profile); the spreading ring keeps energy = 0 only when
         * isothermal -- init_gas_energy runs after init_gas_density when Adiabatic */
        if (adi) {
            for (int i = 0; i < nr; ++i)
                for (int j = 0; j < nphi; ++j) {
                    const double e = initial_energy(d, Rmed[i], M);
                    const double e_floor = d->minimum_temperature * sigma[(size_t)i * nphi + j] /
                                           d->mu * d->Rgas / (d->adiabatic_index - 1.0);
                    energy[(size_t)i * nphi + j] = e > e_floor ? e : e_floor;
                }
            /* init.cpp:1363-1450 the same cutoffs on the energy */
            for (int outer = 1; outer >= 0; --outer) {
                if (!(outer ? d->profile_cutoff_outer : d->profile_cutoff_inner))
                    continue;
                for (int n_radial = 0; n_radial < nr; ++n_radial)
                    for (int n_azimuthal = 0; n_azimuthal < nphi; ++n_azimuthal) {
                        const double r = Rmed[n_radial];
                        const double cut =
                            outer ? 1.0 / (1.0 + exp((r - d->profile_cutoff_point_outer) / d->profile_cutoff_width_outer))
                                  : 1.0 / (1.0 + exp((d->profile_cutoff_point_inner - r) / d->profile_cutoff_width_inner));
                        const double energy_damped = energy[(size_t)n_radial * nphi + n_azimuthal] * cut;
                        const double energy_floor = d->minimum_temperature *
                                                    sigma[(size_t)n_radial * nphi + n_azimuthal] / d->mu * d->Rgas /
                                                    (d->adiabatic_index - 1.0);
                        energy[(size_t)n_radial * nphi + n_azimuthal] = fmax(energy_damped, energy_floor);
                    }
            }
        }
        /* init.cpp:1150-1185 renormalize_sigma_and_report; gas_total_mass
         * (quantities.cpp:50-73) over this slab's active rings only -- exact for
         * a single slab, multi-slab callers must pre-normalise (MPI_Allreduce) */
        if (d->set_sigma0) {
            double total_mass = 0.0;
            for (int i = s.radial_first_active; i < s.radial_active_size; ++i)
                for (int j = 0; j < nphi; ++j)
                    if (Rmed[i] <= 2.0 * d->rmax)
                        total_mass += Surf[i] * sigma[(size_t)i * nphi + j];
            sigma0 *= d->disk_mass / total_mass;
            d->sigma0 = sigma0; /* parameters::sigma0 is rescaled in place (init.cpp:1155) */
            for (int i = 0; i < nr; ++i)
                for (int j = 0; j < nphi; ++j) {
                    sigma[(size_t)i * nphi + j] *= d->disk_mass / total_mass;
                    if (adi)
                        energy[(size_t)i * nphi + j] *= d->disk_mass / total_mass;
                }
        }
    }

    /* init.cpp:1616-1631 / 1725-1772 init_gas_velocities; v_radial row nr is not
     * touched and stays 0 (polargrid.cpp:51-64 clears on allocation) */
    for (int j = 0; j < nphi; ++j)
        vrad[(size_t)nr * nphi + j] = 0.0;
    for (int i = 0; i < nr; ++i) {
        const double r = Rmed[i], ri = Rinf[i];
        for (int j = 0; j < nphi; ++j) {
            if (d->initialize_pure_keplerian) {
                vrad[(size_t)i * nphi + j] = initial_viscous_radial_speed(d, r, M);
                vazi[(size_t)i * nphi + j] = sqrt(d->G * M / r) - d->omega_frame * r;
                continue;
            }
            double v = smoothed_v_az(d, r, M);
            v -= d->omega_frame * r;
            vazi[(size_t)i * nphi + j] = v;
            double vr = 0.0; /* IMPOSEDDISKDRIFT = 0 */
            if (!d->initialize_vradial_zero)
                vr += vs_get_vr(d, ri, M);
            else
                vr = 0.0;
            vrad[(size_t)i * nphi + j] = vr;
        }
    }
    free(Rmed);
    free(Rinf);
    free(Rsup);
    free(Surf);
    return FCPT_OK;
}

/* ------------------------------------------------------------------------ */
static double *dalloc(size_t n)
{
    double *p = (double *)calloc(n ? n : 1, sizeof(double));
    return p;
}

int orc_create(const fcpt_desc *d, const double *radii, orc_ctx **out)
{
    if (!d || !radii || !out)
        return FCPT_EINVAL;
    if (d->struct_size != sizeof(fcpt_desc) || d->abi_version != FCPT_ABI_VERSION)
        return FCPT_EINVAL;
    if (d->stabilize_viscosity < 0 || d->stabilize_viscosity > 2)
        return FCPT_EINVAL;
    /* (cooling switches of an isothermal setup are inert: SubStep3 is only called `if (parameters::Adiabatic)`,
     * simulation.cpp:205-207) */
    if (d->cooling_surface && (d->opacity < FCPT_OPACITY_LIN || d->opacity > FCPT_OPACITY_SIMPLE))
        return FCPT_EINVAL;
    /* the interfaces must be finite and strictly increasing (the exponential spacing's Newton iteration,
     * init.cpp:113-131, collapses to NaN for coarse grids) */
    if (d->nr_global < 1)
        return FCPT_EINVAL;
    for (int i = 0; i <= d->nr_global + FCPT_GEOM_PAD; ++i)
        if (!isfinite(radii[i]) || radii[i] <= 0.0 || (i > 0 && !(radii[i] > radii[i - 1])))
            return FCPT_EINVAL;
    orc_ctx *c = (orc_ctx *)calloc(1, sizeof(orc_ctx));
    if (!c)
        return FCPT_ENOMEM;
    c->d = *d;
    int rc = orc_split_domain(d, &c->s);
    if (rc) {
        free(c);
        return rc;
    }
    c->nr = c->s.nr;
    c->nphi = d->nphi;
    c->big = (long)c->nr * c->nphi >= 32768;
    c->dphi = 2.0 * M_PI / (double)c->nphi; /* Interpret.cpp:230-231 */
    c->invdphi = (double)c->nphi / (2.0 * M_PI);
    const size_t ng = (size_t)d->nr_global + FCPT_GEOM_PAD + 1;
    c->radii = dalloc(ng);
    memcpy(c->radii, radii, ng * sizeof(double));
    const size_t n1 = (size_t)c->nr + FCPT_GEOM_PAD + 1;
    c->Rmed = dalloc(n1);
    c->Rinf = dalloc(n1);
    c->Rsup = dalloc(n1);
    c->Surf = dalloc(n1);
    c->InvRmed = dalloc(n1);
    c->InvRinf = dalloc(n1);
    c->InvSurf = dalloc(n1);
    c->InvDiffRmed = dalloc(n1);
    c->InvDiffRsup = dalloc(n1);
    c->InvDiffRsupRb = dalloc(n1);
    init_geometry(c);
    const size_t ns = (size_t)c->nr * c->nphi, nv = (size_t)(c->nr + 1) * c->nphi;
    c->sigma = dalloc(ns);
    c->vrad = dalloc(nv);
    c->vazi = dalloc(ns);
    c->energy = dalloc(ns);
    c->pressure = dalloc(ns);
    c->soundspeed = dalloc(ns);
    c->scale_height = dalloc(ns);
    c->viscosity = dalloc(ns);
    c->temperature = dalloc(ns);
    c->potential = dalloc(ns);
    c->accel_r = dalloc(nv);
    c->accel_az = dalloc(nv);
    c->sigma0 = dalloc(ns);
    c->vrad0 = dalloc(nv);
    c->vazi0 = dalloc(ns);
    c->energy0 = dalloc(ns);
    c->qr = dalloc(ns);
    c->qphi = dalloc(ns);
    c->divv = dalloc(ns);
    c->trr = dalloc(ns);
    c->tpp = dalloc(ns);
    c->trp = dalloc(nv);
    c->qplus = dalloc(ns);
    c->qminus = dalloc(ns);
    c->density_int = dalloc(ns);
    c->nusig = dalloc(ns);
    c->nusig_rp = dalloc(nv);
    c->cfac_phi = dalloc(ns);
    c->cfac_r = dalloc(ns);
    c->tau_eff = dalloc(ns);
    c->massflow = c->d.write_massflow ? dalloc(nv) : NULL;
    c->rmp = dalloc(ns);
    c->rmm = dalloc(ns);
    c->lp = dalloc(ns);
    c->lm = dalloc(ns);
    c->vres = dalloc(ns);
    c->vmean = dalloc((size_t)c->nr);
    c->work = dalloc(ns);
    c->qrstar = dalloc(nv);
    c->densstar = dalloc(nv);
    c->tempshift = dalloc(ns);
    c->dq = dalloc(ns);
    c->nshift = (int *)calloc((size_t)c->nr, sizeof(int));
    c->nosplit = (int *)calloc((size_t)c->nr, sizeof(int));
    c->cfl_vmean = dalloc((size_t)c->nr + 1);
    c->cfl_vres = dalloc(ns);
    /* default body: the central star at the origin */
    c->nbodies = 1;
    c->bx[0] = c->by[0] = 0.0;
    c->bm[0] = d->hydro_center_mass;
    c->brsm[0] = 0.0;
    c->clk.time = 0.0;
    c->clk.last_dt = d->first_dt; /* Interpret.cpp:86 */
    *out = c;
    return FCPT_OK;
}

int orc_destroy(orc_ctx *c)
{
    if (!c)
        return FCPT_OK;
    double **ps[] = {&c->radii,      &c->Rmed,      &c->Rinf,        &c->Rsup,        &c->Surf,
                     &c->InvRmed,    &c->InvRinf,   &c->InvSurf,     &c->InvDiffRmed, &c->InvDiffRsup,
                     &c->InvDiffRsupRb, &c->sigma,  &c->vrad,        &c->vazi,        &c->energy,
                     &c->pressure,   &c->soundspeed, &c->scale_height, &c->viscosity, &c->temperature,
                     &c->potential,  &c->sigma0,    &c->vrad0,       &c->vazi0,       &c->energy0,
                     &c->qr,         &c->qphi,      &c->divv,        &c->trr,         &c->tpp,
                     &c->trp,        &c->qplus,     &c->qminus,      &c->density_int, &c->tau_eff, &c->nusig, &c->nusig_rp, &c->cfac_phi, &c->cfac_r,
                     &c->rmp,        &c->rmm,       &c->lp,          &c->lm,          &c->vres,
                     &c->vmean,      &c->work,      &c->qrstar,      &c->densstar,    &c->tempshift,
                     &c->dq,         &c->cfl_vmean, &c->cfl_vres,    &c->massflow,
                     &c->accel_r,    &c->accel_az};
    for (size_t i = 0; i < sizeof(ps) / sizeof(ps[0]); ++i)
        free(*ps[i]);
    free(c->nshift);
    free(c->nosplit);
    free(c);
    return FCPT_OK;
}

int orc_get_split(const orc_ctx *c, fcpt_split *o)
{
    if (!c || !o)
        return FCPT_EINVAL;
    *o = c->s;
    return FCPT_OK;
}
int orc_get_clock(const orc_ctx *c, fcpt_clock *o)
{
    if (!c || !o)
        return FCPT_EINVAL;
    *o = c->clk;
    return FCPT_OK;
}
int orc_set_clock(orc_ctx *c, const fcpt_clock *in)
{
    if (!c || !in)
        return FCPT_EINVAL;
    c->clk = *in;
    return FCPT_OK;
}

static double *field_ptr(orc_ctx *c, int32_t f, size_t *n)
{
    const size_t ns = (size_t)c->nr * c->nphi, nv = (size_t)(c->nr + 1) * c->nphi;
    *n = ns;
    switch (f) {
    case FCPT_F_SIGMA: return c->sigma;
    case FCPT_F_VRAD: *n = nv; return c->vrad;
    case FCPT_F_VAZI: return c->vazi;
    case FCPT_F_ENERGY: return c->energy;
    case FCPT_F_PRESSURE: return c->pressure;
    case FCPT_F_SOUNDSPEED: return c->soundspeed;
    case FCPT_F_SCALE_HEIGHT: return c->scale_height;
    case FCPT_F_VISCOSITY: return c->viscosity;
    case FCPT_F_TEMPERATURE: return c->temperature;
    case FCPT_F_POTENTIAL: return c->potential;
    case FCPT_F_ACCEL_RADIAL: *n = nv; return c->d.body_force_from_potential ? NULL : c->accel_r;
    case FCPT_F_ACCEL_AZIMUTHAL: *n = nv; return c->d.body_force_from_potential ? NULL : c->accel_az;
    case FCPT_F_SIGMA0: return c->sigma0;
    case FCPT_F_VRAD0: *n = nv; return c->vrad0;
    case FCPT_F_VAZI0: return c->vazi0;
    case FCPT_F_ENERGY0: return c->energy0;
    case FCPT_F_QPLUS: return c->qplus;
    case FCPT_F_QMINUS: return c->qminus;
    case FCPT_F_VISC_CFAC_PHI: return c->d.stabilize_viscosity ? c->cfac_phi : NULL;
    case FCPT_F_VISC_CFAC_R: return c->d.stabilize_viscosity ? c->cfac_r : NULL;
    case FCPT_F_MASSFLOW: *n = nv; return c->massflow;
    default: return NULL;
    }
}
int orc_upload(orc_ctx *c, int32_t f, const double *host)
{
    size_t n;
    double *p = c ? field_ptr(c, f, &n) : NULL;
    if (!p || !host)
        return FCPT_EINVAL;
    memcpy(p, host, n * sizeof(double));
    return FCPT_OK;
}
int orc_download(orc_ctx *c, int32_t f, double *host)
{
    size_t n;
    double *p = c ? field_ptr(c, f, &n) : NULL;
    if (!p || !host)
        return FCPT_EINVAL;
    memcpy(host, p, n * sizeof(double));
    return FCPT_OK;
}
int orc_geometry(const orc_ctx *c, int32_t which, double *out)
{
    if (!c || !out)
        return FCPT_EINVAL;
    const double *src[] = {c->Rmed, c->Rinf, c->Rsup, c->Surf, c->InvDiffRmed, c->InvDiffRsup};
    if (which < 0 || which > 5)
        return FCPT_EINVAL;
    memcpy(out, src[which], sizeof(double) * ((size_t)c->nr + FCPT_GEOM_PAD + 1));
    return FCPT_OK;
}
int orc_last_nshift(const orc_ctx *c, int32_t *out)
{
    if (!c || !out)
        return FCPT_EINVAL;
    for (int i = 0; i < c->nr; ++i)
        out[i] = c->nshift[i];
    return FCPT_OK;
}
int orc_set_bodies(orc_ctx *c, int32_t n, const double *x, const double *y, const double *m,
                   const double *rsm, double ix, double iy)
{
    if (!c || n < 0 || n > FCPT_MAX_BODIES)
        return FCPT_EINVAL;
    c->nbodies = n;
    for (int k = 0; k < n; ++k) {
        c->bx[k] = x[k];
        c->by[k] = y[k];
        c->bm[k] = m[k];
        c->brsm[k] = rsm ? rsm[k] : 0.0;
    }
    c->indirect_x = ix;
    c->indirect_y = iy;
    c->has_mid = 0;
    return FCPT_OK;
}

int orc_set_body_irradiation(orc_ctx *c, int32_t n, const double *temperature, const double *radius,
                             const double *rampup_time)
{
    if (!c || n < 0 || n > FCPT_MAX_BODIES || (n > 0 && (!temperature || !radius)))
        return FCPT_EINVAL;
    if (c->d.eos != FCPT_EOS_IDEAL)
        return FCPT_EINVAL;
    c->heating_star = 0;
    for (int k = 0; k < FCPT_MAX_BODIES; ++k) {
        c->btemp[k] = k < n ? temperature[k] : 0.0;
        c->bradius[k] = k < n ? radius[k] : 0.0;
        c->bramp[k] = (k < n && rampup_time) ? rampup_time[k] : 0.0;
        if (c->btemp[k] > 0.0)
            c->heating_star = 1; /* planetary_system.cpp:137-146 */
    }
    return FCPT_OK;
}

/* Force.cpp:23-122 ComputeDiskOnPlanetAccel (local sums, no Allreduce; correct_disk_selfgravity
 * off); smoothing: Force.cpp:124-159 compute_smoothing */
int orc_disk_on_body_accel(orc_ctx *c, double x, double y, double r_object, double smoothing_fixed,
                           double cubic_smoothing_radius, double out[4])
{
    if (!c || !out)
        return FCPT_EINVAL;
    const int Nphi = c->nphi;
    double axi = 0.0, ayi = 0.0, axo = 0.0, ayo = 0.0;
    for (int nr = c->s.radial_first_active; nr < c->s.radial_active_size; ++nr) {
        for (int naz = 0; naz < Nphi; ++naz) {
            const double smooth = smoothing_fixed >= 0.0
                                      ? smoothing_fixed
                                      : c->d.thickness_smoothing * c->scale_height[IDX(c, nr, naz)];
            const double xc = c->Rmed[nr] * cos(c->dphi * (double)naz); /* SideEuler.cpp:60-63 */
            const double yc = c->Rmed[nr] * sin(c->dphi * (double)naz);
            const double cellmass = c->Surf[nr] * c->sigma[IDX(c, nr, naz)];
            const double dx = xc - x;
            const double dy = yc - y;
            const double dist_2 = dx * dx + dy * dy;
            const double dist_sm_2 = dist_2 + smooth * smooth;
            const double dist_sm = sqrt(dist_sm_2);
            const double dist_sm_3 = dist_sm_2 * dist_sm;
            const double inv_dist_sm_3 = 1.0 / dist_sm_3;
            double smooth_factor_klahr = 1.0;
            if (cubic_smoothing_radius > 0.0 && dist_sm < cubic_smoothing_radius) {
                const double q = dist_sm / cubic_smoothing_radius;
                smooth_factor_klahr = -(3.0 * ((q * q) * (q * q)) - 4.0 * (q * q * q));
            }
            if (c->Rmed[nr] < r_object) {
                axi += c->d.G * cellmass * dx * inv_dist_sm_3 * smooth_factor_klahr;
                ayi += c->d.G * cellmass * dy * inv_dist_sm_3 * smooth_factor_klahr;
            } else {
                axo += c->d.G * cellmass * dx * inv_dist_sm_3 * smooth_factor_klahr;
                ayo += c->d.G * cellmass * dy * inv_dist_sm_3 * smooth_factor_klahr;
            }
        }
    }
    out[0] = axi;
    out[1] = ayi;
    out[2] = axo;
    out[3] = ayo;
    return FCPT_OK;
}

/* ------------------------------------------------------------------------ */
/* EOS helpers                                                               */

/* SourceEuler.cpp:1054-1092 compute_sound_speed_normal */
static void compute_sound_speed(orc_ctx *c)
{
    const int Nr = c->nr, Nphi = c->nphi;
    const fcpt_desc *d = &c->d;
#pragma omp parallel for if (c->big)
    for (int nr = 0; nr < Nr; ++nr)
        for (int naz = 0; naz < Nphi; ++naz) {
            if (d->eos == FCPT_EOS_IDEAL) {
                const double g = d->adiabatic_index;
                c->soundspeed[IDX(c, nr, naz)] =
                    sqrt(g * (g - 1.0) * c->energy[IDX(c, nr, naz)] / c->sigma[IDX(c, nr, naz)]);
            } else {
                const double vK = sqrt(d->G * d->hydro_center_mass / c->Rmed[nr]);
                const double h = d->aspect_ratio * pow(c->Rmed[nr], d->flaring_index);
                c->soundspeed[IDX(c, nr, naz)] = h * vK;
            }
        }
}
/* SourceEuler.cpp:1218-1251 compute_scale_height_old */
static void compute_scale_height(orc_ctx *c)
{
    const int Nr = c->nr, Nphi = c->nphi;
#pragma omp parallel for if (c->big)
    for (int nr = 0; nr < Nr; ++nr) {
        const double inv_omega_kepler = 1.0 / omega_kepler(c, c->Rmed[nr]);
        for (int naz = 0; naz < Nphi; ++naz) {
            if (c->d.eos == FCPT_EOS_IDEAL)
                c->scale_height[IDX(c, nr, naz)] =
                    c->soundspeed[IDX(c, nr, naz)] / (sqrt(c->d.adiabatic_index)) * inv_omega_kepler;
            else
                c->scale_height[IDX(c, nr, naz)] = c->soundspeed[IDX(c, nr, naz)] * inv_omega_kepler;
        }
    }
}
/* SourceEuler.cpp:1442-1473 compute_pressure */
static void compute_pressure(orc_ctx *c)
{
    const int Nr = c->nr, Nphi = c->nphi;
#pragma omp parallel for if (c->big)
    for (int nr = 0; nr < Nr; ++nr)
        for (int naz = 0; naz < Nphi; ++naz) {
            if (c->d.eos == FCPT_EOS_IDEAL)
                c->pressure[IDX(c, nr, naz)] =
                    (c->d.adiabatic_index - 1.0) * c->energy[IDX(c, nr, naz)];
            else {
                const double cs = c->soundspeed[IDX(c, nr, naz)];
                c->pressure[IDX(c, nr, naz)] = c->sigma[IDX(c, nr, naz)] * (cs * cs);
            }
        }
}
/* SourceEuler.cpp:1475-1505 compute_temperature */
static void compute_temperature(orc_ctx *c)
{
    const int Nr = c->nr, Nphi = c->nphi;
    const double Rgas = c->d.Rgas;
#pragma omp parallel for if (c->big)
    for (int nr = 0; nr < Nr; ++nr)
        for (int naz = 0; naz < Nphi; ++naz) {
            if (c->d.eos == FCPT_EOS_IDEAL) {
                const double c_v_inv = c->d.mu / Rgas * (c->d.adiabatic_index - 1.0);
                c->temperature[IDX(c, nr, naz)] =
                    c_v_inv * c->energy[IDX(c, nr, naz)] / c->sigma[IDX(c, nr, naz)];
            } else {
                c->temperature[IDX(c, nr, naz)] =
                    c->d.mu / Rgas * c->pressure[IDX(c, nr, naz)] / c->sigma[IDX(c, nr, naz)];
            }
        }
}
/* viscosity/viscosity.cpp:98-137 update_viscosity (AlphaMode 0) */
static void update_viscosity(orc_ctx *c)
{
    const int Nr = c->nr, Nphi = c->nphi;
    if (c->d.viscous_alpha > 0) {
#pragma omp parallel for if (c->big)
        for (int nr = 0; nr < Nr; ++nr)
            for (int naz = 0; naz < Nphi; ++naz) {
                const double alpha = c->d.viscous_alpha;
                const double cs = c->soundspeed[IDX(c, nr, naz)];
                const double H = c->scale_height[IDX(c, nr, naz)];
                c->viscosity[IDX(c, nr, naz)] = alpha * H * cs;
            }
    } else {
        if (!c->viscosity_calculated) {
            for (int nr = 0; nr < Nr; ++nr)
                for (int naz = 0; naz < Nphi; ++naz)
                    c->viscosity[IDX(c, nr, naz)] = c->d.constant_viscosity;
        }
        c->viscosity_calculated = 1;
    }
}
/* SourceEuler.cpp:102-134 assure_minimum_value */
static void assure_minimum_value(orc_ctx *c, double *dst, double minimum_value)
{
    const size_t n = (size_t)c->nr * c->nphi;
    for (size_t i = 0; i < n; ++i)
        if (dst[i] < minimum_value)
            dst[i] = minimum_value;
}
/* SourceEuler.cpp:136-202 assure_temperature_range */
static void assure_temperature_range(orc_ctx *c)
{
    const int Nr = c->nr, Nphi = c->nphi;
    const double Tmin = c->d.minimum_temperature, Tmax = c->d.maximum_temperature;
    const double mu = c->d.mu, g = c->d.adiabatic_index, R = c->d.Rgas;
#pragma omp parallel for if (c->big)
    for (int nr = 0; nr < Nr; ++nr)
        for (int naz = 0; naz < Nphi; ++naz) {
            const double rho = c->sigma[IDX(c, nr, naz)];
            const double minimum_energy = Tmin * rho / mu * R / (g - 1.0);
            const double maximum_energy = Tmax * rho / mu * R / (g - 1.0);
            if (!(c->energy[IDX(c, nr, naz)] > minimum_energy))
                c->energy[IDX(c, nr, naz)] = Tmin * rho / mu * R / (g - 1.0);
            if (!(c->energy[IDX(c, nr, naz)] < maximum_energy))
                c->energy[IDX(c, nr, naz)] = Tmax * rho / mu * R / (g - 1.0);
        }
}

/* ------------------------------------------------------------------------ */
/* Pframeforce.cpp:21-94 CalculateNbodyPotential, Force.cpp:124-159 smoothing */
static void calculate_potential(orc_ctx *c)
{
    const int Nr = c->nr, Nphi = c->nphi;
#pragma omp parallel for if (c->big)
    for (int nr = 0; nr < Nr; ++nr)
        for (int naz = 0; naz < Nphi; ++naz) {
            const double x = c->Rmed[nr] * cos(c->dphi * (double)naz); /* SideEuler.cpp:60-63 */
            const double y = c->Rmed[nr] * sin(c->dphi * (double)naz);
            double pot = 0.0;
            for (int k = 0; k < c->nbodies; ++k) {
                const double smooth = c->d.thickness_smoothing * c->scale_height[IDX(c, nr, naz)];
                const double dx = x - c->bx[k];
                const double dy = y - c->by[k];
                const double dist_2 = dx * dx + dy * dy;
                const double d_smoothed = sqrt(dist_2 + smooth * smooth);
                double smooth_factor_klahr = 1.0;
                if (c->brsm[k] > 0.0) {
                    const double r_sm = c->brsm[k];
                    if (d_smoothed < r_sm) {
                        const double q = d_smoothed / r_sm;
                        smooth_factor_klahr = ((q * q) * (q * q) - 2.0 * (q * q * q) + 2.0 * d_smoothed / r_sm);
                    }
                }
                pot += -c->d.G * c->bm[k] / d_smoothed * smooth_factor_klahr;
            }
            pot += -c->indirect_x * x - c->indirect_y * y;
            c->potential[IDX(c, nr, naz)] = pot;
        }
}

/* Pframeforce.cpp:96-189 CalculateAccelOnGas (BodyForceFromPotential: no): the bodies' pull and the indirect term as
 * cell-centred accelerations; ACCEL_RADIAL / ACCEL_AZIMUTHAL are vector grids of Nr+1 rows of which rows 1 .. Nr-1 are
 * written (rows 0 and Nr stay at their initial zero) */
static void calculate_accel_on_gas(orc_ctx *c)
{
    const int Nr = c->nr, Nphi = c->nphi;
#pragma omp parallel for if (c->big)
    for (int nr = 1; nr < Nr; ++nr)
        for (int naz = 0; naz < Nphi; ++naz) {
            const double r = c->Rmed[nr];
            const double x = c->Rmed[nr] * cos(c->dphi * (double)naz);
            const double y = c->Rmed[nr] * sin(c->dphi * (double)naz);
            double ax = c->indirect_x, ay = c->indirect_y;
            for (int k = 0; k < c->nbodies; ++k) {
                const double smooth = c->d.thickness_smoothing * c->scale_height[IDX(c, nr, naz)];
                const double dx = x - c->bx[k];
                const double dy = y - c->by[k];
                const double dist_2 = dx * dx + dy * dy;
                const double dist_2_sm = dist_2 + smooth * smooth;
                const double dist_sm = sqrt(dist_2_sm);
                const double dist_3_sm = dist_sm * dist_2_sm;
                const double inv_dist_3_sm = 1.0 / dist_3_sm;
                double smooth_factor_klahr = 1.0;
                if (c->brsm[k] > 0.0) {
                    const double r_sm = c->brsm[k];
                    if (dist_sm < r_sm) {
                        const double q = dist_sm / r_sm;
                        smooth_factor_klahr = -(3.0 * ((q * q) * (q * q)) - 4.0 * (q * q * q));
                    }
                }
                ax -= dx * c->d.G * c->bm[k] * inv_dist_3_sm * smooth_factor_klahr;
                ay -= dy * c->d.G * c->bm[k] * inv_dist_3_sm * smooth_factor_klahr;
            }
            c->accel_r[IDX(c, nr, naz)] = (x * ax + y * ay) / r;
            c->accel_az[IDX(c, nr, naz)] = (x * ay - y * ax) / r;
        }
}
/* simulation.cpp:167-175: the potential, or the accelerations */
static void calculate_body_force(orc_ctx *c)
{
    if (c->d.body_force_from_potential)
        calculate_potential(c);
    else
        calculate_accel_on_gas(c);
}

/* ------------------------------------------------------------------------ */
/* boundary conditions                                                       */

/* boundary_conditions/zero_gradient.cpp, reference.cpp, reflecting.cpp, outflow.cpp,
 * keplerian_azimuthal.cpp, keplerian_radial.cpp, zero_shear.cpp */
static void bc_scalar(orc_ctx *c, double *x, const double *x0, int type, int outer)
{
    const int Nphi = c->nphi, Irad = c->nr - 1;
    if (type == FCPT_BC_NONE)
        return;
    if (!outer) {
        if (!c->s.is_first)
            return;
        for (int j = 0; j < Nphi; ++j) {
            if (type == FCPT_BC_ZEROGRADIENT)
                x[IDX(c, 0, j)] = x[IDX(c, 1, j)];
            else if (type == FCPT_BC_REFERENCE)
                x[IDX(c, 0, j)] = x0[IDX(c, 0, j)];
        }
    } else {
        if (!c->s.is_last)
            return;
        for (int j = 0; j < Nphi; ++j) {
            if (type == FCPT_BC_ZEROGRADIENT)
                x[IDX(c, Irad, j)] = x[IDX(c, Irad - 1, j)];
            else if (type == FCPT_BC_REFERENCE)
                x[IDX(c, Irad, j)] = x0[IDX(c, Irad, j)];
        }
    }
}
static void bc_vrad(orc_ctx *c, int type, int outer)
{
    const int Nphi = c->nphi, Irad = c->nr; /* vector grid: max_radial = Nr */
    double *vr = c->vrad;
    const double *v0 = c->vrad0;
    if (type == FCPT_BC_NONE)
        return;
    /* reflecting.cpp:15-40 has NO rank guard (reference quirk, kept) */
    if (type == FCPT_BC_REFLECTING) {
        for (int j = 0; j < Nphi; ++j) {
            if (!outer) {
                vr[IDX(c, 0, j)] = -vr[IDX(c, 2, j)];
                vr[IDX(c, 1, j)] = 0;
            } else {
                vr[IDX(c, Irad, j)] = -vr[IDX(c, Irad - 2, j)];
                vr[IDX(c, Irad - 1, j)] = 0;
            }
        }
        return;
    }
    if (!outer && !c->s.is_first)
        return;
    if (outer && !c->s.is_last)
        return;
    for (int j = 0; j < Nphi; ++j) {
        if (!outer) {
            switch (type) {
            case FCPT_BC_ZEROGRADIENT:
                vr[IDX(c, 0, j)] = vr[IDX(c, 2, j)];
                vr[IDX(c, 1, j)] = vr[IDX(c, 2, j)];
                break;
            case FCPT_BC_REFERENCE:
                vr[IDX(c, 0, j)] = v0[IDX(c, 0, j)];
                vr[IDX(c, 1, j)] = v0[IDX(c, 1, j)];
                break;
            case FCPT_BC_OUTFLOW:
                if (vr[IDX(c, 2, j)] > 0.0) {
                    vr[IDX(c, 1, j)] = 0.0;
                    vr[IDX(c, 0, j)] = 0.0;
                } else {
                    vr[IDX(c, 1, j)] = vr[IDX(c, 2, j)];
                    vr[IDX(c, 0, j)] = vr[IDX(c, 2, j)];
                }
                break;
            case FCPT_BC_KEPLERIAN:
                for (int k = 0; k <= 1; k++) {
                    const double vKep = sqrt(c->d.G * c->d.hydro_center_mass / c->Rmed[k]);
                    vr[IDX(c, k, j)] = c->d.keplerian_vrad_factor[0] * vKep;
                }
                break;
            default: break;
            }
        } else {
            switch (type) {
            case FCPT_BC_ZEROGRADIENT:
                vr[IDX(c, Irad, j)] = vr[IDX(c, Irad - 2, j)];
                vr[IDX(c, Irad - 1, j)] = vr[IDX(c, Irad - 2, j)];
                break;
            case FCPT_BC_REFERENCE:
                vr[IDX(c, Irad, j)] = v0[IDX(c, Irad, j)];
                vr[IDX(c, Irad - 1, j)] = v0[IDX(c, Irad - 1, j)];
                break;
            case FCPT_BC_OUTFLOW:
                if (vr[IDX(c, Irad - 2, j)] < 0.0) {
                    vr[IDX(c, Irad - 1, j)] = 0.0;
                    vr[IDX(c, Irad, j)] = 0.0;
                } else {
                    vr[IDX(c, Irad - 1, j)] = vr[IDX(c, Irad - 2, j)];
                    vr[IDX(c, Irad, j)] = vr[IDX(c, Irad - 2, j)];
                }
                break;
            case FCPT_BC_KEPLERIAN:
                for (int k = Irad; k >= Irad - 1; k--) {
                    const double vKep = sqrt(c->d.G * c->d.hydro_center_mass / c->Rmed[k]);
                    vr[IDX(c, k, j)] = c->d.keplerian_vrad_factor[1] * vKep;
                }
                break;
            default: break;
            }
        }
    }
}
static void bc_vaz(orc_ctx *c, int type, int outer)
{
    const int Nphi = c->nphi, Irad = c->nr - 1;
    double *v = c->vazi;
    const double *v0 = c->vazi0;
    if (type == FCPT_BC_NONE)
        return;
    if (!outer && !c->s.is_first)
        return;
    if (outer && !c->s.is_last)
        return;
    const int row = outer ? Irad : 0;
    const int act = outer ? Irad - 1 : 1;
    const double vKep = sqrt(c->d.G * c->d.hydro_center_mass / c->Rmed[row]);
    const double r = c->Rmed[row];
    for (int j = 0; j < Nphi; ++j) {
        switch (type) {
        case FCPT_BC_ZEROGRADIENT: v[IDX(c, row, j)] = v[IDX(c, act, j)]; break;
        case FCPT_BC_REFERENCE: v[IDX(c, row, j)] = v0[IDX(c, row, j)]; break;
        case FCPT_BC_KEPLERIAN:
            v[IDX(c, row, j)] = c->d.keplerian_vaz_factor[outer] * vKep - r * c->d.omega_frame;
            break;
        case FCPT_BC_ZEROSHEAR: {
            const double Omega_active = v[IDX(c, act, j)] / c->Rmed[act];
            v[IDX(c, row, j)] = r * Omega_active;
            break;
        }
        default: break;
        }
    }
}

/* boundary_conditions/damping.cpp:311-427 (reference), :429-557 (zero), :559-700 (mean) */
static void damping_single(orc_ctx *c, double *q, double *q0, int is_vector, int is_density,
                           int type, int outer, double dt)
{
    if (type == FCPT_DAMP_NONE)
        return;
    const double *radius = is_vector ? c->Rinf : c->Rmed;
    const int Nphi = c->nphi;
    const int size_radial = is_vector ? c->nr + 1 : c->nr;
    const double RMIN = c->d.rmin, RMAX = c->d.rmax;
    int lo, hi; /* inclusive row range */
    double rlim, redge, tau;
    if (!outer) {
        if (!((c->d.damping_inner_limit > 1.0) && (radius[0] < RMIN * c->d.damping_inner_limit)))
            return;
        unsigned limit;
        if (!is_vector)
            limit = clamp_r_id(c, get_rmed_id(c, RMIN * c->d.damping_inner_limit), 0);
        else
            limit = clamp_r_id(c, get_rinf_id(c, RMIN * c->d.damping_inner_limit), 1);
        lo = 0;
        hi = (int)limit;
        rlim = RMIN * c->d.damping_inner_limit;
        redge = RMIN;
        tau = c->d.damping_time_factor * 2.0 * M_PI / omega_kepler(c, RMIN);
    } else {
        if (!((c->d.damping_outer_limit < 1.0) &&
              (radius[size_radial - 1] > RMAX * c->d.damping_outer_limit)))
            return;
        unsigned limit;
        if (!is_vector)
            limit = clamp_r_id(c, get_rmed_id(c, RMAX * c->d.damping_outer_limit) + 1, 0);
        else
            limit = clamp_r_id(c, get_rinf_id(c, RMAX * c->d.damping_outer_limit) + 1, 1);
        lo = (int)limit;
        hi = size_radial - 1;
        rlim = RMAX * c->d.damping_outer_limit;
        redge = RMAX;
        tau = c->d.damping_time_factor * 2.0 * M_PI / omega_kepler(c, c->d.damping_time_radius_outer);
    }
    if (type == FCPT_DAMP_MEAN) {
        for (int i = lo; i <= hi; ++i) {
            q0[IDX(c, i, 0)] = 0.0;
            for (int j = 0; j < Nphi; ++j)
                q0[IDX(c, i, 0)] += q[IDX(c, i, j)];
            q0[IDX(c, i, 0)] /= Nphi;
        }
    }
    for (int i = lo; i <= hi; ++i) {
        const double t = (radius[i] - rlim) / (redge - rlim);
        const double factor = t * t;
        const double exp_factor = exp(-dt * factor / tau);
        for (int j = 0; j < Nphi; ++j) {
            const double X = q[IDX(c, i, j)];
            double X0;
            if (type == FCPT_DAMP_REFERENCE)
                X0 = q0[IDX(c, i, j)];
            else if (type == FCPT_DAMP_MEAN)
                X0 = q0[IDX(c, i, 0)];
            else
                X0 = is_density ? c->d.sigma_floor * c->d.sigma0 : 0.0;
            q[IDX(c, i, j)] = (X - X0) * exp_factor + X0;
        }
    }
}
/* damping.cpp:754-774 damping(); order of damping_vector (damping.cpp:218-270):
 * vrad, vaz, sigma, energy (energy dropped when not adiabatic, Interpret.cpp:560-565) */
static void damping(orc_ctx *c, double dt)
{
    if (!c->d.damping)
        return;
    damping_single(c, c->vrad, c->vrad0, 1, 0, c->d.damp_vrad[0], 0, dt);
    damping_single(c, c->vrad, c->vrad0, 1, 0, c->d.damp_vrad[1], 1, dt);
    damping_single(c, c->vazi, c->vazi0, 0, 0, c->d.damp_vaz[0], 0, dt);
    damping_single(c, c->vazi, c->vazi0, 0, 0, c->d.damp_vaz[1], 1, dt);
    damping_single(c, c->sigma, c->sigma0, 0, 1, c->d.damp_sigma[0], 0, dt);
    damping_single(c, c->sigma, c->sigma0, 0, 1, c->d.damp_sigma[1], 1, dt);
    if (c->d.eos == FCPT_EOS_IDEAL) {
        damping_single(c, c->energy, c->energy0, 0, 0, c->d.damp_energy[0], 0, dt);
        damping_single(c, c->energy, c->energy0, 0, 0, c->d.damp_energy[1], 1, dt);
    }
}
/* boundary_conditions/boundary_conditions.cpp:65-114 apply_boundary_condition.
 * Note: energy_*_func are invoked on ENERGY even for isothermal runs; harmless
 * (the grid is unused there) and kept. */
static void apply_boundary_condition(orc_ctx *c, double dt, int final)
{
    if (final && c->d.damping)
        damping(c, dt);
    bc_scalar(c, c->sigma, c->sigma0, c->d.bc_sigma[0], 0);
    bc_scalar(c, c->sigma, c->sigma0, c->d.bc_sigma[1], 1);
    bc_scalar(c, c->energy, c->energy0, c->d.bc_energy[0], 0);
    bc_scalar(c, c->energy, c->energy0, c->d.bc_energy[1], 1);
    bc_vrad(c, c->d.bc_vrad[0], 0);
    bc_vrad(c, c->d.bc_vrad[1], 1);
    bc_vaz(c, c->d.bc_vaz[0], 0);
    bc_vaz(c, c->d.bc_vaz[1], 1);
}
/* damping.cpp:297-306 copy_initial_values (always stored here) */
static void copy_initial_values(orc_ctx *c)
{
    const size_t ns = (size_t)c->nr * c->nphi, nv = (size_t)(c->nr + 1) * c->nphi;
    memcpy(c->vrad0, c->vrad, nv * sizeof(double));
    memcpy(c->vazi0, c->vazi, ns * sizeof(double));
    memcpy(c->sigma0, c->sigma, ns * sizeof(double));
    memcpy(c->energy0, c->energy, ns * sizeof(double));
}

/* ------------------------------------------------------------------------ */
/* source terms                                                              */

/* SourceEuler.cpp:325-372 momentum_update_radial */
static void momentum_update_radial(orc_ctx *c, double dt)
{
    const int Nphi = c->nphi;
    const double OmegaF = c->d.omega_frame;
#pragma omp parallel for if (c->big)
    for (int nr = c->s.one_no_ghost_vr; nr < c->s.maxmo_no_ghost_vr; ++nr)
        for (int naz = 0; naz < Nphi; ++naz) {
            double gradp = 2.0 / (c->sigma[IDX(c, nr, naz)] + c->sigma[IDX(c, nr - 1, naz)]);
            gradp *= (c->pressure[IDX(c, nr, naz)] - c->pressure[IDX(c, nr - 1, naz)]);
            gradp *= c->InvDiffRmed[nr];
            const double gradphi =
                c->d.body_force_from_potential
                    ? (c->potential[IDX(c, nr, naz)] - c->potential[IDX(c, nr - 1, naz)]) * c->InvDiffRmed[nr]
                    : -(c->accel_r[IDX(c, nr, naz)] + c->accel_r[IDX(c, nr - 1, naz)]) * 0.5; /* :348-353 */
            const int naz_next = (naz == Nphi - 1 ? 0 : naz + 1);
            const double vsum = c->vazi[IDX(c, nr, naz)] + c->vazi[IDX(c, nr, naz_next)] +
                                c->vazi[IDX(c, nr - 1, naz)] + c->vazi[IDX(c, nr - 1, naz_next)];
            const double vt = 0.25 * vsum + c->Rinf[nr] * OmegaF;
            const double vt2 = vt * vt;
            const double centrifugal_accel = vt2 * c->InvRinf[nr];
            c->vrad[IDX(c, nr, naz)] += dt * (-gradp - gradphi + centrifugal_accel);
        }
}
/* SourceEuler.cpp:375-428 momentum_update_azimuthal (IMPOSEDDISKDRIFT = 0) */
static void momentum_update_azimuthal(orc_ctx *c, double dt)
{
    const int Nphi = c->nphi;
#pragma omp parallel for if (c->big)
    for (int nr = c->s.zero_no_ghost; nr < c->s.max_no_ghost; ++nr) {
        const double invdxtheta = 2.0 / (c->dphi * (c->Rsup[nr] + c->Rinf[nr]));
        for (int naz = 0; naz < Nphi; ++naz) {
            const int naz_prev = (naz == 0 ? Nphi - 1 : naz - 1);
            const double gradp = 2.0 / (c->sigma[IDX(c, nr, naz)] + c->sigma[IDX(c, nr, naz_prev)]) *
                                 (c->pressure[IDX(c, nr, naz)] - c->pressure[IDX(c, nr, naz_prev)]) *
                                 invdxtheta;
            const double gradphi =
                c->d.body_force_from_potential
                    ? (c->potential[IDX(c, nr, naz)] - c->potential[IDX(c, nr, naz_prev)]) * invdxtheta
                    : -(c->accel_az[IDX(c, nr, naz)] + c->accel_az[IDX(c, nr, naz_prev)]) * 0.5; /* :406-411 */
            c->vazi[IDX(c, nr, naz)] = c->vazi[IDX(c, nr, naz)] + dt * (-gradp - gradphi);
        }
    }
}
/* SourceEuler.cpp:459-493 compression_heating */
static void compression_heating(orc_ctx *c, double dt)
{
    if (c->d.eos != FCPT_EOS_IDEAL)
        return;
    const int Nr = c->nr - 1, Nphi = c->nphi;
    const double gamma = c->d.adiabatic_index;
#pragma omp parallel for if (c->big)
    for (int nr = 0; nr < Nr; ++nr)
        for (int naz = 0; naz < Nphi; ++naz) {
            const int naz_next = (naz == Nphi - 1 ? 0 : naz + 1);
            const double DIV_V =
                (c->vrad[IDX(c, nr + 1, naz)] * c->Rinf[nr + 1] - c->vrad[IDX(c, nr, naz)] * c->Rinf[nr]) *
                    c->InvDiffRsupRb[nr] +
                (c->vazi[IDX(c, nr, naz_next)] - c->vazi[IDX(c, nr, naz)]) * c->invdphi * c->InvRmed[nr];
            const double energy_old = c->energy[IDX(c, nr, naz)];
            c->energy[IDX(c, nr, naz)] = energy_old * exp(-(gamma - 1.0) * dt * DIV_V);
        }
}

/* viscosity/artificial_viscosity.cpp:35-140 update_with_artificial_viscosity_TW */
static void artificial_viscosity_TW(orc_ctx *c, double dt)
{
    const int Nr = c->nr, Nphi = c->nphi;
    const int adi = c->d.eos == FCPT_EOS_IDEAL;
    const double C = c->d.artificial_viscosity_factor;
#pragma omp parallel for if (c->big)
    for (int nr = 0; nr < Nr; ++nr)
        for (int naz = 0; naz < Nphi; ++naz) {
            const int naz_next = naz == Nphi - 1 ? 0 : naz + 1;
            const double eps_rr =
                (c->vrad[IDX(c, nr + 1, naz)] - c->vrad[IDX(c, nr, naz)]) * c->InvDiffRsup[nr];
            const double eps_pp =
                c->InvRmed[nr] *
                ((c->vazi[IDX(c, nr, naz_next)] - c->vazi[IDX(c, nr, naz)]) * c->invdphi +
                 0.5 * (c->vrad[IDX(c, nr + 1, naz)] + c->vrad[IDX(c, nr, naz)]));
            const double div_V = fmin(eps_rr + eps_pp, 0.0);
            const double Dr = c->Rinf[nr + 1] - c->Rinf[nr];
            const double rDphi = c->Rmed[nr] * c->dphi;
            double dx;
            if (Nphi <= 16)
                dx = fmin(Dr, rDphi);
            else
                dx = fmax(Dr, rDphi);
            const double dx_sq = dx * dx;
            const double l_sq = (C * C) * dx_sq;
            const double rho = c->sigma[IDX(c, nr, naz)];
            c->qr[IDX(c, nr, naz)] = l_sq * rho * -div_V * (eps_rr - 1.0 / 3.0 * div_V);
            c->qphi[IDX(c, nr, naz)] = l_sq * rho * -div_V * (eps_pp - 1.0 / 3.0 * div_V);
            if (adi && c->d.artificial_viscosity_dissipation) {
                if (nr > c->s.zero_no_ghost && nr < c->s.max_no_ghost) {
                    const double Qplus =
                        -l_sq * div_V * rho * 1.0 / 3.0 *
                        (eps_rr * eps_rr + eps_pp * eps_pp + (eps_rr - eps_pp) * (eps_rr - eps_pp));
                    c->energy[IDX(c, nr, naz)] += Qplus * dt;
                }
            }
        }
#pragma omp parallel for if (c->big)
    for (int nr = 1; nr < Nr - 1; ++nr)
        for (int naz = 0; naz < Nphi; ++naz) {
            const int naz_prev = (naz == 0 ? Nphi - 1 : naz - 1);
            const double sigma_phi_avg = 0.5 * (c->sigma[IDX(c, nr, naz)] + c->sigma[IDX(c, nr, naz_prev)]);
            const double dVp = 2.0 * dt / ((c->Rsup[nr] + c->Rinf[nr]) * sigma_phi_avg) *
                               (c->qphi[IDX(c, nr, naz)] - c->qphi[IDX(c, nr, naz_prev)]) * c->invdphi;
            c->vazi[IDX(c, nr, naz)] += dVp;
        }
#pragma omp parallel for if (c->big)
    for (int nr = c->s.one_no_ghost_vr; nr < c->s.maxmo_no_ghost_vr; ++nr)
        for (int naz = 0; naz < Nphi; ++naz) {
            const double sigma_r_avg = 0.5 * (c->sigma[IDX(c, nr, naz)] + c->sigma[IDX(c, nr - 1, naz)]);
            const double rm = c->Rmed[nr], rmm = c->Rmed[nr - 1];
            const double dVr =
                c->d.radial_viscosity_factor * dt / sigma_r_avg * 2.0 / (rm * rm - rmm * rmm) *
                ((c->qr[IDX(c, nr, naz)] * rm - c->qr[IDX(c, nr - 1, naz)] * rmm) -
                 0.5 * (c->qphi[IDX(c, nr, naz)] + c->qphi[IDX(c, nr - 1, naz)]) * (rm - rmm));
            c->vrad[IDX(c, nr, naz)] += dVr;
        }
}
/* viscosity/artificial_viscosity.cpp:148-250 update_with_artificial_viscosity_SN */
static void artificial_viscosity_SN(orc_ctx *c, double dt)
{
    if (c->d.artificial_viscosity != FCPT_ARTVISC_SN)
        return;
    const int Nr = c->nr, Nphi = c->nphi;
    const double C = c->d.artificial_viscosity_factor;
#pragma omp parallel for if (c->big)
    for (int nr = 0; nr < Nr; ++nr)
        for (int naz = 0; naz < Nphi; ++naz) {
            const double dv_r = c->vrad[IDX(c, nr + 1, naz)] - c->vrad[IDX(c, nr, naz)];
            if (dv_r < 0.0)
                c->qr[IDX(c, nr, naz)] = (C * C) * c->sigma[IDX(c, nr, naz)] * (dv_r * dv_r);
            else
                c->qr[IDX(c, nr, naz)] = 0.0;
            const int naz_next = naz == Nphi - 1 ? 0 : naz + 1;
            const double dv_phi = c->vazi[IDX(c, nr, naz_next)] - c->vazi[IDX(c, nr, naz)];
            if (dv_phi < 0.0)
                c->qphi[IDX(c, nr, naz)] = (C * C) * c->sigma[IDX(c, nr, naz)] * (dv_phi * dv_phi);
            else
                c->qphi[IDX(c, nr, naz)] = 0.0;
        }
    if (c->d.eos == FCPT_EOS_IDEAL && c->d.artificial_viscosity_dissipation) {
#pragma omp parallel for if (c->big)
        for (int nr = c->s.zero_no_ghost; nr < c->s.max_no_ghost; ++nr) {
            const double dxtheta = c->dphi * c->Rmed[nr];
            const double invdxtheta = 1.0 / dxtheta;
            for (int naz = 0; naz < Nphi; ++naz) {
                const int naz_next = naz == Nphi - 1 ? 0 : naz + 1;
                const double dv_r = c->vrad[IDX(c, nr + 1, naz)] - c->vrad[IDX(c, nr, naz)];
                const double dv_phi = c->vazi[IDX(c, nr, naz_next)] - c->vazi[IDX(c, nr, naz)];
                c->energy[IDX(c, nr, naz)] = c->energy[IDX(c, nr, naz)] -
                                             dt * c->qr[IDX(c, nr, naz)] * dv_r * c->InvDiffRsup[nr] -
                                             dt * c->qphi[IDX(c, nr, naz)] * dv_phi * invdxtheta;
            }
        }
    }
#pragma omp parallel for if (c->big)
    for (int nr = c->s.one_no_ghost_vr; nr < c->s.maxmo_no_ghost_vr; ++nr)
        for (int naz = 0; naz < Nphi; ++naz)
            c->vrad[IDX(c, nr, naz)] =
                c->vrad[IDX(c, nr, naz)] -
                dt * 2.0 / (c->sigma[IDX(c, nr, naz)] + c->sigma[IDX(c, nr - 1, naz)]) *
                    (c->qr[IDX(c, nr, naz)] - c->qr[IDX(c, nr - 1, naz)]) * c->InvDiffRmed[nr];
#pragma omp parallel for if (c->big)
    for (int nr = c->s.zero_no_ghost; nr < c->s.max_no_ghost; ++nr) {
        const double dxtheta = c->dphi * c->Rmed[nr];
        const double invdxtheta = 1.0 / dxtheta;
        for (int naz = 0; naz < Nphi; ++naz) {
            const int naz_prev = (naz == 0 ? Nphi - 1 : naz - 1);
            c->vazi[IDX(c, nr, naz)] =
                c->vazi[IDX(c, nr, naz)] -
                dt * 2.0 / (c->sigma[IDX(c, nr, naz)] + c->sigma[IDX(c, nr, naz_prev)]) *
                    (c->qphi[IDX(c, nr, naz)] - c->qphi[IDX(c, nr, naz_prev)]) * invdxtheta;
        }
    }
}
/* viscosity/artificial_viscosity.cpp:11-26 */
static void update_with_artificial_viscosity(orc_ctx *c, double dt)
{
    if (c->d.artificial_viscosity == FCPT_ARTVISC_TW)
        artificial_viscosity_TW(c, dt);
    else
        artificial_viscosity_SN(c, dt);
    if (c->d.eos == FCPT_EOS_IDEAL && c->d.artificial_viscosity_dissipation)
        assure_temperature_range(c);
}

/* SourceEuler.cpp:205-223 recalculate_viscosity (AspectRatioMode 0) */
static void recalculate_viscosity(orc_ctx *c)
{
    if (c->d.eos == FCPT_EOS_IDEAL) {
        compute_sound_speed(c);
        compute_scale_height(c);
    }
    update_viscosity(c);
}

/* viscosity/viscosity.cpp:139-254 compute_viscous_stress_tensor, :256-348 the StabilizeViscosity factors */
static void compute_viscous_stress_tensor(orc_ctx *c)
{
    const int Nr = c->nr, Nphi = c->nphi;
#pragma omp parallel for if (c->big)
    for (int nr = 0; nr < Nr; ++nr)
        for (int naz = 0; naz < Nphi; ++naz) {
            const int naz_next = (naz == Nphi - 1 ? 0 : naz + 1);
            c->divv[IDX(c, nr, naz)] =
                (c->vrad[IDX(c, nr + 1, naz)] * c->Rinf[nr + 1] - c->vrad[IDX(c, nr, naz)] * c->Rinf[nr]) *
                    c->InvDiffRsupRb[nr] +
                (c->vazi[IDX(c, nr, naz_next)] - c->vazi[IDX(c, nr, naz)]) * c->invdphi * c->InvRmed[nr];
        }
#pragma omp parallel for if (c->big)
    for (int nr = 0; nr < Nr; ++nr)
        for (int naz = 0; naz < Nphi; ++naz) {
            const double drr =
                (c->vrad[IDX(c, nr + 1, naz)] - c->vrad[IDX(c, nr, naz)]) * c->InvDiffRsup[nr];
            c->trr[IDX(c, nr, naz)] = 2.0 * c->viscosity[IDX(c, nr, naz)] * c->sigma[IDX(c, nr, naz)] *
                                      (drr - 1.0 / 3.0 * c->divv[IDX(c, nr, naz)]);
        }
#pragma omp parallel for if (c->big)
    for (int nr = 0; nr < Nr; ++nr)
        for (int naz = 0; naz < Nphi; ++naz) {
            const int naz_next = (naz == Nphi - 1 ? 0 : naz + 1);
            const double dpp =
                (c->vazi[IDX(c, nr, naz_next)] - c->vazi[IDX(c, nr, naz)]) * c->invdphi * c->InvRmed[nr] +
                0.5 * (c->vrad[IDX(c, nr + 1, naz)] + c->vrad[IDX(c, nr, naz)]) * c->InvRmed[nr];
            const double nu = c->viscosity[IDX(c, nr, naz)];
            const double sigma = c->sigma[IDX(c, nr, naz)];
            c->tpp[IDX(c, nr, naz)] = 2.0 * nu * sigma * (dpp - 1.0 / 3.0 * c->divv[IDX(c, nr, naz)]);
        }
#pragma omp parallel for if (c->big)
    for (int nr = 1; nr < Nr; ++nr)
        for (int naz = 0; naz < Nphi; ++naz) {
            const int naz_prev = (naz == 0 ? Nphi - 1 : naz - 1);
            const double dvazirdr = (c->vazi[IDX(c, nr, naz)] * c->InvRmed[nr] -
                                     c->vazi[IDX(c, nr - 1, naz)] * c->InvRmed[nr - 1]) *
                                    c->InvDiffRmed[nr];
            const double dvrdphi =
                (c->vrad[IDX(c, nr, naz)] - c->vrad[IDX(c, nr, naz_prev)]) * c->invdphi;
            const double drp = c->Rinf[nr] * dvazirdr + dvrdphi * c->InvRinf[nr];
            const double nu =
                0.25 * (c->viscosity[IDX(c, nr, naz)] + c->viscosity[IDX(c, nr - 1, naz)] +
                        c->viscosity[IDX(c, nr, naz_prev)] + c->viscosity[IDX(c, nr - 1, naz_prev)]);
            const double sigma = 0.25 * (c->sigma[IDX(c, nr, naz)] + c->sigma[IDX(c, nr - 1, naz)] +
                                         c->sigma[IDX(c, nr, naz_prev)] + c->sigma[IDX(c, nr - 1, naz_prev)]);
            c->trp[IDX(c, nr, naz)] = nu * sigma * drp;
            c->nusig_rp[IDX(c, nr, naz)] = nu * sigma; /* VISCOSITY_SIGMA_RP */
        }
    if (!c->d.stabilize_viscosity)
        return;
    /* viscosity.cpp:256-348: correction factors of the pseudo-implicit viscosity */
    const double dphi = c->dphi;
#pragma omp parallel for if (c->big)
    for (int nr = 0; nr < Nr; ++nr)
        for (int naz = 0; naz < Nphi; ++naz)
            c->nusig[IDX(c, nr, naz)] = c->viscosity[IDX(c, nr, naz)] * c->sigma[IDX(c, nr, naz)]; /* VISCOSITY_SIGMA */
#pragma omp parallel for if (c->big)
    for (int nr = 1; nr < Nr; ++nr)
        for (int naz = 0; naz < Nphi; ++naz) {
            const int naz_prev = (naz == 0 ? Nphi - 1 : naz - 1);
            const int naz_next = (naz == Nphi - 1 ? 0 : naz + 1);
            const double NuSig_rp = c->nusig_rp[IDX(c, nr, naz)];
            const double NuSig_rp_ip = c->nusig_rp[IDX(c, nr + 1, naz)];
            const double NuSig_rp_jp = c->nusig_rp[IDX(c, nr, naz_next)];
            const double NuSigma = c->nusig[IDX(c, nr, naz)];
            const double NuSigma_jm = c->nusig[IDX(c, nr, naz_prev)];
            const double NuSigma_im = c->nusig[IDX(c, nr - 1, naz)];
            const double Ra = c->Rinf[nr], Rap = c->Rinf[nr + 1];
            const double TwoDiffRaSq = 2.0 / (c->Rsup[nr] * c->Rsup[nr] - c->Rinf[nr] * c->Rinf[nr]);
            const double FourThirdInvRbInvdphiSq = 4.0 / 3.0 / c->Rmed[nr] * c->invdphi * c->invdphi;
            const double Ra3NuSigmaInvDiffRmed = NuSig_rp * pow(Ra, 3) * c->InvDiffRmed[nr];
            const double Ra3NuSigmaInvDiffRmed_p = NuSig_rp_ip * pow(Rap, 3) * c->InvDiffRmed[nr + 1];
            const double cphi_rp = -c->InvRmed[nr] * TwoDiffRaSq * (Ra3NuSigmaInvDiffRmed_p + Ra3NuSigmaInvDiffRmed);
            const double cphi_pp = -FourThirdInvRbInvdphiSq * (NuSigma + NuSigma_jm);
            const double sigma_avg_phi = 0.5 * (c->sigma[IDX(c, nr, naz)] + c->sigma[IDX(c, nr, naz_prev)]);
            c->cfac_phi[IDX(c, nr, naz)] = (cphi_rp + cphi_pp) / (sigma_avg_phi * c->Rmed[nr]);
            const double sigma_avg_r = 0.5 * (c->sigma[IDX(c, nr, naz)] + c->sigma[IDX(c, nr - 1, naz)]);
            const double cr_rp = -(NuSig_rp_jp + NuSig_rp) / (dphi * dphi * Ra);
            const double cr_pp_1 = 2.0 * NuSigma * (0.5 * c->InvRmed[nr] + 1.0 / 3.0 * Ra * c->InvDiffRsupRb[nr]);
            const double cr_pp_2 =
                2.0 * NuSigma_im * (0.5 * c->InvRmed[nr - 1] - 1.0 / 3.0 * Ra * c->InvDiffRsupRb[nr - 1]);
            const double cr_rr_1 =
                c->Rmed[nr] * 2.0 * NuSigma * (-c->InvDiffRsup[nr] + 1.0 / 3.0 * Ra * c->InvDiffRsupRb[nr]);
            const double cr_rr_2 = -1.0 * c->Rmed[nr - 1] * 2.0 * NuSigma_im *
                                   (c->InvDiffRsup[nr - 1] - 1.0 / 3.0 * Ra * c->InvDiffRsupRb[nr - 1]);
            const double cr_pp = -0.5 * (cr_pp_1 + cr_pp_2);
            const double cr_rr = c->InvDiffRmed[nr] * (cr_rr_1 + cr_rr_2);
            const double Rmed_mid = 0.5 * (c->Rmed[nr] + c->Rmed[nr - 1]);
            c->cfac_r[IDX(c, nr, naz)] =
                c->d.radial_viscosity_factor * (cr_rr + cr_rp + cr_pp) / (sigma_avg_r * Rmed_mid);
        }
}
/* viscosity/viscosity.cpp:355-426 update_velocities_with_viscosity */
static void update_velocities_with_viscosity(orc_ctx *c, double dt)
{
    const int Nr = c->nr, Nphi = c->nphi;
#pragma omp parallel for if (c->big)
    for (int nr = 1; nr < Nr - 1; ++nr)
        for (int naz = 0; naz < Nphi; ++naz) {
            const int naz_prev = (naz == 0 ? Nphi - 1 : naz - 1);
            const double sigma_avg = 0.5 * (c->sigma[IDX(c, nr, naz)] + c->sigma[IDX(c, nr, naz_prev)]);
            const double ra1 = c->Rinf[nr + 1], ra0 = c->Rinf[nr];
            const double dVp =
                dt * c->InvRmed[nr] / (sigma_avg) *
                ((2.0 / (ra1 * ra1 - ra0 * ra0)) *
                     (ra1 * ra1 * c->trp[IDX(c, nr + 1, naz)] - ra0 * ra0 * c->trp[IDX(c, nr, naz)]) +
                 (c->tpp[IDX(c, nr, naz)] - c->tpp[IDX(c, nr, naz_prev)]) * c->invdphi);
            double dVp_ = dVp;
            if (c->d.stabilize_viscosity == 1) { /* viscosity.cpp:386-391 */
                const double cphi = c->cfac_phi[IDX(c, nr, naz)];
                dVp_ *= 1.0 / (fmax(1.0 + dt * cphi, 0.0) - dt * cphi);
            }
            c->vazi[IDX(c, nr, naz)] += dVp_;
        }
#pragma omp parallel for if (c->big)
    for (int nr = c->s.one_no_ghost_vr; nr < c->s.maxmo_no_ghost_vr; ++nr)
        for (int naz = 0; naz < Nphi; ++naz) {
            const int naz_next = (naz == Nphi - 1 ? 0 : naz + 1);
            const double sigma_avg = 0.5 * (c->sigma[IDX(c, nr, naz)] + c->sigma[IDX(c, nr - 1, naz)]);
            const double dVr =
                dt / (sigma_avg)*c->d.radial_viscosity_factor * 2.0 / (c->Rmed[nr] + c->Rmed[nr - 1]) *
                ((c->Rmed[nr] * c->trr[IDX(c, nr, naz)] - c->Rmed[nr - 1] * c->trr[IDX(c, nr - 1, naz)]) *
                     c->InvDiffRmed[nr] +
                 (c->trp[IDX(c, nr, naz_next)] - c->trp[IDX(c, nr, naz)]) * c->invdphi -
                 0.5 * (c->tpp[IDX(c, nr, naz)] + c->tpp[IDX(c, nr - 1, naz)]));
            double dVr_ = dVr;
            if (c->d.stabilize_viscosity == 1) { /* viscosity.cpp:413-417 */
                const double cr = c->cfac_r[IDX(c, nr, naz)];
                dVr_ *= 1.0 / (fmax(1.0 + dt * cr, 0.0) - dt * cr);
            }
            c->vrad[IDX(c, nr, naz)] += dVr_;
        }
}

/* SourceEuler.cpp:496-536 viscous_heating, :614-630 calculate_qplus, :931-950 calculate_qminus */

/* opacity.cpp:45-168 lin(): Lin & Papaloizou (1985) opacities, cgs in / cgs out */
static double opacity_lin(double density, double temperature)
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
                return pow(pow(o6an * o7an / (o6an + o7an), 2) +
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
            const double o4an = pow(o4, 4), o3an = pow(o3, 4);
            return pow((o4an * o3an / (o4an + o3an)) + pow(o5 / (1.0 + 6.561e-5 / ts48), 4), 0.25);
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
        return pow(pow(o1an * o2an / (o1an + o2an), 2) + pow(o3 / (1 + 1.e22 / t10), 4), 0.25);
    }
}
/* opacity.cpp:170-297 bell(): Bell & Lin (1994) opacities in eight regions, smoothed across their borders; cgs in / cgs out */
static double opacity_bell(double density, double temperature)
{
    const double power1 = 2.8369e-2, power2 = 1.1464e-2, power3 = 2.2667e-1;
    const double t234 = 1.46e3, t456 = 4.51e3, t678 = 2.37e6;
    const double ak1 = 2.e-4, ak2 = 2.e16, ak3 = 0.1e0;
    const double bk3 = 10., bk4 = 2.e-15, bk5 = 1e4, bk6 = 1e4, bk7 = 1.5e10, bk8 = 0.348;
    if (temperature < 1.0)
        temperature = 10.0;
    if (temperature > t234 * pow(density, power1)) {
        const double ts4 = 1.e-4 * temperature; /* to avoid overflow */
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
/* opacity.cpp:10-43 opacity(): code units in / code units out */
static double opacity_of(const fcpt_desc *d, double density, double temperature)
{
    const double temperatureCGS = temperature * d->temperature_cgs;
    const double densityCGS = density * d->density_cgs;
    double rv;
    switch (d->opacity) {
    case FCPT_OPACITY_LIN: rv = opacity_lin(densityCGS, temperatureCGS) * (1.0 / d->opacity_cgs); break;
    case FCPT_OPACITY_BELL: rv = opacity_bell(densityCGS, temperatureCGS) * (1.0 / d->opacity_cgs); break;
    case FCPT_OPACITY_CONST: rv = d->kappa_const; break;
    case FCPT_OPACITY_SIMPLE: rv = d->kappa_const * (temperatureCGS * temperatureCGS); break;
    default: rv = 0.0; break;
    }
    return d->kappa_factor * rv;
}
/* compute.cpp:17-35 midplane_density + :41-87 kappa_eff at one cell: returns tau_eff */
static double tau_eff_of(const fcpt_desc *d, int heating_star, double sigma, double H, double temperature)
{
    const double rho = sigma / (d->density_factor * H);
    const double kappa = opacity_of(d, rho, temperature);
    const double tau = d->tau_factor * (1.0 / d->density_factor) * kappa * sigma;
    if (d->opacity == FCPT_OPACITY_SIMPLE)
        return 3.0 / 8.0 * tau; /* D'Angelo et al. 2003 eq. (28) */
    if (heating_star) /* irradiated disk, D'Angelo & Marzari 2012 */
        return 3.0 / 8.0 * tau + 0.5 + 1.0 / (4.0 * tau + d->tau_min);
    return 3.0 / 8.0 * tau + sqrt(3.0) / 4.0 + 1.0 / (4.0 * tau + d->tau_min);
}
/* SourceEuler.cpp:632-786 thermal_relaxation (beta cooling; the opacity-based Ziampras variants are
 * not part of the path) */
static void thermal_relaxation(orc_ctx *c, double current_time)
{
    const int Nr = c->nr - 1, Nphi = c->nphi;
    const fcpt_desc *d = &c->d;
#pragma omp parallel for if (c->big)
    for (int nr = 1; nr < Nr; ++nr)
        for (int naz = 0; naz < Nphi; ++naz) {
            const double r = c->Rmed[nr];
            const double omega_k = sqrt(d->G * d->hydro_center_mass / (r * r * r));
            const double E = c->energy[IDX(c, nr, naz)];
            const double t_ramp_up = d->cooling_beta_ramp_up;
            double beta_inv = 1 / d->cooling_beta_value;
            if (t_ramp_up > 0.0) {
                const double x = 2 * current_time / t_ramp_up;
                const double ramp_factor = 1 - exp(-(x * x));
                beta_inv = beta_inv * ramp_factor;
            }
            double delta_E = E;
            if (d->cooling_beta_reference == FCPT_BETAREF_REFERENCE) {
                delta_E -= c->energy0[IDX(c, nr, naz)] / c->sigma0[IDX(c, nr, naz)] * c->sigma[IDX(c, nr, naz)];
            } else if (d->cooling_beta_reference == FCPT_BETAREF_MODEL) {
                const double E0 = 1.0 / (d->adiabatic_index - 1.0) * (d->aspect_ratio * d->aspect_ratio) *
                                  pow(c->Rmed[nr], 2.0 * d->flaring_index - 1.0) * d->G * d->hydro_center_mass *
                                  c->sigma[IDX(c, nr, naz)];
                delta_E -= E0;
            } else if (d->cooling_beta_reference == FCPT_BETAREF_FLOOR) {
                const double minimum_energy =
                    d->minimum_temperature * c->sigma[IDX(c, nr, naz)] / d->mu * d->Rgas / (d->adiabatic_index - 1.0);
                delta_E -= minimum_energy;
            }
            c->qminus[IDX(c, nr, naz)] += delta_E * omega_k * beta_inv;
        }
}
/* SourceEuler.cpp:790-820 thermal_cooling (SurfaceCooling: thermal) */
static void kappa_eff(orc_ctx *c)
{
    const int Nphi = c->nphi;
    const fcpt_desc *d = &c->d;
    /* midplane_density recomputes the scale height from the current sound speed (compute.cpp:19) */
    compute_scale_height(c);
#pragma omp parallel for if (c->big)
    for (int nr = 0; nr < c->nr; ++nr)
        for (int naz = 0; naz < Nphi; ++naz)
            c->tau_eff[IDX(c, nr, naz)] = tau_eff_of(d, c->heating_star, c->sigma[IDX(c, nr, naz)],
                                                     c->scale_height[IDX(c, nr, naz)], c->temperature[IDX(c, nr, naz)]);
}
static void thermal_cooling(orc_ctx *c)
{
    const int Nr = c->nr - 1, Nphi = c->nphi;
    const fcpt_desc *d = &c->d;
    kappa_eff(c);
#pragma omp parallel for if (c->big)
    for (int nr = 1; nr < Nr; ++nr)
        for (int naz = 0; naz < Nphi; ++naz) {
            const double T = c->temperature[IDX(c, nr, naz)];
            const double T2 = T * T, Tm2 = d->minimum_temperature * d->minimum_temperature;
            const double T4 = T2 * T2, Tmin4 = Tm2 * Tm2;
            c->qminus[IDX(c, nr, naz)] +=
                d->cooling_radiative_factor * 2 * d->sigma_sb * (T4 - Tmin4) / c->tau_eff[IDX(c, nr, naz)];
        }
}
static void calculate_qminus(orc_ctx *c, double current_time)
{
    memset(c->qminus, 0, sizeof(double) * (size_t)c->nr * c->nphi);
    if (c->d.cooling_beta)
        thermal_relaxation(c, current_time);
    if (c->d.cooling_surface)
        thermal_cooling(c);
}
static void calculate_qplus(orc_ctx *c)
{
    const int Nr_m1 = c->nr - 1, Nphi = c->nphi;
    memset(c->qplus, 0, sizeof(double) * (size_t)c->nr * c->nphi);
    if (!c->d.heating_viscous)
        return;
#pragma omp parallel for if (c->big)
    for (int nr = 1; nr < Nr_m1; ++nr)
        for (int naz = 0; naz < Nphi; ++naz) {
            if (c->viscosity[IDX(c, nr, naz)] != 0.0) {
                const int naz_next = (naz == Nphi - 1 ? 0 : naz + 1);
                const double tau_r_phi = 0.25 * (c->trp[IDX(c, nr, naz)] + c->trp[IDX(c, nr + 1, naz)] +
                                                 c->trp[IDX(c, nr, naz_next)] + c->trp[IDX(c, nr + 1, naz_next)]);
                const double trr = c->trr[IDX(c, nr, naz)], tpp = c->tpp[IDX(c, nr, naz)];
                const double dv = c->divv[IDX(c, nr, naz)];
                double qplus = 1.0 / (2.0 * c->viscosity[IDX(c, nr, naz)] * c->sigma[IDX(c, nr, naz)]) *
                               (trr * trr + 2 * (tau_r_phi * tau_r_phi) + tpp * tpp);
                qplus += (2.0 / 9.0) * c->viscosity[IDX(c, nr, naz)] * c->sigma[IDX(c, nr, naz)] * (dv * dv);
                qplus *= c->d.heating_viscous_factor;
                c->qplus[IDX(c, nr, naz)] += qplus;
            }
        }
}
/* SourceEuler.cpp:538-612 irradiation_single / irradiation: heating by the bodies that carry a temperature */
static void irradiation(orc_ctx *c, double current_time)
{
    const int Nrad = c->nr - 1, Nphi = c->nphi;
    const fcpt_desc *d = &c->d;
    for (int k = 0; k < c->nbodies; ++k) {
        if (!(c->btemp[k] > 0.0))
            continue;
        const double rampup_time = c->bramp[k];
        double ramping = 1.0;
        if (current_time < rampup_time) {
            const double cs = cos(current_time * M_PI / 2.0 / rampup_time);
            ramping = 1.0 - cs * cs;
        }
        const double x = c->bx[k], y = c->by[k];
        const double R_star = c->bradius[k], T_star = c->btemp[k];
        /* l1 * cubic smoothing factor is the cubic smoothing radius handed to fcpt_set_bodies */
        const double min_dist = (x * x + y * y > 1e-10) ? fmax(R_star, c->brsm[k]) : R_star;
        const double Ts2 = T_star * T_star;
#pragma omp parallel for if (c->big)
        for (int nrad = 1; nrad < Nrad; ++nrad)
            for (int naz = 0; naz < Nphi; ++naz) {
                const double xc = c->Rmed[nrad] * cos(c->dphi * (double)naz);
                const double yc = c->Rmed[nrad] * sin(c->dphi * (double)naz);
                const double distance_measured = sqrt((x - xc) * (x - xc) + (y - yc) * (y - yc));
                const double distance = fmax(distance_measured, min_dist);
                const double roverd = distance < R_star ? 1.0 : R_star / distance;
                const double HoverR = c->scale_height[IDX(c, nrad, naz)] / c->Rmed[nrad]; /* ASPECTRATIO */
                const double tau_eff = c->tau_eff[IDX(c, nrad, naz)];
                const double eps = 0.5;
                const double dlogH_dlogr = 9.0 / 7.0; /* Chiang & Goldreich (1997) */
                const double W_G = 0.4 * roverd + HoverR * (dlogH_dlogr - 1.0);
                const double T_irrad_pow4 = (1.0 - eps) * (Ts2 * Ts2) * (roverd * roverd) * W_G;
                const double qplus = 2.0 * d->sigma_sb * T_irrad_pow4 / tau_eff;
                c->qplus[IDX(c, nrad, naz)] += ramping * qplus;
            }
    }
}
/* the alpha factor shared by SubStep3 (SourceEuler.cpp:1005-1024) and
 * compute_heating_cooling_for_CFL (:1520-1545) */
static double substep3_alpha(const orc_ctx *c, double H, double sigma, double energy)
{
    const double mu = c->d.mu, gamma = c->d.adiabatic_index, Rgas = c->d.Rgas;
    const double b = mu * (gamma - 1.0) / (Rgas * sigma);
    const double b2 = b * b;
    const double inv_pow4 = b2 * b2;
    return 1.0 + 2.0 * H * 4.0 * c->d.sigma_sb / c->d.c_light * inv_pow4 * (energy * energy * energy);
}
/* SourceEuler.cpp:956-1051 SubStep3 */
static void substep3(orc_ctx *c, double current_time, double dt)
{
    const int Nr = c->nr, Nphi = c->nphi;
    compute_temperature(c);
    calculate_qminus(c, current_time);
    calculate_qplus(c);
    if (c->heating_star) {
        if (!c->d.cooling_surface)
            kappa_eff(c);
        irradiation(c, current_time);
    }
#pragma omp parallel for if (c->big)
    for (int nr = 1; nr < Nr - 1; ++nr)
        for (int naz = 0; naz < Nphi; ++naz) {
            const double H = c->scale_height[IDX(c, nr, naz)];
            const double sigma = c->sigma[IDX(c, nr, naz)];
            const double energy = c->energy[IDX(c, nr, naz)];
            const double alpha = substep3_alpha(c, H, sigma, energy);
            c->qplus[IDX(c, nr, naz)] /= alpha;
            c->qminus[IDX(c, nr, naz)] /= alpha;
            const double Qplus = c->qplus[IDX(c, nr, naz)];
            const double Qminus = c->qminus[IDX(c, nr, naz)];
            double energy_new = energy + dt * (Qplus - Qminus);
            const double SigmaFloor = 10.0 * c->d.sigma0 * c->d.sigma_floor;
            if (sigma < SigmaFloor) {
                const double tau_eff = c->tau_eff[IDX(c, nr, naz)];
                const double e4 = Qplus * tau_eff / (2.0 * c->d.sigma_sb);
                const double constant = (c->d.Rgas / c->d.mu * sigma / (c->d.adiabatic_index - 1.0));
                const double eq_energy = sqrt(sqrt(e4)) * constant;
                c->qminus[IDX(c, nr, naz)] = Qplus;
                energy_new = eq_energy;
            }
            c->energy[IDX(c, nr, naz)] = energy_new;
        }
    assure_temperature_range(c);
}
/* SourceEuler.cpp:1507-1547 compute_heating_cooling_for_CFL */
static void compute_heating_cooling_for_CFL(orc_ctx *c)
{
    if (c->d.eos != FCPT_EOS_IDEAL)
        return;
    update_viscosity(c);
    compute_viscous_stress_tensor(c);
    compute_temperature(c);
    {
        /* at this point of init_euler the reference copies (SIGMA0, ENERGY0) do not exist yet
         * (copy_initial_values follows, init.cpp:337): the beta term relative to them is 0 */
        const int ref = c->d.cooling_beta_reference;
        const int beta = c->d.cooling_beta;
        if (ref == FCPT_BETAREF_REFERENCE)
            c->d.cooling_beta = 0;
        calculate_qminus(c, c->clk.time);
        c->d.cooling_beta = beta;
    }
    calculate_qplus(c);
    if (c->heating_star) {
        if (!c->d.cooling_surface)
            kappa_eff(c);
        irradiation(c, c->clk.time);
    }
    const int Nr = c->nr - 1, Nphi = c->nphi;
    for (int nr = 1; nr < Nr; ++nr)
        for (int naz = 0; naz < Nphi; ++naz) {
            const double alpha = substep3_alpha(c, c->scale_height[IDX(c, nr, naz)],
                                                c->sigma[IDX(c, nr, naz)], c->energy[IDX(c, nr, naz)]);
            c->qplus[IDX(c, nr, naz)] /= alpha;
            c->qminus[IDX(c, nr, naz)] /= alpha;
        }
}

/* ------------------------------------------------------------------------ */
/* transport                                                                 */

/* TransportEuler.cpp:306-337 */
static double van_leer_lim(double a, double b)
{
    if (a * b > 0.0)
        return 2.0 * a * b / (a + b);
    return 0;
}
static double minmod(double a, double b)
{
    if (a * b > 0.0)
        return fabs(a) < fabs(b) ? a : b;
    return 0.0;
}
static double flux_limiter(const orc_ctx *c, double a, double b)
{
    if (c->d.flux_limiter == FCPT_LIMITER_MC)
        return minmod(0.5 * (a + b), 2.0 * minmod(a, b));
    return van_leer_lim(a, b);
}
/* TransportEuler.cpp:471-493 compute_momenta_from_velocities */
static void compute_momenta_from_velocities(orc_ctx *c)
{
    const int Nr = c->nr, Nphi = c->nphi;
    const double OmegaF = c->d.omega_frame;
#pragma omp parallel for if (c->big)
    for (int nr = 0; nr < Nr; ++nr)
        for (int naz = 0; naz < Nphi; ++naz) {
            const double S = c->sigma[IDX(c, nr, naz)];
            c->rmp[IDX(c, nr, naz)] = S * c->vrad[IDX(c, nr + 1, naz)];
            c->rmm[IDX(c, nr, naz)] = S * c->vrad[IDX(c, nr, naz)];
            const int naz_ind = naz == (Nphi - 1) ? 0 : naz + 1;
            const double vnext = c->vazi[IDX(c, nr, naz_ind)];
            const double r = c->Rmed[nr];
            c->lp[IDX(c, nr, naz)] = S * (vnext + r * OmegaF) * r;
            c->lm[IDX(c, nr, naz)] = S * (c->vazi[IDX(c, nr, naz)] + r * OmegaF) * r;
        }
}
/* TransportEuler.cpp:498-535 compute_velocities_from_momenta */
static void compute_velocities_from_momenta(orc_ctx *c)
{
    const int Nr = c->nr, Nphi = c->nphi;
#pragma omp parallel for if (c->big)
    for (int nr = 0; nr < Nr; ++nr)
        for (int naz = 0; naz < Nphi; ++naz) {
            const int nm = (naz == 0 ? (Nphi - 1) : naz - 1);
            if (nr == 0)
                c->vrad[IDX(c, nr, naz)] = 0.0;
            else
                c->vrad[IDX(c, nr, naz)] = (c->rmp[IDX(c, nr - 1, naz)] + c->rmm[IDX(c, nr, naz)]) /
                                           (c->sigma[IDX(c, nr - 1, naz)] + c->sigma[IDX(c, nr, naz)]);
            c->vazi[IDX(c, nr, naz)] = (c->lp[IDX(c, nr, nm)] + c->lm[IDX(c, nr, naz)]) /
                                           (c->sigma[IDX(c, nr, nm)] + c->sigma[IDX(c, nr, naz)]) *
                                           c->InvRmed[nr] -
                                       c->Rmed[nr] * c->d.omega_frame;
        }
}
/* TransportEuler.cpp:349-406 compute_star_radial */
static void compute_star_radial(orc_ctx *c, const double *Qbase, const double *VRadial, double *QStar,
                                double dt)
{
    const int Nr = c->nr, Nphi = c->nphi;
    double *dq = c->dq;
#pragma omp parallel for if (c->big)
    for (int nr = 0; nr < Nr; ++nr)
        for (int naz = 0; naz < Nphi; ++naz) {
            const size_t cell = IDX(c, nr, naz);
            if ((nr == 0) || (nr == Nr - 1)) {
                dq[cell] = 0.0;
            } else {
                const double dqm = (Qbase[cell] - Qbase[cell - Nphi]) * c->InvDiffRmed[nr];
                const double dqp = (Qbase[cell + Nphi] - Qbase[cell]) * c->InvDiffRmed[nr + 1];
                dq[cell] = flux_limiter(c, dqp, dqm);
            }
        }
#pragma omp parallel for if (c->big)
    for (int nr = 1; nr < Nr; ++nr)
        for (int naz = 0; naz < Nphi; ++naz) {
            const size_t cell = IDX(c, nr, naz);
            const size_t prev = cell - Nphi;
            if (VRadial[cell] > 0.0)
                QStar[cell] =
                    Qbase[prev] + (c->Rmed[nr] - c->Rmed[nr - 1] - VRadial[cell] * dt) * 0.5 * dq[prev];
            else
                QStar[cell] =
                    Qbase[cell] - (c->Rmed[nr + 1] - c->Rmed[nr] + VRadial[cell] * dt) * 0.5 * dq[cell];
        }
    for (int naz = 0; naz < Nphi; ++naz)
        QStar[IDX(c, 0, naz)] = 0.0;
}
/* SideEuler.cpp:26-40 divise_polargrid */
static void divise(const orc_ctx *c, const double *num, const double *den, double *res)
{
    const size_t n = (size_t)c->nr * c->nphi;
#pragma omp parallel for if (c->big)
    for (size_t i = 0; i < n; ++i)
        res[i] = num[i] / den[i];
}
/* TransportEuler.cpp:545-620 VanLeerRadial (of the mass-flow bookkeeping: the MASSFLOW grid, :609-616) */
static void VanLeerRadial(orc_ctx *c, const double *VRadial, double *Qbase, double dt)
{
    const int Nr = c->nr, Nphi = c->nphi;
    const int is_density = Qbase == c->sigma;
    divise(c, Qbase, c->density_int, c->work);
    compute_star_radial(c, c->work, VRadial, c->qrstar, dt);
#pragma omp parallel for if (c->big)
    for (int nr = 0; nr < Nr; ++nr)
        for (int naz = 0; naz < Nphi; ++naz) {
            const size_t cell = IDX(c, nr, naz);
            const size_t lip = cell + Nphi;
            const double varq_inf =
                dt * c->dphi * c->Rinf[nr] * c->qrstar[cell] * c->densstar[cell] * VRadial[cell];
            const double varq_sup =
                dt * c->dphi * c->Rsup[nr] * c->qrstar[lip] * c->densstar[lip] * VRadial[lip];
            Qbase[cell] += (varq_inf - varq_sup) * c->InvSurf[nr];
            if (is_density && c->massflow) { /* parameters::write_massflow */
                c->massflow[cell] += varq_inf;
                if (c->s.is_last && nr == Nr - 1) /* Qbase->get_max_radial() */
                    c->massflow[cell] += varq_sup;
            }
        }
}
/* TransportEuler.cpp:138-167 OneWindRad */
static void OneWindRad(orc_ctx *c, double dt)
{
    const int adi = c->d.eos == FCPT_EOS_IDEAL;
    compute_star_radial(c, c->sigma, c->vrad, c->densstar, dt);
    memcpy(c->density_int, c->sigma, sizeof(double) * (size_t)c->nr * c->nphi);
    VanLeerRadial(c, c->vrad, c->rmp, dt);
    VanLeerRadial(c, c->vrad, c->rmm, dt);
    VanLeerRadial(c, c->vrad, c->lp, dt);
    VanLeerRadial(c, c->vrad, c->lm, dt);
    if (adi)
        VanLeerRadial(c, c->vrad, c->energy, dt);
    VanLeerRadial(c, c->vrad, c->sigma, dt); /* MUST be the last line */
}
/* TransportEuler.cpp:416-466 ComputeStarTheta */
static void ComputeStarTheta(orc_ctx *c, const double *Qbase, const double *VAzimuthal, double *QStar,
                             double dt)
{
    const int Nr = c->nr, Nphi = c->nphi;
    double *dq = c->dq;
#pragma omp parallel for if (c->big)
    for (int nr = 0; nr < Nr; ++nr) {
        const double dxtheta = c->dphi * c->Rmed[nr];
        const double invdxtheta = 1.0 / dxtheta;
        for (int naz = 0; naz < Nphi; ++naz) {
            const size_t cell = IDX(c, nr, naz);
            size_t ljp = cell + 1, ljm = cell - 1;
            if (naz == 0)
                ljm = (size_t)nr * Nphi + Nphi - 1;
            if (naz == Nphi - 1)
                ljp = (size_t)nr * Nphi;
            const double dqm = (Qbase[cell] - Qbase[ljm]);
            const double dqp = (Qbase[ljp] - Qbase[cell]);
            dq[cell] = 0.5 * flux_limiter(c, dqp, dqm) * invdxtheta;
        }
    }
#pragma omp parallel for if (c->big)
    for (int nr = 0; nr < Nr; ++nr) {
        const double dxtheta = c->dphi * c->Rmed[nr];
        for (int naz = 0; naz < Nphi; ++naz) {
            const size_t cell = IDX(c, nr, naz);
            int jm = naz - 1;
            if (naz == 0)
                jm = Nphi - 1;
            const size_t ljm = (size_t)jm + (size_t)nr * Nphi;
            const double ksi = VAzimuthal[cell] * dt;
            if (ksi > 0.0)
                QStar[cell] = Qbase[ljm] + (dxtheta - ksi) * dq[ljm];
            else
                QStar[cell] = Qbase[cell] - (dxtheta + ksi) * dq[cell];
        }
    }
}
/* TransportEuler.cpp:630-664 VanLeerTheta */
static void VanLeerTheta(orc_ctx *c, const double *VAzimuthal, double *Qbase, double dt, int uniform)
{
    const int Nr = c->nr, Nphi = c->nphi;
    divise(c, Qbase, c->density_int, c->work);
    ComputeStarTheta(c, c->work, VAzimuthal, c->qrstar, dt);
#pragma omp parallel for if (c->big)
    for (int nr = 0; nr < Nr; ++nr) {
        const double dxrad = (c->Rsup[nr] - c->Rinf[nr]) * dt;
        const double invsurf = c->InvSurf[nr];
        if (!uniform || !c->nosplit[nr]) {
            for (int naz = 0; naz < Nphi; ++naz) {
                const size_t cell = IDX(c, nr, naz);
                size_t ljp = cell + 1;
                if (naz == Nphi - 1)
                    ljp = (size_t)nr * Nphi;
                double varq = dxrad * c->qrstar[cell] * c->densstar[cell] * VAzimuthal[cell];
                varq -= dxrad * c->qrstar[ljp] * c->densstar[ljp] * VAzimuthal[ljp];
                Qbase[cell] += varq * invsurf;
            }
        }
    }
}
/* TransportEuler.cpp:292-304 QuantitiesAdvection */
static void QuantitiesAdvection(orc_ctx *c, const double *VAzimuthal, double dt, int uniform)
{
    const int adi = c->d.eos == FCPT_EOS_IDEAL;
    ComputeStarTheta(c, c->sigma, VAzimuthal, c->densstar, dt);
    memcpy(c->density_int, c->sigma, sizeof(double) * (size_t)c->nr * c->nphi);
    VanLeerTheta(c, VAzimuthal, c->rmp, dt, uniform);
    VanLeerTheta(c, VAzimuthal, c->rmm, dt, uniform);
    VanLeerTheta(c, VAzimuthal, c->lp, dt, uniform);
    VanLeerTheta(c, VAzimuthal, c->lm, dt, uniform);
    if (adi)
        VanLeerTheta(c, VAzimuthal, c->energy, dt, uniform);
    VanLeerTheta(c, VAzimuthal, c->sigma, dt, uniform); /* MUST be the last line */
}
/* TransportEuler.cpp:238-268 AdvectSHIFT */
static void AdvectSHIFT(orc_ctx *c, double *val)
{
    const int nr = c->nr, ns = c->nphi;
#pragma omp parallel for if (c->big)
    for (int i = 0; i < nr; i++)
        for (int j = 0; j < ns; j++) {
            int ji = j - c->nshift[i];
            while (ji < 0)
                ji += ns;
            while (ji >= ns)
                ji -= ns;
            c->tempshift[j + (size_t)i * ns] = val[ji + (size_t)i * ns];
        }
    memcpy(val, c->tempshift, sizeof(double) * (size_t)nr * ns);
}
/* TransportEuler.cpp:270-288 OneWindTheta, :174-236 helpers */
static void OneWindTheta(orc_ctx *c, double dt)
{
    const int Nr = c->nr, Nphi = c->nphi;
    const int adi = c->d.eos == FCPT_EOS_IDEAL;
    /* compute_average_azimuthal_velocity */
#pragma omp parallel for if (c->big)
    for (int nr = 0; nr < Nr; ++nr) {
        double sum = 0.0;
        for (int naz = 0; naz < Nphi; ++naz)
            sum += c->vazi[IDX(c, nr, naz)];
        c->vmean[nr] = sum / (double)Nphi;
    }
    /* compute_residual_velocity */
#pragma omp parallel for if (c->big)
    for (int nr = 0; nr < Nr; ++nr)
        for (int naz = 0; naz < Nphi; ++naz)
            c->vres[IDX(c, nr, naz)] = c->vazi[IDX(c, nr, naz)] - c->vmean[nr];
    /* ComputeConstantResidual */
    const double invdt = 1.0 / dt;
#pragma omp parallel for if (c->big)
    for (int i = 0; i < Nr; i++) {
        const double Ntilde = c->vmean[i] * c->InvRmed[i] * dt * c->invdphi;
        const double Nround = floor(Ntilde + 0.5);
        c->nshift[i] = (int)Nround;
        for (int j = 0; j < Nphi; j++)
            c->vazi[IDX(c, i, j)] = (Ntilde - Nround) * c->Rmed[i] * invdt * c->dphi;
        if (!c->d.fast_transport) {
            c->nosplit[i] = 1;
            for (int j = 0; j < Nphi; j++) {
                c->vres[IDX(c, i, j)] = c->vazi[IDX(c, i, j)] + c->vres[IDX(c, i, j)];
                c->vazi[IDX(c, i, j)] = 0.0;
            }
        } else {
            c->nosplit[i] = 0;
        }
    }
    QuantitiesAdvection(c, c->vres, dt, 0);
    QuantitiesAdvection(c, c->vazi, dt, 1);
    AdvectSHIFT(c, c->rmp);
    AdvectSHIFT(c, c->rmm);
    AdvectSHIFT(c, c->lp);
    AdvectSHIFT(c, c->lm);
    if (adi)
        AdvectSHIFT(c, c->energy);
    AdvectSHIFT(c, c->sigma);
}
/* TransportEuler.cpp:112-136 Transport */
static void Transport(orc_ctx *c, double dt)
{
    compute_momenta_from_velocities(c);
    OneWindRad(c, dt);
    OneWindTheta(c, dt);
    compute_velocities_from_momenta(c);
    assure_minimum_value(c, c->sigma, c->d.sigma_floor * c->d.sigma0);
    if (c->d.eos == FCPT_EOS_IDEAL)
        assure_temperature_range(c);
}

/* ------------------------------------------------------------------------ */
/* cfl.cpp:185-376 condition_cfl without the Allreduce */
int orc_cfl(orc_ctx *c, double *dt_local)
{
    if (!c || !dt_local)
        return FCPT_EINVAL;
    const int Nr = c->nr, Nphi = c->nphi;
    const fcpt_desc *d = &c->d;
    double *v_mean = c->cfl_vmean;
    for (int nr = 0; nr < Nr; ++nr) {
        v_mean[nr] = 0.0;
        for (int naz = 0; naz < Nphi; ++naz)
            v_mean[nr] += c->vazi[IDX(c, nr, naz)];
        v_mean[nr] /= (double)Nphi;
    }
    const double denom0 = fabs(v_mean[0] * c->InvRmed[0] - v_mean[1] * c->InvRmed[1]) + 1.0e-100;
    double dt_core = d->cfl * c->dphi / denom0;
    const double lf = d->integrator == FCPT_INTEGRATOR_LEAPFROG ? 0.6 : 1.0;
    const double C = d->artificial_viscosity_factor;
    for (int nr = c->s.radial_first_active; nr < c->s.radial_active_size; ++nr) {
        const double denom = fabs(v_mean[nr] * c->InvRmed[nr] - v_mean[nr + 1] * c->InvRmed[nr + 1]) + 1.0e-100;
        const double shear_dt = d->cfl * c->dphi / denom;
        if (shear_dt < dt_core)
            dt_core = shear_dt;
        const double dxRadial = c->Rsup[nr] - c->Rinf[nr];
        const double dxAzimuthal = c->Rmed[nr] * c->dphi;
        const double cell_size = fmin(dxRadial, dxAzimuthal);
        for (int naz = 0; naz < Nphi; ++naz) {
            if (d->fast_transport)
                c->cfl_vres[IDX(c, nr, naz)] = c->vazi[IDX(c, nr, naz)] - v_mean[nr];
            else
                c->cfl_vres[IDX(c, nr, naz)] = c->vazi[IDX(c, nr, naz)];
        }
        for (int naz = 0; naz < Nphi; ++naz) {
            const double invdt1 = c->soundspeed[IDX(c, nr, naz)] / cell_size;
            const double invdt2 = c->vrad[IDX(c, nr, naz)] / dxRadial;
            const double invdt3 = c->cfl_vres[IDX(c, nr, naz)] / dxAzimuthal;
            const int naz_next = naz == Nphi - 1 ? 0 : naz + 1;
            double invdt4;
            if (d->artificial_viscosity == FCPT_ARTVISC_SN) {
                double dvRadial = c->vrad[IDX(c, nr + 1, naz)] - c->vrad[IDX(c, nr, naz)];
                double dvAzimuthal = c->vazi[IDX(c, nr, naz_next)] - c->vazi[IDX(c, nr, naz)];
                dvRadial = dvRadial > 0.0 ? 0.0 : -dvRadial;
                dvAzimuthal = dvAzimuthal > 0.0 ? 0.0 : -dvAzimuthal;
                invdt4 = 4.0 * (C * C) * fmax(dvRadial / dxRadial, dvAzimuthal / dxAzimuthal) * lf;
            } else { /* TW formula also when the type is None (reference quirk, cfl.cpp:292) */
                const double eps_rr =
                    (c->vrad[IDX(c, nr + 1, naz)] - c->vrad[IDX(c, nr, naz)]) * c->InvDiffRsup[nr];
                const double eps_pp =
                    c->InvRmed[nr] *
                    ((c->vazi[IDX(c, nr, naz_next)] - c->vazi[IDX(c, nr, naz)]) * c->invdphi +
                     0.5 * (c->vrad[IDX(c, nr + 1, naz)] + c->vrad[IDX(c, nr, naz)]));
                const double mdiv_V = -fmin(eps_rr + eps_pp, 0.0);
                invdt4 = 4.0 * (C * C) * mdiv_V * lf;
            }
            const double invdt5 = 4.0 * c->viscosity[IDX(c, nr, naz)] / (cell_size * cell_size) * lf;
            double invdt6;
            if (d->eos == FCPT_EOS_IDEAL) {
                const double inv_limit = 1.0 / d->heating_cooling_cfl_limit;
                const double Qp = c->qplus[IDX(c, nr, naz)], Qm = c->qminus[IDX(c, nr, naz)];
                const double E = c->energy[IDX(c, nr, naz)];
                invdt6 = inv_limit * fabs((Qp - Qm) / E) * lf;
            } else {
                invdt6 = 0.0;
            }
            double dt_cell = d->cfl / sqrt(invdt1 * invdt1 + invdt2 * invdt2 + invdt3 * invdt3 +
                                           invdt4 * invdt4 + invdt5 * invdt5 + invdt6 * invdt6);
            if (d->stabilize_viscosity == 2) { /* cfl.cpp:331-351 */
                const double cc = fmin(c->cfac_phi[IDX(c, nr, naz)], c->cfac_r[IDX(c, nr, naz)]);
                if (cc != 0.0)
                    dt_cell = fmin(dt_cell, -d->cfl / cc);
            }
            if (dt_cell < dt_core)
                dt_core = dt_cell;
        }
    }
    *dt_local = dt_core;
    return FCPT_OK;
}

/* simulation.cpp:100-118 CalculateTimeStep */
int orc_calculate_timestep(orc_ctx *c, double cfl_dt, double *dt)
{
    if (!c || !dt)
        return FCPT_EINVAL;
    const double a = c->d.cfl_max_var * c->clk.last_dt;
    const double rv = a < cfl_dt ? a : cfl_dt; /* std::min(a, b) returns a unless b < a */
    c->clk.last_dt = rv;
    *dt = rv;
    return FCPT_OK;
}
/* simulation.cpp:528-540 */
int orc_snap_to_monitor(const orc_ctx *c, double cfl_dt, double *step_dt)
{
    if (!c || !step_dt)
        return FCPT_EINVAL;
    const double time_next_monitor = (c->clk.n_monitor + 1) * c->d.monitor_timestep;
    const double time_left_till_write = time_next_monitor - c->clk.time;
    const int overshoot = cfl_dt > time_left_till_write;
    const double dt_stretch_factor = 0.05;
    const int almost_there = time_left_till_write < cfl_dt * (1 + dt_stretch_factor);
    *step_dt = (overshoot || almost_there) ? time_left_till_write : cfl_dt;
    return FCPT_OK;
}

/* SourceEuler.cpp:251-285 init_euler + init.cpp:337-341 */
int orc_init_physics(orc_ctx *c)
{
    if (!c)
        return FCPT_EINVAL;
    if (c->d.eos == FCPT_EOS_ISOTHERMAL) {
        compute_sound_speed(c);
        compute_pressure(c);
        compute_temperature(c);
        compute_scale_height(c);
    } else {
        compute_temperature(c);
        compute_sound_speed(c);
        compute_scale_height(c);
        compute_pressure(c);
    }
    update_viscosity(c);
    /* compute_heating_cooling_for_CFL runs in init_euler BEFORE the velocities
     * are initialised (init.cpp:331-332): the stress tensor it sees is that of a
     * fluid at rest, so Q+ = 0.  Reproduced by evaluating it on zero velocities. */
    if (c->d.eos == FCPT_EOS_IDEAL) {
        const size_t ns = (size_t)c->nr * c->nphi, nv = (size_t)(c->nr + 1) * c->nphi;
        double *svr = c->vrad, *sva = c->vazi;
        c->vrad = dalloc(nv);
        c->vazi = dalloc(ns);
        compute_heating_cooling_for_CFL(c);
        free(c->vrad);
        free(c->vazi);
        c->vrad = svr;
        c->vazi = sva;
    }
    copy_initial_values(c);
    apply_boundary_condition(c, 0.0, 0);
    copy_initial_values(c);
    return FCPT_OK;
}

/* the gas "kick" shared by both integrators (simulation.cpp:190-203 / :324-337 / :379-392) */
static void gas_kick(orc_ctx *c, double current_time, double dt)
{
    /* update_with_sourceterms, SourceEuler.cpp:435-452 */
    momentum_update_radial(c, dt);
    momentum_update_azimuthal(c, dt);
    compression_heating(c, dt);
    update_with_artificial_viscosity(c, dt);
    recalculate_viscosity(c);
    compute_viscous_stress_tensor(c);
    update_velocities_with_viscosity(c, dt);
    if (c->d.eos == FCPT_EOS_IDEAL)
        substep3(c, current_time, dt);
}

/* simulation.cpp:167-217 (step_Euler: potential .. Transport) and :316-393 (step_LeapFrog:
 * kick 1/2, drift 1/1, kick 2/2), gas part */
int orc_step(orc_ctx *c, double dt)
{
    if (!c)
        return FCPT_EINVAL;
    if (c->d.integrator == FCPT_INTEGRATOR_LEAPFROG) {
        const double frog_dt = dt / 2;
        calculate_body_force(c);
        gas_kick(c, c->clk.time, frog_dt); /* start_time (simulation.cpp:333) */
        apply_boundary_condition(c, 0.0, 0);
        Transport(c, dt);
        if (c->has_mid) { /* bodies at x_{i+1/2} */
            double sx[FCPT_MAX_BODIES], sy[FCPT_MAX_BODIES], sm[FCPT_MAX_BODIES], sr[FCPT_MAX_BODIES];
            memcpy(sx, c->bx, sizeof sx); memcpy(sy, c->by, sizeof sy);
            memcpy(sm, c->bm, sizeof sm); memcpy(sr, c->brsm, sizeof sr);
            memcpy(c->bx, c->mx, sizeof sx); memcpy(c->by, c->my, sizeof sy);
            memcpy(c->bm, c->mm, sizeof sm); memcpy(c->brsm, c->mrsm, sizeof sr);
            calculate_body_force(c);
            memcpy(c->bx, sx, sizeof sx); memcpy(c->by, sy, sizeof sy);
            memcpy(c->bm, sm, sizeof sm); memcpy(c->brsm, sr, sizeof sr);
        } else {
            calculate_body_force(c);
        }
        compute_pressure(c);
        gas_kick(c, c->clk.time + frog_dt, frog_dt); /* midstep_time (simulation.cpp:388) */
    } else {
        calculate_body_force(c);
        gas_kick(c, c->clk.time, dt);
        apply_boundary_condition(c, 0.0, 0);
        Transport(c, dt);
    }
    c->clk.time += dt;
    c->clk.n_hydro_iter += 1;
    return FCPT_OK;
}

int orc_set_bodies_midstep(orc_ctx *c, int32_t n, const double *x, const double *y, const double *m,
                           const double *rsm)
{
    if (!c || n != c->nbodies)
        return FCPT_EINVAL;
    for (int k = 0; k < n; ++k) {
        c->mx[k] = x[k];
        c->my[k] = y[k];
        c->mm[k] = m[k];
        c->mrsm[k] = rsm ? rsm[k] : 0.0;
    }
    c->has_mid = 1;
    return FCPT_OK;
}

/* commbound.cpp:47-59,98-125,163-180 */
int orc_exchange_count(const orc_ctx *c, uint64_t *count)
{
    if (!c || !count)
        return FCPT_EINVAL;
    *count = (uint64_t)(c->d.eos == FCPT_EOS_IDEAL ? 4 : 3) * c->nphi * FCPT_OVERLAP;
    return FCPT_OK;
}
int orc_exchange_pack(orc_ctx *c, double *send_inner, double *send_outer)
{
    if (!c)
        return FCPT_EINVAL;
    const size_t l = (size_t)FCPT_OVERLAP * c->nphi;
    const size_t o = (size_t)(c->nr - 2 * FCPT_OVERLAP) * c->nphi;
    const int adi = c->d.eos == FCPT_EOS_IDEAL;
    if (send_inner) {
        memcpy(send_inner, c->sigma + l, l * sizeof(double));
        memcpy(send_inner + l, c->vrad + l, l * sizeof(double));
        memcpy(send_inner + 2 * l, c->vazi + l, l * sizeof(double));
        if (adi)
            memcpy(send_inner + 3 * l, c->energy + l, l * sizeof(double));
    }
    if (send_outer) {
        memcpy(send_outer, c->sigma + o, l * sizeof(double));
        memcpy(send_outer + l, c->vrad + o, l * sizeof(double));
        memcpy(send_outer + 2 * l, c->vazi + o, l * sizeof(double));
        if (adi)
            memcpy(send_outer + 3 * l, c->energy + o, l * sizeof(double));
    }
    return FCPT_OK;
}
int orc_exchange_unpack(orc_ctx *c, const double *recv_inner, const double *recv_outer)
{
    if (!c)
        return FCPT_EINVAL;
    const size_t l = (size_t)FCPT_OVERLAP * c->nphi;
    const size_t oo = (size_t)(c->nr - FCPT_OVERLAP) * c->nphi;
    const int adi = c->d.eos == FCPT_EOS_IDEAL;
    if (recv_inner) {
        memcpy(c->sigma, recv_inner, l * sizeof(double));
        memcpy(c->vrad, recv_inner + l, l * sizeof(double));
        memcpy(c->vazi, recv_inner + 2 * l, l * sizeof(double));
        if (adi)
            memcpy(c->energy, recv_inner + 3 * l, l * sizeof(double));
    }
    if (recv_outer) {
        memcpy(c->sigma + oo, recv_outer, l * sizeof(double));
        memcpy(c->vrad + oo, recv_outer + l, l * sizeof(double));
        memcpy(c->vazi + oo, recv_outer + 2 * l, l * sizeof(double));
        if (adi)
            memcpy(c->energy + oo, recv_outer + 3 * l, l * sizeof(double));
    }
    return FCPT_OK;
}

/* simulation.cpp:244-265: BC(final=true) + recalculate_derived_disk_quantities
 * (SourceEuler.cpp:225-249, AspectRatioMode 0) */
int orc_post(orc_ctx *c, double dt)
{
    if (!c)
        return FCPT_EINVAL;
    apply_boundary_condition(c, dt, 1);
    if (c->d.eos == FCPT_EOS_ISOTHERMAL) {
        compute_pressure(c);
    } else {
        compute_temperature(c);
        compute_sound_speed(c);
        compute_scale_height(c);
        compute_pressure(c);
    }
    update_viscosity(c);
    return FCPT_OK;
}

/* SourceEuler.cpp:225-249 recalculate_derived_disk_quantities */
int orc_recalculate_derived(orc_ctx *c)
{
    if (!c)
        return FCPT_EINVAL;
    if (c->d.eos == FCPT_EOS_ISOTHERMAL) {
        compute_pressure(c);
    } else {
        compute_temperature(c);
        compute_sound_speed(c);
        compute_scale_height(c);
        compute_pressure(c);
    }
    update_viscosity(c);
    return FCPT_OK;
}

int orc_apply_boundary(orc_ctx *c, double dt, int32_t final)
{
    if (!c)
        return FCPT_EINVAL;
    apply_boundary_condition(c, dt, final);
    return FCPT_OK;
}

/* simulation.cpp:515-553, single slab */
int orc_run_steps(orc_ctx *c, int64_t nsteps, int32_t snap, int64_t *done)
{
    if (!c)
        return FCPT_EINVAL;
    const double t_final = (double)c->d.nsnapshots * c->d.nmonitor * c->d.monitor_timestep;
    int64_t n = 0;
    for (; n < nsteps; ++n) {
        if (snap && t_final > 0 && !(c->clk.time < t_final))
            break;
        double cfl_dt, dt, step_dt;
        orc_cfl(c, &cfl_dt);
        orc_calculate_timestep(c, cfl_dt, &dt);
        step_dt = dt;
        double time_next_monitor = 0;
        if (snap) {
            orc_snap_to_monitor(c, dt, &step_dt);
            time_next_monitor = (c->clk.n_monitor + 1) * c->d.monitor_timestep;
        }
        orc_step(c, step_dt);
        orc_post(c, step_dt);
        if (snap) {
            const int towrite = fabs(time_next_monitor - c->clk.time) < 1e-6 * dt;
            if (towrite) {
                c->clk.n_monitor++;
                c->clk.n_snapshot = c->clk.n_monitor / (uint32_t)(c->d.nmonitor > 0 ? c->d.nmonitor : 1);
            }
        }
    }
    if (done)
        *done = n;
    return FCPT_OK;
}
