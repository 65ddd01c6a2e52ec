// Host-only part of the C ABI: defaults, radial split, grid construction and
// initial conditions.  No HIP here; these run on any machine.
//
// Reference counterparts are cited per function (paths relative to the
// reference's src/).
#include "fcpt_internal.h"

#include <cmath>
#include <cstdarg>
#include <cstdio>
#include <cstring>
#include <vector>

namespace fcpt {

static thread_local char g_err[512] = "";

void set_error(const char *fmt, ...)
{
    va_list ap;
    va_start(ap, fmt);
    vsnprintf(g_err, sizeof(g_err), fmt, ap);
    va_end(ap);
}

// ---- Theo.cpp:128-180 --------------------------------------------------------
static double support_azi_pressure(const fcpt_desc &d, double R)
{
    const double h = d.aspect_ratio * std::pow(R, d.flaring_index);
    return (2.0 * d.flaring_index - 1.0 - d.sigma_slope) * (h * h);
}
static double support_azi_smoothing_derivative(const fcpt_desc &d, double R)
{
    const double F = d.flaring_index;
    const double h = d.aspect_ratio * std::pow(R, F);
    const double he = h * d.thickness_smoothing;
    const double s = std::sqrt(1 + he * he);
    return (1.0 + (F + 1.0) * (he * he)) / (s * s * s);
}
// initial_locally_isothermal_smoothed_v_az
static double smoothed_v_az(const fcpt_desc &d, double R, double M)
{
    const double support = support_azi_smoothing_derivative(d, R) + support_azi_pressure(d, R);
    return std::sqrt(d.G * M / R * support);
}
// Theo.cpp:86-99 initial_energy
static double initial_energy(const fcpt_desc &d, double R, double M)
{
    return 1.0 / (d.adiabatic_index - 1.0) * d.sigma0 * (d.aspect_ratio * d.aspect_ratio) *
           std::pow(R, -d.sigma_slope - 1.0 + 2.0 * d.flaring_index) * d.G * M;
}
// Theo.cpp:215-243 initial_viscous_radial_speed
static double initial_viscous_radial_speed(const fcpt_desc &d, double R, double M)
{
    if (d.viscous_alpha > 0) {
        const double sqrt_gamma = d.eos == FCPT_EOS_IDEAL ? std::sqrt(d.adiabatic_index) : 1.0;
        const double v_k = std::sqrt(d.G * M / R);
        const double h = d.aspect_ratio * std::pow(R, d.flaring_index);
        const double nu = d.viscous_alpha * (sqrt_gamma * h * v_k) * (h * R);
        return -3.0 * nu / R * (-d.sigma_slope + 2.0 * d.flaring_index + 1.0);
    }
    return -3.0 * d.constant_viscosity / R * (-d.sigma_slope + .5);
}

// ---- viscosity/viscous_radial_speed.cpp:39-206 (no profile cut-offs) ---------
namespace viscous_speed {
static double sigma_of(const fcpt_desc &d, double R)
{
    const double rho = d.sigma0 * std::pow(R, -d.sigma_slope);
    const double floor = d.sigma_floor * d.sigma0;
    return rho > floor ? rho : floor;
}
static double nu_of(const fcpt_desc &d, double R, double M, double Sigma)
{
    const double v_k = std::sqrt(d.G * M / R);
    const double h = d.aspect_ratio * std::pow(R, d.flaring_index);
    double cs_adb, H;
    if (d.eos == FCPT_EOS_IDEAL) {
        const double g = d.adiabatic_index;
        double e = 1.0 * 1.0 / (g - 1.0) * Sigma * ((h * v_k) * (h * v_k));
        const double e_lo = d.minimum_temperature * Sigma / d.mu * d.Rgas / (g - 1.0);
        const double e_hi = d.maximum_temperature * Sigma / d.mu * d.Rgas / (g - 1.0);
        e = e > e_lo ? e : e_lo;
        e = e < e_hi ? e : e_hi;
        cs_adb = std::sqrt(g * (g - 1.0) * e / Sigma);
        H = std::sqrt((g - 1.0) * e / Sigma) / (v_k / R);
    } else {
        cs_adb = h * v_k;
        H = h * R;
    }
    return d.viscous_alpha * cs_adb * H;
}
template <class F> static double derive(double r, F f)
{
    const double h = 8.0e-4 * r;
    const double f1 = -1.0 * f(r + 2.0 * h);
    const double f2 = 8.0 * f(r + h);
    const double f3 = -8.0 * f(r - h);
    const double f4 = 1.0 * f(r - 2.0 * h);
    return (f1 + f2 + f3 + f4) / (12.0 * h);
}
// get_vr_with_numerical_viscous_speed
static double vr(const fcpt_desc &d, double r, double M)
{
    auto w = [&](double x) { return smoothed_v_az(d, x, M) / x; };
    auto r2w = [&](double x) { return (x * x) * w(x); };
    auto nuSr3dw = [&](double x) {
        const double dw = derive(x, w);
        const double S = sigma_of(d, x);
        return nu_of(d, x, M, S) * S * (x * x * x) * dw;
    };
    const double num = 1.0 / r * derive(r, nuSr3dw);
    const double den = sigma_of(d, r) * derive(r, r2w);
    return num / den;
}
} // namespace viscous_speed

// src/init.cpp:92-145
int build_radii(const fcpt_desc &d, double *Radii)
{
    const int N = d.nr_global;
    const double RMIN = d.rmin, RMAX = d.rmax;
    switch (d.radial_spacing) {
    case FCPT_SPACING_LOGARITHMIC: {
        const double f = std::pow(RMAX / RMIN, 1.0 / ((double)N - 2.0));
        for (int n = 0; n <= N + FCPT_GEOM_PAD; ++n)
            Radii[n] = RMIN * std::pow(f, (double)n - 1.0);
        return FCPT_OK;
    }
    case FCPT_SPACING_ARITHMETIC: {
        const double interval = (RMAX - RMIN) / (double)(N - 2.0);
        for (int n = 0; n <= N + FCPT_GEOM_PAD; ++n)
            Radii[n] = RMIN + interval * (double)(n - 1.0);
        return FCPT_OK;
    }
    case FCPT_SPACING_EXPONENTIAL: {
        const double cgf = std::pow(RMAX / RMIN, 1.0 / ((double)N - 2.0));
        const double first = RMIN * (cgf - 1.0) * d.exponential_cell_size_factor;
        const double f = (RMAX - RMIN) / first;
        const double Nr = (double)N - 2.0;
        double g = 1.02;
        for (int i = 0; i < 500000; ++i)
            g = g - ((std::pow(g, Nr) - g * f + f - 1)) / (Nr * std::pow(g, Nr - 1.0) - f);
        for (int n = 0; n <= N + FCPT_GEOM_PAD; ++n)
            Radii[n] = RMIN + first * (std::pow(g, (double)n - 1.0) - 1.0) / (g - 1.0);
        return FCPT_OK;
    }
    default:
        set_error("unknown radial_spacing %d", d.radial_spacing);
        return FCPT_EINVAL;
    }
}

// src/split.cpp:34-88
int split_domain(const fcpt_desc &d, fcpt_split &o)
{
    if (d.nranks < 1 || d.rank < 0 || d.rank >= d.nranks || d.nr_global < 3 || d.nphi < 1) {
        set_error("bad grid/decomposition: nr=%d nphi=%d rank=%d/%d", d.nr_global, d.nphi, d.rank,
                  d.nranks);
        return FCPT_EINVAL;
    }
    const int NR = d.nr_global, P = d.nranks, rank = d.rank;
    const int size_low = NR / P, size_high = size_low + 1, remainder = NR % P;
    if (P > 1 && size_low < 2 * FCPT_OVERLAP) {
        set_error("The number of processes is too large or the mesh is radially too narrow.");
        return FCPT_ESPLIT;
    }
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
    const bool first = rank == 0, last = rank == P - 1;
    o.nr = nr;
    o.imin = imin;
    o.imax = imax;
    o.zero_no_ghost = first ? 1 : 0;
    o.one_no_ghost_vr = first ? 2 : 1;
    o.max_no_ghost = nr - (last ? 1 : 0);
    o.maxmo_no_ghost_vr = nr + 1 - (last ? 2 : 1);
    o.zero_or_active = first ? 0 : FCPT_OVERLAP;
    o.radial_first_active = first ? FCPT_GHOSTCELLS_B : FCPT_OVERLAP;
    o.max_or_active = nr - (last ? 0 : FCPT_OVERLAP);
    o.radial_active_size = nr - (last ? FCPT_GHOSTCELLS_B : FCPT_OVERLAP);
    o.is_first = first;
    o.is_last = last;
    return FCPT_OK;
}

// src/init.cpp:169-225 (local part) and src/find_cell_id.cpp:47-106
void build_geometry(const fcpt_desc &d, const fcpt_split &s, const double *Radii, HostGeometry &g)
{
    const int n1 = s.nr + FCPT_GEOM_PAD;
    const size_t n = (size_t)n1 + 1;
    for (auto *v : {&g.Rmed, &g.Rinf, &g.Rsup, &g.Surf, &g.InvRmed, &g.InvRinf, &g.InvSurf,
                    &g.InvDiffRmed, &g.InvDiffRsup, &g.InvDiffRsupRb})
        v->assign(n, 0.0);
    const double invdphi = (double)d.nphi / (2.0 * M_PI);
    (void)invdphi;
    for (int i = 0; i < n1; ++i) {
        const double ri = Radii[i + s.imin], rs = Radii[i + s.imin + 1];
        g.Rinf[i] = ri;
        g.Rsup[i] = rs;
        double rm = 2.0 / 3.0 * (rs * rs * rs - ri * ri * ri);
        rm = rm / (rs * rs - ri * ri);
        g.Rmed[i] = rm;
        g.Surf[i] = M_PI * (rs * rs - ri * ri) / (double)d.nphi;
        g.InvRmed[i] = 1.0 / rm;
        g.InvSurf[i] = 1.0 / g.Surf[i];
        g.InvDiffRsup[i] = 1.0 / (rs - ri);
        g.InvDiffRsupRb[i] = 1.0 / ((rs - ri) * rm);
        g.InvRinf[i] = 1.0 / ri;
    }
    for (int i = 1; i < s.nr + 1; ++i)
        g.InvDiffRmed[i] = 1.0 / (g.Rmed[i] - g.Rmed[i - 1]);
    g.dphi = 2.0 * M_PI / (double)d.nphi; // Interpret.cpp:230-231
    g.invdphi = (double)d.nphi / (2.0 * M_PI);

    // cell finder constants
    const double N = (double)d.nr_global;
    g.cf_growth = g.cf_inv_log_growth = g.cf_opt_const = 0.0;
    if (d.radial_spacing == FCPT_SPACING_LOGARITHMIC) {
        const double gf = std::pow(d.rmax / d.rmin, 1.0 / (N - 2.0));
        g.cf_growth = gf;
        g.cf_opt_const = 3.0 / 2.0 / d.rmin * (1 - std::pow(gf, 2.0)) / (1 - std::pow(gf, 3.0));
        g.cf_inv_log_growth = 1.0 / std::log(gf);
    } else if (d.radial_spacing == FCPT_SPACING_ARITHMETIC) {
        g.cf_growth = (N - 2.0) / (d.rmax - d.rmin);
    } else {
        const double cgf = std::pow(d.rmax / d.rmin, 1.0 / (N - 2.0));
        const double first = d.rmin * (cgf - 1.0) * d.exponential_cell_size_factor;
        const double f = (d.rmax - d.rmin) / first;
        const double Nr = N - 2.0;
        double gg = 1.02;
        for (int i = 0; i < 500000; ++i)
            gg = gg - ((std::pow(gg, Nr) - gg * f + f - 1)) / (Nr * std::pow(gg, Nr - 1.0) - f);
        g.cf_growth = gg;
        g.cf_inv_log_growth = 1.0 / std::log(gg);
        g.cf_opt_const = (gg - 1.0) / first;
    }
}

// src/find_cell_id.cpp:217-285 with the NDEBUG behaviour
int rmed_id(const fcpt_desc &d, const fcpt_split &s, const HostGeometry &g, double r)
{
    int id;
    if (d.radial_spacing == FCPT_SPACING_LOGARITHMIC) {
        id = (int)std::floor(std::log(r * g.cf_opt_const) * g.cf_inv_log_growth) - s.imin + 1;
    } else if (d.radial_spacing == FCPT_SPACING_ARITHMETIC) {
        id = (int)std::floor((r - d.rmin) * g.cf_growth) - s.imin + 1;
        if (id >= 0 && id < s.nr + FCPT_GEOM_PAD && g.Rmed[id] > r)
            id--;
    } else {
        const double tmp = (r - d.rmin) * g.cf_opt_const + 1.0;
        id = (int)std::floor(std::log(tmp) * g.cf_inv_log_growth) - s.imin + 1;
        if (id >= 0 && id < s.nr + FCPT_GEOM_PAD && g.Rmed[id] > r)
            id--;
    }
    return id;
}
int rinf_id(const fcpt_desc &d, const fcpt_split &s, const HostGeometry &g, double r)
{
    if (d.radial_spacing == FCPT_SPACING_LOGARITHMIC)
        return (int)std::floor(std::log(r / d.rmin) * g.cf_inv_log_growth) - s.imin + 1;
    if (d.radial_spacing == FCPT_SPACING_ARITHMETIC)
        return (int)std::floor((r - d.rmin) * g.cf_growth) - s.imin + 1;
    const double tmp = (r - d.rmin) * g.cf_opt_const + 1.0;
    return (int)std::floor(std::log(tmp) * g.cf_inv_log_growth) - s.imin + 1;
}

} // namespace fcpt

using namespace fcpt;

extern "C" {

const char *fcpt_last_error(void) { return g_err; }

// Defaults of src/parameters.cpp:520-900, src/Interpret.cpp:73-700,
// src/boundary_conditions/config.cpp and damping.cpp:185-271.
int fcpt_desc_default(fcpt_desc *d)
{
    if (!d)
        return FCPT_EINVAL;
    std::memset(d, 0, sizeof(*d));
    d->struct_size = sizeof(fcpt_desc);
    d->abi_version = FCPT_ABI_VERSION;
    d->nr_global = 64; // Nrad
    d->nphi = 64;      // Naz
    d->rank = 0;
    d->nranks = 1;
    d->radial_spacing = FCPT_SPACING_ARITHMETIC;
    d->rmin = 0.4;
    d->rmax = 2.5;
    d->exponential_cell_size_factor = 1.41;
    d->eos = FCPT_EOS_ISOTHERMAL;
    d->adiabatic_index = 7.0 / 5.0;
    d->mu = 1.0;
    d->aspect_ratio = 0.05;
    d->flaring_index = 0.0;
    d->sigma_slope = 0.0;
    d->sigma_floor = 1e-9;
    d->viscous_alpha = 0.0;
    d->constant_viscosity = 0.0;
    d->radial_viscosity_factor = 1.0;
    d->stabilize_viscosity = 0;
    d->artificial_viscosity = FCPT_ARTVISC_SN;
    d->artificial_viscosity_factor = 1.41;
    d->artificial_viscosity_dissipation = 1;
    d->heating_viscous = 1;
    d->heating_viscous_factor = 1.0;
    d->fast_transport = 1;
    d->flux_limiter = FCPT_LIMITER_VANLEER;
    d->integrator = FCPT_INTEGRATOR_EULER;
    d->cfl = 0.5;
    d->cfl_max_var = 1.1;
    d->first_dt = 1e-9;
    d->heating_cooling_cfl_limit = 10.0;
    d->monitor_timestep = 1.0;
    d->nmonitor = 10;
    d->nsnapshots = 1000;
    d->omega_frame = 0.0;
    d->thickness_smoothing = 0.6;
    d->body_force_from_potential = 1;
    d->hydro_center_mass = 1.0;
    for (int k = 0; k < 2; ++k) {
        // the reference has no default composite; "zerogradient"+"keplerian vaz" is the
        // behaviour of InnerBoundary/OuterBoundary: zerogradient (config.cpp:345-440)
        d->bc_sigma[k] = FCPT_BC_ZEROGRADIENT;
        d->bc_energy[k] = FCPT_BC_ZEROGRADIENT;
        d->bc_vrad[k] = FCPT_BC_ZEROGRADIENT;
        d->bc_vaz[k] = FCPT_BC_KEPLERIAN; // config.cpp:263,305
        d->keplerian_vaz_factor[k] = 1.0;
        d->keplerian_vrad_factor[k] = 0.1;
    }
    d->damping = 0;
    d->damping_inner_limit = 1.05;
    d->damping_outer_limit = 0.95;
    d->damping_time_factor = 1.0;
    d->damping_time_radius_outer = d->rmax;

    // Code units: L0 = 1 au, M0 = 1 solMass, T0 = sqrt(L0^3/(G M0)), Temp0 = G mu/kB M0/L0
    // (src/units.cpp:158-185) => G = 1, R = kB/mu = 1 (src/constants.cpp:236-262).
    const double G_cgs = 6.67430e-8, kB = 1.380649e-16, mu_cgs = 1.66053906660e-24;
    const double h_cgs = 6.62607015e-27, c_cgs = 2.99792458e10;
    const double L0 = 1.495978707e13, M0 = 1.988409870698051e33; // au, GM_sun(IAU)/G
    const double T0 = std::sqrt(L0 * L0 * L0 / (G_cgs * M0));
    const double Temp0 = G_cgs * mu_cgs / kB * M0 / L0;
    const double E0 = M0 * L0 * L0 / (T0 * T0);
    const double sigma_cgs = 2. * std::pow(M_PI, 5) * std::pow(kB, 4) /
                             (15. * std::pow(h_cgs, 3) * std::pow(c_cgs, 2));
    d->G = 1.0;
    d->Rgas = 1.0;
    d->c_light = c_cgs / (L0 / T0);
    d->sigma_sb = sigma_cgs / (E0 / (L0 * L0 * T0 * Temp0 * Temp0 * Temp0 * Temp0));
    d->minimum_temperature = 3.0 / Temp0;      // "3 K"
    d->maximum_temperature = 1.0e300 / Temp0;  // "1.0e300 K"
    d->sigma0 = 173.0 / (M0 / (L0 * L0));      // "173 g/cm2"
    d->ic = FCPT_IC_PROFILE;
    d->set_sigma0 = 0;
    d->disk_mass = 0.01;
    d->initialize_vradial_zero = 0;
    d->initialize_pure_keplerian = 0;
    // cooling (src/parameters.cpp:399-490,661-665)
    d->cooling_surface = 0;
    d->opacity = FCPT_OPACITY_LIN;
    d->cooling_radiative_factor = 1.0;
    d->kappa_const = 1.0;
    d->kappa_factor = 1.0;
    d->tau_factor = 0.5;
    d->tau_min = 0.01;
    d->density_factor = std::sqrt(2.0 * M_PI);
    d->cooling_beta = 0;
    d->cooling_beta_reference = FCPT_BETAREF_ZERO;
    d->cooling_beta_value = 1.0;
    d->cooling_beta_ramp_up = 0.0;
    d->temperature_cgs = Temp0;
    d->density_cgs = M0 / (L0 * L0 * L0);
    d->opacity_cgs = L0 * L0 / M0;
    d->profile_cutoff_inner = d->profile_cutoff_outer = 0; // parameters.cpp:762-776
    d->profile_cutoff_point_inner = 0.0;
    d->profile_cutoff_point_outer = 1.0e300;
    d->profile_cutoff_width_inner = d->profile_cutoff_width_outer = 1.0;
    return FCPT_OK;
}

int fcpt_split_domain(const fcpt_desc *d, fcpt_split *out)
{
    if (!d || !out) {
        set_error("null argument");
        return FCPT_EINVAL;
    }
    return split_domain(*d, *out);
}

int fcpt_radii(const fcpt_desc *d, double *radii)
{
    if (!d || !radii || d->nr_global < 3) {
        set_error("null argument or nr_global < 3");
        return FCPT_EINVAL;
    }
    return build_radii(*d, radii);
}

// init_physics up to the velocities (src/init.cpp:255-343): init_gas_density
// (:937-1004), init_spreading_ring_test (:358-413), init_shock_tube_test
// (:423-441), renormalize_sigma_and_report (:1150-1185), init_gas_energy
// (:1257-1300), init_gas_velocities (:1616-1631, :1725-1772).
int fcpt_initial_fields(fcpt_desc *d, const double *Radii, double *sigma, double *vrad, double *vazi,
                        double *energy)
{
    if (!d || !Radii || !sigma || !vrad || !vazi) {
        set_error("null argument");
        return FCPT_EINVAL;
    }
    fcpt_split s;
    if (int rc = split_domain(*d, s))
        return rc;
    const bool adi = d->eos == FCPT_EOS_IDEAL;
    if (adi && !energy) {
        set_error("energy buffer required for EquationOfState: ideal");
        return FCPT_EINVAL;
    }
    const int nr = s.nr, nphi = d->nphi;
    std::vector<double> Rmed(nr), Rinf(nr), Rsup(nr), Surf(nr);
    for (int i = 0; i < nr; ++i) {
        const double ri = Radii[i + s.imin], rs = Radii[i + s.imin + 1];
        Rinf[i] = ri;
        Rsup[i] = rs;
        Rmed[i] = 2.0 / 3.0 * (rs * rs * rs - ri * ri * ri);
        Rmed[i] = Rmed[i] / (rs * rs - ri * ri);
        Surf[i] = M_PI * (rs * rs - ri * ri) / (double)nphi;
    }
    const double M = d->hydro_center_mass;
    auto at = [nphi](int i, int j) { return (size_t)i * nphi + j; };

    if (d->ic == FCPT_IC_SHOCKTUBE) {
        const double r0 = Radii[0], r1 = Radii[1];
        double g0 = 2.0 / 3.0 * (r1 * r1 * r1 - r0 * r0 * r0);
        g0 = g0 / (r1 * r1 - r0 * r0); // GlobalRmed[0]
        for (int i = 0; i < nr; ++i)
            for (int j = 0; j < nphi; ++j) {
                const bool right = Rmed[i] - g0 > 0.5;
                sigma[at(i, j)] = right ? 0.125 : 1.0;
                if (energy)
                    energy[at(i, j)] = right ? 2.0 * 0.125 : 2.5;
            }
    } else {
        for (int i = 0; i < nr; ++i) {
            const double rho = d->sigma0 * std::pow(Rmed[i], -d->sigma_slope);
            const double floor = d->sigma_floor * d->sigma0;
            for (int j = 0; j < nphi; ++j)
                sigma[at(i, j)] = rho > floor ? rho : floor;
        }
        if (d->ic == FCPT_IC_SPREADING_RING) {
            const double R0 = 1.0, tau0 = 0.016;
            int R0_id = 0;
            for (int i = 0; i < nr; ++i)
                if (Rsup[i] > R0 && R0 > Rinf[i])
                    R0_id = i;
            auto ring = [&](double x) {
                const double I = std::cyl_bessel_i(0.25, 2.0 * x / tau0);
                return d->disk_mass / (M_PI * R0 * R0) * 1.0 / (tau0 * std::pow(x, 0.25)) * I *
                       std::exp(-(1.0 + x * x) / tau0);
            };
            const double Sigma0 = ring(Rmed[R0_id] / R0);
            for (int i = 0; i < nr; ++i) {
                const double floor = Sigma0 * d->sigma_floor;
                const double rho = ring(Rmed[i] / R0);
                for (int j = 0; j < nphi; ++j) {
                    sigma[at(i, j)] = rho > floor ? rho : floor;
                    if (energy)
                        energy[at(i, j)] = 0.0;
                }
            }
        }
        // profile cutoffs of Sigma, outer then inner (init.cpp:1063-1146; util.cpp:69-93)
        auto cut_outer = [&](double r) { return 1.0 / (1.0 + std::exp((r - d->profile_cutoff_point_outer) / d->profile_cutoff_width_outer)); };
        auto cut_inner = [&](double r) { return 1.0 / (1.0 + std::exp((d->profile_cutoff_point_inner - r) / d->profile_cutoff_width_inner)); };
        for (int pass = 0; pass < 2; ++pass) {
            if (!(pass == 0 ? d->profile_cutoff_outer : d->profile_cutoff_inner))
                continue;
            for (int i = 0; i < nr; ++i) {
                const double f = pass == 0 ? cut_outer(Rmed[i]) : cut_inner(Rmed[i]);
                const double floor = d->sigma_floor * d->sigma0;
                for (int j = 0; j < nphi; ++j) {
                    const double v = sigma[at(i, j)] * f;
                    sigma[at(i, j)] = v > floor ? v : floor;
                }
            }
        }
        if (adi) {
            for (int i = 0; i < nr; ++i)
                for (int j = 0; j < nphi; ++j) {
                    const double e = initial_energy(*d, Rmed[i], M);
                    const double e_floor = d->minimum_temperature * sigma[at(i, j)] / d->mu * d->Rgas /
                                           (d->adiabatic_index - 1.0);
                    energy[at(i, j)] = e > e_floor ? e : e_floor;
                }
            // the same cutoffs on the energy (init.cpp:1363-1450)
            for (int pass = 0; pass < 2; ++pass) {
                if (!(pass == 0 ? d->profile_cutoff_outer : d->profile_cutoff_inner))
                    continue;
                for (int i = 0; i < nr; ++i) {
                    const double f = pass == 0 ? cut_outer(Rmed[i]) : cut_inner(Rmed[i]);
                    for (int j = 0; j < nphi; ++j) {
                        const double v = energy[at(i, j)] * f;
                        const double e_floor = d->minimum_temperature * sigma[at(i, j)] / d->mu * d->Rgas /
                                               (d->adiabatic_index - 1.0);
                        energy[at(i, j)] = v > e_floor ? v : e_floor;
                    }
                }
            }
        }
        if (d->set_sigma0) {
            // gas_total_mass (quantities.cpp:50-73) over this slab's active rings; exact
            // for one slab (multi-slab callers normalise before splitting)
            double total = 0.0;
            for (int i = s.radial_first_active; i < s.radial_active_size; ++i)
                for (int j = 0; j < nphi; ++j)
                    if (Rmed[i] <= 2.0 * d->rmax)
                        total += Surf[i] * sigma[at(i, j)];
            const double f = d->disk_mass / total;
            d->sigma0 *= f; // parameters::sigma0 is rescaled in place (init.cpp:1155)
            for (int i = 0; i < nr; ++i)
                for (int j = 0; j < nphi; ++j) {
                    sigma[at(i, j)] *= f;
                    if (adi)
                        energy[at(i, j)] *= f;
                }
        }
    }

    for (int j = 0; j < nphi; ++j)
        vrad[at(nr, j)] = 0.0; // row Nr is never initialised by the reference (stays 0)
    for (int i = 0; i < nr; ++i) {
        const double r = Rmed[i], ri = Rinf[i];
        double va, vr;
        if (d->initialize_pure_keplerian) {
            vr = initial_viscous_radial_speed(*d, r, M);
            va = std::sqrt(d->G * M / r) - d->omega_frame * r;
        } else {
            va = smoothed_v_az(*d, r, M);
            va -= d->omega_frame * r;
            vr = 0.0;
            if (!d->initialize_vradial_zero)
                vr += viscous_speed::vr(*d, ri, M);
        }
        for (int j = 0; j < nphi; ++j) {
            vazi[at(i, j)] = va;
            vrad[at(i, j)] = vr;
        }
    }
    return FCPT_OK;
}

} // extern "C"
