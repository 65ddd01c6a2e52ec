// Context management: creation and destruction, options, host <-> device transfers, bodies, initial physics and the
// per-kernel profiler of the C ABI (include/fargocpt_hip.h).
#include "fcpt_ctx.h"

using namespace fcpt;

namespace fcpt {

void drop_graph(fcpt_ctx *c)
{
    if (c->graph_exec)
        (void)hipGraphExecDestroy(c->graph_exec);
    if (c->graph)
        (void)hipGraphDestroy(c->graph);
    c->graph_exec = nullptr;
    c->graph = nullptr;
    c->graph_cycle = 0;
}

// the caller's stream waits for the interior transport forked by fcpt_step_device_begin
void join_side(fcpt_ctx *c)
{
    if (c->join_pending) {
        (void)hipStreamWaitEvent(c->stream, c->e_join, 0);
        c->join_pending = false;
    }
}

int read_clock(fcpt_ctx *c, DevClock *out)
{
    join_side(c);
    HIPCHK(hipMemcpyAsync(c->h_clk, c->P.clk, sizeof(DevClock), hipMemcpyDeviceToHost, c->stream));
    HIPCHK(hipStreamSynchronize(c->stream));
    *out = *c->h_clk;
    if (out->shear_error) {
        set_error("a step exceeded the FARGO shear limit (|Nshift[i]-Nshift[i-1]| > 1) with the option "
                  "transport_fallback = 0, i.e. without the two-kernel transport queued behind the fused one; "
                  "the state is invalid");
        return FCPT_ESHEAR;
    }
    return FCPT_OK;
}

} // namespace fcpt

namespace {

int *option_slot(Options &o, const char *name)
{
#define X(n)                     \
    if (!std::strcmp(name, #n)) \
        return &o.n;
    FCPT_OPTION_NAMES
#undef X
    return nullptr;
}

// defaults of the switches: FCPT_<NAME> in the environment, read here and nowhere else
void options_from_environment(Options &o)
{
    o.transport_fused = o.transport_rows = o.source_rows = o.theta_rows = -1;
    o.transport_graded = 1;
    o.source_graded = -1;
    o.transport_big = o.transport_ladder = o.transport_rank_grade = -1;
    o.transport_fallback = o.transport_split = o.fused_source = o.march_source = o.march_source_adi = 1;
    o.theta_march = o.theta_fused = o.cfl_rings = o.cfl_split = o.source_ring_parts = o.fused_damping = 1;
    o.cfl_wide_blocks = -1;
    o.gate_in_boundary = 1;
    o.cfl_fold_in_source = -1;
    o.inline_potential = 1;
    o.cfl_thermal = 0; // measured: the kernel that stores the terms spills (28 B) and loses more than the CFL pass gains
    o.bc_fold = 1;
    o.bc_in_cfl = 1;
    o.comm_overlap = 0;
    o.comm_loopback = 0;
    o.graph_steps = -1;
    o.profile_stride = 1;
#define X(n)                                                           \
    {                                                                  \
        char env[64] = "FCPT_";                                        \
        size_t k = 5;                                                  \
        for (const char *q = #n; *q && k + 1 < sizeof(env); ++q)       \
            env[k++] = (char)((*q >= 'a' && *q <= 'z') ? *q - 32 : *q); \
        env[k] = 0;                                                    \
        if (const char *e = getenv(env))                               \
            if (e[0])                                                  \
                o.n = atoi(e);                                         \
    }
    FCPT_OPTION_NAMES
#undef X
}

int dev_upload(fcpt_ctx *c, const double **dst, const std::vector<double> &src)
{
    double *p = nullptr;
    if (int rc = dev_alloc(c, &p, src.size()))
        return rc;
    hipError_t e = hipMemcpy(p, src.data(), src.size() * sizeof(double), hipMemcpyHostToDevice);
    if (e != hipSuccess) {
        set_error("hipMemcpy H2D failed: %s", hipGetErrorString(e));
        return FCPT_EHIP;
    }
    *dst = p;
    return FCPT_OK;
}

// damping.cpp:311-427: which rows a damping call touches and its time scale
DampRange damp_range(const fcpt_ctx *c, int is_vector, int type, int outer)
{
    DampRange r;
    r.lo = 0;
    r.hi = -1;
    r.type = type;
    r.rlim = r.redge = r.tau = 0;
    if (!c->d.damping || type == FCPT_DAMP_NONE)
        return r;
    const fcpt_desc &d = c->d;
    const std::vector<double> &radius = is_vector ? c->geo.Rinf : c->geo.Rmed;
    const int nr = c->s.nr;
    const int size_radial = is_vector ? nr + 1 : nr;
    auto clamp = [&](int id) {
        const int mx = nr - (is_vector ? 0 : 1);
        return id < 0 ? 0 : (id > mx ? mx : id);
    };
    auto omega_k = [&](double rr) { return std::sqrt(d.G * d.hydro_center_mass / (rr * rr * rr)); };
    if (!outer) {
        if (!((d.damping_inner_limit > 1.0) && (radius[0] < d.rmin * d.damping_inner_limit)))
            return r;
        const double rl = d.rmin * d.damping_inner_limit;
        const int limit = clamp(is_vector ? rinf_id(d, c->s, c->geo, rl) : rmed_id(d, c->s, c->geo, rl));
        r.lo = 0;
        r.hi = limit;
        r.rlim = rl;
        r.redge = d.rmin;
        r.tau = d.damping_time_factor * 2.0 * M_PI / omega_k(d.rmin);
    } else {
        if (!((d.damping_outer_limit < 1.0) && (radius[size_radial - 1] > d.rmax * d.damping_outer_limit)))
            return r;
        const double rl = d.rmax * d.damping_outer_limit;
        const int limit =
            clamp((is_vector ? rinf_id(d, c->s, c->geo, rl) : rmed_id(d, c->s, c->geo, rl)) + 1);
        r.lo = limit;
        r.hi = size_radial - 1;
        r.rlim = rl;
        r.redge = d.rmax;
        r.tau = d.damping_time_factor * 2.0 * M_PI / omega_k(d.damping_time_radius_outer);
    }
    return r;
}

// the context's derived switches after its options changed (fcpt_create, fcpt_set_option)
void apply_options(fcpt_ctx *c, bool at_create = true)
{
    const Options &o = c->P.opt;
    c->fused_source = o.fused_source != 0;
    c->march_source = o.march_source != 0;
    {
        // StabilizeViscosity is implemented by the marching kernels and by the per-loop kernels, not by the three
        // intermediate fused ones: where the march does not run (narrow rings, switched off) fall back to the loops
        const bool march_runs = c->fused_source && c->march_source && c->P.nphi >= 128 &&
                                (!c->P.adiabatic || o.march_source_adi != 0);
        if ((c->P.stabilize || c->P.accel_force) && !march_runs) // (likewise BodyForceFromPotential: no)
            c->fused_source = c->march_source = false;
    }
    const bool adi_march =
        c->P.adiabatic && c->fused_source && c->march_source && c->P.nphi >= 128 && o.march_source_adi != 0;
    c->P.lazy_derived = adi_march ? 1 : 0;
    c->P.inline_potential = (adi_march && !c->P.leapfrog && o.inline_potential != 0 && !c->P.accel_force) ? 1 : 0;
    // the transport leaves the cell-local CFL terms only where the CFL kernel that reads them will run (lazy derived
    // quantities, Euler: the leapfrog's second kick changes e after the transport)
    c->P.cfl_thermal = (adi_march && !c->P.leapfrog && o.cfl_thermal != 0) ? c->thermal_grid : nullptr;
    c->thermal_valid = false;
    c->P.damp_in_step = (c->damp_foldable && o.fused_damping != 0) ? 1 : 0;
    c->cfl_interior = false;
    c->potential_valid = false;
    c->pressure_valid = false;
    // the wavefront table of the marching source kernels (the caller has synchronised the stream, or nothing has run yet)
    c->P.sm_sched = nullptr;
    c->P.sm_sched_n = 0;
    c->sm_sched_host.clear();
    if (c->sm_sched_dev && c->fused_source && c->march_source) {
        const std::vector<int> sched = source_schedule(c->P);
        if (!sched.empty() && sched.size() <= c->sm_sched_cap &&
            hipMemcpy(c->sm_sched_dev, sched.data(), sched.size() * sizeof(int), hipMemcpyHostToDevice) == hipSuccess) {
            c->P.sm_sched = c->sm_sched_dev;
            c->P.sm_sched_n = (int)(sched.size() / 4);
            c->sm_sched_host = sched;
        }
    }
    // the chunk table of the fused transport
    c->P.tf_sched = nullptr;
    c->P.tf_sched_n = 0;
    if (c->tf_sched_dev) {
        std::vector<int> slow(c->P.nr, 0);
        if (c->P.damp_in_step)
            slow = c->ring_ref_damped;
        const std::vector<int> sched = transport_schedule(c->P, slow, &c->tf_lengths);
        c->tf_sched_host.clear();
        if (!sched.empty() && sched.size() <= c->tf_sched_cap &&
            hipMemcpy(c->tf_sched_dev, sched.data(), sched.size() * sizeof(int), hipMemcpyHostToDevice) == hipSuccess) {
            c->P.tf_sched = c->tf_sched_dev;
            c->P.tf_sched_n = (int)(sched.size() / 4);
            c->tf_sched_host = sched;
        }
    }
    if (!at_create && c->P.adiabatic && !c->P.lazy_derived) {
        launch_derived(c->P, c->stream); // the kernels that read c_s, H, nu, T from the grids expect them current
        c->pressure_valid = true;
    }
}

int copy_initial_values(fcpt_ctx *c)
{
    const Dev &P = c->P;
    const size_t ns = (size_t)P.nr * P.nphi * sizeof(double), nv = (size_t)(P.nr + 1) * P.nphi * sizeof(double);
    HIPCHK(hipMemcpyAsync(P.vrad0, P.vrad, nv, hipMemcpyDeviceToDevice, c->stream));
    HIPCHK(hipMemcpyAsync(P.vazi0, P.vazi, ns, hipMemcpyDeviceToDevice, c->stream));
    HIPCHK(hipMemcpyAsync(P.sigma0, P.sigma, ns, hipMemcpyDeviceToDevice, c->stream));
    HIPCHK(hipMemcpyAsync(P.energy0, P.energy, ns, hipMemcpyDeviceToDevice, c->stream));
    return FCPT_OK;
}

} // namespace

extern "C" {

int fcpt_device_count(int32_t *n)
{
    if (!n)
        return FCPT_EINVAL;
    int ndev = 0;
    if (hipGetDeviceCount(&ndev) != hipSuccess) {
        (void)hipGetLastError();
        ndev = 0;
    }
    *n = ndev;
    return FCPT_OK;
}

int fcpt_set_device(int32_t device)
{
    int ndev = 0;
    if (hipGetDeviceCount(&ndev) != hipSuccess || ndev <= 0) {
        set_error("no HIP device available");
        return FCPT_ENODEV;
    }
    if (device < 0 || device >= ndev) {
        set_error("fcpt_set_device(%d): %d device(s) visible", device, ndev);
        return FCPT_EINVAL;
    }
    HIPCHK(hipSetDevice(device));
    return FCPT_OK;
}

int fcpt_create(const fcpt_desc *d, const double *radii, fcpt_ctx **out)
{
    if (!d || !radii || !out) {
        set_error("null argument");
        return FCPT_EINVAL;
    }
    if (d->struct_size != sizeof(fcpt_desc) || d->abi_version != FCPT_ABI_VERSION) {
        set_error("descriptor ABI mismatch: size %u (expected %zu), version %u (expected %d)", d->struct_size,
                  sizeof(fcpt_desc), d->abi_version, FCPT_ABI_VERSION);
        return FCPT_EINVAL;
    }
    if (d->stabilize_viscosity < 0 || d->stabilize_viscosity > 2) {
        set_error("StabilizeViscosity must be 0, 1 or 2");
        return FCPT_EINVAL;
    }
    if (d->cooling_surface && d->eos == FCPT_EOS_IDEAL && (d->opacity < FCPT_OPACITY_LIN || d->opacity > FCPT_OPACITY_SIMPLE)) {
        set_error("Opacity: Lin, Bell, Constant and Simple are the laws of the path (src/opacity.cpp:10-43)");
        return FCPT_EINVAL;
    }
    // the exponential spacing's Newton iteration (init.cpp:113-131) collapses to NaN for coarse grids
    for (int i = 0; d->nr_global >= 1 && i <= d->nr_global + FCPT_GEOM_PAD; ++i)
        if (!std::isfinite(radii[i]) || radii[i] <= 0.0 || (i > 0 && !(radii[i] > radii[i - 1]))) {
            set_error("radii[%d] = %g: the interfaces must be finite, positive and strictly increasing", i, radii[i]);
            return FCPT_EINVAL;
        }
    if ((long long)(d->nr_global + 1) * (long long)d->nphi >= (1ll << 31)) {
        set_error("grid too large for 32-bit cell indices");
        return FCPT_EINVAL;
    }
    int ndev = 0;
    if (hipGetDeviceCount(&ndev) != hipSuccess || ndev <= 0) {
        set_error("no HIP device available: the HIP path cannot run (there is no CPU fallback)");
        return FCPT_ENODEV;
    }
    fcpt_ctx *c = new (std::nothrow) fcpt_ctx();
    if (!c)
        return FCPT_ENOMEM;
    c->d = *d;
    if (d->eos != FCPT_EOS_IDEAL) // no energy equation: SubStep3 and with it every cooling term is not called
        c->d.cooling_surface = c->d.cooling_beta = 0; // (simulation.cpp:205-207 `if (parameters::Adiabatic)`)
    d = &c->d;
    (void)hipGetDevice(&c->device);
    if (int rc = split_domain(*d, c->s)) {
        delete c;
        return rc;
    }
    const int nr = c->s.nr, nphi = d->nphi;
    c->radii.assign(radii, radii + d->nr_global + FCPT_GEOM_PAD + 1);
    build_geometry(*d, c->s, radii, c->geo);

    Dev &P = c->P;
    std::memset(&P, 0, sizeof(P));
    P.nr = nr;
    P.nphi = nphi;
    P.dphi = c->geo.dphi;
    P.invdphi = c->geo.invdphi;
    int rc = FCPT_OK;
#define UP(field) \
    if (!rc)      \
        rc = dev_upload(c, &P.field.p, c->geo.field);
    UP(Rmed) UP(Rinf) UP(Rsup) UP(Surf) UP(InvRmed) UP(InvRinf) UP(InvSurf) UP(InvDiffRmed)
    UP(InvDiffRsup) UP(InvDiffRsupRb)
#undef UP
    std::vector<double> cp(nphi), sp(nphi), cs_ring(nr);
    for (int j = 0; j < nphi; ++j) { // SideEuler.cpp:60-63
        cp[j] = std::cos(c->geo.dphi * (double)j);
        sp[j] = std::sin(c->geo.dphi * (double)j);
    }
    for (int i = 0; i < nr; ++i) { // SourceEuler.cpp:1080-1088
        const double vK = std::sqrt(d->G * d->hydro_center_mass / c->geo.Rmed[i]);
        const double h = d->aspect_ratio * std::pow(c->geo.Rmed[i], d->flaring_index);
        cs_ring[i] = h * vK;
    }
    if (!rc)
        rc = dev_upload(c, &P.cosphi, cp);
    if (!rc)
        rc = dev_upload(c, &P.sinphi, sp);
    const double *csr = nullptr;
    if (!rc)
        rc = dev_upload(c, &csr, cs_ring);
    c->d_cs_ring = const_cast<double *>(csr);
    P.cs_ring.p = csr;
    {
        const HostGeometry &g = c->geo;
        std::vector<double> a(nr + 1, 0.0), b(nr + 1, 0.0), c2(nr + 1, 0.0), d2(nr + 1, 0.0), e2(nr + 1, 0.0);
        for (int i = 0; i <= nr; ++i) {
            a[i] = 2.0 / (g.dphi * (g.Rsup[i] + g.Rinf[i]));
            b[i] = 1.0 / (g.Rsup[i] + g.Rinf[i]);
            d2[i] = 1.0 / (g.Rinf[i + 1] * g.Rinf[i + 1] - g.Rinf[i] * g.Rinf[i]);
            if (i >= 1) {
                c2[i] = 1.0 / (g.Rmed[i] * g.Rmed[i] - g.Rmed[i - 1] * g.Rmed[i - 1]);
                e2[i] = 1.0 / (g.Rmed[i] + g.Rmed[i - 1]);
            }
        }
        if (!rc) rc = dev_upload(c, &P.g_inv_dxt_src.p, a);
        if (!rc) rc = dev_upload(c, &P.g_inv_rsum.p, b);
        if (!rc) rc = dev_upload(c, &P.g_inv_drmed2.p, c2);
        if (!rc) rc = dev_upload(c, &P.g_inv_dra2.p, d2);
        if (!rc) rc = dev_upload(c, &P.g_inv_rmsum.p, e2);
        std::vector<double> t1(nr + 1, 0.0), t2(nr + 1, 0.0), t3(nr + 1, 0.0), t4(nr + 1, 0.0);
        for (int i = 0; i < nr; ++i) {
            t1[i] = g.dphi * g.Rmed[i];
            t2[i] = 1.0 / t1[i];
            t3[i] = (g.Rsup[i] - g.Rinf[i]) * g.InvSurf[i];
            t4[i] = g.Rmed[i] * d->omega_frame;
        }
        if (!rc) rc = dev_upload(c, &P.g_dxtheta.p, t1);
        if (!rc) rc = dev_upload(c, &P.g_inv_dxtheta.p, t2);
        if (!rc) rc = dev_upload(c, &P.g_dr_invsurf.p, t3);
        if (!rc) rc = dev_upload(c, &P.g_r_omega.p, t4);
        std::vector<double> t5(nr + 1, 0.0);
        for (int i = 0; i < nr; ++i) {
            const double rm = g.Rmed[i];
            t5[i] = 1.0 / std::sqrt(d->G * d->hydro_center_mass / (rm * rm * rm));
        }
        if (!rc) rc = dev_upload(c, &P.g_inv_omk.p, t5);
        std::vector<double> t6(nr + 1, 0.0);
        for (int i = 0; i < nr; ++i) {
            const double rm = g.Rmed[i];
            t6[i] = std::sqrt(d->G * d->hydro_center_mass / (rm * rm * rm));
        }
        if (!rc) rc = dev_upload(c, &P.g_omk.p, t6);
        std::vector<double> t7(nr + 2, 0.0);
        for (int i = 0; i <= nr; ++i)
            t7[i] = std::pow(g.Rinf[i], 3);
        if (!rc) rc = dev_upload(c, &P.g_ra3.p, t7);
        std::vector<RadRow> rt(nr + 2);
        for (int k = -1; k <= nr; ++k) {
            const bool open = k > 0 && k < nr;
            const int kk = open ? k : 1;
            RadRow &r = rt[k + 1];
            r.dr_lo = g.Rmed[kk] - g.Rmed[kk - 1];
            r.dr_hi = g.Rmed[kk + 1] - g.Rmed[kk];
            r.gphi = g.dphi * g.Rinf[kk];
            r.idr_up = (k + 1 >= 1 && k + 1 <= nr - 1) ? g.InvDiffRmed[k + 1] : 0.0;
        }
        std::vector<ThetaRow> tt(nr);
        for (int i = 0; i < nr; ++i)
            tt[i] = ThetaRow{g.InvSurf[i], t1[i], t2[i], t3[i], g.InvRmed[i], t4[i], g.Rmed[i], 0.0};
        if (!rc) rc = dev_upload_raw(c, &P.rad_tab, rt);
        if (!rc) rc = dev_upload_raw(c, &P.theta_tab, tt);
        if (!rc) rc = dev_alloc(c, &P.shift_tab, (size_t)nr);
    }
    std::vector<double> nu_ring(nr);
    {
        // isothermal alpha viscosity per ring, exactly as k_iso_cs_h + k_viscosity evaluate it:
        // H = cs * (1 / Omega_K), nu = alpha * H * cs
        for (int i = 0; i < nr; ++i) {
            const double r = c->geo.Rmed[i];
            const double inv_omega_kepler = 1.0 / std::sqrt(d->G * d->hydro_center_mass / (r * r * r));
            const double H = cs_ring[i] * inv_omega_kepler;
            nu_ring[i] = d->viscous_alpha * H * cs_ring[i];
        }
        if (!rc)
            rc = dev_upload(c, &P.nu_ring.p, nu_ring);
    }
    {
        // per-iteration rows of the marching source kernel (k_source_march): every per-ring factor
        // with the index expression and operation order of the kernel's former array reads
        const HostGeometry &g = c->geo;
        auto at = [](const std::vector<double> &v, int i) { return (i >= 0 && i < (int)v.size()) ? v[i] : 0.0; };
        auto crow = [nr](int r) { return r < 0 ? 0 : (r > nr - 1 ? nr - 1 : r); };
        const bool alpha = d->viscous_alpha > 0;
        auto nu_of = [&](int r) { return alpha ? nu_ring[crow(r)] : d->constant_viscosity; };
        const double C2 = d->artificial_viscosity_factor * d->artificial_viscosity_factor;
        std::vector<SrcRow> rows(nr + 8);
        for (int m = -2; m <= nr + 5; ++m) {
            SrcRow &R = rows[m + 2];
            std::memset(&R, 0, sizeof(R));
            {
                const int rc0 = crow(m), rc1 = crow(m - 1);
                R.cs2_m = cs_ring[rc0] * cs_ring[rc0];
                R.cs2_m1 = cs_ring[rc1] * cs_ring[rc1];
                R.idr_m = at(g.InvDiffRmed, m);
                R.rinf_om_m = at(g.Rinf, m) * d->omega_frame;
                R.inv_rinf_m = at(g.InvRinf, m);
                R.inv_dxt_m = (m >= 0 && m <= nr) ? 2.0 / (g.dphi * (g.Rsup[m] + g.Rinf[m])) : 0.0;
            }
            {
                const int r = crow(m - 1);
                R.inv_drsup_b = g.InvDiffRsup[r];
                R.inv_rmed_b = g.InvRmed[r];
                const double Dr = g.Rinf[r + 1] - g.Rinf[r];
                const double rDphi = g.Rmed[r] * g.dphi;
                const double dx = nphi <= 16 ? std::min(Dr, rDphi) : std::max(Dr, rDphi);
                R.lsq_b = C2 * (dx * dx);
                R.rinf_b1 = g.Rinf[r + 1];
                R.rinf_b0 = g.Rinf[r];
                R.inv_drsuprb_b = g.InvDiffRsupRb[r];
                const double rm = g.Rmed[r];
                R.inv_omk_b = 1.0 / std::sqrt(d->G * d->hydro_center_mass / (rm * rm * rm));
                R.inv_dxtheta_b = 1.0 / (g.dphi * rm);
            }
            {
                const int r = m - 1;
                if (r >= 1 && r <= nr) {
                    R.inv_rsum_c = 1.0 / (g.Rsup[r] + g.Rinf[r]);
                    R.rmed_c = g.Rmed[r];
                    R.rmed_cm1 = g.Rmed[r - 1];
                    R.inv_drmed2_c = 1.0 / (g.Rmed[r] * g.Rmed[r] - g.Rmed[r - 1] * g.Rmed[r - 1]);
                }
                R.idr_c = at(g.InvDiffRmed, r);
                R.inv_dxtheta_c = at(g.InvRmed, r) * g.invdphi;
            }
            {
                const int r = crow(m - 2);
                R.rinf_d1 = g.Rinf[r + 1];
                R.rinf_d0 = g.Rinf[r];
                R.inv_drsuprb_d = g.InvDiffRsupRb[r];
                R.inv_rmed_d = g.InvRmed[r];
                R.inv_drsup_d = g.InvDiffRsup[r];
                R.nu_d = nu_of(m - 2);
            }
            {
                const int r = m - 1;
                if (r >= 1 && r <= nr - 1) {
                    R.inv_rmed_r = g.InvRmed[r];
                    R.inv_rmed_rm1 = g.InvRmed[r - 1];
                    R.idr_r = g.InvDiffRmed[r];
                    R.rinf_r = g.Rinf[r];
                    R.inv_rinf_r = g.InvRinf[r];
                    const double nu1 = nu_of(r), nu2 = nu_of(r - 1);
                    R.nu_avg_r = 0.25 * (nu1 + nu2 + nu1 + nu2);
                }
            }
            {
                const int k = m - 2;
                if (k >= 1 && k <= nr) {
                    const double ra1 = at(g.Rinf, k + 1), ra0 = g.Rinf[k];
                    R.inv_rmed_k = at(g.InvRmed, k);
                    R.two_inv_dra2_k = 2.0 * (1.0 / (ra1 * ra1 - ra0 * ra0));
                    R.ra1sq_k = ra1 * ra1;
                    R.ra0sq_k = ra0 * ra0;
                    R.inv_rmsum_k = 1.0 / (g.Rmed[k] + g.Rmed[k - 1]);
                    R.rmed_k = g.Rmed[k];
                    R.rmed_km1 = g.Rmed[k - 1];
                    R.idr_k = at(g.InvDiffRmed, k);
                }
            }
        }
        if (!rc) rc = dev_upload_raw(c, &P.src_tab, rows);
    }

    const size_t ns = (size_t)nr * nphi, nv = (size_t)(nr + 1) * nphi;
#define AL(field, n) \
    if (!rc)         \
        rc = dev_alloc(c, &P.field, (n));
    AL(sigma, ns) AL(vrad, nv) AL(vazi, ns) AL(energy, ns) AL(vrad_b, nv) AL(vazi_b, ns) AL(energy_b, ns)
    AL(pressure, ns) AL(soundspeed, ns) AL(scale_height, ns) AL(viscosity, ns) AL(temperature, ns)
    AL(potential, ns)
    AL(sigma0, ns) AL(vrad0, nv) AL(vazi0, ns) AL(energy0, ns)
    AL(qr, ns) AL(qphi, ns) AL(divv, ns) AL(trr, ns) AL(tpp, ns) AL(trp, nv) AL(qplus, ns) AL(qminus, ns)
    AL(rmpA, ns) AL(rmmA, ns) AL(lpA, ns) AL(lmA, ns) AL(sigA, ns) AL(eA, ns)
    AL(rmpB, ns) AL(rmmB, ns) AL(lpB, ns) AL(lmB, ns) AL(sigB, ns) AL(eB, ns)
    AL(vmean, (size_t)nr + 1) AL(vconst, (size_t)nr) AL(nshift, (size_t)nr) AL(clk, 1) AL(shift_jump, 8)
    AL(cfl_part, (size_t)(nr + 256) * (size_t)((nphi + 255) / 256 + 1))
    AL(cfl_tickets, 32)
    P.ring_pstride = nphi / 32 + 4;
    AL(ring_part, (size_t)nr * P.ring_pstride)
    P.stabilize = d->stabilize_viscosity;
    if (P.stabilize) {
        AL(cfac_phi, ns) AL(cfac_r, ns)
    }
    if (d->write_massflow) {
        AL(massflow, nv)
    }
    P.accel_force = d->body_force_from_potential ? 0 : 1;
    if (P.accel_force) {
        AL(accel_r, nv) AL(accel_az, nv) // zero-filled: rows 0 and nr are never written (Pframeforce.cpp:121-123)
    }
    double *thermal_grid = nullptr; // ideal EOS: the cell-local CFL terms left by the marching transport
    if (!rc && d->eos == FCPT_EOS_IDEAL)
        rc = dev_alloc(c, &thermal_grid, ns);
    c->thermal_grid = thermal_grid;
    if (!rc && d->eos == FCPT_EOS_IDEAL)
        rc = dev_alloc(c, &P.qdiff, ns);
#undef AL
    if (!rc && hipHostMalloc((void **)&c->h_clk, sizeof(DevClock)) != hipSuccess) {
        set_error("hipHostMalloc failed");
        rc = FCPT_ENOMEM;
    }
    if (rc) {
        fcpt_destroy(c);
        return rc;
    }
    P.vmean_c.p = P.vmean;
    P.vconst_c.p = P.vconst;
    P.nshift_c.p = P.nshift;
    c->grid[FCPT_F_SIGMA] = P.sigma;
    c->grid[FCPT_F_VRAD] = P.vrad;
    c->grid[FCPT_F_VAZI] = P.vazi;
    c->grid[FCPT_F_ENERGY] = P.energy;
    c->grid[FCPT_F_PRESSURE] = P.pressure;
    c->grid[FCPT_F_SOUNDSPEED] = P.soundspeed;
    c->grid[FCPT_F_SCALE_HEIGHT] = P.scale_height;
    c->grid[FCPT_F_VISCOSITY] = P.viscosity;
    c->grid[FCPT_F_TEMPERATURE] = P.temperature;
    c->grid[FCPT_F_POTENTIAL] = P.potential;
    c->grid[FCPT_F_SIGMA0] = P.sigma0;
    c->grid[FCPT_F_VRAD0] = P.vrad0;
    c->grid[FCPT_F_VAZI0] = P.vazi0;
    c->grid[FCPT_F_ENERGY0] = P.energy0;
    c->grid[FCPT_F_QPLUS] = P.qplus;
    c->grid[FCPT_F_QMINUS] = P.qminus;
    c->grid[FCPT_F_VISC_CFAC_PHI] = P.cfac_phi; // null unless StabilizeViscosity
    c->grid[FCPT_F_VISC_CFAC_R] = P.cfac_r;
    c->grid[FCPT_F_MASSFLOW] = P.massflow; // null unless WriteMassFlow
    c->grid[FCPT_F_ACCEL_RADIAL] = P.accel_r; // null unless BodyForceFromPotential: no
    c->grid[FCPT_F_ACCEL_AZIMUTHAL] = P.accel_az;

    P.zero_no_ghost = c->s.zero_no_ghost;
    P.one_no_ghost_vr = c->s.one_no_ghost_vr;
    P.max_no_ghost = c->s.max_no_ghost;
    P.maxmo_no_ghost_vr = c->s.maxmo_no_ghost_vr;
    P.first_active = c->s.radial_first_active;
    P.active_size = c->s.radial_active_size;
    P.is_first = c->s.is_first;
    P.is_last = c->s.is_last;
    P.adiabatic = d->eos == FCPT_EOS_IDEAL;
    P.art_visc = d->artificial_viscosity;
    P.art_visc_dissipation = d->artificial_viscosity_dissipation;
    P.heating_viscous = d->heating_viscous;
    P.fast_transport = d->fast_transport;
    P.limiter = d->flux_limiter;
    P.leapfrog = d->integrator == FCPT_INTEGRATOR_LEAPFROG;
    P.alpha_viscosity = d->viscous_alpha > 0;
    P.gamma = d->adiabatic_index;
    P.mu = d->mu;
    P.Rgas = d->Rgas;
    P.G = d->G;
    P.Mc = d->hydro_center_mass;
    P.sigma_sb = d->sigma_sb;
    P.c_light = d->c_light;
    P.aspect_ratio = d->aspect_ratio;
    P.flaring_index = d->flaring_index;
    P.tmin = d->minimum_temperature;
    P.cooling_surface = d->cooling_surface;
    P.opacity = d->opacity;
    P.cooling_beta = d->cooling_beta;
    P.cooling_beta_reference = d->cooling_beta_reference;
    P.cooling_radiative_factor = d->cooling_radiative_factor;
    P.kappa_const = d->kappa_const;
    P.kappa_factor = d->kappa_factor;
    P.tau_factor = d->tau_factor;
    P.tau_min = d->tau_min;
    P.density_factor = d->density_factor;
    P.cooling_beta_value = d->cooling_beta_value;
    P.cooling_beta_ramp_up = d->cooling_beta_ramp_up;
    P.temperature_cgs = d->temperature_cgs;
    P.density_cgs = d->density_cgs;
    P.opacity_cgs = d->opacity_cgs;
    P.emin_fac = d->minimum_temperature / d->mu * d->Rgas / (d->adiabatic_index - 1.0);
    P.emax_fac = d->maximum_temperature / d->mu * d->Rgas / (d->adiabatic_index - 1.0);
    P.b_fac = d->mu * (d->adiabatic_index - 1.0) / d->Rgas;
    P.alpha_fac = 2.0 * 4.0 * d->sigma_sb / d->c_light;
    P.tmax = d->maximum_temperature;
    P.sigma_floor_abs = d->sigma_floor * d->sigma0;
    P.sigma_floor_rel = d->sigma_floor;
    P.sigma0_val = d->sigma0;
    P.alpha = d->viscous_alpha;
    P.nu_const = d->constant_viscosity;
    P.radial_viscosity_factor = d->radial_viscosity_factor;
    P.art_visc_factor = d->artificial_viscosity_factor;
    P.heating_viscous_factor = d->heating_viscous_factor;
    P.omega_frame = d->omega_frame;
    P.thickness_smoothing = d->thickness_smoothing;
    P.cfl = d->cfl;
    P.cfl_max_var = d->cfl_max_var;
    P.heating_cooling_cfl_limit = d->heating_cooling_cfl_limit;
    P.monitor_timestep = d->monitor_timestep;
    for (int k = 0; k < 2; ++k) {
        P.bc_sigma[k] = d->bc_sigma[k];
        P.bc_energy[k] = d->bc_energy[k];
        P.bc_vrad[k] = d->bc_vrad[k];
        P.bc_vaz[k] = d->bc_vaz[k];
        P.kep_vaz[k] = d->keplerian_vaz_factor[k];
        P.kep_vrad[k] = d->keplerian_vrad_factor[k];
    }
    // default body: the central star at the origin
    P.nbodies = 1;
    P.bm[0] = d->hydro_center_mass;

    const int dtype[4][2] = {{d->damp_vrad[0], d->damp_vrad[1]},
                             {d->damp_vaz[0], d->damp_vaz[1]},
                             {d->damp_sigma[0], d->damp_sigma[1]},
                             {d->damp_energy[0], d->damp_energy[1]}};
    for (int q = 0; q < 4; ++q)
        for (int o = 0; o < 2; ++o)
            c->damp[q][o] = damp_range(c, q == 0, dtype[q][o], o);

    {
        // per-ring damping tables for the fused end-of-transport kernel (reference / zero targets;
        // "mean" needs a ring reduction and keeps the separate k_damping launches)
        std::vector<double> fs(nr + 1, 0.0), ts(nr + 1, 1.0), fv(nr + 1, 0.0), tv(nr + 1, 1.0);
        std::vector<int> ty[4];
        bool any = false, mean = false;
        for (int q = 0; q < 4; ++q) {
            ty[q].assign(nr + 1, 0);
            for (int o = 0; o < 2; ++o) {
                const DampRange &r = c->damp[q][o];
                if (r.type == FCPT_DAMP_NONE || r.lo > r.hi)
                    continue;
                any = true;
                mean = mean || r.type == FCPT_DAMP_MEAN;
                const std::vector<double> &radius = q == 0 ? c->geo.Rinf : c->geo.Rmed;
                for (int i = r.lo; i <= r.hi; ++i) {
                    const double t = (radius[i] - r.rlim) / (r.redge - r.rlim);
                    (q == 0 ? fv : fs)[i] = t * t;
                    (q == 0 ? tv : ts)[i] = r.tau;
                    ty[q][i] = r.type == FCPT_DAMP_REFERENCE ? 1 : 2;
                }
            }
        }
        auto upi = [&](CArrI &dst, const std::vector<int> &src) {
            int *p = nullptr;
            if (int e = dev_alloc(c, &p, src.size()))
                return e;
            if (hipMemcpy(p, src.data(), src.size() * sizeof(int), hipMemcpyHostToDevice) != hipSuccess)
                return (int)FCPT_EHIP;
            dst.p = p;
            return (int)FCPT_OK;
        };
        if (!rc) rc = dev_upload(c, &P.dfac_s.p, fs);
        if (!rc) rc = dev_upload(c, &P.dtau_s.p, ts);
        if (!rc) rc = dev_upload(c, &P.dfac_v.p, fv);
        if (!rc) rc = dev_upload(c, &P.dtau_v.p, tv);
        if (!rc) rc = upi(P.dtype_vr, ty[0]);
        if (!rc) rc = upi(P.dtype_va, ty[1]);
        if (!rc) rc = upi(P.dtype_sig, ty[2]);
        if (!rc) rc = upi(P.dtype_e, ty[3]);
        std::vector<DampRow> dr(nr + 1);
        for (int i = 0; i <= nr; ++i)
            dr[i] = DampRow{fs[i], ts[i], fv[i], tv[i], ty[0][i], ty[1][i], ty[2][i], ty[3][i], {0.0, 0.0}};
        if (!rc) rc = dev_upload_raw(c, &P.damp_tab, dr);
        c->ring_ref_damped.assign(nr, 0);
        for (int i = 0; i < nr; ++i)
            c->ring_ref_damped[i] = (ty[0][i] == 1 || ty[1][i] == 1 || ty[2][i] == 1 || ty[3][i] == 1) ? 1 : 0;
        c->tf_sched_cap = (size_t)4 * 65536; // graded chunks: two to three rounds of a 512-CU device's wavefront slots, and the padding
        if (!rc) rc = dev_alloc(c, &c->tf_sched_dev, c->tf_sched_cap);
        c->sm_sched_cap = (size_t)4 * (8 * 4 * 8 * 64 + 64); // one entry per wavefront slot of a 512-CU device at 8 per SIMD
        if (!rc) rc = dev_alloc(c, &c->sm_sched_dev, c->sm_sched_cap);
        if (const char *q = getenv("FCPT_TF_SCHEDULE")) { // tuning runs: "28x56,12x24,6" = 56 chunks of 28 rings, 24 of 12, the rest of 6
            while (*q) {
                char *e = nullptr;
                const long a = strtol(q, &e, 10);
                if (e == q)
                    break;
                long n = 1;
                q = e;
                if (*q == 'x' || *q == 'X') {
                    n = strtol(q + 1, &e, 10);
                    q = e;
                }
                for (long k = 0; k < n && (long)c->tf_lengths.size() < 4l * nr; ++k)
                    c->tf_lengths.push_back((int)(a < 1 ? 1 : a));
                while (*q == ',' || *q == ' ')
                    ++q;
            }
        }
        if (rc) {
            fcpt_destroy(c);
            return rc;
        }
        // leapfrog kicks the gas once more after the transport, so its damping cannot be folded in
        c->damp_any = d->damping && any;
        c->damp_foldable = d->damping && any && !mean && d->integrator == FCPT_INTEGRATOR_EULER;
    }

    DevClock clk;
    std::memset(&clk, 0, sizeof(clk));
    clk.last_dt = d->first_dt; // Interpret.cpp:86
    clk.dt = d->first_dt;
    clk.dt_min = 1.0e300;
    clk.dt_max = 0.0;
    if (hipMemcpy(P.clk, &clk, sizeof(clk), hipMemcpyHostToDevice) != hipSuccess) {
        set_error("hipMemcpy of the clock failed");
        fcpt_destroy(c);
        return FCPT_EHIP;
    }
    options_from_environment(c->P.opt);
    apply_options(c);
    *out = c;
    return FCPT_OK;
}

int fcpt_set_option(fcpt_ctx *c, const char *name, int32_t value)
{
    if (!c || !name)
        return FCPT_EINVAL;
    int *slot = option_slot(c->P.opt, name);
    if (!slot) {
        set_error("fcpt_set_option: unknown option '%s'", name);
        return FCPT_EINVAL;
    }
    if (*slot == value)
        return FCPT_OK;
    if (c->stepped) {
        set_error("fcpt_set_option between fcpt_step and fcpt_post");
        return FCPT_EINVAL;
    }
    join_side(c);
    // a switch may change which grids hold the derived quantities: finish what is queued, then start clean
    HIPCHK(hipStreamSynchronize(c->stream));
    *slot = value;
    apply_options(c, false);
    HIPCHK(hipGetLastError());
    return FCPT_OK;
}

int fcpt_get_option(const fcpt_ctx *c, const char *name, int32_t *value)
{
    if (!c || !name || !value)
        return FCPT_EINVAL;
    // read-only counters of the hipGraph replay in fcpt_run_steps
    if (!std::strcmp(name, "graph_replays")) {
        *value = (int32_t)(c->graph_replays > 0x7fffffffll ? 0x7fffffffll : c->graph_replays);
        return FCPT_OK;
    }
    if (!std::strcmp(name, "transport_fell_back")) { // 1: the last transport met a ring pair beyond the one-lane shift
        fcpt_ctx *m = const_cast<fcpt_ctx *>(c);      //    and took the two-kernel path (blocks: reads the device stamps)
        int stamps[4] = {0, 0, 0, 0};
        join_side(m);
        HIPCHK(hipMemcpyAsync(stamps, c->P.shift_jump, sizeof(stamps), hipMemcpyDeviceToHost, c->stream));
        HIPCHK(hipStreamSynchronize(c->stream));
        *value = (stamps[2] != 0 && stamps[0] == stamps[2]) ? 1 : 0;
        return FCPT_OK;
    }
    if (!std::strcmp(name, "coop_active")) { // 1: fcpt_run_steps takes the one-kernel-per-step path on this grid
        *value = c->coop_active ? 1 : 0;
        return FCPT_OK;
    }
    if (!std::strcmp(name, "graph_cycle")) {
        *value = c->graph_exec ? c->graph_cycle : 0;
        return FCPT_OK;
    }
    const int *slot = option_slot(const_cast<Options &>(c->P.opt), name);
    if (!slot) {
        set_error("fcpt_get_option: unknown option '%s'", name);
        return FCPT_EINVAL;
    }
    *value = *slot;
    return FCPT_OK;
}

int fcpt_set_transport_chunks(fcpt_ctx *c, const int32_t *lengths, int32_t n)
{
    if (!c || n < 0 || (n > 0 && !lengths))
        return FCPT_EINVAL;
    if (c->stepped) {
        set_error("fcpt_set_transport_chunks between fcpt_step and fcpt_post");
        return FCPT_EINVAL;
    }
    for (int k = 0; k < n; ++k)
        if (lengths[k] < 1) {
            set_error("fcpt_set_transport_chunks: chunk %d has %d rings", k, (int)lengths[k]);
            return FCPT_EINVAL;
        }
    join_side(c);
    HIPCHK(hipStreamSynchronize(c->stream));
    c->tf_lengths.assign(lengths, lengths + n);
    apply_options(c, false);
    HIPCHK(hipGetLastError());
    return FCPT_OK;
}

int fcpt_transport_chunks(const fcpt_ctx *c, int32_t *tile_first_last, int32_t capacity, int32_t *n_wavefronts)
{
    if (!c || !n_wavefronts || capacity < 0 || (capacity > 0 && !tile_first_last))
        return FCPT_EINVAL;
    const int n = (int)(c->tf_sched_host.size() / 4);
    *n_wavefronts = n;
    for (int k = 0; k < n && k < capacity; ++k)
        for (int q = 0; q < 3; ++q)
            tile_first_last[3 * k + q] = c->tf_sched_host[4 * k + q];
    return FCPT_OK;
}

int fcpt_source_chunks(const fcpt_ctx *c, int32_t *seg_first_last, int32_t capacity, int32_t *n_wavefronts)
{
    if (!c || !n_wavefronts || capacity < 0 || (capacity > 0 && !seg_first_last))
        return FCPT_EINVAL;
    const int n = (int)(c->sm_sched_host.size() / 4);
    *n_wavefronts = n;
    for (int k = 0; k < n && k < capacity; ++k)
        for (int q = 0; q < 3; ++q)
            seg_first_last[3 * k + q] = c->sm_sched_host[4 * k + q];
    return FCPT_OK;
}

int fcpt_destroy(fcpt_ctx *c)
{
    if (!c)
        return FCPT_OK;
    (void)fcpt_comm_destroy(c);
    drop_graph(c);
    if (c->capture_stream)
        (void)hipStreamDestroy(c->capture_stream);
    for (void *p : c->allocs)
        (void)hipFree(p);
    if (c->h_clk)
        (void)hipHostFree(c->h_clk);
    for (hipEvent_t e : c->prof.events)
        (void)hipEventDestroy(e);
    if (c->side) {
        (void)hipStreamSynchronize(c->side);
        (void)hipStreamDestroy(c->side);
    }
    if (c->e_fork)
        (void)hipEventDestroy(c->e_fork);
    if (c->e_join)
        (void)hipEventDestroy(c->e_join);
    delete c;
    return FCPT_OK;
}

int fcpt_set_stream(fcpt_ctx *c, void *hip_stream)
{
    if (!c)
        return FCPT_EINVAL;
    join_side(c);
    c->stream = (hipStream_t)hip_stream;
    return FCPT_OK;
}

int fcpt_synchronize(fcpt_ctx *c)
{
    if (!c)
        return FCPT_EINVAL;
    DevClock k;
    return read_clock(c, &k); // synchronises, and reports a step beyond the shear limit
}

int fcpt_get_split(const fcpt_ctx *c, fcpt_split *out)
{
    if (!c || !out)
        return FCPT_EINVAL;
    *out = c->s;
    return FCPT_OK;
}

int fcpt_get_clock(const fcpt_ctx *cc, fcpt_clock *out)
{
    fcpt_ctx *c = const_cast<fcpt_ctx *>(cc);
    if (!c || !out)
        return FCPT_EINVAL;
    DevClock k;
    if (int rc = read_clock(c, &k))
        return rc;
    out->time = k.time;
    out->last_dt = k.last_dt;
    out->n_hydro_iter = k.n_hydro_iter;
    out->n_monitor = k.n_monitor;
    out->n_snapshot = k.n_snapshot;
    return FCPT_OK;
}

// hydro_dt_logger (hydro_dt_logger.h:13-34)
int fcpt_dt_statistics(fcpt_ctx *c, double *dt_min, double *dt_max, int32_t reset)
{
    if (!c)
        return FCPT_EINVAL;
    DevClock k;
    if (int rc = read_clock(c, &k))
        return rc;
    if (dt_min)
        *dt_min = k.dt_min;
    if (dt_max)
        *dt_max = k.dt_max;
    if (reset) {
        k.dt_min = 1.0e300;
        k.dt_max = 0.0;
        *c->h_clk = k;
        HIPCHK(hipMemcpyAsync(c->P.clk, c->h_clk, sizeof(DevClock), hipMemcpyHostToDevice, c->stream));
        HIPCHK(hipStreamSynchronize(c->stream));
    }
    return FCPT_OK;
}

int fcpt_set_clock(fcpt_ctx *c, const fcpt_clock *in)
{
    if (!c || !in)
        return FCPT_EINVAL;
    DevClock k;
    if (int rc = read_clock(c, &k))
        return rc;
    k.time = in->time;
    k.last_dt = in->last_dt;
    k.n_hydro_iter = in->n_hydro_iter;
    k.n_monitor = in->n_monitor;
    k.n_snapshot = in->n_snapshot;
    // (the shift-jump stamps of k_ring_mean are sequence numbers derived from n_hydro_iter: none may survive a clock
    //  that is set back)
    HIPCHK(hipMemsetAsync(c->P.shift_jump, 0, 8 * sizeof(int), c->stream));
    *c->h_clk = k;
    HIPCHK(hipMemcpyAsync(c->P.clk, c->h_clk, sizeof(DevClock), hipMemcpyHostToDevice, c->stream));
    HIPCHK(hipStreamSynchronize(c->stream));
    return FCPT_OK;
}

static size_t grid_count(const fcpt_ctx *c, int32_t f)
{
    const bool vec = f == FCPT_F_VRAD || f == FCPT_F_VRAD0 || f == FCPT_F_MASSFLOW || f == FCPT_F_ACCEL_RADIAL ||
                     f == FCPT_F_ACCEL_AZIMUTHAL;
    return (size_t)(c->s.nr + (vec ? 1 : 0)) * c->d.nphi;
}

int fcpt_upload(fcpt_ctx *c, int32_t f, const double *host)
{
    if (!c || !host || f < 0 || f >= FCPT_F_COUNT || !c->grid[f]) {
        set_error("bad argument to fcpt_upload");
        return FCPT_EINVAL;
    }
    join_side(c);
    HIPCHK(hipMemcpyAsync(c->grid[f], host, grid_count(c, f) * sizeof(double), hipMemcpyHostToDevice, c->stream));
    HIPCHK(hipStreamSynchronize(c->stream));
    if (f == FCPT_F_SIGMA || f == FCPT_F_VRAD || f == FCPT_F_VAZI || f == FCPT_F_ENERGY) {
        // a replaced state grid: nothing derived from the old one may be reused.  The non-lazy ideal-EOS paths read
        // c_s, H, nu, T from grids that only fcpt_init_physics / fcpt_recalculate_derived / fcpt_post refresh --
        // that contract (restart_load, restart.cpp:18-139) stays; what the library evaluates lazily is redone.
        c->pressure_valid = false;
        c->potential_valid = false;
        c->stepped = false;
        c->thermal_valid = false;
        c->ghosts_unknown = true;
        drop_graph(c); // its launches were chosen for ghost rings that satisfied the boundary conditions
    }
    if (f == FCPT_F_QPLUS || f == FCPT_F_QMINUS) {
        c->thermal_valid = false;
        c->qdiff_valid = false;
    }
    if (f == FCPT_F_SCALE_HEIGHT)
        c->potential_valid = false;
    c->cfl_interior = false;
    return FCPT_OK;
}

int fcpt_download(fcpt_ctx *c, int32_t f, double *host)
{
    if (!c || !host || f < 0 || f >= FCPT_F_COUNT || !c->grid[f]) {
        set_error("bad argument to fcpt_download");
        return FCPT_EINVAL;
    }
    join_side(c);
    if (f == FCPT_F_PRESSURE || (c->P.adiabatic && (f == FCPT_F_SOUNDSPEED || f == FCPT_F_SCALE_HEIGHT ||
                                                     f == FCPT_F_VISCOSITY || f == FCPT_F_TEMPERATURE)))
        ensure_pressure(c);
    if (f == FCPT_F_POTENTIAL && (!c->potential_valid || c->P.accel_force)) {
        // evaluated inside the source march (ideal EOS), or not at all (BodyForceFromPotential: no): the grid on
        // request, from the current state and bodies
        launch_potential(c->P, c->stream);
        if (!c->P.accel_force) // (there the flag speaks for the acceleration grids)
            c->potential_valid = !c->P.adiabatic;
    }
    HIPCHK(hipMemcpyAsync(host, c->grid[f], grid_count(c, f) * sizeof(double), hipMemcpyDeviceToHost, c->stream));
    HIPCHK(hipStreamSynchronize(c->stream));
    return FCPT_OK;
}

int fcpt_device_ptr(fcpt_ctx *c, int32_t f, void **dptr, uint64_t *count)
{
    if (!c || !dptr || f < 0 || f >= FCPT_F_COUNT || !c->grid[f])
        return FCPT_EINVAL;
    *dptr = c->grid[f];
    if (count)
        *count = grid_count(c, f);
    return FCPT_OK;
}

int fcpt_set_bodies(fcpt_ctx *c, int32_t n, const double *x, const double *y, const double *m,
                    const double *rsm, double ix, double iy)
{
    if (!c || n < 0 || n > FCPT_MAX_BODIES || (n > 0 && (!x || !y || !m))) {
        set_error("bad argument to fcpt_set_bodies (at most %d bodies)", FCPT_MAX_BODIES);
        return FCPT_EINVAL;
    }
    Dev &P = c->P;
    P.nbodies = n;
    for (int k = 0; k < n; ++k) {
        P.bx[k] = x[k];
        P.by[k] = y[k];
        P.bm[k] = m[k];
        P.brsm[k] = rsm ? rsm[k] : 0.0;
    }
    P.indirect_x = ix;
    P.indirect_y = iy;
    c->potential_valid = false;
    c->has_mid = false;
    return FCPT_OK;
}

int fcpt_set_bodies_midstep(fcpt_ctx *c, int32_t n, const double *x, const double *y, const double *m,
                            const double *rsm)
{
    if (!c || n != c->P.nbodies || (n > 0 && (!x || !y || !m))) {
        set_error("fcpt_set_bodies_midstep must follow fcpt_set_bodies with the same number of bodies");
        return FCPT_EINVAL;
    }
    for (int k = 0; k < n; ++k) {
        c->mx[k] = x[k];
        c->my[k] = y[k];
        c->mm[k] = m[k];
        c->mrsm[k] = rsm ? rsm[k] : 0.0;
    }
    c->has_mid = true;
    return FCPT_OK;
}

// init_euler (SourceEuler.cpp:251-285) + the tail of init_physics (init.cpp:337-341)
int fcpt_set_body_irradiation(fcpt_ctx *c, int32_t n, const double *temperature, const double *radius,
                              const double *rampup_time)
{
    if (!c || n < 0 || n > FCPT_MAX_BODIES || (n > 0 && (!temperature || !radius))) {
        set_error("bad argument to fcpt_set_body_irradiation");
        return FCPT_EINVAL;
    }
    if (!c->P.adiabatic) {
        set_error("irradiation needs EquationOfState: ideal");
        return FCPT_EINVAL;
    }
    c->P.heating_star = 0;
    for (int k = 0; k < FCPT_MAX_BODIES; ++k) {
        c->P.btemp[k] = k < n ? temperature[k] : 0.0;
        c->P.bradius[k] = k < n ? radius[k] : 0.0;
        c->P.bramp[k] = (k < n && rampup_time) ? rampup_time[k] : 0.0;
        if (c->P.btemp[k] > 0.0)
            c->P.heating_star = 1; // planetary_system.cpp:137-146
    }
    return FCPT_OK;
}

int fcpt_disk_on_body_accel(fcpt_ctx *c, double x, double y, double r_object, double smoothing_fixed,
                            double cubic_smoothing_radius, double out[4])
{
    if (!c || !out)
        return FCPT_EINVAL;
    join_side(c);
    ProfScope prof_scope(c);
    if (smoothing_fixed < 0.0 && c->P.adiabatic && !c->P.lazy_derived)
        ensure_pressure(c); // the scale-height grid
    double *d_out = c->P.cfl_part + 4 * (size_t)(((c->P.nphi + 255) / 256) * (c->P.nr / DOB_ROWS_HOST + 2));
    launch_disk_on_body(c->P, x, y, r_object, smoothing_fixed, cubic_smoothing_radius, d_out, c->stream);
    HIPCHK(hipGetLastError());
    HIPCHK(hipMemcpyAsync(out, d_out, 4 * sizeof(double), hipMemcpyDeviceToHost, c->stream));
    HIPCHK(hipStreamSynchronize(c->stream));
    return FCPT_OK;
}

int fcpt_init_physics(fcpt_ctx *c)
{
    if (!c)
        return FCPT_EINVAL;
    ProfScope prof_scope(c);
    const Dev &P = c->P;
    hipStream_t st = c->stream;
    if (!P.adiabatic) {
        launch_iso_cs_h(P, c->d_cs_ring, st);
        launch_pressure(P, st);
        launch_temperature(P, st);
    } else {
        launch_temperature(P, st);
        launch_recalculate_viscosity(P, st); // cs, H (and nu when alpha)
        launch_pressure(P, st);
    }
    launch_viscosity_field(P, st);
    // compute_heating_cooling_for_CFL (SourceEuler.cpp:1507-1547) runs before the velocities
    // are initialised (init.cpp:331-332): the gas is at rest, the stress tensor vanishes and
    // Q+ = Q- = 0, which is what the zero-filled grids already hold.
    const size_t ns = (size_t)P.nr * P.nphi * sizeof(double);
    HIPCHK(hipMemsetAsync(P.qplus, 0, ns, st));
    HIPCHK(hipMemsetAsync(P.qminus, 0, ns, st));
    // ... and its compute_viscous_stress_tensor leaves the StabilizeViscosity factors (functions of nu and Sigma
    // only) behind: with StabilizeViscosity 2 they limit the very first time step (cfl.cpp:331-351)
    if (P.adiabatic)
        launch_visc_factors(P, st);
    if (P.adiabatic && (P.cooling_surface || P.cooling_beta || P.heating_star)) {
        // ... but Q- does not vanish: calculate_qminus + the 1/alpha of compute_heating_cooling_for_CFL
        Dev I = P;
        I.cooling_at_init = 1;
        launch_substep3_cooling_only(I, st);
    }
    if (int rc = copy_initial_values(c))
        return rc;
    apply_boundary(c, false);
    if (int rc = copy_initial_values(c))
        return rc;
    c->potential_valid = false;
    c->pressure_valid = true;
    c->thermal_valid = false;
    HIPCHK(hipGetLastError());
    return FCPT_OK;
}


// recalculate_derived_disk_quantities (SourceEuler.cpp:225-249) after the state grids were replaced from outside
int fcpt_recalculate_derived(fcpt_ctx *c)
{
    if (!c)
        return FCPT_EINVAL;
    join_side(c);
    ProfScope prof_scope(c);
    c->stepped = false;
    c->cfl_interior = false;
    c->potential_valid = false;
    c->thermal_valid = false;
    if (c->P.adiabatic && !c->P.lazy_derived) {
        launch_derived(c->P, c->stream);
        c->pressure_valid = true;
    } else {
        c->pressure_valid = false; // evaluated lazily
    }
    HIPCHK(hipGetLastError());
    return FCPT_OK;
}

// test hook: the transport kernels' half_limiter on host-supplied operands
int fcpt_selftest_half_limiter(int32_t limiter, int64_t n, const double *a, const double *b, double *out)
{
    if (n < 0 || (n > 0 && (!a || !b || !out)) || (limiter != FCPT_LIMITER_VANLEER && limiter != FCPT_LIMITER_MC))
        return FCPT_EINVAL;
    if (n == 0)
        return FCPT_OK;
    int ndev = 0;
    if (hipGetDeviceCount(&ndev) != hipSuccess || ndev <= 0) {
        set_error("no HIP device available");
        return FCPT_ENODEV;
    }
    double *d = nullptr;
    const size_t nb = (size_t)n * sizeof(double);
    HIPCHK(hipMalloc((void **)&d, 3 * nb));
    int rc = FCPT_OK;
    if (hipMemcpy(d, a, nb, hipMemcpyHostToDevice) != hipSuccess || hipMemcpy(d + n, b, nb, hipMemcpyHostToDevice) != hipSuccess)
        rc = FCPT_EHIP;
    if (!rc) {
        launch_selftest_half_limiter(limiter, n, d, d + n, d + 2 * n, nullptr);
        if (hipDeviceSynchronize() != hipSuccess || hipMemcpy(out, d + 2 * n, nb, hipMemcpyDeviceToHost) != hipSuccess)
            rc = FCPT_EHIP;
    }
    (void)hipFree(d);
    if (rc)
        set_error("fcpt_selftest_half_limiter: a HIP call failed");
    return rc;
}

// test hook: the chunk tables of the marching kernels as pure host logic
int fcpt_selftest_chunk_tables(int32_t nr, int32_t nphi, int32_t n_cu, int32_t adiabatic, int32_t damp_inner, int32_t damp_outer,
                               int32_t *transport_first_last, int32_t transport_capacity, int32_t *n_transport,
                               int32_t *source_seg_first_last, int32_t source_capacity, int32_t *n_source)
{
    if (nr < 1 || nphi < 1 || n_cu < 8 || damp_inner < 0 || damp_outer < 0 || !n_transport || !n_source || transport_capacity < 0 ||
        source_capacity < 0 || (transport_capacity > 0 && !transport_first_last) || (source_capacity > 0 && !source_seg_first_last))
        return FCPT_EINVAL;
    Options opt;
    options_from_environment(opt);
    std::vector<int> t, s;
    selftest_chunk_tables(nr, nphi, n_cu, adiabatic != 0, damp_inner, damp_outer, opt, t, s);
    *n_transport = (int32_t)(t.size() / 4);
    *n_source = (int32_t)(s.size() / 4);
    for (int k = 0; k < *n_transport && k < transport_capacity; ++k)
        for (int q = 0; q < 3; ++q)
            transport_first_last[3 * k + q] = t[4 * k + q];
    for (int k = 0; k < *n_source && k < source_capacity; ++k)
        for (int q = 0; q < 3; ++q)
            source_seg_first_last[3 * k + q] = s[4 * k + q];
    return FCPT_OK;
}

int32_t fcpt_kernel_count(void) { return KID_COUNT; }
const char *fcpt_kernel_name(int32_t id) { return (id >= 0 && id < KID_COUNT) ? kKernelNames[id] : ""; }

int fcpt_profile_start(fcpt_ctx *c, uint64_t mask, int32_t max_launches)
{
    if (!c || max_launches < 0)
        return FCPT_EINVAL;
    Profiler &p = c->prof;
    while ((int)p.events.size() < 2 * max_launches) {
        hipEvent_t e;
        HIPCHK(hipEventCreate(&e));
        p.events.push_back(e);
    }
    p.mask = mask;
    p.used = 0;
    p.open_id = -1;
    p.stride = c->P.opt.profile_stride > 1 ? c->P.opt.profile_stride : 1;
    p.seen = 0;
    p.ids.clear();
    c->profiling = true;
    return FCPT_OK;
}

int fcpt_profile_stop(fcpt_ctx *c, double *ms_total, int64_t *launches)
{
    if (!c || !ms_total || !launches)
        return FCPT_EINVAL;
    c->profiling = false;
    join_side(c);
    HIPCHK(hipStreamSynchronize(c->stream));
    for (int k = 0; k < KID_COUNT; ++k) {
        ms_total[k] = 0.0;
        launches[k] = 0;
    }
    Profiler &p = c->prof;
    for (size_t n = 0; n < p.ids.size(); ++n) {
        float ms = 0.f;
        HIPCHK(hipEventElapsedTime(&ms, p.events[2 * n], p.events[2 * n + 1]));
        ms_total[p.ids[n]] += ms;
        launches[p.ids[n]] += 1;
    }
    return FCPT_OK;
}

} // extern "C"
