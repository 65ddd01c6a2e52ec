// Part of fcpt_kernels.hip (one translation unit, namespace fcpt): limiters, radial sweep, ring means, azimuthal sweeps (per-loop, fused, ring-marching), velocities.
// Not a stand-alone header: included once, in the order given there.

// ---------------------------------------------------------------------------
// Transport (TransportEuler.cpp).

// TransportEuler.cpp:306-337 flux_limiter
__device__ __forceinline__ double limiter(int type, double a, double b)
{
    if (type == FCPT_LIMITER_MC) {
        // minmod(0.5*(a+b), 2*minmod(a,b))
        double m = 0.0;
        if (a * b > 0.0)
            m = fabs(a) < fabs(b) ? a : b;
        const double c = 0.5 * (a + b), d = 2.0 * m;
        if (c * d > 0.0)
            return fabs(c) < fabs(d) ? c : d;
        return 0.0;
    }
    const double ab = a * b;
    return ab > 0.0 ? 2.0 * ab * fast_rcp1(a + b) : 0.0;
}

// The cell-local part of the ideal-EOS CFL sum (cfl.cpp:243-247,319-328): invdt1^2 + invdt5^2 + invdt6^2 -- sound
// speed, viscosity and |Q+ - Q-| / e -- depends on Sigma, e, Q+ and Q- of the cell alone.  The marching transport
// kernels hold the cell's final Sigma and e when they store them, so they leave that sum in one grid (cfl_thermal)
// and the next k_cfl_rings reads it instead of Sigma, e, Q+ and Q- (3 grids instead of 6).  c_s^2 = gamma (gamma-1)
// e / Sigma and nu = alpha H c_s = alpha c_s^2 / (sqrt(gamma) Omega_K) need no root.
struct ThermalRing {
    double inv_cell2, nu_fac, lf, inv_limit;
};
__device__ __forceinline__ ThermalRing thermal_ring(const Dev &P, int i)
{
    ThermalRing t;
    const double inv_dxr = P.InvDiffRsup[i], inv_dxa = P.InvRmed[i] * P.invdphi;
    const double inv_cell = dmax(inv_dxr, inv_dxa);
    t.inv_cell2 = inv_cell * inv_cell;
    t.nu_fac = P.alpha * (1.0 / sqrt(P.gamma)) * P.g_inv_omk[i];
    t.lf = P.leapfrog ? 0.6 : 1.0;
    t.inv_limit = 1.0 / P.heating_cooling_cfl_limit;
    return t;
}
__device__ __forceinline__ double cfl_thermal_term(const Dev &P, const ThermalRing &t, double sg, double e, double qp, double qm)
{
    const double re = fast_rcp(e);
    const double cs2 = P.gamma * (P.gamma - 1.0) * e * fast_rcp(sg);
    const double nu = P.alpha_viscosity ? t.nu_fac * cs2 : P.nu_const;
    const double invdt5 = 4.0 * nu * t.inv_cell2 * t.lf;
    const double invdt6 = t.inv_limit * fabs((qp - qm) * re) * t.lf;
    return cs2 * t.inv_cell2 + invdt5 * invdt5 + invdt6 * invdt6;
}

// Upwind "star" state at radial interface k (between rings k-1 and k),
// compute_star_radial (TransportEuler.cpp:349-406).  wm2..wp1 = Q at rings k-2..k+1.
// Per-interface geometry of compute_star_radial, loaded once with the (wavefront-uniform)
// interface index so it lives in scalar registers; the upwind choice then only selects
// between preloaded values instead of issuing lane-divergent loads.
struct StarGeo {
    double idr_m, idr_0, idr_p; // InvDiffRmed[k-1], [k], [k+1]
    double dr_lo, dr_hi;        // Rmed[k]-Rmed[k-1], Rmed[k+1]-Rmed[k]
    bool lim_lo, lim_hi;        // slope of ring k-1 / ring k is limited (not a closed boundary ring)
    bool open;                  // interface carries a flux (0 < k < Nr)
};
__device__ __forceinline__ StarGeo star_geo(const Dev &P, int k)
{
    StarGeo g;
    g.open = k > 0 && k < P.nr;
    const int kk = g.open ? k : 1;
    g.idr_m = P.InvDiffRmed[kk - 1];
    g.idr_0 = P.InvDiffRmed[kk];
    g.idr_p = P.InvDiffRmed[kk + 1];
    g.dr_lo = P.Rmed[kk] - P.Rmed[kk - 1];
    g.dr_hi = P.Rmed[kk + 1] - P.Rmed[kk];
    g.lim_lo = (kk - 1 != 0) && (kk - 1 != P.nr - 1);
    g.lim_hi = (kk != 0) && (kk != P.nr - 1);
    return g;
}
__device__ __forceinline__ double star_radial(const Dev &P, const StarGeo &g, double v, double dt,
                                              double wm2, double wm1, double w0, double wp1)
{
    if (!g.open)
        return 0.0; // row 0 is zeroed on every call, row Nr is never written
    // upwind cell c = k-1 (v > 0) or k; one limiter evaluation on the selected stencil
    const bool up = v > 0.0;
    const double x0 = up ? wm2 : wm1, x1 = up ? wm1 : w0, x2 = up ? w0 : wp1;
    const double ihi = up ? g.idr_0 : g.idr_p, ilo = up ? g.idr_m : g.idr_0;
    const bool lim = up ? g.lim_lo : g.lim_hi;
    const double dq = lim ? limiter(P.limiter, (x2 - x1) * ihi, (x1 - x0) * ilo) : 0.0;
    const double dist = up ? (g.dr_lo - v * dt) : -(g.dr_hi + v * dt);
    return x1 + dist * 0.5 * dq;
}

// WriteMassFlow: the mass VanLeerRadial carries through the inner interface of every cell, accumulated in the
// MASSFLOW grid (TransportEuler.cpp:566-571,609-616).  For the density QRStar is Sigma / Sigma = 1 on every open
// interface (0 on row 0, unset on row Nr), so varq_inf = dt dphi Rinf[i] DensityStar v_r and the reference's extra
// "+= varq_sup" of the last ring adds nothing.  A monitoring quantity: its own small kernel ahead of the
// transport (same Sigma, same post-kick v_r, same dt), so that the marching kernels stay as they are.
template <bool ROWU> __global__ void k_massflow(const Dev P)
{
    CELL(1, P.nr - 1);
    const double dt = P.clk->dt;
    const double v = P.vrad[IDX(i, j)];
    const StarGeo geo = star_geo(P, i);
    const double wm2 = i >= 2 ? P.sigma[IDX(i - 2, j)] : 0.0, wm1 = P.sigma[IDX(i - 1, j)];
    const double w0 = P.sigma[IDX(i, j)], wp1 = i + 1 < P.nr ? P.sigma[IDX(i + 1, j)] : 0.0;
    const double rho = star_radial(P, geo, v, dt, wm2, wm1, w0, wp1);
    const double g = dt * P.dphi * P.Rinf[i];
    P.massflow[IDX(i, j)] += g * 1.0 * rho * v;
}

// compute_momenta_from_velocities (:471-493) + OneWindRad (:138-167) with all
// VanLeerRadial calls (:545-620) in one pass.  Reads Sigma, v_r, v_phi(, e) and
// writes the transported momenta / density / energy to set B, so the in-place
// ordering constraint of the reference ("Sigma MUST be last") is met by
// construction: every quantity sees the pre-transport density.
//
// One thread owns a phi column and marches RADIAL_ROWS rings outward keeping the
// 4-ring stencil of every specific quantity in registers, so each interface flux
// is evaluated once and each ring is loaded once per chunk (+4 halo rings).
// The specific momenta Work = (Sigma v)/Sigma are formed as v directly (equal to
// the reference's quotient to within 1 ulp).
#define RADIAL_ROWS 16 /* rings per thread on grids that fill the GPU; fewer on small ones (march_len, launch.h) */

struct RadialRow { // specific quantities of one ring at this column (er: the energy itself)
    double s, rmp, rmm, lp, lm, e, er;
};
__device__ __forceinline__ RadialRow radial_load(const Dev &P, int k, int j, int jn, double vr_k,
                                                 double vr_k1)
{
    RadialRow w;
    if (k >= 0 && k < P.nr) {
        const double r = P.Rmed[k];
        w.s = P.sigma[IDX(k, j)];
        w.rmp = vr_k1;
        w.rmm = vr_k;
        w.lp = (P.vazi[IDX(k, jn)] + r * P.omega_frame) * r;
        w.lm = (P.vazi[IDX(k, j)] + r * P.omega_frame) * r;
        w.er = P.adiabatic ? P.energy[IDX(k, j)] : 0.0;
        w.e = P.adiabatic ? w.er / w.s : 0.0;
    } else {
        w.s = w.rmp = w.rmm = w.lp = w.lm = w.e = w.er = 0.0;
    }
    return w;
}
struct RadialFlux {
    double s, rmp, rmm, lp, lm, e;
};
// fluxes through interface k given rings k-2..k+1 (a,b,c,d) and v_r(k)
__device__ __forceinline__ RadialFlux radial_flux(const Dev &P, int k, double v, double dt,
                                                  const RadialRow &a, const RadialRow &b,
                                                  const RadialRow &c, const RadialRow &d)
{
    RadialFlux f;
    const StarGeo geo = star_geo(P, k);
    if (!geo.open) { // closed: QRStar/DensityStar row 0 zeroed, row Nr never written
        f.s = f.rmp = f.rmm = f.lp = f.lm = f.e = 0.0;
        return f;
    }
    const double rho = star_radial(P, geo, v, dt, a.s, b.s, c.s, d.s);
    const double g = dt * P.dphi * P.Rinf[k];
    f.s = g * 1.0 * rho * v;
    f.rmp = g * star_radial(P, geo, v, dt, a.rmp, b.rmp, c.rmp, d.rmp) * rho * v;
    f.rmm = g * star_radial(P, geo, v, dt, a.rmm, b.rmm, c.rmm, d.rmm) * rho * v;
    f.lp = g * star_radial(P, geo, v, dt, a.lp, b.lp, c.lp, d.lp) * rho * v;
    f.lm = g * star_radial(P, geo, v, dt, a.lm, b.lm, c.lm, d.lm) * rho * v;
    f.e = P.adiabatic ? g * star_radial(P, geo, v, dt, a.e, b.e, c.e, d.e) * rho * v : 0.0;
    return f;
}
template <bool ROWU> __device__ __forceinline__ void transport_radial_block(const Dev &P, int vb, int gx, int nvb, int rows)
{
    const int lb = xcd_block(vb, nvb);
    const int j = (lb % gx) * blockDim.x + threadIdx.x;
    const int r0_ = ((lb / gx) * blockDim.y + threadIdx.y) * rows;
    if (j >= P.nphi || r0_ >= P.nr)
        return;
    const int r0 = ROWU ? __builtin_amdgcn_readfirstlane(r0_) : r0_;
    const double dt = P.clk->dt;
    const int jn = JNEXT;
    const int nr = P.nr;
    const int r1 = r0 + rows < nr ? r0 + rows : nr;
    auto vr_at = [&](int k) { return (k >= 0 && k <= nr) ? P.vrad[IDX(k, j)] : 0.0; };
    // rings r0-2 .. r0+2 (the last one is the software-prefetched ring of the next iteration:
    // the marching loop is latency-bound unless each ring's loads are issued one iteration
    // before their first use)
    double v0 = vr_at(r0 - 2), v1 = vr_at(r0 - 1), v2 = vr_at(r0), v3 = vr_at(r0 + 1), v4 = vr_at(r0 + 2),
           v5 = vr_at(r0 + 3);
    RadialRow a = radial_load(P, r0 - 2, j, jn, v0, v1);
    RadialRow b = radial_load(P, r0 - 1, j, jn, v1, v2);
    RadialRow c = radial_load(P, r0, j, jn, v2, v3);
    RadialRow d = radial_load(P, r0 + 1, j, jn, v3, v4);
    RadialRow e = radial_load(P, r0 + 2, j, jn, v4, v5);
    RadialFlux fin = radial_flux(P, r0, v2, dt, a, b, c, d);
    for (int i = r0; i < r1; ++i) {
        // prefetch ring i+3 (used by the next iteration)
        const double v6 = vr_at(i + 4);
        const RadialRow f = radial_load(P, i + 3, j, jn, v5, v6);
        // rings i-1..i+2 around interface i+1
        const RadialFlux fout = radial_flux(P, i + 1, v3, dt, b, c, d, e);
        const double invsurf = P.InvSurf[i];
        const double s0 = c.s;
        // momenta of T1 for ring i (TransportEuler.cpp:484-490); c.lp/c.lm = (v_phi + r Omega) r
        P.rmpB[IDX(i, j)] = s0 * v3 + (fin.rmp - fout.rmp) * invsurf;
        P.rmmB[IDX(i, j)] = s0 * v2 + (fin.rmm - fout.rmm) * invsurf;
        P.lpB[IDX(i, j)] = s0 * c.lp + (fin.lp - fout.lp) * invsurf;
        P.lmB[IDX(i, j)] = s0 * c.lm + (fin.lm - fout.lm) * invsurf;
        if (P.adiabatic)
            P.eB[IDX(i, j)] = c.er + (fin.e - fout.e) * invsurf;
        P.sigB[IDX(i, j)] = s0 + (fin.s - fout.s) * invsurf;
        a = b; b = c; c = d; d = e; e = f;
        v2 = v3; v3 = v4; v4 = v5; v5 = v6;
        fin = fout;
    }
}

// The kernel proper walks gx * gy virtual blocks with a grid stride: launched with one block per
// virtual block in normal use, and with a small grid as the in-stream fallback of
// k_transport_fused (only_if: runs only when that kernel gave up; an idle fallback then costs a
// few hundred blocks that return at once, not thousands).
template <bool ROWU> __global__ void __launch_bounds__(256) k_transport_radial(const Dev P, const int *only_if, int gx, int gy, int rows)
{
    if (only_if && !shift_jump_raised(only_if))
        return;
    for (int vb = blockIdx.x; vb < gx * gy; vb += gridDim.x)
        transport_radial_block<ROWU>(P, vb, gx, gx * gy, rows);
}

// compute_average_azimuthal_velocity (:174-189) + ComputeConstantResidual (:207-236):
// one wavefront per ring (4 rings per block), 16-byte loads with 16 in flight per lane, a
// butterfly for the ring sum; the per-ring scalars of the epilogue are fetched up front so the
// last lane-0 instructions do not queue behind three dependent memory round trips.
//
// with_shift: the launch also decides whether this transport can take the one-kernel form: k_transport_fused covers
// |Nshift[i] - Nshift[i-1]| <= 1 (cyclic) for all ring pairs.  Every wavefront compares its ring's shift with the
// previous ring's (its block neighbour's through LDS; the first wavefront of a block sums ring i-1 once more itself)
// and a pair beyond the limit stamps shift_jump[0] with this transport's sequence number, which the wavefront of ring
// 0 leaves in shift_jump[2]: the kernels behind this one test shift_jump[0] == shift_jump[2].  Nothing is ever reset
// (a reset by one wavefront would race with the stamps of the others), and because the flag is known before
// k_transport_fused starts, that launch itself runs the radial sweep of the two-kernel form when it is raised --
// one gated azimuthal launch behind it completes the step, with no grid-wide barrier anywhere.
__device__ __forceinline__ double ring_sum_vphi(const Dev &P, int i, int lane, const double *part, int nparts, int pstride)
{
    // part != nullptr: the producer kernel left nparts partial sums per ring (fixed order, so the
    // result is deterministic); rings rewritten afterwards by a boundary condition are re-summed
    // from the grid.
    const bool ghost = (i == 0 && P.is_first && P.bc_vaz[0] != FCPT_BC_NONE) ||
                       (i == P.nr - 1 && P.is_last && P.bc_vaz[1] != FCPT_BC_NONE) ||
                       (!P.is_first && i < FCPT_OVERLAP) || (!P.is_last && i >= P.nr - FCPT_OVERLAP);
    double acc = 0.0;
    if (part && !ghost) {
        for (int n = lane; n < nparts; n += 64)
            acc += part[i * pstride + n];
    } else {
        const double *row = P.vazi + (size_t)i * P.nphi;
        const int npair = P.nphi >> 1;
        double acc2 = 0.0;
        int n = lane;
        // (32 requests per lane in flight: a ring of 4096 cells in one memory round trip -- the two ghost rings' wavefronts
        //  are the critical path of k_ring_mean, every other ring being 70 partial sums)
        for (; n + 31 * 64 < npair; n += 32 * 64) {
            D2 v[32];
#pragma unroll
            for (int u = 0; u < 32; ++u)
                v[u] = *(const D2 *)(row + 2 * (n + u * 64));
#pragma unroll
            for (int u = 0; u < 32; ++u) {
                acc += v[u].x;
                acc2 += v[u].y;
            }
        }
        for (; n + 7 * 64 < npair; n += 8 * 64) {
            D2 v[8];
#pragma unroll
            for (int u = 0; u < 8; ++u)
                v[u] = *(const D2 *)(row + 2 * (n + u * 64));
#pragma unroll
            for (int u = 0; u < 8; ++u) {
                acc += v[u].x;
                acc2 += v[u].y;
            }
        }
        if (n < npair) { // the last, ragged batch: all of its requests at once too (round 3: they went one by one, 512 x 1536: 6.6 -> 5.2 us)
            D2 v[8];
#pragma unroll
            for (int u = 0; u < 8; ++u) {
                v[u] = D2{0.0, 0.0};
                if (n + u * 64 < npair)
                    v[u] = *(const D2 *)(row + 2 * (n + u * 64));
            }
#pragma unroll
            for (int u = 0; u < 8; ++u)
                if (n + u * 64 < npair) {
                    acc += v[u].x;
                    acc2 += v[u].y;
                }
        }
        if ((P.nphi & 1) && lane == 0)
            acc += row[P.nphi - 1];
        acc += acc2;
    }
    for (int off = 32; off > 0; off >>= 1)
        acc += __shfl_down(acc, off, 64);
    return acc; // lane 0 holds the sum
}
// Nshift of a ring from its mean (ComputeConstantResidual, :207-236)
__device__ __forceinline__ double ring_ntilde(const Dev &P, int i, double mean, double dt) { return mean * P.InvRmed[i] * dt * P.invdphi; }

// (vb: the block's index among the ring-mean blocks of its launch; any 256-thread block shape)
__device__ __forceinline__ void ring_mean_block(const Dev &P, int vb, int with_shift, const double *part, int nparts, int pstride)
{
    __shared__ int s_nshift[4];
    const int tid = threadIdx.y * blockDim.x + threadIdx.x;
    const int lane = tid & 63, w = tid >> 6;
    const int i = __builtin_amdgcn_readfirstlane(vb * 4 + w);
    const bool active = i < P.nr;
    const double dt = with_shift ? P.clk->dt : 1.0;
    int my_shift = 0;
    if (active) {
        const double invr = P.InvRmed[i], rmed = P.Rmed[i];
        const double acc = ring_sum_vphi(P, i, lane, part, nparts, pstride);
        if (with_shift) { // (all lanes: the shift is needed wave-uniform below)
            const double mean0 = __shfl(acc, 0, 64) / (double)P.nphi;
            my_shift = (int)floor(ring_ntilde(P, i, mean0, dt) + 0.5);
        }
        if (lane == 0) {
            const double mean = acc / (double)P.nphi;
            P.vmean[i] = mean;
            if (with_shift && i == 0)
                P.shift_jump[2] = SHIFT_SEQ(P.clk); // this transport's sequence number
            if (with_shift) {
                const double invdt = 1.0 / dt;
                const double Ntilde = mean * invr * dt * P.invdphi;
                const double Nround = floor(Ntilde + 0.5);
                P.nshift[i] = (int)Nround;
                const double vc = (Ntilde - Nround) * rmed * invdt * P.dphi;
                P.vconst[i] = vc;
                ShiftRow sr;
                sr.mean = mean, sr.vconst = vc, sr.nshift = (int)Nround, sr.pad0 = 0, sr.pad1[0] = sr.pad1[1] = 0.0;
                const DampRow dr = P.damp_tab[i]; // damping.cpp:311-427: X <- (X - X0) exp(-dt f / tau) + X0
                sr.es = exp(-dt * dr.fs / dr.ts);
                sr.ev = exp(-dt * dr.fv / dr.tv);
                sr.ev_top = 1.0;
                if (i == P.nr - 1) {
                    const DampRow dn = P.damp_tab[P.nr];
                    sr.ev_top = exp(-dt * dn.fv / dn.tv);
                }
                P.shift_tab[i] = sr;
            }
        }
    }
    if (!with_shift)
        return;
    // ---- ring pairs beyond the one-lane shift -----------------------------------------------------------------------
    if (lane == 0)
        s_nshift[w] = my_shift;
    __syncthreads();
    if (!active || i == 0)
        return;
    int prev_shift;
    if (w > 0) {
        prev_shift = s_nshift[w - 1];
    } else { // the ring below belongs to another workgroup: its sum once more
        const double acc = ring_sum_vphi(P, i - 1, lane, part, nparts, pstride);
        const double mean_p = __shfl(acc, 0, 64) / (double)P.nphi;
        prev_shift = (int)floor(ring_ntilde(P, i - 1, mean_p, dt) + 0.5);
    }
    const int nphi = P.nphi;
    int dd = my_shift % nphi - prev_shift % nphi;
    dd = dd < 0 ? -dd : dd;
    dd = dd > nphi / 2 ? nphi - dd : dd;
    if (dd > 1 && lane == 0)
        P.shift_jump[0] = SHIFT_SEQ(P.clk);
}
__global__ void __launch_bounds__(256) k_ring_mean(const Dev P, int with_shift, const double *part, int nparts, int pstride)
{
    ring_mean_block(P, blockIdx.x, with_shift, part, nparts, pstride);
}
// The two independent first kernels of the unfused Transport() in one launch (narrow rings, where every launch is a
// visible share of the step): blocks [0, gx gy) do the radial sweep, the (nr + 3) / 4 blocks behind them the ring
// means, shifts and uniform residuals.
template <bool ROWU> __global__ void __launch_bounds__(256) k_transport_radial_means(const Dev P, int gx, int gy, int rows, const double *part,
                                                                                    int nparts, int pstride)
{
    if ((int)blockIdx.x < gx * gy)
        transport_radial_block<ROWU>(P, blockIdx.x, gx, gx * gy, rows);
    else
        ring_mean_block(P, (int)blockIdx.x - gx * gy, 1, part, nparts, pstride);
}

// Upwind star state at azimuthal interface k (between cells k-1 and k),
// ComputeStarTheta (:416-466); wm2..wp1 = Q at cells k-2..k+1.
__device__ __forceinline__ double star_theta(const Dev &P, double v, double dt, double dxtheta,
                                             double invdxtheta, double wm2, double wm1, double w0,
                                             double wp1)
{
    const double ksi = v * dt;
    const bool up = ksi > 0.0;
    const double x0 = up ? wm2 : wm1, x1 = up ? wm1 : w0, x2 = up ? w0 : wp1;
    const double dq = 0.5 * limiter(P.limiter, (x2 - x1), (x1 - x0)) * invdxtheta;
    const double dist = up ? (dxtheta - ksi) : -(dxtheta + ksi);
    return x1 + dist * dq;
}

struct ThetaSet {
    const double *rmp, *rmm, *lp, *lm, *sig, *e;
};
struct ThetaOut {
    double *rmp, *rmm, *lp, *lm, *sig, *e;
};

// QuantitiesAdvection (:292-304) with all VanLeerTheta calls (:630-664) in one pass,
// out of place.  PASS 1: residual velocity v_phi - <v_phi> (+ constant residual when the
// FARGO split is off).  PASS 2: uniform residual, and the integer shift AdvectSHIFT
// (:238-268) is applied by the store (cell j lands in j + Nshift).
template <int PASS, bool ROWU> __global__ void k_transport_theta(const Dev P, ThetaSet in, ThetaOut out)
{
    CELL(0, P.nr);
    const double dt = P.clk->dt;
    const int nphi = P.nphi;
    int jj[5];
    jj[2] = j;
    jj[1] = j == 0 ? nphi - 1 : j - 1;
    jj[0] = jj[1] == 0 ? nphi - 1 : jj[1] - 1;
    jj[3] = j == nphi - 1 ? 0 : j + 1;
    jj[4] = jj[3] == nphi - 1 ? 0 : jj[3] + 1;

    int jout = j;
    if (PASS == 2) {
        int s = j + P.nshift[i];
        s %= nphi;
        if (s < 0)
            s += nphi;
        jout = s;
    }
    const bool skip = (PASS == 2) && !P.fast_transport; // NoSplitAdvection rows (:646)
    if (skip) {
        out.rmp[IDX(i, jout)] = in.rmp[IDX(i, j)];
        out.rmm[IDX(i, jout)] = in.rmm[IDX(i, j)];
        out.lp[IDX(i, jout)] = in.lp[IDX(i, j)];
        out.lm[IDX(i, jout)] = in.lm[IDX(i, j)];
        out.sig[IDX(i, jout)] = in.sig[IDX(i, j)];
        if (P.adiabatic)
            out.e[IDX(i, jout)] = in.e[IDX(i, j)];
        return;
    }
    double v0, v1; // velocity at interfaces j and j+1
    if (PASS == 1) {
        const double m = P.vmean[i];
        v0 = P.vazi[IDX(i, j)] - m;
        v1 = P.vazi[IDX(i, jj[3])] - m;
        if (!P.fast_transport) {
            v0 = P.vconst[i] + v0;
            v1 = P.vconst[i] + v1;
        }
    } else {
        v0 = v1 = P.vconst[i];
    }
    const double dxtheta = P.dphi * P.Rmed[i];
    const double invdxtheta = 1.0 / dxtheta;
    const double dxrad = (P.Rsup[i] - P.Rinf[i]) * dt;
    const double invsurf = P.InvSurf[i];
    double S[5], rS[5]; // Work = Q * (1/Sigma): within 1 ulp of the reference's Q / Sigma
#pragma unroll
    for (int a = 0; a < 5; ++a) {
        S[a] = in.sig[IDX(i, jj[a])];
        rS[a] = fast_rcp(S[a]);
    }
    const double rho0 = star_theta(P, v0, dt, dxtheta, invdxtheta, S[0], S[1], S[2], S[3]);
    const double rho1 = star_theta(P, v1, dt, dxtheta, invdxtheta, S[1], S[2], S[3], S[4]);
#define THETA_UPDATE(IN, OUT)                                                                    \
    {                                                                                            \
        double W[5];                                                                             \
        _Pragma("unroll") for (int a = 0; a < 5; ++a) W[a] = IN[IDX(i, jj[a])] * rS[a];         \
        const double q0 = star_theta(P, v0, dt, dxtheta, invdxtheta, W[0], W[1], W[2], W[3]);    \
        const double q1 = star_theta(P, v1, dt, dxtheta, invdxtheta, W[1], W[2], W[3], W[4]);    \
        double varq = dxrad * q0 * rho0 * v0;                                                    \
        varq -= dxrad * q1 * rho1 * v1;                                                          \
        OUT[IDX(i, jout)] = IN[IDX(i, j)] + varq * invsurf;                                      \
    }
    THETA_UPDATE(in.rmp, out.rmp);
    THETA_UPDATE(in.rmm, out.rmm);
    THETA_UPDATE(in.lp, out.lp);
    THETA_UPDATE(in.lm, out.lm);
    if (P.adiabatic)
        THETA_UPDATE(in.e, out.e);
    {
        // density: Work = 1 => star state 1
        double varq = dxrad * 1.0 * rho0 * v0;
        varq -= dxrad * 1.0 * rho1 * v1;
        out.sig[IDX(i, jout)] = S[2] + varq * invsurf;
    }
#undef THETA_UPDATE
}



// compute_velocities_from_momenta (:498-535) + assure_minimum_value and the
// temperature floor/ceiling of Transport (:121-131); reads set B, writes the state.
// Wave damping of one value (damping.cpp:311-557): X <- (X - X0) exp(-dt f / tau) + X0 on rings
// whose per-ring type is non-zero (1: reference field, 2: zero / density floor).
__device__ __forceinline__ double damp_value(const Dev &P, double X, int type, double fac, double tau,
                                             double dt, const double *ref, int cell, double zero_target)
{
    if (type == 0)
        return X;
    const double exp_factor = exp(-dt * fac / tau);
    const double X0 = type == 1 ? ref[cell] : zero_target;
    return (X - X0) * exp_factor + X0;
}
// compute_velocities_from_momenta (:498-535) + assure_minimum_value and the temperature
// floor/ceiling of Transport (:121-131); reads a momenta set, writes the state.  With
// DAMP the reference/zero wave damping of the final boundary call (damping.cpp:754-774) is
// applied to the fresh values in the same pass (the per-cell operations commute with the
// ghost exchange that sits between them in the reference).
template <bool DAMP, bool ROWU> __global__ void k_velocities(const Dev P, ThetaSet in, const double *vr_src)
{
    CELL(0, P.nr);
    const double dt = P.clk->dt;
    if (i_ == P.nr - 1) { // v_r row Nr is not transported: it keeps its post-boundary value
        double v = vr_src[IDX(P.nr, j)];
        if (DAMP)
            v = damp_value(P, v, P.dtype_vr[P.nr], P.dfac_v[P.nr], P.dtau_v[P.nr], dt, P.vrad0, IDX(P.nr, j), 0.0);
        P.vrad[IDX(P.nr, j)] = v;
    }
    const int jp = JPREV;
    const double s = in.sig[IDX(i, j)];
    double vr = 0.0;
    if (i != 0)
        vr = (in.rmp[IDX(i - 1, j)] + in.rmm[IDX(i, j)]) / (in.sig[IDX(i - 1, j)] + s);
    double va = (in.lp[IDX(i, jp)] + in.lm[IDX(i, j)]) / (in.sig[IDX(i, jp)] + s) * P.InvRmed[i] -
                P.Rmed[i] * P.omega_frame;
    double sf = s < P.sigma_floor_abs ? P.sigma_floor_abs : s;
    double e = 0.0;
    if (P.adiabatic)
        e = clamp_energy(P, in.e[IDX(i, j)], sf);
    if (DAMP) {
        const int c = IDX(i, j);
        vr = damp_value(P, vr, P.dtype_vr[i], P.dfac_v[i], P.dtau_v[i], dt, P.vrad0, c, 0.0);
        const double fs = P.dfac_s[i], ts = P.dtau_s[i];
        va = damp_value(P, va, P.dtype_va[i], fs, ts, dt, P.vazi0, c, 0.0);
        sf = damp_value(P, sf, P.dtype_sig[i], fs, ts, dt, P.sigma0, c, P.sigma_floor_abs);
        if (P.adiabatic)
            e = damp_value(P, e, P.dtype_e[i], fs, ts, dt, P.energy0, c, 0.0);
    }
    P.vrad[IDX(i, j)] = vr;
    P.vazi[IDX(i, j)] = va;
    P.sigma[IDX(i, j)] = sf;
    if (P.adiabatic)
        P.energy[IDX(i, j)] = e;
}

// ---------------------------------------------------------------------------
// Azimuthal transport + velocities + floors + wave damping in ONE kernel, marching over rings.
// As k_transport_theta_fused, but a wavefront owns a phi segment in POST-shift coordinates and
// walks THETA_ROWS rings outward: for ring i it reads the cells that the integer shift maps
// onto its segment (input index = output index - Nshift[i]), runs both passes in registers,
// and -- because ring i-1 was processed by the same lanes one iteration earlier -- forms
//   v_r(i)   = (rm+(i-1) + rm-(i)) / (Sigma(i-1) + Sigma(i))                (:515-523)
//   v_phi(i) = (L+(j-1) + L-(j)) / (Sigma(j-1) + Sigma(j)) / r - r Omega    (:526-532)
// applies the density floor / temperature range (:121-131) and the reference/zero wave damping
// of the final boundary call, and stores the new state.  The transported momenta never go to
// memory: the sweep reads 6 (7) grids and writes 3 (4) instead of 11 + 8 (13 + 10) doubles per
// cell for k_transport_theta_fused + k_velocities.
// Validity: 4 cells at either end of a segment are lost to the two passes, one more on the
// left to the L+(j-1) neighbour.
#define THETA_ROWS 8
#define THETA_LO 6 /* even, so that a lane's two cells are both final or both halo */
#define THETA_HI 4

// 0.5 * flux_limiter(a, b) (TransportEuler.cpp:306-337): the factor 2 of van Leer's 2ab/(a+b)
// and the 0.5 of the half-cell slope cancel exactly.
__device__ __forceinline__ double half_limiter(int type, double a, double b)
{
    if (type == FCPT_LIMITER_MC)
        return 0.5 * limiter(type, a, b);
    // ab > 0 ? ab / (a + b) : 0 without the compare and the two selects: max(ab, 0) is the numerator, and the
    // reciprocal is kept finite where it is not needed.  For ab > 0 the operations are those of the select form, bit
    // for bit.  Where a + b is +-0 or a denormal, v_rcp_f64 returns +-inf and the Newton step NaN; the clamp BEHIND
    // the Newton step turns that (v_min_f64 returns its non-NaN operand, as the compare-and-select it is written as)
    // and every reciprocal beyond 1e300 into 1e300 -- reached only where |a|, |b| < 1e-300, i.e. where ab has
    // underflowed to 0 or is <= 0 and the numerator is an exact 0 (round 2 clamped before the Newton step, which let a
    // NEGATIVE denormal sum through as -inf -> NaN: ADVICE round 2).
    const double ab = a * b, d = a + b;
    double x = __builtin_amdgcn_rcp(d);
    x = fma(x, fma(-d, x, 1.0), x);
    x = x < 1e300 ? x : 1e300; // v_min_f64; NaN -> 1e300
    return (ab > 0.0 ? ab : 0.0) * x; // v_max_f64
}
// fcpt_selftest_half_limiter: the device function on host-supplied operands
__global__ void k_selftest_half_limiter(int type, long long n, const double *a, const double *b, double *out)
{
    const long long k = (long long)blockIdx.x * blockDim.x + threadIdx.x;
    if (k < n)
        out[k] = half_limiter(type, a[k], b[k]);
}

// Upwind star states of one quantity on the C cells of a lane (compute_star_theta,
// TransportEuler.cpp:408-441): st = q_upwind + (dist / dxtheta) * half_limited_difference_upwind.
// MODE 0: per-cell upwind direction up[c] and distance factor d2[c]; MODE 1 / 2: the whole ring
// moves with one velocity > 0 / <= 0 (second FARGO pass), so the upwind choice is made at compile
// time and the selects disappear.
template <int C, bool PER, int MODE>
__device__ __forceinline__ void theta_star(int lim, int lsrc_l, int lsrc_r, const double (&W)[C], const bool (&up)[C],
                                           const double (&d2)[C], double d2u, double (&st)[C])
{
#define SH_PREV(x) (PER ? __shfl((x), lsrc_l, 64) : lane_prev(x))
#define SH_NEXT(x) (PER ? __shfl((x), lsrc_r, 64) : lane_next(x))
    const double wl = SH_PREV(W[C - 1]); // cell -1
    const double wr = SH_NEXT(W[0]);     // cell C
    double h[C];
#pragma unroll
    for (int c = 0; c < C; ++c) {
        const double wm = c == 0 ? wl : W[c == 0 ? 0 : c - 1];
        const double wp = c == C - 1 ? wr : W[c == C - 1 ? C - 1 : c + 1];
        h[c] = half_limiter(lim, wp - W[c], W[c] - wm);
    }
    if (MODE == 2) {
#pragma unroll
        for (int c = 0; c < C; ++c)
            st[c] = W[c] + d2u * h[c];
        return;
    }
    const double hl = SH_PREV(h[C - 1]); // slope of cell -1
#pragma unroll
    for (int c = 0; c < C; ++c) {
        const double wm = c == 0 ? wl : W[c == 0 ? 0 : c - 1];
        const double hm = c == 0 ? hl : h[c == 0 ? 0 : c - 1];
        if (MODE == 1)
            st[c] = wm + d2u * hm;
        else {
            // both candidates with the lane's own distance factor, then ONE select (two v_cndmask) instead of two
            // selects of the operands: the same value bit for bit
            const double st_up = wm + d2[c] * hm, st_dn = W[c] + d2[c] * h[c];
            st[c] = up[c] ? st_up : st_dn;
        }
    }
}

// One azimuthal pass (OneWindTheta's VanLeerTheta calls, TransportEuler.cpp:443-496,583-628) on
// the cells of a lane.  geo_dt = (Rsup-Rinf) * InvSurf * dt; V the per-cell velocity (MODE 0) or vu
// the ring velocity (MODE 1/2).  The interface mass flux F = geo_dt * v * rho* is formed once and
// shared by all quantities: Q += q*(c) F(c) - q*(c+1) F(c+1).
template <int C, bool ADI, bool PER, int MODE>
__device__ __forceinline__ void theta_pass(int lim, int lsrc_l, int lsrc_r, double geo_dt, double dxtheta, double invdx,
                                           double dt, const double (&V)[C], double vu, double (&S)[C], double (&Q)[4][C],
                                           double (&E)[C])
{
    bool up[C];
    double d2[C], d2u = 0.0;
    if (MODE == 0) {
#pragma unroll
        for (int c = 0; c < C; ++c) {
            const double ksi = V[c] * dt;
            up[c] = ksi > 0.0;
            d2[c] = (up[c] ? (dxtheta - ksi) : -(dxtheta + ksi)) * invdx;
        }
    } else {
        const double ksi = vu * dt;
#pragma unroll
        for (int c = 0; c < C; ++c) {
            up[c] = MODE == 1;
            d2[c] = 0.0;
        }
        d2u = (MODE == 1 ? (dxtheta - ksi) : -(dxtheta + ksi)) * invdx;
    }
    double rho[C], F[C + 1], rS[C];
    theta_star<C, PER, MODE>(lim, lsrc_l, lsrc_r, S, up, d2, d2u, rho);
    const double gvu = geo_dt * vu;
#pragma unroll
    for (int c = 0; c < C; ++c) {
        F[c] = (MODE == 0 ? geo_dt * V[c] : gvu) * rho[c];
        rS[c] = FAST_RCP_TR(S[c]);
    }
    F[C] = SH_NEXT(F[0]);
    auto advect = [&](double (&X)[C]) {
        double Wq[C], qs[C], fl[C + 1];
#pragma unroll
        for (int c = 0; c < C; ++c)
            Wq[c] = X[c] * rS[c];
        theta_star<C, PER, MODE>(lim, lsrc_l, lsrc_r, Wq, up, d2, d2u, qs);
#pragma unroll
        for (int c = 0; c < C; ++c)
            fl[c] = qs[c] * F[c];
        fl[C] = SH_NEXT(fl[0]);
#pragma unroll
        for (int c = 0; c < C; ++c)
            X[c] += fl[c] - fl[c + 1];
    };
    advect(Q[0]);
    advect(Q[1]);
    advect(Q[2]);
    advect(Q[3]);
    if (ADI)
        advect(E);
#pragma unroll
    for (int c = 0; c < C; ++c)
        S[c] += F[c] - F[c + 1];
}

template <int C, bool ADI, bool DAMP, bool PER>
__device__ __forceinline__ void transport_theta_march_block(const Dev &P, const double *va_pre, const double *vr_pre, const ThetaSet &in,
                                                            int tiles, int rows, int advance_clock, int vb, int nvb)
{
    constexpr int periodic = PER ? 1 : 0;
    // va_pre / vr_pre: the pre-transport (post-source, post-boundary) velocities; the new state goes to P's grids
    const int lane = threadIdx.x & 63;
    const int wave = __builtin_amdgcn_readfirstlane(xcd_block(vb, nvb) * (blockDim.x >> 6) + (threadIdx.x >> 6));
    const int chunk = wave / tiles;
    const int r0 = chunk * rows;
    const int nr = P.nr;
    if (r0 >= nr)
        return;
    const int r1 = r0 + rows < nr ? r0 + rows : nr;
    const int tile = wave - chunk * tiles;
    const int nphi = P.nphi;
    const int nl = periodic ? nphi / C : 64;
    const int stride = periodic ? nphi : 64 * C - (THETA_LO + THETA_HI);
    const int a = periodic ? 0 : tile * stride - THETA_LO; // first (output) cell of the segment
    const bool act = lane < nl;
    const int ln = act ? lane : 0;
    int lsrc_l = ln - 1, lsrc_r = ln + 1;
    if (periodic) {
        lsrc_l = lsrc_l < 0 ? nl - 1 : lsrc_l;
        lsrc_r = lsrc_r >= nl ? 0 : lsrc_r;
    } else {
        lsrc_l = lsrc_l < 0 ? 0 : lsrc_l;
        lsrc_r = lsrc_r > 63 ? 63 : lsrc_r;
    }
    const double dt = P.clk->dt;
    const int lim = P.limiter;
    auto wrap = [nphi](int j) { return j < 0 ? j + nphi : (j >= nphi ? j - nphi : j); };

    int jout[C];
    bool valid[C];
#pragma unroll
    for (int c = 0; c < C; ++c) {
        const int pos = lane * C + c;
        jout[c] = wrap(a + ln * C + c);
        valid[c] = act && (periodic || (pos >= THETA_LO && pos < 64 * C - THETA_HI && a + pos < nphi));
    }
    // 16-byte stores when, for every lane, the two cells are adjacent in memory and final together
    const bool pair_out =
        C == 2 && !PER && __builtin_amdgcn_ballot_w64(jout[C - 1] != jout[0] + 1 || valid[0] != valid[C - 1]) == 0;
    double rmp_prev[C], S_prev[C]; // transported rm+ and Sigma of ring i-1 at the same output cells
#pragma unroll
    for (int c = 0; c < C; ++c)
        rmp_prev[c] = S_prev[c] = 0.0;

    const int i0 = r0 > 0 ? r0 - 1 : 0;
    for (int i = i0; i < r1; ++i) {
        const int row = i * nphi;
        const double mean = P.vmean_c[i];
        const double vconst = P.vconst_c[i];
        const double vadd = P.fast_transport ? 0.0 : vconst;
        double S[C], Q[4][C], E[C], V[C];
        {
            int ns = P.nshift_c[i] % nphi;
            ns = ns < 0 ? ns + nphi : ns;
            int gin[C];
#pragma unroll
            for (int c = 0; c < C; ++c)
                gin[c] = row + wrap(jout[c] - ns); // the cell that AdvectSHIFT moves onto jout
            // 2 cells per lane: one 16-byte load per grid unless the ring seam falls inside a lane's pair
            const bool pairs = C == 2 && !PER && __builtin_amdgcn_ballot_w64(gin[C - 1] != gin[0] + 1) == 0;
            if (pairs) {
                const D2 s2 = LD2(in.sig + gin[0]), a2 = LD2(in.rmp + gin[0]), b2 = LD2(in.rmm + gin[0]);
                const D2 c2 = LD2(in.lp + gin[0]), d2 = LD2(in.lm + gin[0]), v2 = LD2(va_pre + gin[0]);
                D2 e2 = {0.0, 0.0};
                if (ADI)
                    e2 = LD2(in.e + gin[0]);
                S[0] = s2.x, S[C - 1] = s2.y;
                Q[0][0] = a2.x, Q[0][C - 1] = a2.y;
                Q[1][0] = b2.x, Q[1][C - 1] = b2.y;
                Q[2][0] = c2.x, Q[2][C - 1] = c2.y;
                Q[3][0] = d2.x, Q[3][C - 1] = d2.y;
                E[0] = e2.x, E[C - 1] = e2.y;
                V[0] = vadd + (v2.x - mean), V[C - 1] = vadd + (v2.y - mean);
            } else {
#pragma unroll
                for (int c = 0; c < C; ++c) {
                    const int g = gin[c];
                    S[c] = in.sig[g];
                    Q[0][c] = in.rmp[g];
                    Q[1][c] = in.rmm[g];
                    Q[2][c] = in.lp[g];
                    Q[3][c] = in.lm[g];
                    E[c] = ADI ? in.e[g] : 0.0;
                    V[c] = vadd + (va_pre[g] - mean);
                }
            }
        }
        const double dxtheta = P.g_dxtheta[i];
        const double invdx = P.g_inv_dxtheta[i];
        const double geo_dt = P.g_dr_invsurf[i] * dt;
#ifndef EXP_THETA_NOCOMP
        theta_pass<C, ADI, PER, 0>(lim, lsrc_l, lsrc_r, geo_dt, dxtheta, invdx, dt, V, 0.0, S, Q, E);
#else
#pragma unroll
        for (int c = 0; c < C; ++c) { Q[1][c] += Q[0][c] * 1e-9; Q[3][c] += (Q[2][c] + V[c]) * 1e-9; }
#endif
#ifndef EXP_THETA_NOCOMP
        if (P.fast_transport) {
            if (vconst * dt > 0.0)
                theta_pass<C, ADI, PER, 1>(lim, lsrc_l, lsrc_r, geo_dt, dxtheta, invdx, dt, V, vconst, S, Q, E);
            else
                theta_pass<C, ADI, PER, 2>(lim, lsrc_l, lsrc_r, geo_dt, dxtheta, invdx, dt, V, vconst, S, Q, E);
        }
#endif
        // compute_velocities_from_momenta + floors + damping for ring i (rings < r0 only prime rmp/S)
        if (i >= r0) {
            const double lp_l = SH_PREV(Q[2][C - 1]); // L+ and Sigma of cell j-1
            const double s_l = SH_PREV(S[C - 1]);
            const double fs = DAMP ? P.dfac_s[i] : 0.0, ts = DAMP ? P.dtau_s[i] : 1.0;
            const int tvr = DAMP ? P.dtype_vr[i] : 0, tva = DAMP ? P.dtype_va[i] : 0;
            const int tsg = DAMP ? P.dtype_sig[i] : 0, ten = DAMP ? P.dtype_e[i] : 0;
            const double fv = DAMP ? P.dfac_v[i] : 0.0, tv = DAMP ? P.dtau_v[i] : 1.0;
            const double invr = P.InvRmed[i], romega = P.g_r_omega[i];
            double o_vr[C], o_va[C], o_s[C], o_e[C];
#pragma unroll
            for (int c = 0; c < C; ++c) {
                const double lpm = c == 0 ? lp_l : Q[2][c == 0 ? 0 : c - 1];
                const double sm = c == 0 ? s_l : S[c == 0 ? 0 : c - 1];
                double vr = 0.0;
                if (i != 0)
                    vr = (rmp_prev[c] + Q[1][c]) * fast_rcp(S_prev[c] + S[c]);
                double va = (lpm + Q[3][c]) * FAST_RCP_TR(sm + S[c]) * invr - romega;
                double sf = S[c] < P.sigma_floor_abs ? P.sigma_floor_abs : S[c];
                double e = ADI ? clamp_energy(P, E[c], sf) : 0.0;
                const int g = row + jout[c];
                if (DAMP) {
                    vr = damp_value(P, vr, tvr, fv, tv, dt, P.vrad0, g, 0.0);
                    va = damp_value(P, va, tva, fs, ts, dt, P.vazi0, g, 0.0);
                    sf = damp_value(P, sf, tsg, fs, ts, dt, P.sigma0, g, P.sigma_floor_abs);
                    if (ADI)
                        e = damp_value(P, e, ten, fs, ts, dt, P.energy0, g, 0.0);
                }
                o_vr[c] = vr, o_va[c] = va, o_s[c] = sf, o_e[c] = e;
            }
            if (ADI && P.cfl_thermal) {
                const ThermalRing tr = thermal_ring(P, i);
#pragma unroll
                for (int c = 0; c < C; ++c)
                    if (valid[c]) {
                        const int g = row + jout[c];
                        P.cfl_thermal[g] = cfl_thermal_term(P, tr, o_s[c], o_e[c], P.qplus[g], P.qminus[g]);
                    }
            }
            if (pair_out) { // both cells of the lane are final and adjacent in memory
                if (valid[0]) {
                    const int g = row + jout[0];
                    ST2(P.vrad + g, (D2{o_vr[0], o_vr[C - 1]}));
                    ST2(P.vazi + g, (D2{o_va[0], o_va[C - 1]}));
                    ST2(P.sigma + g, (D2{o_s[0], o_s[C - 1]}));
                    if (ADI)
                        ST2(P.energy + g, (D2{o_e[0], o_e[C - 1]}));
                }
            } else {
#pragma unroll
                for (int c = 0; c < C; ++c)
                    if (valid[c]) {
                        const int g = row + jout[c];
                        P.vrad[g] = o_vr[c];
                        P.vazi[g] = o_va[c];
                        P.sigma[g] = o_s[c];
                        if (ADI)
                            P.energy[g] = o_e[c];
                    }
            }
            if (i == nr - 1) { // v_r row Nr is not transported: it keeps its post-boundary value
#pragma unroll
                for (int c = 0; c < C; ++c)
                    if (valid[c]) {
                        double v = vr_pre[nr * nphi + jout[c]];
                        if (DAMP)
                            v = damp_value(P, v, P.dtype_vr[nr], P.dfac_v[nr], P.dtau_v[nr], dt, P.vrad0,
                                           nr * nphi + jout[c], 0.0);
                        P.vrad[nr * nphi + jout[c]] = v;
                    }
            }
        }
#pragma unroll
        for (int c = 0; c < C; ++c) {
            rmp_prev[c] = Q[0][c];
            S_prev[c] = S[c];
        }
    }
    if (wave == 0 && lane == 0 && advance_clock) { // sim::time += dt; N_hydro_iter++ (simulation.cpp:226-227)
        clock_advance(P.clk, dt);
    }
#undef SH_PREV
#undef SH_NEXT
}

// The gated azimuthal march of the fallback transport AND the final boundary call of the step in one launch
// (fcpt_run_steps on grids whose boundary call is its own kernel: one dependent launch of ~5 us less per step).
// Workgroups [0, ntheta): the march, as k_transport_theta_march with only_if -- they return at once unless
// k_ring_mean raised the flag; workgroups behind them: boundary_column on the view B (the state after the
// transport's pointer swap), one column per thread.  Only in the rare step that falls back do the boundary
// workgroups wait -- for the stamp the last marching workgroup publishes (flag[4]: arrivals, flag[5]: stamp = this
// transport's sequence number; the marching workgroups have the lower indices and never wait: no deadlock).
template <int C, bool ADI, bool DAMP, bool PER>
__global__ void __launch_bounds__(256) k_theta_march_gated_boundary(const Dev P, const double *va_pre, const double *vr_pre, ThetaSet in,
                                                                    int tiles, int rows, int nvb, int ntheta, const Dev B)
{
    int *flag = P.shift_jump;
    const bool raised = shift_jump_raised(flag);
    if ((int)blockIdx.x < ntheta) {
        if (!raised)
            return;
        for (int vb = blockIdx.x; vb < nvb; vb += ntheta)
            transport_theta_march_block<C, ADI, DAMP, PER>(P, va_pre, vr_pre, in, tiles, rows, 0, vb, nvb);
        __syncthreads();
        if (threadIdx.x == 0) {
            __threadfence(); // the workgroup's part of the new state is visible device-wide before it reports in
            if (__hip_atomic_fetch_add(flag + 4, 1, __ATOMIC_ACQ_REL, __HIP_MEMORY_SCOPE_AGENT) == ntheta - 1) {
                __hip_atomic_store(flag + 4, 0, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                __hip_atomic_store(flag + 5, flag[2], __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_AGENT);
            }
        }
        return;
    }
    if (raised) {
        const int seq = flag[2];
        while (__hip_atomic_load(flag + 5, __ATOMIC_ACQUIRE, __HIP_MEMORY_SCOPE_AGENT) != seq)
            __builtin_amdgcn_s_sleep(2);
    }
    const int j = ((int)blockIdx.x - ntheta) * blockDim.x + threadIdx.x;
    if (j < B.nphi)
        boundary_column(B, j, 3);
}

// grid-stride wrapper, as k_transport_radial
template <int C, bool ADI, bool DAMP, bool PER>
__global__ void __launch_bounds__(256) k_transport_theta_march(const Dev P, const double *va_pre, const double *vr_pre, ThetaSet in,
                                                              int tiles, int rows, int advance_clock, const int *only_if, int nvb)
{
    if (only_if && !shift_jump_raised(only_if))
        return;
    for (int vb = blockIdx.x; vb < nvb; vb += gridDim.x)
        transport_theta_march_block<C, ADI, DAMP, PER>(P, va_pre, vr_pre, in, tiles, rows, advance_clock, vb, nvb);
}
