// Part of fcpt_kernels.hip (one translation unit, namespace fcpt): disk-on-body force, boundary conditions, wave damping.
// Not a stand-alone header: included once, in the order given there.

// ---------------------------------------------------------------------------
// ComputeDiskOnPlanetAccel (Force.cpp:23-122): specific force of the slab's gas on an object.
// A thread owns a phi column over DOB_ROWS active rings; block sums in a fixed order into
// part[block][4] = {inner a_x, inner a_y, outer a_x, outer a_y}, folded by k_disk_on_body_final
// (two fixed-order stages: the result is deterministic, unlike an atomic accumulation).
#define DOB_ROWS 8
__global__ void __launch_bounds__(256) k_disk_on_body(const Dev P, double x, double y, double r_object,
                                                     double smoothing_fixed, double r_sm, double *part)
{
    const int j = blockIdx.x * blockDim.x + threadIdx.x;
    const int r0 = P.first_active + blockIdx.y * DOB_ROWS;
    const int r1 = r0 + DOB_ROWS < P.active_size ? r0 + DOB_ROWS : P.active_size;
    double a[4] = {0.0, 0.0, 0.0, 0.0};
    if (j < P.nphi) {
        const double cj = P.cosphi[j], sj = P.sinphi[j];
        for (int i = r0; i < r1; ++i) {
            const double rm = P.Rmed[i];
            double smooth = smoothing_fixed;
            if (smoothing_fixed < 0.0) { // compute_smoothing_scaleheight (Force.cpp:124-131)
                double H;
                if (!P.adiabatic) {
                    H = P.cs_ring[i] * P.g_inv_omk[i];
                } else if (P.lazy_derived) {
                    const double cs = sqrt(P.gamma * (P.gamma - 1.0) * P.energy[IDX(i, j)] / P.sigma[IDX(i, j)]);
                    H = cs / (sqrt(P.gamma)) * P.g_inv_omk[i];
                } else {
                    H = P.scale_height[IDX(i, j)];
                }
                smooth = P.thickness_smoothing * H;
            }
            const double cellmass = P.Surf[i] * P.sigma[IDX(i, j)];
            const double dx = rm * cj - x;
            const double dy = rm * sj - y;
            const double dist_sm_2 = dx * dx + dy * dy + smooth * smooth;
            const double dist_sm = sqrt(dist_sm_2);
            const double inv_dist_sm_3 = 1.0 / (dist_sm_2 * dist_sm);
            double klahr = 1.0;
            if (r_sm > 0.0 && dist_sm < r_sm) {
                const double q = dist_sm / r_sm;
                klahr = -(3.0 * ((q * q) * (q * q)) - 4.0 * (q * q * q));
            }
            const double fx = P.G * cellmass * dx * inv_dist_sm_3 * klahr;
            const double fy = P.G * cellmass * dy * inv_dist_sm_3 * klahr;
            const int o = rm < r_object ? 0 : 2;
            a[o] += fx;
            a[o + 1] += fy;
        }
    }
    __shared__ double s_a[4][4];
#pragma unroll
    for (int q = 0; q < 4; ++q) {
        double v = a[q];
        for (int off = 32; off > 0; off >>= 1)
            v += __shfl_down(v, off, 64);
        if ((threadIdx.x & 63) == 0)
            s_a[threadIdx.x >> 6][q] = v;
    }
    __syncthreads();
    if (threadIdx.x < 4)
        part[(blockIdx.y * gridDim.x + blockIdx.x) * 4 + threadIdx.x] =
            (s_a[0][threadIdx.x] + s_a[1][threadIdx.x]) + (s_a[2][threadIdx.x] + s_a[3][threadIdx.x]);
}
__global__ void __launch_bounds__(256) k_disk_on_body_final(const double *part, int nblocks, double *out)
{
    __shared__ double s_a[4][4];
#pragma unroll
    for (int q = 0; q < 4; ++q) {
        double v = 0.0;
        for (int n = threadIdx.x; n < nblocks; n += blockDim.x)
            v += part[n * 4 + q];
        for (int off = 32; off > 0; off >>= 1)
            v += __shfl_down(v, off, 64);
        if ((threadIdx.x & 63) == 0)
            s_a[threadIdx.x >> 6][q] = v;
    }
    __syncthreads();
    if (threadIdx.x < 4)
        out[threadIdx.x] = (s_a[0][threadIdx.x] + s_a[1][threadIdx.x]) + (s_a[2][threadIdx.x] + s_a[3][threadIdx.x]);
}

// ---------------------------------------------------------------------------
// boundary_conditions/{zero_gradient,reference,reflecting,outflow,keplerian_*,zero_shear}.cpp
// called in the order of boundary_conditions.cpp:65-114; one thread per phi column.
// All loads first, then all stores: the ghost values only depend on active rings (or on the
// reference fields), so the ~10 memory round trips of the sequential form collapse into one.
struct BcScalar {
    bool on;
    double v;
};
__device__ __forceinline__ BcScalar bc_scalar_load(const Dev &P, const double *x, const double *x0, int type, int outer,
                                                   int j)
{
    const int Irad = P.nr - 1;
    BcScalar r = {false, 0.0};
    if ((!outer && !P.is_first) || (outer && !P.is_last))
        return r;
    if (type == FCPT_BC_ZEROGRADIENT) {
        r.on = true;
        r.v = x[IDX(outer ? Irad - 1 : 1, j)];
    } else if (type == FCPT_BC_REFERENCE) {
        r.on = true;
        r.v = x0[IDX(outer ? Irad : 0, j)];
    }
    return r;
}
// One column of the ghost rings; `sides`: bit 0 = inner edge, bit 1 = outer edge.  Also called by the marching source
// kernels for the columns of their edge chunks (the pre-transport boundary call folded into the kick).
__device__ __forceinline__ void boundary_column(const Dev &P, int j, int sides)
{
    const int Irad = P.nr - 1;
    const int Iv = P.nr; // max_radial of the vector grid
    // ---- loads ---------------------------------------------------------------------------
    BcScalar sg[2] = {{false, 0.0}, {false, 0.0}}, en[2] = {{false, 0.0}, {false, 0.0}};
    for (int outer = 0; outer < 2; ++outer) {
        if (!((sides >> outer) & 1))
            continue;
        sg[outer] = bc_scalar_load(P, P.sigma, P.sigma0, P.bc_sigma[outer], outer, j);
        en[outer] = bc_scalar_load(P, P.energy, P.energy0, P.bc_energy[outer], outer, j);
    }
    bool vr_on[2] = {false, false}, va_on[2] = {false, false};
    double vr_g0[2] = {0.0, 0.0}, vr_g1[2] = {0.0, 0.0}, va_g[2] = {0.0, 0.0};
    for (int outer = 0; outer < 2; ++outer) {
        if (!((sides >> outer) & 1))
            continue;
        const int type = P.bc_vrad[outer];
        const int g0 = outer ? Iv : 0, g1 = outer ? Iv - 1 : 1, a = outer ? Iv - 2 : 2;
        if (type == FCPT_BC_REFLECTING) { // no rank guard in the reference (reflecting.cpp:15-40)
            vr_on[outer] = true;
            vr_g0[outer] = -P.vrad[IDX(a, j)];
            vr_g1[outer] = 0.0;
            continue;
        }
        if ((!outer && !P.is_first) || (outer && !P.is_last))
            continue;
        switch (type) {
        case FCPT_BC_ZEROGRADIENT:
            vr_on[outer] = true;
            vr_g0[outer] = vr_g1[outer] = P.vrad[IDX(a, j)];
            break;
        case FCPT_BC_REFERENCE:
            vr_on[outer] = true;
            vr_g0[outer] = P.vrad0[IDX(g0, j)];
            vr_g1[outer] = P.vrad0[IDX(g1, j)];
            break;
        case FCPT_BC_OUTFLOW: {
            const double va = P.vrad[IDX(a, j)];
            const bool inflow = outer ? (va < 0.0) : (va > 0.0);
            vr_on[outer] = true;
            vr_g0[outer] = vr_g1[outer] = inflow ? 0.0 : va;
            break;
        }
        case FCPT_BC_KEPLERIAN:
            vr_on[outer] = true;
            vr_g0[outer] = P.kep_vrad[outer] * sqrt(P.G * P.Mc / P.Rmed[g0]);
            vr_g1[outer] = P.kep_vrad[outer] * sqrt(P.G * P.Mc / P.Rmed[g1]);
            break;
        default:
            break;
        }
    }
    for (int outer = 0; outer < 2; ++outer) {
        const int type = P.bc_vaz[outer];
        if (!((sides >> outer) & 1) || (!outer && !P.is_first) || (outer && !P.is_last))
            continue;
        const int row = outer ? Irad : 0, act = outer ? Irad - 1 : 1;
        const double r = P.Rmed[row];
        switch (type) {
        case FCPT_BC_ZEROGRADIENT:
            va_on[outer] = true;
            va_g[outer] = P.vazi[IDX(act, j)];
            break;
        case FCPT_BC_REFERENCE:
            va_on[outer] = true;
            va_g[outer] = P.vazi0[IDX(row, j)];
            break;
        case FCPT_BC_KEPLERIAN:
            va_on[outer] = true;
            va_g[outer] = P.kep_vaz[outer] * sqrt(P.G * P.Mc / r) - r * P.omega_frame;
            break;
        case FCPT_BC_ZEROSHEAR:
            va_on[outer] = true;
            va_g[outer] = r * (P.vazi[IDX(act, j)] / P.Rmed[act]);
            break;
        default:
            break;
        }
    }
    // ---- stores --------------------------------------------------------------------------
    for (int outer = 0; outer < 2; ++outer) {
        const int row = outer ? Irad : 0;
        if (sg[outer].on)
            P.sigma[IDX(row, j)] = sg[outer].v;
        if (en[outer].on)
            P.energy[IDX(row, j)] = en[outer].v;
        if (vr_on[outer]) {
            P.vrad[IDX(outer ? Iv : 0, j)] = vr_g0[outer];
            P.vrad[IDX(outer ? Iv - 1 : 1, j)] = vr_g1[outer];
        }
        if (va_on[outer])
            P.vazi[IDX(row, j)] = va_g[outer];
    }
}
__global__ void k_boundary(const Dev P)
{
    const int j = blockIdx.x * blockDim.x + threadIdx.x;
    if (j >= P.nphi)
        return;
    boundary_column(P, j, 3);
}

// boundary_conditions/damping.cpp:311-427 (reference), :429-557 (zero), :559-700 (mean):
// one block per damped ring.
__global__ void k_damping(const Dev P, double *q, double *q0, const double *radius, int lo, int type,
                          double rlim, double redge, double tau, int is_density)
{
    const int i = lo + blockIdx.x;
    const double dt = P.clk->dt;
    __shared__ double s_part[256];
    __shared__ double s_mean;
    if (type == FCPT_DAMP_MEAN) {
        double acc = 0.0;
        for (int j = threadIdx.x; j < P.nphi; j += blockDim.x)
            acc += q[IDX(i, j)];
        s_part[threadIdx.x] = acc;
        __syncthreads();
        for (int s = blockDim.x / 2; s > 0; s >>= 1) {
            if ((int)threadIdx.x < s)
                s_part[threadIdx.x] += s_part[threadIdx.x + s];
            __syncthreads();
        }
        if (threadIdx.x == 0) {
            s_mean = s_part[0] / P.nphi;
            q0[IDX(i, 0)] = s_mean;
        }
        __syncthreads();
    }
    const double t = (radius[i] - rlim) / (redge - rlim);
    const double factor = t * t;
    const double exp_factor = exp(-dt * factor / tau);
    for (int j = threadIdx.x; j < P.nphi; j += blockDim.x) {
        const double X = q[IDX(i, j)];
        double X0;
        if (type == FCPT_DAMP_REFERENCE)
            X0 = q0[IDX(i, j)];
        else if (type == FCPT_DAMP_MEAN)
            X0 = s_mean;
        else
            X0 = is_density ? P.sigma_floor_abs : 0.0;
        q[IDX(i, j)] = (X - X0) * exp_factor + X0;
    }
}

// ---------------------------------------------------------------------------
// commbound.cpp:108-125|163-180: the 7 overlap rings of Sigma, v_r, v_phi(, e) to / from both neighbours in ONE
// launch (six to eight hipMemcpyAsync of 7 rings each cost 5 us apiece on the step's critical path).
// blockIdx.y = 2 * field + side; rings are contiguous, so this is a flat 16-byte copy.  `row0[side]`: first ring
// of the field's block; pack copies field -> buffer, unpack buffer -> field.
struct ExchangeArgs {
    double *field[4];
    double *buf[2]; // inner, outer (null: no neighbour on that side)
    int row0[2];
    int nq, nphi, unpack;
};
__global__ void __launch_bounds__(256) k_exchange_copy(const ExchangeArgs a)
{
    const int q = blockIdx.y >> 1, side = blockIdx.y & 1;
    double *buf = a.buf[side];
    if (!buf)
        return;
    const size_t l = (size_t)FCPT_OVERLAP * a.nphi;
    double *f = a.field[q] + (size_t)a.row0[side] * a.nphi;
    double *b = buf + (size_t)q * l;
    const double *src = a.unpack ? b : f;
    double *dst = a.unpack ? f : b;
    const size_t npair = l >> 1; // rings of even Nphi; the odd tail below
    for (size_t p = (size_t)blockIdx.x * blockDim.x + threadIdx.x; p < npair; p += (size_t)gridDim.x * blockDim.x)
        ST2(dst + 2 * p, LD2(src + 2 * p));
    if ((l & 1) && blockIdx.x == 0 && threadIdx.x == 0)
        dst[l - 1] = src[l - 1];
}
