// Part of fcpt_kernels.hip (one translation unit, namespace fcpt): CFL reduction and the device clock.
// Not a stand-alone header: included once, in the order given there.

// ---------------------------------------------------------------------------
// cfl.cpp:185-376 condition_cfl.  k_ring_mean gives <v_phi>.  dt_cell = CFL / sqrt(sum of
// the squared inverse limits) and both sqrt and the quotient are monotone, so
// min_cells dt_cell == CFL / sqrt(max_cells sum): k_cfl_cells reduces the per-cell sums to one
// maximum per block (no atomics); k_cfl_final folds the block maxima, applies sqrt and the
// quotient once, and adds the per-ring FARGO shear limit (:207-220).
#define CFL_ROWS 8 /* on grids that fill the GPU; fewer on small ones (march_len, launch.h) */
// One thread owns a phi column and walks CFL_ROWS rings (v_r(i+1) of one ring is v_r(i) of the
// next, so every value is loaded once); per-block maxima, no atomics.
template <bool ROWU> __global__ void k_cfl_cells(const Dev P, double *part, int rows)
{
    const int j = blockIdx.x * blockDim.x + threadIdx.x;
    const int r0_ = P.first_active + (blockIdx.y * blockDim.y + threadIdx.y) * rows;
    const int r0 = ROWU ? __builtin_amdgcn_readfirstlane(r0_) : r0_;
    double s = 0.0;
    if (j < P.nphi && r0 < P.active_size) {
        const int jn = JNEXT;
        const int r1 = r0 + rows < P.active_size ? r0 + rows : P.active_size;
        const double lf = P.leapfrog ? 0.6 : 1.0;
        const double C2 = P.art_visc_factor * P.art_visc_factor;
        const double gg1 = P.gamma * (P.gamma - 1.0), inv_sqrt_gamma = 1.0 / sqrt(P.gamma);
        const double inv_limit = 1.0 / P.heating_cooling_cfl_limit;
        double vr0 = P.vrad[IDX(r0, j)];
        for (int i = r0; i < r1; ++i) {
            const double vr1 = P.vrad[IDX(i + 1, j)];
            const double inv_dxr = P.InvDiffRsup[i];         // 1 / (Rsup - Rinf)
            const double inv_dxa = P.InvRmed[i] * P.invdphi; // 1 / (Rmed dphi)
            const double inv_cell = dmax(inv_dxr, inv_dxa);  // 1 / min(dxRadial, dxAzimuthal)
            const double va = P.vazi[IDX(i, j)];
            const double van = P.vazi[IDX(i, jn)];
            const double vres = P.fast_transport ? va - P.vmean_c[i] : va;
            // isothermal: c_s and the alpha viscosity are per-ring constants (set once at init)
            double cs, nu;
            if (P.adiabatic && P.lazy_derived) { // k_adi_cs_h + k_viscosity in registers
                cs = sqrt(gg1 * P.energy[IDX(i, j)] * fast_rcp(P.sigma[IDX(i, j)]));
                const double H = cs * inv_sqrt_gamma * P.g_inv_omk[i];
                nu = P.alpha_viscosity ? P.alpha * H * cs : P.nu_const;
            } else {
                cs = P.adiabatic ? P.soundspeed[IDX(i, j)] : P.cs_ring[i];
                nu = P.adiabatic ? P.viscosity[IDX(i, j)] : (P.alpha_viscosity ? P.nu_ring[i] : P.nu_const);
            }
            const double invdt1 = cs * inv_cell;
            const double invdt2 = vr0 * inv_dxr;
            const double invdt3 = vres * inv_dxa;
            double invdt4;
            if (P.art_visc == FCPT_ARTVISC_SN) {
                double dvRadial = vr1 - vr0;
                double dvAzimuthal = van - va;
                dvRadial = dvRadial > 0.0 ? 0.0 : -dvRadial;
                dvAzimuthal = dvAzimuthal > 0.0 ? 0.0 : -dvAzimuthal;
                invdt4 = 4.0 * C2 * dmax(dvRadial * inv_dxr, dvAzimuthal * inv_dxa) * lf;
            } else { // the TW formula is also used for ArtificialViscosity: None (cfl.cpp:292)
                const double eps_rr = (vr1 - vr0) * P.InvDiffRsup[i];
                const double eps_pp = P.InvRmed[i] * ((van - va) * P.invdphi + 0.5 * (vr1 + vr0));
                const double mdiv_V = -dmin(eps_rr + eps_pp, 0.0);
                invdt4 = 4.0 * C2 * mdiv_V * lf;
            }
            const double invdt5 = 4.0 * nu * (inv_cell * inv_cell) * lf;
            double invdt6 = 0.0;
            if (P.adiabatic) {
                if (P.lazy_derived)
                    invdt6 = inv_limit * fabs((P.qplus[IDX(i, j)] - P.qminus[IDX(i, j)]) * fast_rcp(P.energy[IDX(i, j)])) * lf;
                else
                    invdt6 = inv_limit * fabs((P.qplus[IDX(i, j)] - P.qminus[IDX(i, j)]) / P.energy[IDX(i, j)]) * lf;
            }
            s = dmax(s, invdt1 * invdt1 + invdt2 * invdt2 + invdt3 * invdt3 + invdt4 * invdt4 + invdt5 * invdt5 +
                            invdt6 * invdt6);
            if (P.stabilize == 2) { // cfl.cpp:331-351: dt_cell = min(dt_cell, -CFL / c) == CFL / sqrt(max(sum, c^2)), c < 0
                const double c = dmin(P.cfac_phi[IDX(i, j)], P.cfac_r[IDX(i, j)]);
                if (c < 0.0)
                    s = dmax(s, c * c);
            }
            vr0 = vr1;
        }
    }
    for (int off = 32; off > 0; off >>= 1)
        s = dmax(s, __shfl_down(s, off, 64));
    __shared__ double s_w[4];
    const int tid = threadIdx.y * blockDim.x + threadIdx.x;
    if ((tid & 63) == 0)
        s_w[tid >> 6] = s;
    __syncthreads();
    if (tid == 0)
        part[blockIdx.y * gridDim.x + blockIdx.x] = dmax(dmax(s_w[0], s_w[1]), dmax(s_w[2], s_w[3]));
}
__device__ __forceinline__ void cfl_fold(const Dev &P, const double *part, int nparts, int apply_policy);
__device__ __forceinline__ bool cfl_last_workgroup(int *tickets, int b, int nb);
// Ring mean and per-cell limits in one pass: a block owns a ring, keeps its v_phi in registers
// (CFL_MAXP pairs per thread), sums them (<v_phi>, cfl.cpp:196-205), then evaluates the cells of
// the ring against that mean (:222-330) -- v_phi is read once instead of once by k_ring_mean and
// once by k_cfl_cells.  One partial maximum per ring.
#define CFL_MAXP 8 // pairs of cells per thread (Nphi <= 4096); 16 for rings up to 8192 cells
// The launch covers rings [r1, r1+n1) and [r2, r2+n2): all of them in one go, or (slabs with neighbours) the
// interior while the ghost rings are on the wire and the rings next to them after the unpack (fcpt_cfl_begin).
// finalize: 0 = partial maxima only (the interior rings ahead of the ghost exchange), 1 + apply_policy = the last
// workgroup also folds them (cfl_fold over all nr rings)
// NT threads per ring, MAXP pairs of cells per thread (NT x MAXP x 2 >= Nphi)
// (P BY VALUE, as a kernel's own argument: with `const Dev &P` the scheduler hoists the loads of all eight cell pairs of
//  the 256-thread form -- 224 instead of 100 registers, two instead of four wavefronts per SIMD -- and the ideal-EOS
//  launch takes 97 instead of 74 us: occupancy beats loads in flight per thread here)
// HOIST (the 1024-thread forms, two cell pairs per thread): everything else a cell pair needs -- v_r of rings i and i+1,
// v_phi of cell j+2, and for the ideal EOS e, Sigma, Q+ - Q- -- is loaded TOGETHER with v_phi, ahead of the ring
// sum and its barrier, which none of it depends on: one memory round trip per workgroup instead of two.
struct CflPair {
    D2 r0, r1, e2, s2, qp, qm, th;
    double van1;
};
template <bool ADI> __device__ __forceinline__ CflPair cfl_load_pair(const Dev &P, size_t row, int nphi, int j)
{
    CflPair d;
    d.r0 = *(const D2 *)(P.vrad + row + j), d.r1 = *(const D2 *)(P.vrad + row + nphi + j);
    d.van1 = P.vazi[row + (j + 2 >= nphi ? 0 : j + 2)]; // v_phi of cell j+2
    d.e2 = D2{0.0, 0.0}, d.s2 = D2{1.0, 1.0}, d.qp = D2{0.0, 0.0}, d.qm = D2{0.0, 0.0}, d.th = D2{0.0, 0.0};
    const bool thermal = ADI && P.cfl_thermal_on != 0; // invdt1^2 + invdt5^2 + invdt6^2 left by the transport
    if (thermal) {
        d.th = *(const D2 *)(P.cfl_thermal + row + j);
    } else if (ADI) {
        d.e2 = *(const D2 *)(P.energy + row + j);
        d.s2 = *(const D2 *)(P.sigma + row + j);
        if (P.qdiff_on) { // Q+ - Q- as one grid, left by the source march
            d.qp = *(const D2 *)(P.qdiff + row + j);
        } else {
            d.qp = *(const D2 *)(P.qplus + row + j);
            d.qm = *(const D2 *)(P.qminus + row + j);
        }
    }
    return d;
}
template <bool ADI, int MAXP, int NT, bool HOIST = false> __device__ __forceinline__ void cfl_ring_block(const Dev P, double *part, const int i)
{
    constexpr int NW = NT / 64;
    const int nphi = P.nphi, npair = nphi >> 1;
    const int t = threadIdx.x;
    const size_t row = (size_t)i * nphi;
    const bool active = i >= P.first_active && i < P.active_size;
    D2 va[MAXP];
    CflPair pd[HOIST ? MAXP : 1];
    double acc = 0.0, acc2 = 0.0;
#pragma unroll
    for (int n = 0; n < MAXP; ++n) {
        const int p = t + n * NT;
        va[n] = D2{0.0, 0.0};
        if (p < npair)
            va[n] = *(const D2 *)(P.vazi + row + 2 * p);
        if (HOIST && active && p < npair)
            pd[n] = cfl_load_pair<ADI>(P, row, nphi, 2 * p);
    }
#pragma unroll
    for (int n = 0; n < MAXP; ++n) {
        acc += va[n].x;
        acc2 += va[n].y;
    }
    acc += acc2;
    for (int off = 32; off > 0; off >>= 1)
        acc += __shfl_down(acc, off, 64);
    __shared__ double s_w[NW], s_m[NW];
    if ((t & 63) == 0)
        s_w[t >> 6] = acc;
    __syncthreads();
    double ring_sum = (s_w[0] + s_w[1]) + (s_w[2] + s_w[3]); // (the 256-thread order; more wavefronts append pairwise)
#pragma unroll
    for (int q = 4; q < NW; q += 4)
        ring_sum += (s_w[q] + s_w[q + 1]) + (s_w[q + 2] + s_w[q + 3]);
    const double mean = ring_sum / (double)nphi;
    if (t == 0)
        P.vmean[i] = mean;
    double s = 0.0;
    if (active) {
        const double lf = P.leapfrog ? 0.6 : 1.0;
        const double C2 = P.art_visc_factor * P.art_visc_factor;
        const double inv_dxr = P.InvDiffRsup[i];         // 1 / (Rsup - Rinf)
        const double inv_rmed = P.InvRmed[i];
        const double inv_dxa = inv_rmed * P.invdphi;     // 1 / (Rmed dphi)
        const double inv_cell = dmax(inv_dxr, inv_dxa);  // 1 / min(dxRadial, dxAzimuthal)
        const double gg1 = P.gamma * (P.gamma - 1.0), inv_sqrt_gamma = 1.0 / sqrt(P.gamma);
        const double inv_limit = 1.0 / P.heating_cooling_cfl_limit;
        const double inv_omk = ADI ? P.g_inv_omk[i] : 0.0;
        const double cs_iso = ADI ? 0.0 : P.cs_ring[i];
        const double nu_iso = ADI ? 0.0 : (P.alpha_viscosity ? P.nu_ring[i] : P.nu_const);
        const double sub = P.fast_transport ? mean : 0.0;
#pragma unroll
        for (int n = 0; n < MAXP; ++n) {
            const int p = t + n * NT;
            if (p < npair) {
                const int j = 2 * p;
                const CflPair d = HOIST ? pd[HOIST ? n : 0] : cfl_load_pair<ADI>(P, row, nphi, j);
                const D2 r0 = d.r0, r1 = d.r1, e2 = d.e2, s2 = d.s2, qp = d.qp, qm = d.qm, th = d.th;
                const double van1 = d.van1;
                const bool thermal = ADI && P.cfl_thermal_on != 0;
#pragma unroll
                for (int c = 0; c < 2; ++c) {
                    const double vr0 = c ? r0.y : r0.x, vr1 = c ? r1.y : r1.x;
                    const double v = c ? va[n].y : va[n].x, van = c ? van1 : va[n].y;
                    double cs = cs_iso, nu = nu_iso;
                    if (ADI && !thermal) { // k_adi_cs_h + k_viscosity in registers
                        const double e = c ? e2.y : e2.x, sg = c ? s2.y : s2.x;
                        cs = sqrt(gg1 * e * fast_rcp(sg));
                        const double H = cs * inv_sqrt_gamma * inv_omk;
                        nu = P.alpha_viscosity ? P.alpha * H * cs : P.nu_const;
                    }
                    const double invdt1 = cs * inv_cell;
                    const double invdt2 = vr0 * inv_dxr;
                    const double invdt3 = (v - sub) * inv_dxa;
                    double invdt4;
                    if (P.art_visc == FCPT_ARTVISC_SN) {
                        double dvRadial = vr1 - vr0;
                        double dvAzimuthal = van - v;
                        dvRadial = dvRadial > 0.0 ? 0.0 : -dvRadial;
                        dvAzimuthal = dvAzimuthal > 0.0 ? 0.0 : -dvAzimuthal;
                        invdt4 = 4.0 * C2 * dmax(dvRadial * inv_dxr, dvAzimuthal * inv_dxa) * lf;
                    } else { // the TW formula is also used for ArtificialViscosity: None (cfl.cpp:292)
                        const double eps_rr = (vr1 - vr0) * inv_dxr;
                        const double eps_pp = inv_rmed * ((van - v) * P.invdphi + 0.5 * (vr1 + vr0));
                        const double mdiv_V = -dmin(eps_rr + eps_pp, 0.0);
                        invdt4 = 4.0 * C2 * mdiv_V * lf;
                    }
                    const double invdt5 = 4.0 * nu * (inv_cell * inv_cell) * lf;
                    double invdt6 = 0.0;
                    if (ADI && !thermal) {
                        const double e = c ? e2.y : e2.x;
                        invdt6 = inv_limit * fabs(((c ? qp.y : qp.x) - (c ? qm.y : qm.x)) * fast_rcp(e)) * lf;
                    }
                    if (thermal)
                        s = dmax(s, (c ? th.y : th.x) + invdt2 * invdt2 + invdt3 * invdt3 + invdt4 * invdt4);
                    else
                        s = dmax(s, invdt1 * invdt1 + invdt2 * invdt2 + invdt3 * invdt3 + invdt4 * invdt4 +
                                        invdt5 * invdt5 + invdt6 * invdt6);
                }
            }
        }
    }
    for (int off = 32; off > 0; off >>= 1)
        s = dmax(s, __shfl_down(s, off, 64));
    if ((t & 63) == 0)
        s_m[t >> 6] = s;
    __syncthreads();
    if (t == 0) {
        double m = dmax(dmax(s_m[0], s_m[1]), dmax(s_m[2], s_m[3]));
#pragma unroll
        for (int q = 4; q < NW; ++q)
            m = dmax(m, s_m[q]);
        part[i] = m;
    }
}
#define CFL_RINGS_ATTR __launch_bounds__(NT)
template <bool ADI, int MAXP, int NT = 256> __global__ void CFL_RINGS_ATTR k_cfl_rings(const Dev P, double *part, int r1, int n1, int r2, int finalize)
{
    const int b = xcd_block(blockIdx.x, gridDim.x); // neighbouring rings share the v_r row between them: same L2
    const int i = b < n1 ? r1 + b : r2 + (b - n1);
    cfl_ring_block<ADI, MAXP, NT, (NT >= 512)>(P, part, i);
    if (finalize && cfl_last_workgroup(P.cfl_tickets, blockIdx.x, gridDim.x))
        cfl_fold(P, part, P.nr, finalize - 1);
}
// condition_cfl of step n + 1 and the final boundary call of step n (boundary_conditions.cpp:65-114 without its damping,
// which the transport kernel applied) in ONE launch -- the device-resident loop of fcpt_run_steps, where the two are
// neighbours in the stream.  The boundary kernel is a chain of dependent loads of 5-7 us on its own; here its nbc
// workgroups (one column per thread) are the first of the grid and run under the CFL terms of the rings that read
// nothing it writes.  Four rings do: 0 and nr-1 (their <v_phi> enters the shear limit) and 1 and nr-2 (v_r rows 1 and
// nr-1).  Their workgroups are the last of the grid and wait until the boundary workgroups -- running or done by then,
// never waiting for anything themselves -- have published this step's stamp (sequence number as for the shift-jump
// flag: nothing is ever reset by a kernel).  flag[1]: arrivals of the boundary workgroups, flag[3]: the stamp.
template <bool ADI, int MAXP, int NT = 256> __global__ void CFL_RINGS_ATTR k_cfl_rings_bc(const Dev P, double *part, int nbc)
{
    const int b = blockIdx.x, t = threadIdx.x;
    int *flag = P.shift_jump;
    const int seq = SHIFT_SEQ(P.clk);
    if (b < nbc) {
        const int j = b * NT + t;
        if (j < P.nphi)
            boundary_column(P, j, 3);
        __syncthreads();
        if (t == 0) {
            __threadfence(); // the workgroup's ghost rings are visible device-wide before it reports in
            if (__hip_atomic_fetch_add(flag + 1, 1, __ATOMIC_ACQ_REL, __HIP_MEMORY_SCOPE_AGENT) == nbc - 1) {
                __hip_atomic_store(flag + 1, 0, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                __hip_atomic_store(flag + 3, seq, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_AGENT);
            }
        }
        return;
    }
    const int nr = P.nr, nfree = nr - 4, q = b - nbc;
    int i;
    if (q < nfree) {
        i = 2 + xcd_block(q, nfree);
    } else {
        const int d = q - nfree;
        i = d == 0 ? 0 : (d == 1 ? 1 : (d == 2 ? nr - 2 : nr - 1));
        while (__hip_atomic_load(flag + 3, __ATOMIC_ACQUIRE, __HIP_MEMORY_SCOPE_AGENT) != seq)
            __builtin_amdgcn_s_sleep(2);
    }
    cfl_ring_block<ADI, MAXP, NT, (NT >= 512)>(P, part, i);
}
// The last step of the reduction by one workgroup: fold the partial maxima, add the FARGO shear limit, leave the
// result in the device clock (and apply the CalculateTimeStep policy for device-resident loops).
// (every thread of the workgroup returns the same value: the maxima and minima fold exactly, in any order)
__device__ __forceinline__ double cfl_fold_value(const Dev &P, const double *part, int nparts)
{
    double smax = 0.0;
    for (int n = threadIdx.x; n < nparts; n += blockDim.x)
        smax = dmax(smax, part[n]);
    // FARGO shear limit, rings 0|1 (:207-208) and the active rings (:213-220): the smallest of cfl dphi / denom over the
    // ring pairs IS cfl dphi / (the largest denom) -- a correctly rounded quotient is monotone in its divisor -- so the
    // pairs are folded by their denominators and divided once (an IEEE division per pair was most of this function's
    // vector work, which every workgroup of the marching source kernel repeats when the fold runs in its prologue)
    double dmx = 0.0;
    for (int n = threadIdx.x; n < P.active_size; n += blockDim.x) {
        if (n == 0 || n >= P.first_active) {
            const double denom = fabs(P.vmean[n] * P.InvRmed[n] - P.vmean[n + 1] * P.InvRmed[n + 1]) + 1.0e-100;
            dmx = dmax(dmx, denom);
        }
    }
    for (int off = 32; off > 0; off >>= 1) {
        smax = dmax(smax, __shfl_down(smax, off, 64));
        dmx = dmax(dmx, __shfl_down(dmx, off, 64));
    }
    __shared__ double s_s[16], s_d[16];
    const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
    if (lane == 0) {
        s_s[wave] = smax;
        s_d[wave] = dmx;
    }
    __syncthreads();
    smax = s_s[0], dmx = s_d[0];
    for (int w = 1; w < (int)(blockDim.x >> 6); ++w) {
        smax = dmax(smax, s_s[w]);
        dmx = dmax(dmx, s_d[w]);
    }
    double dt = dmx > 0.0 ? P.cfl * P.dphi / dmx : 1.0e300; // (no pair at all: no limit)
    if (nparts > 0)
        dt = dmin(dt, P.cfl / sqrt(smax));
    return dt;
}
__device__ __forceinline__ void cfl_fold(const Dev &P, const double *part, int nparts, int apply_policy)
{
    const double dt = cfl_fold_value(P, part, nparts);
    if (threadIdx.x == 0) {
        P.clk->cfl_bits = (unsigned long long)__double_as_longlong(dt);
        if (P.cfl_export)
            *P.cfl_export = dt;
        if (apply_policy) { // sim::CalculateTimeStep (simulation.cpp:100-118) for single-slab device loops
            const double a = P.cfl_max_var * P.clk->last_dt;
            const double rv = dt < a ? dt : a;
            P.clk->cfl_dt = rv;
            P.clk->last_dt = rv;
            P.clk->dt = rv;
        }
    }
}
// The same inside the marching source kernel (fcpt_run_steps on one slab: the kernel is the next launch behind the ring
// kernel of the CFL reduction): every workgroup folds the nr partial maxima and the shear limit for itself and takes
// the step length from its own result -- 48 KB of L2 reads per workgroup instead of a 5 us launch of one workgroup in
// front of the kernel.  Workgroup 0 leaves the result in the device clock for the kernels behind this one.  last_dt is
// an INPUT of the policy that other workgroups may still have to read: it is completed by clock_advance (the
// transport of the same step).
__device__ __forceinline__ double cfl_fold_in_step(const Dev &P)
{
    const double dt = cfl_fold_value(P, P.cfl_part, P.nr);
    const double a = P.cfl_max_var * P.clk->last_dt;
    const double rv = dt < a ? dt : a;
    if (blockIdx.x == 0 && threadIdx.x == 0) {
        P.clk->cfl_bits = (unsigned long long)__double_as_longlong(dt);
        if (P.cfl_export)
            *P.cfl_export = dt;
        P.clk->cfl_dt = rv;
        P.clk->dt = rv;
        P.clk->last_dt_pending = 1;
    }
    return rv;
}
__global__ void __launch_bounds__(1024) k_cfl_final(const Dev P, const double *part, int nparts, int apply_policy)
{
    cfl_fold(P, part, nparts, apply_policy);
}
// "Last workgroup folds": every workgroup of k_cfl_rings takes a ticket after its partial maximum is in memory, the
// one that draws the last ticket runs cfl_fold -- no separate launch for the final fold.  Two levels of tickets
// (CFL_TICKET_LANES counters, then one) keep the same-address atomics, ~10 ns each on this GPU, off the critical path.
#define CFL_TICKET_LANES 16
__device__ __forceinline__ bool cfl_last_workgroup(int *tickets, int b, int nb)
{
    __shared__ int s_last;
    if (threadIdx.x == 0) {
        const int lane = b % CFL_TICKET_LANES;
        const int in_lane = nb / CFL_TICKET_LANES + (lane < nb % CFL_TICKET_LANES ? 1 : 0);
        const int lanes_used = nb < CFL_TICKET_LANES ? nb : CFL_TICKET_LANES;
        int last = 0;
        // release: this workgroup's partial result (written by this thread) is visible before the ticket
        if (__hip_atomic_fetch_add(tickets + 1 + lane, 1, __ATOMIC_ACQ_REL, __HIP_MEMORY_SCOPE_AGENT) == in_lane - 1) {
            tickets[1 + lane] = 0;
            if (__hip_atomic_fetch_add(tickets, 1, __ATOMIC_ACQ_REL, __HIP_MEMORY_SCOPE_AGENT) == lanes_used - 1) {
                tickets[0] = 0; // ready for the next launch (stream order)
                last = 1;
            }
        }
        s_last = last;
    }
    __syncthreads();
    const bool last = s_last != 0;
    if (last)
        __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "agent"); // the other workgroups' partials, not this CU's cache
    return last;
}

// ---------------------------------------------------------------------------
// clock kernels (single thread)
__global__ void k_clock_set_dt(DevClock *clk, double dt) { clk->dt = dt; }
// leapfrog sub-steps: mode 0: step <- dt (host value), mode 1: step <- clk->dt (device value),
// mode 2: step <- the saved one; then clk->dt = factor * step.  The full step is parked in cfl_dt.
__global__ void k_clock_scale_dt(DevClock *clk, int mode, double dt, double factor)
{
    if (mode == 0)
        clk->cfl_dt = dt;
    else if (mode == 1)
        clk->cfl_dt = clk->dt;
    clk->dt = mode == 2 && factor == 1.0 ? clk->cfl_dt : clk->cfl_dt * factor;
}
__global__ void k_clock_advance(DevClock *clk)
{
    clock_advance(clk, clk->dt);
}
__global__ void k_clock_export_cfl(const DevClock *clk, double *out)
{
    *out = __longlong_as_double((long long)clk->cfl_bits);
}
__global__ void k_clock_policy_ptr(DevClock *clk, double cfl_max_var, const double *cfl_global)
{
    const double cfl_dt = *cfl_global;
    const double a = cfl_max_var * clk->last_dt;
    const double rv = cfl_dt < a ? cfl_dt : a;
    clk->cfl_dt = rv;
    clk->last_dt = rv;
    clk->dt = rv;
}
// sim::CalculateTimeStep (simulation.cpp:100-118): rv = min(CFLmaxVar*last_dt, cfl_dt)
__global__ void k_clock_policy(DevClock *clk, double cfl_max_var, int use_device_cfl, double cfl_global)
{
    const double cfl_dt = use_device_cfl ? __longlong_as_double((long long)clk->cfl_bits) : cfl_global;
    const double a = cfl_max_var * clk->last_dt;
    const double rv = cfl_dt < a ? cfl_dt : a;
    clk->cfl_dt = rv;
    clk->last_dt = rv;
    clk->dt = rv;
}
