// Shared by the translation units of the device side of the C ABI: the context structure and the helpers that more
// than one of them uses.  fcpt_context.hip: creation, options, transfers, initial physics, profiling;
// fcpt_step.hip: CFL, the step, the final boundary call, the run loop (+ hipGraph replay);
// fcpt_exchange.hip: ghost exchange and the RCCL entry points.
#ifndef FCPT_CTX_H
#define FCPT_CTX_H

#include <hip/hip_runtime.h>

#include <cmath>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <new>
#include <string>
#include <vector>

#include "fcpt_comm.h"
#include "fcpt_kernels.h"

using namespace fcpt; // private header of three translation units of namespace-fcpt code around one C struct

#define HIPCHK(call)                                                                        \
    do {                                                                                    \
        hipError_t e_ = (call);                                                             \
        if (e_ != hipSuccess) {                                                             \
            set_error("%s failed: %s (%s:%d)", #call, hipGetErrorString(e_), __FILE__, __LINE__); \
            return FCPT_EHIP;                                                               \
        }                                                                                   \
    } while (0)

struct fcpt_ctx {
    fcpt_desc d;
    fcpt_split s;
    HostGeometry geo;
    std::vector<double> radii;
    Dev P;
    hipStream_t stream = nullptr;
    std::vector<void *> allocs;
    double *d_cs_ring = nullptr;
    double *grid[FCPT_F_COUNT] = {};
    DampRange damp[4][2]; // [vrad, vaz, sigma, energy][inner, outer]
    bool potential_valid = false;
    DevClock *h_clk = nullptr; // pinned staging copy
    Profiler prof;
    bool profiling = false;
    bool fused_source = true;
    int src_parts = 0; // segments of ring sums left by the last k_source_march
    bool kick_energy_b = false;
    bool kick_bc_folded = false; // the last kick applied the boundary conditions that follow it
    bool ghosts_unknown = true;  // a state grid was uploaded since the last boundary call
    // the dt of the next step is the CFL policy's (set by calculate_timestep*, consumed by the step):
    // with CFL <= 0.8 that keeps it inside the FARGO shear limit
    bool policy_dt_dev = false;
    double policy_dt_host = -1.0; // the last kick left the energy in energy_b (marching source step, ideal EOS)
    bool march_source = true;
    bool stepped = false; // fcpt_step ran since the last fcpt_post
    // fcpt_run_steps (single slab): the final boundary call of the step just taken has not been launched yet -- the next
    // iteration's CFL launch carries it (k_cfl_rings_bc), or flush_deferred_boundary() does.  Never set when the function returns.
    bool bc_deferred = false;
    bool skip_q_store = false; // fcpt_run_steps: the step about to be queued is not the call's last (Dev::q_skip for its kick)
    // fcpt_run_steps (single slab): the gated azimuthal launch of the fallback transport is queued together with the final
    // boundary call of the step (one launch); never pending when a function of the ABI returns
    bool want_gated_deferred = false, gated_pending = false;
    GatedTheta gated;
    // fcpt_run_steps (single slab): the CFL fold + time-step policy of the step about to be taken has not been launched --
    // the marching source kernel queued next does it in its prologue (or enqueue_kick launches k_cfl_final first)
    bool fold_pending = false;
    bool cfl_interior = false; // fcpt_cfl_begin evaluated the interior rings of the current state
    bool damp_any = false;     // this slab holds rings of a damping zone
    double *thermal_grid = nullptr; // storage of Dev::cfl_thermal (the view's pointer is null when the option is off)
    bool thermal_valid = false;     // ... and it describes the current state (set by a marching-transport step)
    bool qdiff_valid = false;       // Dev::qdiff holds Q+ - Q- of the grids (written by the last kick's march)
    std::vector<int> ring_ref_damped; // per ring: 1 if the folded damping loads reference values there (costlier rings of the transport)
    int *sm_sched_dev = nullptr;      // storage of Dev::sm_sched
    size_t sm_sched_cap = 0;          // ... in ints
    std::vector<int> sm_sched_host;
    int *tf_sched_dev = nullptr;      // storage of Dev::tf_sched
    size_t tf_sched_cap = 0;          // ... in ints
    std::vector<int> tf_lengths;      // explicit chunk lengths in dispatch order (fcpt_set_transport_chunks; FCPT_TF_SCHEDULE at fcpt_create); empty: built-in
    std::vector<int> tf_sched_host;   // what Dev::tf_sched holds
    bool damp_foldable = false; // ... and its damping can be folded into the transport (no "mean" target, Euler)
    // fcpt_step_device_begin: the interior chunks of the transport run on `side` while the caller's stream
    // marches the chunks with the neighbours' ghost rings, packs and sends them
    hipStream_t side = nullptr;
    hipEvent_t e_fork = nullptr, e_join = nullptr;
    bool join_pending = false;
    bool pressure_valid = false;
    // leapfrog: bodies at the mid-step time (simulation.cpp:359-366)
    bool has_mid = false;
    double mx[FCPT_MAX_BODIES], my[FCPT_MAX_BODIES], mm[FCPT_MAX_BODIES], mrsm[FCPT_MAX_BODIES];
    // radial slabs over RCCL (fcpt_comm_init): communicator, packed ghost-ring buffers [send inner, send outer,
    // recv inner, recv outer], the device scalar of the MIN all-reduce, a stream for transfers that overlap the CFL
    Comm *comm = nullptr;
    double *xbuf[4] = {};
    double *d_cfl = nullptr;
    int peer_inner = -1, peer_outer = -1;
    hipStream_t comm_stream = nullptr;
    hipEvent_t e_packed = nullptr, e_received = nullptr;
    int device = 0; // HIP device the context was created on
    // fcpt_run_steps on launch-bound grids: a captured hipGraph of `graph_cycle` consecutive steps (the out-of-place
    // transport swaps grid pointers, so the launch arguments repeat with period 2), replayed while the host-side state
    // that decided the launches (the whole Dev view, the lazy-evaluation flags) is what it was at capture
    hipGraphExec_t graph_exec = nullptr;
    hipGraph_t graph = nullptr;
    hipStream_t capture_stream = nullptr;
    int graph_cycle = 0;
    bool coop_active = false; // the last fcpt_run_steps ran its steps as one cooperative kernel each
    long long graph_replays = 0; // hipGraphLaunch calls issued so far (read-only option "graph_replays")
    bool graph_failed = false;
    Dev graph_P;
    unsigned graph_flags = 0;
};


#define DOB_ROWS_HOST 8 /* = DOB_ROWS of k_disk_on_body */

namespace fcpt {

// routes this thread's launches to the context's profiler while it is recording
struct ProfScope {
    Profiler *outer;
    explicit ProfScope(fcpt_ctx *c) : outer(g_prof) { g_prof = c->profiling ? &c->prof : nullptr; }
    ~ProfScope() { g_prof = outer; }
};

template <class T> int dev_alloc(fcpt_ctx *c, T **p, size_t n)
{
    void *q = nullptr;
    hipError_t e = hipMalloc(&q, (n ? n : 1) * sizeof(T));
    if (e != hipSuccess) {
        set_error("hipMalloc(%zu bytes) failed: %s", n * sizeof(T), hipGetErrorString(e));
        return FCPT_ENOMEM;
    }
    e = hipMemset(q, 0, (n ? n : 1) * sizeof(T));
    if (e != hipSuccess) {
        set_error("hipMemset failed: %s", hipGetErrorString(e));
        return FCPT_EHIP;
    }
    c->allocs.push_back(q);
    *p = (T *)q;
    return FCPT_OK;
}

template <class T> int dev_upload_raw(fcpt_ctx *c, const T **dst, const std::vector<T> &src)
{
    T *p = nullptr;
    if (int e = dev_alloc(c, &p, src.size()))
        return e;
    if (hipMemcpy(p, src.data(), src.size() * sizeof(T), hipMemcpyHostToDevice) != hipSuccess) {
        set_error("hipMemcpy of a per-ring table failed");
        return FCPT_EHIP;
    }
    *dst = p;
    return FCPT_OK;
}

// fcpt_context.hip
void drop_graph(fcpt_ctx *c);
void join_side(fcpt_ctx *c);
int read_clock(fcpt_ctx *c, DevClock *out);
// fcpt_step.hip
void apply_boundary_view(fcpt_ctx *c, const Dev &P, bool final, bool damping_done = false);
void apply_boundary(fcpt_ctx *c, bool final);
void ensure_pressure(fcpt_ctx *c);
void enqueue_cfl(fcpt_ctx *c, int apply_policy);
void enqueue_step(fcpt_ctx *c, bool dt_dev, double dt, bool shear_safe, bool split = false);
void enqueue_post(fcpt_ctx *c, bool may_defer_boundary = false);
void flush_deferred_boundary(fcpt_ctx *c);
// fcpt_exchange.hip
int enqueue_exchange(fcpt_ctx *c);
int enqueue_cfl_allreduce(fcpt_ctx *c);

} // namespace fcpt
#endif
