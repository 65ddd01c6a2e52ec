// CommunicateBoundaries and the MIN all-reduce of the C ABI (include/fargocpt_hip.h): pack / unpack of the overlap
// rings, and the RCCL communicator of a context (fcpt_comm.h does the RCCL calls).
#include "fcpt_ctx.h"

using namespace fcpt;

extern "C" {

int fcpt_exchange_count(const fcpt_ctx *c, uint64_t *count)
{
    if (!c || !count)
        return FCPT_EINVAL;
    *count = (uint64_t)(c->d.eos == FCPT_EOS_IDEAL ? 4 : 3) * c->d.nphi * FCPT_OVERLAP;
    return FCPT_OK;
}

} // extern "C"
namespace {
// device buffers (RCCL sends them in place) go through one copy kernel; host buffers (slabs of one process,
// staged exchange) through hipMemcpyAsync per field and side
bool exchange_on_device(const void *p)
{
    if (!p)
        return true;
    hipPointerAttribute_t a;
    if (hipPointerGetAttributes(&a, p) != hipSuccess) {
        (void)hipGetLastError(); // plain host memory: not an error here
        return false;
    }
    return a.type == hipMemoryTypeDevice;
}
int exchange_memcpy(fcpt_ctx *c, double *inner, double *outer, int unpack)
{
    const Dev &P = c->P;
    const size_t l = (size_t)FCPT_OVERLAP * P.nphi, lb = l * sizeof(double);
    const size_t row_in = unpack ? 0 : l, row_out = (size_t)(P.nr - (unpack ? 1 : 2) * FCPT_OVERLAP) * P.nphi;
    double *field[4] = {P.sigma, P.vrad, P.vazi, P.energy};
    const int nq = P.adiabatic ? 4 : 3;
    for (int q = 0; q < nq; ++q) {
        if (inner)
            HIPCHK(unpack ? hipMemcpyAsync(field[q] + row_in, inner + q * l, lb, hipMemcpyDefault, c->stream)
                          : hipMemcpyAsync(inner + q * l, field[q] + row_in, lb, hipMemcpyDefault, c->stream));
        if (outer)
            HIPCHK(unpack ? hipMemcpyAsync(field[q] + row_out, outer + q * l, lb, hipMemcpyDefault, c->stream)
                          : hipMemcpyAsync(outer + q * l, field[q] + row_out, lb, hipMemcpyDefault, c->stream));
    }
    return FCPT_OK;
}
} // namespace
extern "C" {

// commbound.cpp:108-125: rows [7,14) -> inner neighbour, rows [nr-14,nr-7) -> outer
int fcpt_exchange_pack(fcpt_ctx *c, double *send_inner, double *send_outer)
{
    if (!c)
        return FCPT_EINVAL;
    if (c->P.nr < 2 * FCPT_OVERLAP)
        return FCPT_EINVAL;
    if (!exchange_on_device(send_inner) || !exchange_on_device(send_outer))
        return exchange_memcpy(c, send_inner, send_outer, 0);
    if (send_inner || send_outer)
        launch_exchange_copy(c->P, send_inner, send_outer, 0, c->stream);
    return FCPT_OK;
}

// commbound.cpp:163-180: inner neighbour's data -> rows [0,7), outer's -> rows [nr-7,nr)
int fcpt_exchange_unpack(fcpt_ctx *c, const double *recv_inner, const double *recv_outer)
{
    if (!c)
        return FCPT_EINVAL;
    if (c->P.nr < 2 * FCPT_OVERLAP)
        return FCPT_EINVAL;
    join_side(c);
    if (!exchange_on_device(recv_inner) || !exchange_on_device(recv_outer))
        return exchange_memcpy(c, const_cast<double *>(recv_inner), const_cast<double *>(recv_outer), 1);
    if (recv_inner || recv_outer)
        launch_exchange_copy(c->P, const_cast<double *>(recv_inner), const_cast<double *>(recv_outer), 1, c->stream);
    return FCPT_OK;
}

// ---- radial slabs over RCCL ----------------------------------------------------------------------------------

int fcpt_comm_unique_id(void *id128)
{
    const int rc = comm_unique_id(id128);
    return rc == FCPT_EHIP ? FCPT_ECOMM : rc;
}

} // extern "C"
namespace {
// what a context needs around either communicator: neighbours, ghost-ring buffers, the all-reduce's scalar, the
// communication stream of comm_overlap
int comm_attach(fcpt_ctx *c, bool loopback, int rank)
{
    c->peer_inner = loopback ? 0 : (c->s.is_first ? -1 : rank - 1);
    c->peer_outer = loopback ? 0 : (c->s.is_last ? -1 : rank + 1);
    uint64_t cnt = 0;
    (void)fcpt_exchange_count(c, &cnt);
    int rc = FCPT_OK;
    for (int k = 0; k < 4 && !rc; ++k)
        if (!c->xbuf[k])
            rc = dev_alloc(c, &c->xbuf[k], (size_t)cnt);
    if (!rc && !c->d_cfl)
        rc = dev_alloc(c, &c->d_cfl, 1);
    if (rc)
        return rc;
    HIPCHK(hipStreamCreateWithFlags(&c->comm_stream, hipStreamNonBlocking));
    HIPCHK(hipEventCreateWithFlags(&c->e_packed, hipEventDisableTiming));
    HIPCHK(hipEventCreateWithFlags(&c->e_received, hipEventDisableTiming));
    return FCPT_OK;
}
} // namespace
extern "C" {

int fcpt_comm_init(fcpt_ctx *c, const void *id128)
{
    if (!c || !id128)
        return FCPT_EINVAL;
    if (c->comm) {
        set_error("fcpt_comm_init: the context already has a communicator");
        return FCPT_EINVAL;
    }
    const bool loopback = c->P.opt.comm_loopback != 0;
    if (!loopback && c->P.nr < 2 * FCPT_OVERLAP && c->d.nranks > 1)
        return FCPT_ESPLIT;
    HIPCHK(hipSetDevice(c->device));
    // rehearsal on one GPU: a communicator of one rank whose slab sends its ghost rings to itself
    const int rank = loopback ? 0 : c->d.rank, nranks = loopback ? 1 : c->d.nranks;
    if (int rc = comm_create(id128, rank, nranks, &c->comm))
        return rc == FCPT_EHIP ? FCPT_ECOMM : rc;
    return comm_attach(c, loopback, rank);
}

// The host-staged transport (fcpt_comm.h): the same fcpt_exchange / fcpt_cfl_allreduce / fcpt_run_steps, for ranks
// that share a GPU
int fcpt_comm_init_host(fcpt_ctx *c, const char *path)
{
    if (!c || !path)
        return FCPT_EINVAL;
    if (c->comm) {
        set_error("fcpt_comm_init_host: the context already has a communicator");
        return FCPT_EINVAL;
    }
    if (c->P.opt.comm_loopback != 0) {
        set_error("fcpt_comm_init_host: the loopback rehearsal is an RCCL mode");
        return FCPT_EINVAL;
    }
    if (c->P.nr < 2 * FCPT_OVERLAP && c->d.nranks > 1)
        return FCPT_ESPLIT;
    HIPCHK(hipSetDevice(c->device));
    uint64_t cnt = 0;
    (void)fcpt_exchange_count(c, &cnt);
    if (int rc = comm_create_host(path, c->d.rank, c->d.nranks, (size_t)cnt, &c->comm))
        return rc == FCPT_EHIP ? FCPT_ECOMM : rc;
    return comm_attach(c, false, c->d.rank);
}

int fcpt_comm_barrier(fcpt_ctx *c)
{
    if (!c)
        return FCPT_EINVAL;
    if (!c->comm) {
        set_error("fcpt_comm_barrier needs fcpt_comm_init or fcpt_comm_init_host");
        return FCPT_EINVAL;
    }
    join_side(c);
    const int rc = comm_barrier(c->comm, c->stream);
    return rc == FCPT_EHIP ? FCPT_ECOMM : rc;
}

int fcpt_comm_destroy(fcpt_ctx *c)
{
    if (!c)
        return FCPT_EINVAL;
    if (c->comm) {
        join_side(c);
        (void)hipStreamSynchronize(c->stream);
        if (c->comm_stream)
            (void)hipStreamSynchronize(c->comm_stream);
        comm_destroy(c->comm);
        c->comm = nullptr;
    }
    if (c->comm_stream)
        (void)hipStreamDestroy(c->comm_stream);
    if (c->e_packed)
        (void)hipEventDestroy(c->e_packed);
    if (c->e_received)
        (void)hipEventDestroy(c->e_received);
    c->comm_stream = nullptr;
    c->e_packed = c->e_received = nullptr;
    return FCPT_OK; // the ghost buffers go with the context
}

} // extern "C"
namespace fcpt {
// commbound.cpp:98-182
int enqueue_exchange(fcpt_ctx *c)
{
    if (!c->comm) {
        set_error("fcpt_exchange needs fcpt_comm_init");
        return FCPT_EINVAL;
    }
    if (c->peer_inner < 0 && c->peer_outer < 0)
        return FCPT_OK; // a single slab: CommunicateBoundaries returns at once (commbound.cpp:104)
    join_side(c);
    double *s_in = c->peer_inner >= 0 ? c->xbuf[0] : nullptr, *s_out = c->peer_outer >= 0 ? c->xbuf[1] : nullptr;
    double *r_in = c->peer_inner >= 0 ? c->xbuf[2] : nullptr, *r_out = c->peer_outer >= 0 ? c->xbuf[3] : nullptr;
    uint64_t cnt = 0;
    (void)fcpt_exchange_count(c, &cnt);
    launch_exchange_copy(c->P, s_in, s_out, 0, c->stream);
    int rc;
    if (c->P.opt.comm_overlap != 0) {
        // transfers on the communication stream; under them, on the context's stream, the CFL terms of the rings
        // that neither the unpack nor the boundary kernels write (fcpt_cfl_begin)
        HIPCHK(hipEventRecord(c->e_packed, c->stream));
        HIPCHK(hipStreamWaitEvent(c->comm_stream, c->e_packed, 0));
        rc = comm_neighbour_exchange(c->comm, c->peer_inner, s_in, r_in, c->peer_outer, s_out, r_out, (size_t)cnt,
                                     c->comm_stream);
        HIPCHK(hipEventRecord(c->e_received, c->comm_stream));
        if (!rc)
            rc = fcpt_cfl_begin(c);
        HIPCHK(hipStreamWaitEvent(c->stream, c->e_received, 0));
    } else {
        rc = comm_neighbour_exchange(c->comm, c->peer_inner, s_in, r_in, c->peer_outer, s_out, r_out, (size_t)cnt,
                                     c->stream);
    }
    if (rc)
        return rc == FCPT_EHIP ? FCPT_ECOMM : rc;
    launch_exchange_copy(c->P, r_in, r_out, 1, c->stream);
    return FCPT_OK;
}

// cfl.cpp:185-379 with the result left in c->d_cfl
int enqueue_cfl_allreduce(fcpt_ctx *c)
{
    c->P.cfl_export = c->d_cfl; // the final fold of the reduction writes the all-reduce's operand itself
    enqueue_cfl(c, 0);
    c->P.cfl_export = nullptr;
    const int rc = comm_allreduce_min(c->comm, c->d_cfl, c->stream);
    return rc == FCPT_EHIP ? FCPT_ECOMM : rc;
}
} // namespace fcpt
extern "C" {

int fcpt_exchange(fcpt_ctx *c)
{
    if (!c)
        return FCPT_EINVAL;
    ProfScope prof_scope(c);
    if (int rc = enqueue_exchange(c))
        return rc;
    HIPCHK(hipGetLastError());
    return FCPT_OK;
}

int fcpt_cfl_allreduce(fcpt_ctx *c, double *dt_global)
{
    if (!c)
        return FCPT_EINVAL;
    if (!c->comm) {
        set_error("fcpt_cfl_allreduce needs fcpt_comm_init");
        return FCPT_EINVAL;
    }
    ProfScope prof_scope(c);
    if (int rc = enqueue_cfl_allreduce(c))
        return rc;
    HIPCHK(hipGetLastError());
    if (dt_global) {
        HIPCHK(hipMemcpyAsync(&c->h_clk->cfl_dt, c->d_cfl, sizeof(double), hipMemcpyDeviceToHost, c->stream));
        HIPCHK(hipStreamSynchronize(c->stream));
        *dt_global = c->h_clk->cfl_dt;
    }
    return FCPT_OK;
}

} // extern "C"
