// RCCL transport of the ghost exchange and of the CFL MIN reduction (see fcpt_comm.h).
#include "fcpt_comm.h"

#include <dlfcn.h>
#include <link.h>
#include <rccl/rccl.h>

#include <cstdlib>
#include <cstring>
#include <mutex>
#include <string>

#include "fcpt_internal.h"

namespace fcpt {

namespace {

// the entry points of librccl the path needs, bound once per process
struct Rccl {
    void *handle = nullptr;
    decltype(&ncclGetUniqueId) GetUniqueId = nullptr;
    decltype(&ncclCommInitRank) CommInitRank = nullptr;
    decltype(&ncclCommDestroy) CommDestroy = nullptr;
    decltype(&ncclGroupStart) GroupStart = nullptr;
    decltype(&ncclGroupEnd) GroupEnd = nullptr;
    decltype(&ncclSend) Send = nullptr;
    decltype(&ncclRecv) Recv = nullptr;
    decltype(&ncclAllReduce) AllReduce = nullptr;
    decltype(&ncclGetErrorString) GetErrorString = nullptr;
    bool ok = false;
};
Rccl g_rccl;
std::once_flag g_rccl_once;

void bind_rccl()
{
    Rccl &r = g_rccl;
    // 1. a copy the process already holds (PyTorch-ROCm ships its own librccl.so next to its own HIP runtime: two
    //    RCCL instances, or one bound to another HIP runtime than the kernels of this library, must be avoided)
    // 2. FCPT_RCCL_PATH  3. the ROCm installation this library was built against
    const char *env = getenv("FCPT_RCCL_PATH");
    {
        // (PyTorch's copy carries no SONAME: look for it among the loaded objects by path)
        std::string loaded;
        dl_iterate_phdr(
            [](struct dl_phdr_info *info, size_t, void *out) {
                if (info->dlpi_name && std::strstr(info->dlpi_name, "librccl.so")) {
                    *static_cast<std::string *>(out) = info->dlpi_name;
                    return 1;
                }
                return 0;
            },
            &loaded);
        if (!loaded.empty())
            r.handle = dlopen(loaded.c_str(), RTLD_NOW | RTLD_NOLOAD | RTLD_GLOBAL);
    }
    for (const char *name : {"librccl.so", "librccl.so.1"})
        if (!r.handle)
            r.handle = dlopen(name, RTLD_NOW | RTLD_NOLOAD | RTLD_GLOBAL);
    if (!r.handle && env && env[0])
        r.handle = dlopen(env, RTLD_NOW | RTLD_GLOBAL);
    for (const char *name : {"librccl.so.1", "/opt/rocm/lib/librccl.so.1"})
        if (!r.handle)
            r.handle = dlopen(name, RTLD_NOW | RTLD_GLOBAL);
    if (!r.handle) {
        set_error("librccl.so.1 could not be loaded (%s): multi-GPU runs need RCCL; FCPT_RCCL_PATH names an alternative",
                  dlerror());
        return;
    }
    bool all = true;
#define BIND(member, symbol)                                                     \
    r.member = reinterpret_cast<decltype(r.member)>(dlsym(r.handle, symbol));  \
    all = all && r.member != nullptr;
    BIND(GetUniqueId, "ncclGetUniqueId")
    BIND(CommInitRank, "ncclCommInitRank")
    BIND(CommDestroy, "ncclCommDestroy")
    BIND(GroupStart, "ncclGroupStart")
    BIND(GroupEnd, "ncclGroupEnd")
    BIND(Send, "ncclSend")
    BIND(Recv, "ncclRecv")
    BIND(AllReduce, "ncclAllReduce")
    BIND(GetErrorString, "ncclGetErrorString")
#undef BIND
    if (!all) {
        set_error("the loaded librccl lacks one of the ncclSend/ncclRecv/ncclAllReduce entry points");
        return;
    }
    r.ok = true;
}

bool have_rccl()
{
    std::call_once(g_rccl_once, bind_rccl);
    return g_rccl.ok;
}

#define NCHK(call)                                                                                  \
    do {                                                                                            \
        ncclResult_t r_ = (call);                                                                   \
        if (r_ != ncclSuccess) {                                                                    \
            set_error("%s failed: %s (%s:%d)", #call, g_rccl.GetErrorString(r_), __FILE__, __LINE__); \
            return FCPT_EHIP;                                                                       \
        }                                                                                           \
    } while (0)

} // namespace

struct Comm {
    ncclComm_t comm = nullptr;
    int rank = 0, nranks = 1;
};

int comm_unique_id(void *id128)
{
    if (!id128)
        return FCPT_EINVAL;
    if (!have_rccl())
        return FCPT_EHIP;
    static_assert(sizeof(ncclUniqueId) == 128, "fcpt_comm_unique_id hands out 128 bytes");
    ncclUniqueId id;
    NCHK(g_rccl.GetUniqueId(&id));
    std::memcpy(id128, &id, sizeof(id));
    return FCPT_OK;
}

int comm_create(const void *id128, int rank, int nranks, Comm **out)
{
    if (!id128 || !out || nranks < 1 || rank < 0 || rank >= nranks) {
        set_error("comm_create: bad rank %d of %d", rank, nranks);
        return FCPT_EINVAL;
    }
    if (!have_rccl())
        return FCPT_EHIP;
    ncclUniqueId id;
    std::memcpy(&id, id128, sizeof(id));
    Comm *c = new Comm();
    c->rank = rank;
    c->nranks = nranks;
    ncclResult_t r = g_rccl.CommInitRank(&c->comm, nranks, id, rank);
    if (r != ncclSuccess) {
        set_error("ncclCommInitRank(rank %d of %d) failed: %s", rank, nranks, g_rccl.GetErrorString(r));
        delete c;
        return FCPT_EHIP;
    }
    *out = c;
    return FCPT_OK;
}

void comm_destroy(Comm *c)
{
    if (!c)
        return;
    if (c->comm && g_rccl.ok)
        (void)g_rccl.CommDestroy(c->comm);
    delete c;
}

int comm_rank(const Comm *c) { return c ? c->rank : 0; }
int comm_size(const Comm *c) { return c ? c->nranks : 1; }

// commbound.cpp:130-158.  One group: RCCL fuses the (up to) two sends and two receives into one kernel, each pair
// on its own xGMI link (the neighbours are different GPUs).  Send before receive towards the inner neighbour and
// the same order towards the outer one is what the neighbour's group mirrors, so the pairs match in both
// directions; within a group the order only matters between operations of one peer (the loopback rehearsal, where
// both "neighbours" are this rank: first send pairs with first receive).
int comm_neighbour_exchange(Comm *c, int peer_inner, const double *send_inner, double *recv_inner, int peer_outer,
                            const double *send_outer, double *recv_outer, size_t count, hipStream_t st)
{
    if (!c || !c->comm)
        return FCPT_EINVAL;
    if (peer_inner < 0 && peer_outer < 0)
        return FCPT_OK;
    NCHK(g_rccl.GroupStart());
    ncclResult_t r = ncclSuccess;
    if (peer_inner >= 0) {
        r = g_rccl.Send(send_inner, count, ncclDouble, peer_inner, c->comm, st);
        if (r == ncclSuccess)
            r = g_rccl.Recv(recv_inner, count, ncclDouble, peer_inner, c->comm, st);
    }
    if (peer_outer >= 0 && r == ncclSuccess) {
        r = g_rccl.Send(send_outer, count, ncclDouble, peer_outer, c->comm, st);
        if (r == ncclSuccess)
            r = g_rccl.Recv(recv_outer, count, ncclDouble, peer_outer, c->comm, st);
    }
    const ncclResult_t e = g_rccl.GroupEnd(); // always closes the group, also after a failed call inside it
    if (r != ncclSuccess || e != ncclSuccess) {
        set_error("ghost exchange over RCCL failed: %s", g_rccl.GetErrorString(r != ncclSuccess ? r : e));
        return FCPT_EHIP;
    }
    return FCPT_OK;
}

// cfl.cpp:379
int comm_allreduce_min(Comm *c, double *d_value, hipStream_t st)
{
    if (!c || !c->comm || !d_value)
        return FCPT_EINVAL;
    if (c->nranks == 1)
        return FCPT_OK;
    NCHK(g_rccl.AllReduce(d_value, d_value, 1, ncclDouble, ncclMin, c->comm, st));
    return FCPT_OK;
}

} // namespace fcpt
