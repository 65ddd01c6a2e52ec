// The two transports of the ghost exchange and of the CFL MIN reduction (see fcpt_comm.h): RCCL, and a host-staged
// one through a shared-memory file for ranks that share a GPU.
#include "fcpt_comm.h"

#include <dlfcn.h>
#include <fcntl.h>
#include <link.h>
#include <rccl/rccl.h>
#include <sched.h>
#include <sys/mman.h>
#include <sys/stat.h>
#include <time.h>
#include <unistd.h>

#include <cerrno>
#include <cstdint>
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

// ---- host-staged transport ------------------------------------------------------------------------------------------
// One file, mapped shared by the ranks of a node: per rank two mailboxes (the ghost rings its inner / its outer
// neighbour sends it) and two reduction slots.  Every operation carries a sequence number that all ranks count
// alike (they issue the same operations in the same order, as MPI ranks do): a sender waits until the receiver has
// consumed message s - 1 before it writes message s, a receiver waits for message s; reductions alternate between two
// slots, so a rank that is one reduction ahead never overwrites a value a slower rank still has to read.
namespace {
constexpr uint64_t kLinkMagic = 0x46435054484c4e4bull; // "FCPTHLNK"
struct alignas(64) LinkHeader {
    uint64_t magic; // written last by rank 0
    uint64_t count; // doubles per mailbox
    uint32_t nranks;
    uint32_t attached; // ranks that have mapped the file
};
struct alignas(64) LinkMailboxHead {
    uint64_t written; // sequence number of the message in `data`
    uint64_t read;    // ... of the last message the owner has taken out
};
struct alignas(64) LinkSlot {
    uint64_t seq;
    double v;
};
struct HostLink {
    int fd = -1;
    unsigned char *base = nullptr;
    size_t bytes = 0, count = 0, box_bytes = 0, rank_bytes = 0;
    uint64_t xseq = 0, rseq = 0;
    double *stage = nullptr; // pinned: send inner, send outer, recv inner, recv outer, one scalar
    std::string path;
    double timeout_s = 120.0;
    LinkHeader *header() const { return reinterpret_cast<LinkHeader *>(base); }
    unsigned char *rank_block(int r) const { return base + sizeof(LinkHeader) + (size_t)r * rank_bytes; }
    LinkMailboxHead *box(int r, int side) const { return reinterpret_cast<LinkMailboxHead *>(rank_block(r) + (size_t)side * box_bytes); }
    double *box_data(int r, int side) const { return reinterpret_cast<double *>(rank_block(r) + (size_t)side * box_bytes + sizeof(LinkMailboxHead)); }
    LinkSlot *slot(int r, int k) const { return reinterpret_cast<LinkSlot *>(rank_block(r) + 2 * box_bytes) + k; }
};
double now_s()
{
    timespec t;
    clock_gettime(CLOCK_MONOTONIC, &t);
    return (double)t.tv_sec + 1e-9 * (double)t.tv_nsec;
}
// waits until *word >= want (acquire); false after the link's timeout
bool link_wait(const HostLink *h, const uint64_t *word, uint64_t want, const char *what)
{
    const double t_end = now_s() + h->timeout_s;
    for (unsigned spin = 0;; ++spin) {
        if (__atomic_load_n(word, __ATOMIC_ACQUIRE) >= want)
            return true;
        if (spin < 2000) {
            sched_yield();
        } else {
            timespec ts = {0, 50000};
            nanosleep(&ts, nullptr);
            if ((spin & 1023) == 0 && now_s() > t_end) {
                set_error("host-staged transport: %s: the other rank did not arrive within %.0f s (%s)", what, h->timeout_s,
                          h->path.c_str());
                return false;
            }
        }
    }
}
void link_close(HostLink *h, bool unlink_file)
{
    if (!h)
        return;
    if (h->stage)
        (void)hipHostFree(h->stage);
    if (h->base)
        munmap(h->base, h->bytes);
    if (h->fd >= 0)
        close(h->fd);
    if (unlink_file)
        unlink(h->path.c_str());
    delete h;
}
} // namespace

struct Comm {
    ncclComm_t comm = nullptr;
    int rank = 0, nranks = 1;
    HostLink *host = nullptr; // the host-staged transport instead of RCCL
    double *d_scratch = nullptr; // RCCL: operand of the barrier's all-reduce
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

// The host-staged communicator: rank 0 creates `path` (which must not be in use by another run), the others wait for it.
int comm_create_host(const char *path, int rank, int nranks, size_t count, Comm **out)
{
    if (!path || !path[0] || !out || nranks < 1 || rank < 0 || rank >= nranks || count == 0) {
        set_error("comm_create_host: bad argument (rank %d of %d)", rank, nranks);
        return FCPT_EINVAL;
    }
    HostLink *h = new HostLink();
    h->path = path;
    h->count = count;
    if (const char *e = getenv("FCPT_HOSTLINK_TIMEOUT"))
        if (atof(e) > 0.0)
            h->timeout_s = atof(e);
    h->box_bytes = (sizeof(LinkMailboxHead) + count * sizeof(double) + 63) / 64 * 64;
    h->rank_bytes = 2 * h->box_bytes + 2 * sizeof(LinkSlot);
    h->bytes = sizeof(LinkHeader) + (size_t)nranks * h->rank_bytes;
    const double t_end = now_s() + h->timeout_s;
    if (rank == 0) {
        unlink(path); // a file left behind by a run that died
        h->fd = open(path, O_RDWR | O_CREAT | O_EXCL, 0600);
        if (h->fd < 0 || ftruncate(h->fd, (off_t)h->bytes) != 0) {
            set_error("host-staged transport: cannot create %s (%s)", path, std::strerror(errno));
            link_close(h, false);
            return FCPT_ECOMM;
        }
    } else {
        for (;;) { // until rank 0 has created the file at its full size
            h->fd = open(path, O_RDWR);
            struct stat sb;
            if (h->fd >= 0 && fstat(h->fd, &sb) == 0 && (size_t)sb.st_size == h->bytes)
                break;
            if (h->fd >= 0)
                close(h->fd);
            h->fd = -1;
            if (now_s() > t_end) {
                set_error("host-staged transport: rank 0 did not create %s (%zu bytes) within %.0f s", path, h->bytes, h->timeout_s);
                link_close(h, false);
                return FCPT_ECOMM;
            }
            timespec ts = {0, 2000000};
            nanosleep(&ts, nullptr);
        }
    }
    void *m = mmap(nullptr, h->bytes, PROT_READ | PROT_WRITE, MAP_SHARED, h->fd, 0);
    if (m == MAP_FAILED) {
        set_error("host-staged transport: mmap of %s failed (%s)", path, std::strerror(errno));
        link_close(h, rank == 0);
        return FCPT_ECOMM;
    }
    h->base = static_cast<unsigned char *>(m);
    LinkHeader *hd = h->header();
    if (rank == 0) {
        hd->count = count;
        hd->nranks = (uint32_t)nranks;
        __atomic_store_n(&hd->magic, kLinkMagic, __ATOMIC_RELEASE);
    } else if (!link_wait(h, &hd->magic, 1, "attach")) {
        link_close(h, false);
        return FCPT_ECOMM;
    }
    if (hd->magic != kLinkMagic || hd->count != count || hd->nranks != (uint32_t)nranks) {
        set_error("host-staged transport: %s belongs to another run (%u ranks, %llu doubles per message)", path, hd->nranks,
                  (unsigned long long)hd->count);
        link_close(h, false);
        return FCPT_ECOMM;
    }
    __atomic_add_fetch(&hd->attached, 1u, __ATOMIC_ACQ_REL);
    if (hipHostMalloc((void **)&h->stage, (4 * count + 8) * sizeof(double)) != hipSuccess) {
        set_error("host-staged transport: hipHostMalloc of the staging buffers failed");
        link_close(h, rank == 0);
        return FCPT_ENOMEM;
    }
    Comm *c = new Comm();
    c->rank = rank;
    c->nranks = nranks;
    c->host = h;
    *out = c;
    return FCPT_OK;
}

void comm_destroy(Comm *c)
{
    if (!c)
        return;
    if (c->comm && g_rccl.ok)
        (void)g_rccl.CommDestroy(c->comm);
    if (c->d_scratch)
        (void)hipFree(c->d_scratch);
    link_close(c->host, c->rank == 0);
    delete c;
}

bool comm_is_host_staged(const Comm *c) { return c && c->host; }

namespace {
// MIN over the ranks of one double through the reduction slots
int link_allreduce_min(Comm *c, double *value)
{
    HostLink *h = c->host;
    const uint64_t s = ++h->rseq;
    LinkSlot *mine = h->slot(c->rank, (int)(s & 1));
    mine->v = *value;
    __atomic_store_n(&mine->seq, s, __ATOMIC_RELEASE);
    double m = *value;
    for (int r = 0; r < c->nranks; ++r) {
        LinkSlot *o = h->slot(r, (int)(s & 1));
        if (!link_wait(h, &o->seq, s, "MIN reduction"))
            return FCPT_ECOMM;
        m = o->v < m ? o->v : m;
    }
    *value = m;
    return FCPT_OK;
}
int link_exchange(Comm *c, int peer_inner, const double *send_inner, double *recv_inner, int peer_outer,
                  const double *send_outer, double *recv_outer, size_t count, hipStream_t st)
{
    HostLink *h = c->host;
    if (count != h->count || peer_inner == c->rank || peer_outer == c->rank) {
        set_error("host-staged transport: message of %zu doubles on a link of %zu (or a slab that is its own neighbour)", count, h->count);
        return FCPT_EINVAL;
    }
    const size_t nb = count * sizeof(double);
    double *s_in = h->stage, *s_out = h->stage + count, *r_in = h->stage + 2 * count, *r_out = h->stage + 3 * count;
#define LCHK(call)                                                                    \
    if ((call) != hipSuccess) {                                                       \
        set_error("host-staged transport: %s failed", #call);                         \
        return FCPT_EHIP;                                                             \
    }
    if (peer_inner >= 0)
        LCHK(hipMemcpyAsync(s_in, send_inner, nb, hipMemcpyDeviceToHost, st));
    if (peer_outer >= 0)
        LCHK(hipMemcpyAsync(s_out, send_outer, nb, hipMemcpyDeviceToHost, st));
    LCHK(hipStreamSynchronize(st)); // (also: the previous exchange's uploads from r_in / r_out are done)
    const uint64_t s = ++h->xseq;
    // my rows [7,14) are the inner neighbour's ghost rows [nr-7,nr): its mailbox "from the outer neighbour" (side 1)
    const int peers[2] = {peer_inner, peer_outer};
    const double *sends[2] = {s_in, s_out};
    for (int k = 0; k < 2; ++k) {
        if (peers[k] < 0)
            continue;
        LinkMailboxHead *b = h->box(peers[k], 1 - k);
        if (!link_wait(h, &b->read, s - 1, "ghost exchange (send)"))
            return FCPT_ECOMM;
        std::memcpy(h->box_data(peers[k], 1 - k), sends[k], nb);
        __atomic_store_n(&b->written, s, __ATOMIC_RELEASE);
    }
    double *recvs[2] = {r_in, r_out};
    for (int k = 0; k < 2; ++k) {
        if (peers[k] < 0)
            continue;
        LinkMailboxHead *b = h->box(c->rank, k);
        if (!link_wait(h, &b->written, s, "ghost exchange (receive)"))
            return FCPT_ECOMM;
        std::memcpy(recvs[k], h->box_data(c->rank, k), nb);
        __atomic_store_n(&b->read, s, __ATOMIC_RELEASE);
    }
    if (peer_inner >= 0)
        LCHK(hipMemcpyAsync(recv_inner, r_in, nb, hipMemcpyHostToDevice, st));
    if (peer_outer >= 0)
        LCHK(hipMemcpyAsync(recv_outer, r_out, nb, hipMemcpyHostToDevice, st));
    return FCPT_OK;
}
} // namespace

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
    if (!c || (!c->comm && !c->host))
        return FCPT_EINVAL;
    if (peer_inner < 0 && peer_outer < 0)
        return FCPT_OK;
    if (c->host)
        return link_exchange(c, peer_inner, send_inner, recv_inner, peer_outer, send_outer, recv_outer, count, st);
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
    if (!c || (!c->comm && !c->host) || !d_value)
        return FCPT_EINVAL;
    if (c->nranks == 1)
        return FCPT_OK;
    if (c->host) {
        double *v = c->host->stage + 4 * c->host->count;
        if (hipMemcpyAsync(v, d_value, sizeof(double), hipMemcpyDeviceToHost, st) != hipSuccess ||
            hipStreamSynchronize(st) != hipSuccess) {
            set_error("host-staged transport: reading the local CFL value failed");
            return FCPT_EHIP;
        }
        if (int rc = link_allreduce_min(c, v))
            return rc;
        if (hipMemcpyAsync(d_value, v, sizeof(double), hipMemcpyHostToDevice, st) != hipSuccess ||
            hipStreamSynchronize(st) != hipSuccess) { // (v is reused by the next call)
            set_error("host-staged transport: writing the reduced CFL value failed");
            return FCPT_EHIP;
        }
        return FCPT_OK;
    }
    NCHK(g_rccl.AllReduce(d_value, d_value, 1, ncclDouble, ncclMin, c->comm, st));
    return FCPT_OK;
}

// MPI_Barrier: returns when every rank has called it and all work queued on `st` before the call is done
int comm_barrier(Comm *c, hipStream_t st)
{
    if (!c || (!c->comm && !c->host))
        return FCPT_EINVAL;
    if (hipStreamSynchronize(st) != hipSuccess) {
        set_error("comm_barrier: hipStreamSynchronize failed");
        return FCPT_EHIP;
    }
    if (c->nranks == 1)
        return FCPT_OK;
    if (c->host) {
        double dummy = 0.0;
        return link_allreduce_min(c, &dummy);
    }
    if (!c->d_scratch && (hipMalloc((void **)&c->d_scratch, sizeof(double)) != hipSuccess ||
                          hipMemset(c->d_scratch, 0, sizeof(double)) != hipSuccess)) {
        set_error("comm_barrier: hipMalloc failed");
        return FCPT_ENOMEM;
    }
    NCHK(g_rccl.AllReduce(c->d_scratch, c->d_scratch, 1, ncclDouble, ncclMin, c->comm, st));
    if (hipStreamSynchronize(st) != hipSuccess) {
        set_error("comm_barrier: hipStreamSynchronize failed");
        return FCPT_EHIP;
    }
    return FCPT_OK;
}

} // namespace fcpt
