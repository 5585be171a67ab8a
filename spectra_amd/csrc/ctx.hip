// Context, error state and the communicators of the row-sharded path.
//
// Multi-GPU model: one process (or, for the loopback test
// communicator, one host thread) per shard; each shard owns a contiguous block
// of rows of A, V, f.  Two collectives exist on the whole path: an all-gather
// of the current Krylov vector before each SpMV and a sum all-reduce of a few
// doubles (alpha, |f|^2, V'f).  The production communicator is RCCL over xGMI;
// librccl.so.1 is dlopen'ed lazily so that a single-GPU user never needs it and
// so that, inside a PyTorch process, torch's already-loaded RCCL is the one used.
#include "common.hpp"

#include <dlfcn.h>
#include <rccl/rccl.h>

#include <algorithm>
#include <cctype>
#include <condition_variable>
#include <exception>
#include <thread>
#include <cstring>
#include <mutex>

namespace mispec {
static thread_local std::string g_last_error;
void set_last_error(const std::string& msg) { g_last_error = msg; }

void* big_host_alloc(size_t bytes)
{
    // plain malloc: asking for transparent huge pages (madvise) made the first touch SLOWER where it was tried (direct
    // compaction and 2 MiB zeroing inside the page faults: 0.28 -> 2.8 s for the mirroring of a 2M-row matrix)
    void* q = std::malloc(bytes ? bytes : 1);
    if (!q)
        throw std::bad_alloc();
    return q;
}

// ---- options (common.hpp) ----------------------------------------------------------------------------------------------
namespace {
struct OptionTable
{
    std::mutex mu;
    std::vector<std::pair<std::string, std::string>> set;  // few entries: a linear scan beats a map
    // values handed out as const char*: a value that is replaced stays alive (the list only grows by what callers set)
    std::vector<std::string*> retired;
};
OptionTable& option_table()
{
    static OptionTable* t = new OptionTable;  // never destroyed: option() may be called from static destructors
    return *t;
}
// every name the library reads (mispec_set_option rejects anything else, so that a typo cannot pass for a measurement)
const char* const kOptionNames[] = {
    "csr_win", "spmv_tiles", "reorder", "spmv_staged", "dia2", "csr_win_iters", "csr_win_pf", "csr_win_nt", "kernel_probe",
    "overlap", "exchange", "small", "spec_corr", "one_reduction", "host_steps", "orth", "restart_sync", "vq", "shift",
    "host_turn", "orth_kernel", "host_threads", nullptr};
}  // namespace

const char* option(const char* name)
{
    OptionTable& t = option_table();
    {
        std::lock_guard<std::mutex> lock(t.mu);
        for (auto& kv : t.set)
            if (kv.first == name)
                return kv.second.c_str();
    }
    std::string env = "MISPEC_";
    for (const char* c = name; *c; c++)
        env.push_back(char(std::toupper(static_cast<unsigned char>(*c))));
    return std::getenv(env.c_str());
}
int option_int(const char* name, int dflt)
{
    const char* v = option(name);
    return v ? std::atoi(v) : dflt;
}
bool option_is(const char* name, const char* value)
{
    const char* v = option(name);
    return v && std::strcmp(v, value) == 0;
}

int ingest_threads()
{
    static const int n = [] {
        const unsigned hw = std::thread::hardware_concurrency();
        return int(std::min(64u, std::max(1u, hw)));
    }();
    // option host_threads (tests): an upper bound on the host threads of the ingest and of the shift solve's host-side
    // factorisation — their results must not depend on it
    const int cap = option_int("host_threads", 0);
    return cap >= 1 ? std::min(n, cap) : n;
}

void parallel_ranges(int64_t n, int parts, const std::function<void(int, int64_t, int64_t)>& fn)
{
    if (n <= 0)
        return;
    parts = int(std::max<int64_t>(1, std::min<int64_t>(parts, n)));
    if (parts == 1)
    {
        fn(0, 0, n);
        return;
    }
    std::vector<std::thread> th;
    std::vector<std::exception_ptr> err(static_cast<size_t>(parts));
    th.reserve(static_cast<size_t>(parts));
    for (int t = 0; t < parts; t++)
    {
        const int64_t b = n * t / parts, e = n * (t + 1) / parts;
        th.emplace_back([&, t, b, e] {
            try
            {
                fn(t, b, e);
            }
            catch (...)
            {
                err[size_t(t)] = std::current_exception();
            }
        });
    }
    for (auto& t : th)
        t.join();
    for (auto& e : err)
        if (e)
            std::rethrow_exception(e);
}
}  // namespace mispec

using namespace mispec;

extern "C" const char* mispec_last_error(void) { return g_last_error.c_str(); }
extern "C" const char* mispec_version(void) { return "0.1.0 (gfx950)"; }

extern "C" int mispec_set_option(const char* name, const char* value)
{
    return guarded([&] {
        MISPEC_REQUIRE(name != nullptr, "mispec_set_option: name is NULL");
        bool known = false;
        for (const char* const* k = kOptionNames; *k; k++)
            known = known || std::strcmp(*k, name) == 0;
        MISPEC_REQUIRE(known, std::string("mispec_set_option: unknown option '") + name + "'");
        auto& t = option_table();
        std::lock_guard<std::mutex> lock(t.mu);
        for (size_t i = 0; i < t.set.size(); i++)
            if (t.set[i].first == name)
            {
                t.retired.push_back(new std::string(std::move(t.set[i].second)));  // a pointer handed out earlier stays valid
                t.set.erase(t.set.begin() + long(i));
                break;
            }
        if (value)
            t.set.emplace_back(name, value);
    });
}

extern "C" const char* mispec_get_option(const char* name) { return name ? option(name) : nullptr; }

extern "C" int mispec_ctx_create(int device, void* hip_stream, mispec_ctx** out)
{
    return guarded([&] {
        MISPEC_REQUIRE(out != nullptr, "mispec_ctx_create: out is NULL");
        int ndev = 0;
        hipError_t e = hipGetDeviceCount(&ndev);
        if (e != hipSuccess || ndev <= 0)
            throw Error(MISPEC_ERUNTIME, "mispec_ctx_create: no HIP device available (this library has no CPU fallback)");
        MISPEC_REQUIRE(device >= 0 && device < ndev, "mispec_ctx_create: device index out of range");
        auto* ctx = new mispec_ctx();
        try
        {
            ctx->device = device;
            MISPEC_HIP(hipSetDevice(device));
            hipDeviceProp_t prop;
            MISPEC_HIP(hipGetDeviceProperties(&prop, device));
            ctx->num_cu = prop.multiProcessorCount > 0 ? prop.multiProcessorCount : 256;
            if (hip_stream)
            {
                ctx->stream = static_cast<hipStream_t>(hip_stream);
                ctx->own_stream = false;
            }
            else
            {
                MISPEC_HIP(hipStreamCreateWithFlags(&ctx->stream, hipStreamNonBlocking));
                ctx->own_stream = true;
            }
        }
        catch (...)
        {
            delete ctx;
            throw;
        }
        *out = ctx;
    });
}

extern "C" int mispec_ctx_destroy(mispec_ctx* ctx)
{
    return guarded([&] {
        if (!ctx)
            return;
        (void) hipSetDevice(ctx->device);
        if (ctx->comm_owner && ctx->comm_owner_free)
            ctx->comm_owner_free(ctx->comm_owner);
        if (ctx->own_stream && ctx->stream)
            (void) hipStreamDestroy(ctx->stream);
        delete ctx;
    });
}

extern "C" int mispec_ctx_sync(mispec_ctx* ctx)
{
    return guarded([&] {
        MISPEC_REQUIRE(ctx, "mispec_ctx_sync: ctx is NULL");
        ctx->make_current();
        MISPEC_HIP(hipStreamSynchronize(ctx->stream));
    });
}

extern "C" void* mispec_ctx_stream(mispec_ctx* ctx) { return ctx ? static_cast<void*>(ctx->stream) : nullptr; }

extern "C" int mispec_ctx_set_comm(mispec_ctx* ctx, const mispec_comm* comm)
{
    return guarded([&] {
        MISPEC_REQUIRE(ctx && comm, "mispec_ctx_set_comm: NULL argument");
        MISPEC_REQUIRE(comm->world >= 1 && comm->rank >= 0 && comm->rank < comm->world,
                       "mispec_ctx_set_comm: need 0 <= rank < world");
        MISPEC_REQUIRE(comm->world == 1 || (comm->allgather && comm->allreduce_sum),
                       "mispec_ctx_set_comm: collectives missing");
        MISPEC_REQUIRE((comm->allgather == nullptr) == (comm->allreduce_sum == nullptr),
                       "mispec_ctx_set_comm: give both collectives or none");
        ctx->comm = *comm;
    });
}

// ---------------------------------------------------------------------------------------------
// Row partition: equal blocks, so that the all-gather is a plain equal-count collective.
// The block is even so every shard's slice of a 16-byte-aligned vector stays 16-byte aligned.
// ---------------------------------------------------------------------------------------------
extern "C" int64_t mispec_shard_block(int64_t n, int world)
{
    if (world <= 1)
        return n;
    int64_t b = (n + world - 1) / world;
    return (b + 1) & ~int64_t(1);
}

extern "C" int mispec_shard_range(int64_t n, int world, int rank, int64_t* begin, int64_t* end)
{
    return guarded([&] {
        MISPEC_REQUIRE(begin && end && world >= 1 && rank >= 0 && rank < world && n >= 0, "mispec_shard_range: bad argument");
        const int64_t b = mispec_shard_block(n, world);
        int64_t lo = b * rank, hi = b * (rank + 1);
        if (lo > n)
            lo = n;
        if (hi > n)
            hi = n;
        *begin = lo;
        *end = hi;
    });
}

// ---------------------------------------------------------------------------------------------
// RCCL communicator (dlopen)
// ---------------------------------------------------------------------------------------------
namespace {
struct RcclApi
{
    void* handle = nullptr;
    ncclResult_t (*GetUniqueId)(ncclUniqueId*) = nullptr;
    ncclResult_t (*CommInitRank)(ncclComm_t*, int, ncclUniqueId, int) = nullptr;
    ncclResult_t (*CommDestroy)(ncclComm_t) = nullptr;
    ncclResult_t (*AllGather)(const void*, void*, size_t, ncclDataType_t, ncclComm_t, hipStream_t) = nullptr;
    ncclResult_t (*AllReduce)(const void*, void*, size_t, ncclDataType_t, ncclRedOp_t, ncclComm_t, hipStream_t) = nullptr;
    ncclResult_t (*Send)(const void*, size_t, ncclDataType_t, int, ncclComm_t, hipStream_t) = nullptr;
    ncclResult_t (*Recv)(void*, size_t, ncclDataType_t, int, ncclComm_t, hipStream_t) = nullptr;
    ncclResult_t (*GroupStart)() = nullptr;
    ncclResult_t (*GroupEnd)() = nullptr;
    const char* (*GetErrorString)(ncclResult_t) = nullptr;
};

RcclApi& rccl()
{
    static RcclApi api;
    static std::once_flag once;
    std::call_once(once, [] {
        const char* names[] = {"librccl.so.1", "librccl.so", "/opt/rocm/lib/librccl.so.1"};
        for (const char* nm : names)
        {
            api.handle = dlopen(nm, RTLD_NOW | RTLD_GLOBAL);
            if (api.handle)
                break;
        }
        if (!api.handle)
            return;
        api.GetUniqueId = reinterpret_cast<decltype(api.GetUniqueId)>(dlsym(api.handle, "ncclGetUniqueId"));
        api.CommInitRank = reinterpret_cast<decltype(api.CommInitRank)>(dlsym(api.handle, "ncclCommInitRank"));
        api.CommDestroy = reinterpret_cast<decltype(api.CommDestroy)>(dlsym(api.handle, "ncclCommDestroy"));
        api.AllGather = reinterpret_cast<decltype(api.AllGather)>(dlsym(api.handle, "ncclAllGather"));
        api.AllReduce = reinterpret_cast<decltype(api.AllReduce)>(dlsym(api.handle, "ncclAllReduce"));
        api.GetErrorString = reinterpret_cast<decltype(api.GetErrorString)>(dlsym(api.handle, "ncclGetErrorString"));
        api.Send = reinterpret_cast<decltype(api.Send)>(dlsym(api.handle, "ncclSend"));
        api.Recv = reinterpret_cast<decltype(api.Recv)>(dlsym(api.handle, "ncclRecv"));
        api.GroupStart = reinterpret_cast<decltype(api.GroupStart)>(dlsym(api.handle, "ncclGroupStart"));
        api.GroupEnd = reinterpret_cast<decltype(api.GroupEnd)>(dlsym(api.handle, "ncclGroupEnd"));
    });
    if (!api.handle || !api.GetUniqueId || !api.CommInitRank || !api.AllGather || !api.AllReduce)
        throw Error(MISPEC_ERUNTIME, "RCCL is not available (dlopen librccl.so.1 failed)");
    return api;
}

void rccl_check(ncclResult_t r, const char* what)
{
    if (r != ncclSuccess)
        throw Error(MISPEC_ERUNTIME, std::string(what) + ": " + (rccl().GetErrorString ? rccl().GetErrorString(r) : "RCCL error"));
}

struct RcclComm
{
    ncclComm_t comm = nullptr;
    int rank = 0, world = 1;
};

int rccl_allgather(void* user, const double* send, double* recv, int64_t count, void* stream)
{
    return guarded([&] {
        auto* c = static_cast<RcclComm*>(user);
        rccl_check(rccl().AllGather(send, recv, size_t(count), ncclDouble, c->comm, static_cast<hipStream_t>(stream)),
                   "ncclAllGather");
    });
}
int rccl_allreduce(void* user, double* buf, int64_t count, void* stream)
{
    return guarded([&] {
        auto* c = static_cast<RcclComm*>(user);
        rccl_check(rccl().AllReduce(buf, buf, size_t(count), ncclDouble, ncclSum, c->comm, static_cast<hipStream_t>(stream)),
                   "ncclAllReduce");
    });
}
// Neighbour exchange: every non-empty range is one ncclSend / ncclRecv of a group, so the transfers to all
// peers run concurrently on their own xGMI links.
int rccl_exchange(void* user, const double* send, const int64_t* send_off, const int64_t* send_count, double* recv,
                  const int64_t* recv_off, const int64_t* recv_count, void* stream)
{
    return guarded([&] {
        auto* c = static_cast<RcclComm*>(user);
        hipStream_t s = static_cast<hipStream_t>(stream);
        rccl_check(rccl().GroupStart(), "ncclGroupStart");
        for (int p = 0; p < c->world; p++)
        {
            if (p == c->rank)
                continue;
            if (send_count[p] > 0)
                rccl_check(rccl().Send(send + send_off[p], size_t(send_count[p]), ncclDouble, p, c->comm, s), "ncclSend");
            if (recv_count[p] > 0)
                rccl_check(rccl().Recv(recv + recv_off[p], size_t(recv_count[p]), ncclDouble, p, c->comm, s), "ncclRecv");
        }
        rccl_check(rccl().GroupEnd(), "ncclGroupEnd");
    });
}
void rccl_free(void* p)
{
    auto* c = static_cast<RcclComm*>(p);
    if (c->comm)
        (void) rccl().CommDestroy(c->comm);
    delete c;
}
}  // namespace

extern "C" int mispec_rccl_unique_id(char out[128])
{
    return guarded([&] {
        MISPEC_REQUIRE(out, "mispec_rccl_unique_id: out is NULL");
        static_assert(sizeof(ncclUniqueId) == 128, "ncclUniqueId is 128 bytes");
        ncclUniqueId id;
        rccl_check(rccl().GetUniqueId(&id), "ncclGetUniqueId");
        std::memcpy(out, &id, 128);
    });
}

extern "C" int mispec_ctx_set_comm_rccl(mispec_ctx* ctx, int rank, int world, const char unique_id[128])
{
    return guarded([&] {
        MISPEC_REQUIRE(ctx && unique_id, "mispec_ctx_set_comm_rccl: NULL argument");
        MISPEC_REQUIRE(world >= 1 && rank >= 0 && rank < world, "mispec_ctx_set_comm_rccl: need 0 <= rank < world");
        ctx->make_current();
        ncclUniqueId id;
        std::memcpy(&id, unique_id, 128);
        auto* c = new RcclComm();
        try
        {
            rccl_check(rccl().CommInitRank(&c->comm, world, id, rank), "ncclCommInitRank");
        }
        catch (...)
        {
            delete c;
            throw;
        }
        if (ctx->comm_owner && ctx->comm_owner_free)
            ctx->comm_owner_free(ctx->comm_owner);
        ctx->comm_owner = c;
        ctx->comm_owner_free = rccl_free;
        c->rank = rank;
        c->world = world;
        const bool p2p = rccl().Send && rccl().Recv && rccl().GroupStart && rccl().GroupEnd;
        ctx->comm = mispec_comm{rank, world, rccl_allgather, rccl_allreduce, c, p2p ? rccl_exchange : nullptr};
    });
}

// ---------------------------------------------------------------------------------------------
// Loopback communicator: `world` host threads of one process, typically all on the same device.
// Used to exercise the sharded code path (partitioning, all-gather assembly, reductions) on a
// 1-GPU box.  Rendezvous through a mutex/condvar barrier; data moves with device-to-device copies.
// Correctness over speed: every collective drains the caller's stream first.
// ---------------------------------------------------------------------------------------------
struct mispec_loopback
{
    int world = 1;
    std::mutex mu;
    std::condition_variable cv;
    int arrived = 0;
    uint64_t generation = 0;
    std::vector<const double*> send;
    std::vector<const int64_t*> send_off, send_count;
    std::vector<double*> bufs;
    std::vector<int> devices;
    struct Endpoint
    {
        mispec_loopback* grp;
        int rank;
    };
    std::vector<Endpoint> endpoints;

    void barrier()
    {
        std::unique_lock<std::mutex> lk(mu);
        const uint64_t gen = generation;
        if (++arrived == world)
        {
            arrived = 0;
            generation++;
            cv.notify_all();
        }
        else
            cv.wait(lk, [&] { return generation != gen; });
    }
};

namespace {
int loopback_allgather(void* user, const double* send, double* recv, int64_t count, void* stream)
{
    return guarded([&] {
        auto* ep = static_cast<mispec_loopback::Endpoint*>(user);
        mispec_loopback* g = ep->grp;
        hipStream_t s = static_cast<hipStream_t>(stream);
        MISPEC_HIP(hipStreamSynchronize(s));
        g->send[ep->rank] = send;
        g->barrier();
        for (int r = 0; r < g->world; r++)
            MISPEC_HIP(hipMemcpyAsync(recv + int64_t(r) * count, g->send[r], size_t(count) * sizeof(double),
                                      hipMemcpyDeviceToDevice, s));
        MISPEC_HIP(hipStreamSynchronize(s));
        g->barrier();
    });
}
// pull model: every rank publishes its send buffer and tables, then copies what the peers address to it
int loopback_exchange(void* user, const double* send, const int64_t* send_off, const int64_t* send_count, double* recv,
                      const int64_t* recv_off, const int64_t* recv_count, void* stream)
{
    return guarded([&] {
        auto* ep = static_cast<mispec_loopback::Endpoint*>(user);
        mispec_loopback* g = ep->grp;
        hipStream_t s = static_cast<hipStream_t>(stream);
        MISPEC_HIP(hipStreamSynchronize(s));
        g->send[ep->rank] = send;
        g->send_off[ep->rank] = send_off;
        g->send_count[ep->rank] = send_count;
        g->barrier();
        for (int r = 0; r < g->world; r++)
        {
            if (r == ep->rank || recv_count[r] <= 0)
                continue;
            MISPEC_REQUIRE(g->send_count[r][ep->rank] == recv_count[r], "loopback exchange: send/recv counts disagree");
            MISPEC_HIP(hipMemcpyAsync(recv + recv_off[r], g->send[r] + g->send_off[r][ep->rank],
                                      size_t(recv_count[r]) * sizeof(double), hipMemcpyDeviceToDevice, s));
        }
        MISPEC_HIP(hipStreamSynchronize(s));
        g->barrier();
    });
}
int loopback_allreduce(void* user, double* buf, int64_t count, void* stream)
{
    return guarded([&] {
        auto* ep = static_cast<mispec_loopback::Endpoint*>(user);
        mispec_loopback* g = ep->grp;
        hipStream_t s = static_cast<hipStream_t>(stream);
        MISPEC_HIP(hipStreamSynchronize(s));
        g->bufs[ep->rank] = buf;
        g->barrier();
        std::vector<double> acc(size_t(count), 0.0), tmp(static_cast<size_t>(count));
        for (int r = 0; r < g->world; r++)  // fixed rank order => identical result on every rank
        {
            MISPEC_HIP(hipMemcpy(tmp.data(), g->bufs[r], size_t(count) * sizeof(double), hipMemcpyDeviceToHost));
            for (int64_t i = 0; i < count; i++)
                acc[size_t(i)] += tmp[size_t(i)];
        }
        g->barrier();  // everybody has read every buffer
        MISPEC_HIP(hipMemcpy(buf, acc.data(), size_t(count) * sizeof(double), hipMemcpyHostToDevice));
        g->barrier();
    });
}
}  // namespace

extern "C" int mispec_loopback_create(int world, mispec_loopback** out)
{
    return guarded([&] {
        MISPEC_REQUIRE(out && world >= 1, "mispec_loopback_create: bad argument");
        auto* g = new mispec_loopback();
        g->world = world;
        g->send.assign(size_t(world), nullptr);
        g->send_off.assign(size_t(world), nullptr);
        g->send_count.assign(size_t(world), nullptr);
        g->bufs.assign(size_t(world), nullptr);
        g->endpoints.resize(size_t(world));
        for (int r = 0; r < world; r++)
            g->endpoints[size_t(r)] = {g, r};
        *out = g;
    });
}

extern "C" int mispec_loopback_attach(mispec_loopback* grp, mispec_ctx* ctx, int rank)
{
    return guarded([&] {
        MISPEC_REQUIRE(grp && ctx && rank >= 0 && rank < grp->world, "mispec_loopback_attach: bad argument");
        ctx->comm = mispec_comm{rank, grp->world, loopback_allgather, loopback_allreduce, &grp->endpoints[size_t(rank)],
                                loopback_exchange};
    });
}

extern "C" int mispec_loopback_destroy(mispec_loopback* grp)
{
    return guarded([&] { delete grp; });
}
