// Shared host-side plumbing of libmispec.so: error propagation across the C ABI,
// the context object, small RAII helpers.  gfx950 / ROCm only — no CUDA shims.
#pragma once

#include <hip/hip_runtime.h>

#include <cstdint>
#include <cstdio>
#include <cstdlib>
#include <new>
#include <functional>
#include <utility>
#include <stdexcept>
#include <string>
#include <vector>

#include "../../include/mispec.h"
#include "../../include/mispec_extras.h"  // the library implements both headers

namespace mispec {

// Error carried through the library and turned into (code, thread-local message) at the C boundary.
struct Error : std::runtime_error
{
    int code;
    Error(int c, const std::string& m) : std::runtime_error(m), code(c) {}
};

void set_last_error(const std::string& msg);

#define MISPEC_HIP(expr)                                                                                   \
    do                                                                                                     \
    {                                                                                                      \
        hipError_t _e = (expr);                                                                            \
        if (_e != hipSuccess)                                                                              \
            throw ::mispec::Error(MISPEC_ERUNTIME, std::string(#expr) + ": " + hipGetErrorString(_e) +     \
                                                       " (" __FILE__ ":" + std::to_string(__LINE__) + ")"); \
    } while (0)

#define MISPEC_REQUIRE(cond, msg)                       \
    do                                                  \
    {                                                   \
        if (!(cond))                                    \
            throw ::mispec::Error(MISPEC_EINVAL, msg);  \
    } while (0)

// Wrap the body of an extern "C" entry point.
template <typename F>
int guarded(F&& f) noexcept
{
    try
    {
        f();
        return MISPEC_OK;
    }
    catch (const Error& e)
    {
        set_last_error(e.what());
        return e.code;
    }
    catch (const std::invalid_argument& e)
    {
        set_last_error(e.what());
        return MISPEC_EINVAL;
    }
    catch (const std::logic_error& e)
    {
        set_last_error(e.what());
        return MISPEC_ELOGIC;
    }
    catch (const std::exception& e)
    {
        set_last_error(e.what());
        return MISPEC_ERUNTIME;
    }
    catch (...)
    {
        set_last_error("unknown error");
        return MISPEC_ERUNTIME;
    }
}

// Device buffer owned by a handle.
template <typename T>
struct DevBuf
{
    T* p = nullptr;
    size_t n = 0;
    DevBuf() {}
    DevBuf(const DevBuf&) = delete;
    DevBuf& operator=(const DevBuf&) = delete;
    ~DevBuf() { release(); }
    void alloc(size_t count)
    {
        release();
        if (count)
            MISPEC_HIP(hipMalloc(reinterpret_cast<void**>(&p), count * sizeof(T)));
        n = count;
    }
    void release()
    {
        if (p)
            (void) hipFree(p);
        p = nullptr;
        n = 0;
    }
    void swap(DevBuf& o)
    {
        std::swap(p, o.p);
        std::swap(n, o.n);
    }
};

// Pinned host buffer (D2H targets of the per-step scalars).
template <typename T>
struct PinnedBuf
{
    T* p = nullptr;
    size_t n = 0;
    PinnedBuf() {}
    PinnedBuf(const PinnedBuf&) = delete;
    PinnedBuf& operator=(const PinnedBuf&) = delete;
    ~PinnedBuf()
    {
        if (p)
            (void) hipHostFree(p);
    }
    void alloc(size_t count)
    {
        if (p)
            (void) hipHostFree(p);
        p = nullptr;
        MISPEC_HIP(hipHostMalloc(reinterpret_cast<void**>(&p), count * sizeof(T), hipHostMallocDefault));
        n = count;
    }
};

inline int64_t round_up(int64_t x, int64_t m) { return (x + m - 1) / m * m; }

// Tuning switches and test hooks: ONE table, set through mispec_set_option(name, value) (include/mispec.h lists the names).
// A name that has not been set falls back to the environment variable MISPEC_<NAME IN UPPER CASE> — kept as the test-only
// override of the A/B tools and the parity tests (VERDICT r05 hygiene item: no getenv() scattered through the kernels' files).
// Returns nullptr when neither is present.  The pointer stays valid until the option is set again.
const char* option(const char* name);
int option_int(const char* name, int dflt);
bool option_is(const char* name, const char* value);

// malloc for the gigabyte-sized host arrays of the ingest stages (free with std::free).  Throws std::bad_alloc.
void* big_host_alloc(size_t bytes);

// Host array whose storage is NOT value-initialised (std::vector<T>::resize zero-fills on one thread and faults every page
// in there; the ingest stages fill gigabyte-sized arrays from many threads instead).  Trivially copyable T only.
template <typename T>
struct RawVec
{
    T* p = nullptr;
    size_t n = 0;
    RawVec() {}
    RawVec(const RawVec&) = delete;
    RawVec& operator=(const RawVec&) = delete;
    RawVec(RawVec&& o) noexcept : p(o.p), n(o.n) { o.p = nullptr; o.n = 0; }
    RawVec& operator=(RawVec&& o) noexcept
    {
        if (this != &o)
        {
            std::free(p);
            p = o.p;
            n = o.n;
            o.p = nullptr;
            o.n = 0;
        }
        return *this;
    }
    ~RawVec() { std::free(p); }
    void resize_uninitialized(size_t count)
    {
        std::free(p);
        p = count ? static_cast<T*>(big_host_alloc(count * sizeof(T))) : nullptr;
        n = count;
    }
    size_t size() const { return n; }
    T* data() { return p; }
    const T* data() const { return p; }
    T* begin() { return p; }
    T& operator[](size_t i) { return p[i]; }
    const T& operator[](size_t i) const { return p[i]; }
};

// Host threads of the ingest stages (format builders, triangle mirroring, validation passes): the machine's hardware threads,
// at most 64.  Every parallel stage produces the bytes the serial loop would.
int ingest_threads();
// fn(t, begin, end) over `parts` contiguous, nearly equal pieces of [0, n), one std::thread each (inline when parts <= 1);
// an exception thrown by a piece is rethrown on the calling thread.
void parallel_ranges(int64_t n, int parts, const std::function<void(int, int64_t, int64_t)>& fn);

}  // namespace mispec

// The context: one device, one stream, optional communicator for the row-sharded path.
struct mispec_ctx
{
    int device = 0;
    hipStream_t stream = nullptr;
    bool own_stream = false;
    int num_cu = 256;
    // communicator (world == 1: no collectives are ever called)
    mispec_comm comm{0, 1, nullptr, nullptr, nullptr, nullptr};
    void* comm_owner = nullptr;                 // built-in communicator state to free with the ctx
    void (*comm_owner_free)(void*) = nullptr;

    void make_current() const { MISPEC_HIP(hipSetDevice(device)); }
    int rank() const { return comm.rank; }
    int world() const { return comm.world; }
};
