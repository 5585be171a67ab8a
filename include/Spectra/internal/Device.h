// Glue between the header-only templates and the C ABI (include/mispec.h): error codes back to the
// exception types the reference throws, a shared default device context, RAII for handles.
#ifndef MISPEC_SPECTRA_DEVICE_H
#define MISPEC_SPECTRA_DEVICE_H

#include <cstdlib>
#include <memory>
#include <stdexcept>
#include <string>

#include "../../mispec.h"

namespace Spectra {
namespace internal {

// MISPEC_EINVAL -> std::invalid_argument (HermEigsBase.h:267-271, Arnoldi.h:148,209),
// MISPEC_ELOGIC -> std::logic_error (UpperHessenbergQR.h:207), anything else -> std::runtime_error.
inline void check(int rc)
{
    if (rc == MISPEC_OK)
        return;
    const std::string msg = mispec_last_error();
    if (rc == MISPEC_EINVAL)
        throw std::invalid_argument(msg);
    if (rc == MISPEC_ELOGIC)
        throw std::logic_error(msg);
    throw std::runtime_error(msg);
}

struct CtxDeleter
{
    void operator()(mispec_ctx* c) const { (void) mispec_ctx_destroy(c); }
};
using CtxPtr = std::shared_ptr<mispec_ctx>;

// One context per process for operators constructed straight from host data.  Device ordinal from
// MISPEC_DEVICE (default 0).  There is no CPU fallback: without a GPU this throws.
inline CtxPtr default_context()
{
    static std::weak_ptr<mispec_ctx> cached;
    CtxPtr p = cached.lock();
    if (!p)
    {
        int dev = 0;
        if (const char* e = std::getenv("MISPEC_DEVICE"))
            dev = std::atoi(e);
        mispec_ctx* raw = nullptr;
        check(mispec_ctx_create(dev, nullptr, &raw));
        p = CtxPtr(raw, CtxDeleter());
        cached = p;
    }
    return p;
}

// Non-owning wrapper of a context created elsewhere (e.g. by the C facade or a binding).
inline CtxPtr borrow_context(mispec_ctx* raw) { return CtxPtr(raw, [](mispec_ctx*) {}); }

}  // namespace internal
}  // namespace Spectra

#endif
