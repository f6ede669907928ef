// fwx_guard.h -- the exception barrier of the C ABI (fwx.h: "no C++ exception or abort crosses this
// boundary").  Every extern "C" entry point that can allocate runs its body through guarded();
// an exception unwinding into a C or Haskell caller would be undefined behaviour.
//
// fail_point(): the test hook behind fwx_test_fail_after (fwx.h).  Entry points call it where they
// are about to allocate host memory; with a countdown armed on the calling thread the n-th call
// throws std::bad_alloc, which is how tests/test_gpu_multi.py proves that the barrier holds and
// that nothing leaks on the way out.  Unarmed it is one thread-local load.
#ifndef FWX_GUARD_H
#define FWX_GUARD_H

#include <new>

#include "fwx.h"

namespace fwxi {

inline thread_local int g_fail_countdown = 0;   // 0 = unarmed

inline void fail_point()
{
    if (g_fail_countdown > 0 && --g_fail_countdown == 0) throw std::bad_alloc();
}

template <typename F> int guarded(F &&body) noexcept
{
    try {
        return body();
    } catch (const std::bad_alloc &) {
        return FWX_ERR_OOM;
    } catch (...) {
        return FWX_ERR_INTERNAL;
    }
}

}  // namespace fwxi

#endif
