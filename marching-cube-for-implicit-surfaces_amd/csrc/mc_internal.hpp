// mc_internal.hpp -- the few hooks the library's own translation units share beside the public C ABI (include/mc_hip.h).
// Hidden visibility: none of this is exported (tests/test_abi.py checks that `nm -D` shows mc_* of the header only).
#pragma once
#include "../../include/mc_hip.h"

// sets the calling thread's mc_last_error() text and returns `code`
int mc_internal_fail(int code, const char* fmt, ...) __attribute__((format(printf, 2, 3)));
// the result of the context's last sweep as the mc_copy_* calls see it (null before the first sweep)
const mc_result* mc_internal_last(const mc_context* ctx);
// the HIP device the context lives on
int mc_internal_device(const mc_context* ctx);
// Marching::seed_mode is on for this context (mc_seed_mode)
bool mc_internal_seed_on(const mc_context* ctx);

// A host thread with a LARGE stack (64 MB).  The library's own threads run hiprtc (LLVM: deep recursion on the kernels'
// large functions) or whole sweeps that may; a default thread stack is as small as 2 MB when the process runs with an
// unlimited stack limit.  fn is deleted after it has run.  Returns false when the thread could not be started (fn is
// then still the caller's).
#include <functional>
#include <pthread.h>
struct McThread {
    pthread_t t{};
    bool started = false;
};
inline void* mc_thread_entry(void* arg) {
    std::function<void()>* fn = static_cast<std::function<void()>*>(arg);
    (*fn)();
    delete fn;
    return nullptr;
}
inline bool mc_thread_start(McThread& th, std::function<void()>* fn) {
    pthread_attr_t attr;
    if (pthread_attr_init(&attr) != 0) return false;
    (void)pthread_attr_setstacksize(&attr, (size_t)64 << 20);
    const int rc = pthread_create(&th.t, &attr, mc_thread_entry, fn);
    pthread_attr_destroy(&attr);
    th.started = rc == 0;
    return th.started;
}
inline void mc_thread_join(McThread& th) {
    if (th.started) (void)pthread_join(th.t, nullptr);
    th.started = false;
}
