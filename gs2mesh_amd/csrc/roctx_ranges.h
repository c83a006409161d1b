// roctx_ranges.h -- optional rocTX ranges around the stages of the hot path (tracing).
// With GS2M_ROCTX=1 in the environment every stage launch is bracketed by roctxRangePushA / roctxRangePop
// ("gs2m:project", "gs2m:blend", "gs2m:tsdf_integrate", ...), so `rocprofv3 --marker-trace --kernel-trace` shows the
// host-side stage structure next to the kernels.  libroctx64.so is looked up at run time (dlopen): the library has no link
// dependency on the profiler, and without the variable the ranges cost one predictable branch.
#pragma once
#include <dlfcn.h>
#include <stdlib.h>

struct Gs2mRoctxApi {
    int (*push)(const char*) = nullptr;
    int (*pop)() = nullptr;
};
static inline const Gs2mRoctxApi& gs2m_roctx_api() {
    static const Gs2mRoctxApi api = [] {
        Gs2mRoctxApi a;
        const char* e = getenv("GS2M_ROCTX");
        if (e && e[0] && e[0] != '0') {
            void* h = dlopen("libroctx64.so", RTLD_NOW | RTLD_GLOBAL);
            if (!h) h = dlopen("libroctx64.so.4", RTLD_NOW | RTLD_GLOBAL);
            if (h) {
                a.push = (int (*)(const char*))dlsym(h, "roctxRangePushA");
                a.pop = (int (*)())dlsym(h, "roctxRangePop");
                if (!a.push || !a.pop) a.push = nullptr, a.pop = nullptr;
            }
        }
        return a;
    }();
    return api;
}
struct Gs2mRange {  // RAII
    bool on;
    explicit Gs2mRange(const char* name) : on(gs2m_roctx_api().push != nullptr) {
        if (on) (void)gs2m_roctx_api().push(name);
    }
    ~Gs2mRange() {
        if (on) (void)gs2m_roctx_api().pop();
    }
    Gs2mRange(const Gs2mRange&) = delete;
    Gs2mRange& operator=(const Gs2mRange&) = delete;
};
