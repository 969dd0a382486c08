// piqp_amd/csrc/trace.hpp -- named ranges on the host timeline of the hot path, the counterpart of the reference's Tracy zones
// (include/piqp/utils/tracy.hpp:11-25: PIQP_TRACY_ZoneScopedN on every hot-path function, e.g. kkt_system.hpp:146,215,257, sparse/ldlt.hpp:109).
// Here the ranges are roctx ranges (roctxRangePushA / roctxRangePop), which rocprofv3 --marker-trace records next to the kernel trace.
// The roctx library is loaded with dlopen at the first zone of a process that sets PIQP_AMD_TRACE=1; without the variable a zone is one
// predictable branch, and the library keeps libamdhip64 as its only link-time dependency.
#pragma once

#include <dlfcn.h>

#include <cstdlib>

namespace pq {
namespace trace {

struct Api {
    int (*push)(const char*) = nullptr;
    int (*pop)() = nullptr;
    bool on = false;
};

inline const Api& api()
{
    static const Api a = [] {
        Api r;
        const char* e = std::getenv("PIQP_AMD_TRACE");
        if (!e || e[0] == '0' || e[0] == '\0') return r;
        for (const char* name : {"librocprofiler-sdk-roctx.so.1", "librocprofiler-sdk-roctx.so", "libroctx64.so.4", "libroctx64.so"}) {
            void* lib = dlopen(name, RTLD_NOW | RTLD_GLOBAL);
            if (!lib) continue;
            r.push = reinterpret_cast<int (*)(const char*)>(dlsym(lib, "roctxRangePushA"));
            r.pop = reinterpret_cast<int (*)()>(dlsym(lib, "roctxRangePop"));
            if (r.push && r.pop) { r.on = true; break; }
        }
        return r;
    }();
    return a;
}

// RAII range: PQ_ZONE("piqp_amd::KKTSystem::solve");
struct Zone {
    bool live;
    explicit Zone(const char* name) : live(api().on) { if (live) api().push(name); }
    ~Zone() { if (live) api().pop(); }
    Zone(const Zone&) = delete;
    Zone& operator=(const Zone&) = delete;
};

}  // namespace trace
}  // namespace pq

#define PQ_ZONE_CAT2(a, b) a##b
#define PQ_ZONE_CAT(a, b) PQ_ZONE_CAT2(a, b)
#define PQ_ZONE(name) ::pq::trace::Zone PQ_ZONE_CAT(pq_zone_, __LINE__)(name)
