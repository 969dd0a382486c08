// piqp_amd/csrc/common.hpp -- shared host-side plumbing for the HIP KKT backend (gfx950 only).
#pragma once

#include <hip/hip_runtime.h>

#include <atomic>
#include <cstdarg>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <ctime>
#include <mutex>
#include <set>
#include <string>
#include <utility>
#include <vector>

#include "../../include/piqp_amd.h"

namespace pq {

// thread-local error text behind pq_last_error_string()
inline std::string& last_error()
{
    static thread_local std::string s;
    return s;
}
inline int fail(int code, const char* fmt, ...)
{
    char buf[512];
    va_list ap;
    va_start(ap, fmt);
    vsnprintf(buf, sizeof(buf), fmt, ap);
    va_end(ap);
    last_error() = buf;
    return code;
}

struct HipError {
    hipError_t err;
    const char* what;
    const char* file;
    int line;
};

#define PQ_HIP(expr)                                                         \
    do {                                                                     \
        hipError_t e_ = (expr);                                              \
        if (e_ != hipSuccess) throw ::pq::HipError{e_, #expr, __FILE__, __LINE__}; \
    } while (0)

// "Once per device": hipFuncSetAttribute (dynamic LDS limits) applies to the CURRENT device only, while handles on several devices -- and host threads -- share
// the launch helpers' call sites.  `static PerDeviceOnce once; once([&] { ... });`
struct PerDeviceOnce {
    std::mutex mu;
    std::set<int> done;
    template <class F>
    void operator()(F&& f)
    {
        int dev = 0;
        PQ_HIP(hipGetDevice(&dev));
        std::lock_guard<std::mutex> lk(mu);
        if (done.count(dev)) return;
        f();
        done.insert(dev);
    }
};

// Waits for a stream: polls it for up to two milliseconds, then blocks.  One interior-point iteration reads three or four scalars back (factorisation status,
// finiteness of a solve: kkt_system.hpp:266,305), each a full drain of the stream; hipStreamSynchronize sleeps on an interrupt, and the wake-up was 30-150 us per
// read on the boxes of this pool (0.4 ms of a 3.4 ms step on a freshly booted one), the poll is a few microseconds.  PIQP_AMD_DEBUG=sync_block keeps the
// blocking wait (a host thread that must not spin).
const char* debug_token(const char* name);
inline void stream_wait(hipStream_t s)
{
    static const bool spin = debug_token("sync_block") == nullptr;
    if (spin) {
        timespec t0;
        clock_gettime(CLOCK_MONOTONIC, &t0);
        for (unsigned it = 0;; ++it) {
            const hipError_t e = hipStreamQuery(s);
            if (e == hipSuccess) {
                if (it > 0) (void)hipGetLastError();  // (the "not ready" answers of the polls before must not look like the error of a later launch check)
                return;
            }
            if (e != hipErrorNotReady) throw ::pq::HipError{e, "hipStreamQuery", __FILE__, __LINE__};
            if ((it & 15u) == 15u) {
                timespec t1;
                clock_gettime(CLOCK_MONOTONIC, &t1);
                if ((t1.tv_sec - t0.tv_sec) * 1000000000LL + (t1.tv_nsec - t0.tv_nsec) > 2000000LL) { (void)hipGetLastError(); break; }
            }
        }
    }
    PQ_HIP(hipStreamSynchronize(s));
}

// Every extern "C" entry wraps its body in this: no exception crosses the ABI.
template <class F>
inline int guarded(F&& f)
{
    try {
        return f();
    } catch (const HipError& e) {
        return fail(PQ_ERR_HIP, "%s failed: %s (%s:%d)", e.what, hipGetErrorString(e.err), e.file, e.line);
    } catch (const std::bad_alloc&) {
        return fail(PQ_ERR_NOMEM, "out of memory");
    } catch (const std::exception& e) {
        return fail(PQ_ERR_INVALID, "%s", e.what());
    } catch (...) {
        return fail(PQ_ERR_INVALID, "unknown error");
    }
}

// every hipMalloc / hipHostMalloc the library makes is counted (pq_debug_alloc_count): the reference's tests assert allocation-free factor / solve
// (fwd.hpp:44-52, PIQP_EIGEN_MALLOC_NOT_ALLOWED); here the same statement is "the counter does not move after *_create"
inline std::atomic<long long>& alloc_counter()
{
    static std::atomic<long long> c{0};
    return c;
}

// RAII device buffer (all allocation happens in constructors / create paths)
template <class T>
struct DBuf {
    T* p = nullptr;
    size_t n = 0;
    DBuf() = default;
    explicit DBuf(size_t count) { alloc(count); }
    DBuf(const DBuf&) = delete;
    DBuf& operator=(const DBuf&) = delete;
    DBuf(DBuf&& o) noexcept : p(o.p), n(o.n) { o.p = nullptr; o.n = 0; }
    DBuf& operator=(DBuf&& o) noexcept
    {
        if (this != &o) { release(); p = o.p; n = o.n; o.p = nullptr; o.n = 0; }
        return *this;
    }
    ~DBuf() { release(); }
    void alloc(size_t count)
    {
        release();
        n = count;
        if (count) { PQ_HIP(hipMalloc((void**)&p, count * sizeof(T))); ++alloc_counter(); }
    }
    void release()
    {
        if (p) (void)hipFree(p);
        p = nullptr;
        n = 0;
    }
    void zero(hipStream_t s) { if (n) PQ_HIP(hipMemsetAsync(p, 0, n * sizeof(T), s)); }
    size_t bytes() const { return n * sizeof(T); }
};

// pinned host staging buffer
template <class T>
struct HBuf {
    T* p = nullptr;
    size_t n = 0;
    HBuf() = default;
    explicit HBuf(size_t count) { alloc(count); }
    HBuf(const HBuf&) = delete;
    HBuf& operator=(const HBuf&) = delete;
    ~HBuf() { release(); }
    void alloc(size_t count)
    {
        release();
        n = count;
        if (count) { PQ_HIP(hipHostMalloc((void**)&p, count * sizeof(T), hipHostMallocDefault)); ++alloc_counter(); }
    }
    void release()
    {
        if (p) (void)hipHostFree(p);
        p = nullptr;
        n = 0;
    }
};

inline void copy_in(void* dst_dev, const void* src, size_t bytes, int src_mem, hipStream_t s)
{
    if (!bytes) return;
    PQ_HIP(hipMemcpyAsync(dst_dev, src, bytes, src_mem == PQ_MEM_DEVICE ? hipMemcpyDeviceToDevice : hipMemcpyHostToDevice, s));
}

inline int div_up(int a, int b) { return (a + b - 1) / b; }

// ONE debugging variable for the whole library: PIQP_AMD_DEBUG="token[=value],token,...".  Returns nullptr when the token is
// absent, "" when it is present without a value, its value otherwise.  Parsed once per process.  Tokens select measured-equivalent device
// schedules for the bitwise-consistency tests (tests/test_sparse_variants_gpu.py) or switch diagnostics on; none is needed for normal use.
inline const char* debug_token(const char* name)
{
    static const std::vector<std::pair<std::string, std::string>> toks = [] {
        std::vector<std::pair<std::string, std::string>> v;
        const char* e = std::getenv("PIQP_AMD_DEBUG");
        std::string s = e ? e : "";
        size_t i = 0;
        while (i < s.size()) {
            size_t j = s.find(',', i);
            if (j == std::string::npos) j = s.size();
            const std::string t = s.substr(i, j - i);
            const size_t eq = t.find('=');
            if (!t.empty()) v.emplace_back(eq == std::string::npos ? t : t.substr(0, eq), eq == std::string::npos ? std::string() : t.substr(eq + 1));
            i = j + 1;
        }
        return v;
    }();
    for (const auto& kv : toks) if (kv.first == name) return kv.second.c_str();
    return nullptr;
}

// hipEvent brackets around stages of a backend, accumulated lazily (events are read at query time,
// after the stream has been synchronised; nothing here blocks the timed region)
struct StageProfiler {
    static constexpr int NSTAGE = 6;  // 0..2 stages of a backend; 3..5 kernel-level brackets of the dense backend (level 2)
    bool enabled = false;
    int level = 0;  // 2: also bracket individual launches (adds event markers between dependent kernels: measurement passes only)
    struct Pair { hipEvent_t a, b; };
    std::vector<Pair> pending[NSTAGE], pool;
    ~StageProfiler()
    {
        for (auto& v : pending) for (auto& pr : v) { (void)hipEventDestroy(pr.a); (void)hipEventDestroy(pr.b); }
        for (auto& pr : pool) { (void)hipEventDestroy(pr.a); (void)hipEventDestroy(pr.b); }
    }
    Pair get()
    {
        if (!pool.empty()) { Pair p = pool.back(); pool.pop_back(); return p; }
        Pair p;
        PQ_HIP(hipEventCreate(&p.a));
        PQ_HIP(hipEventCreate(&p.b));
        return p;
    }
    int begin(int stage, hipStream_t s)
    {
        if (!enabled || (stage >= 3 && level < 2)) return -1;
        Pair p = get();
        PQ_HIP(hipEventRecord(p.a, s));
        pending[stage].push_back(p);
        return (int)pending[stage].size() - 1;
    }
    void end(int stage, int tok, hipStream_t s)
    {
        if (tok >= 0) PQ_HIP(hipEventRecord(pending[stage][tok].b, s));
    }
    void collect(int stage, hipStream_t s, double* total_ms, int* count)
    {
        stream_wait(s);
        double tot = 0.0;
        for (auto& pr : pending[stage]) {
            float ms = 0.f;
            PQ_HIP(hipEventElapsedTime(&ms, pr.a, pr.b));
            tot += ms;
            pool.push_back(pr);
        }
        *count = (int)pending[stage].size();
        *total_ms = tot;
        pending[stage].clear();
    }
};

}  // namespace pq
