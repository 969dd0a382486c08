// piqp_amd/csrc/rccl_transport.cpp -- see rccl_transport.hpp
#include "rccl_transport.hpp"

#include <dlfcn.h>
#include <rccl/rccl.h>

#include <mutex>
#include <stdexcept>
#include <string>

namespace pq {
namespace rccl {

namespace {

struct Api {
    void* lib = nullptr;
    ncclResult_t (*GetUniqueId)(ncclUniqueId*) = nullptr;
    ncclResult_t (*CommInitRank)(ncclComm_t*, int, ncclUniqueId, int) = nullptr;
    ncclResult_t (*CommDestroy)(ncclComm_t) = nullptr;
    ncclResult_t (*AllReduce)(const void*, void*, size_t, ncclDataType_t, ncclRedOp_t, ncclComm_t, hipStream_t) = nullptr;
    ncclResult_t (*AllGather)(const void*, void*, size_t, ncclDataType_t, ncclComm_t, hipStream_t) = nullptr;
    const char* (*GetErrorString)(ncclResult_t) = nullptr;
    ncclResult_t (*CommCount)(const ncclComm_t, int*) = nullptr;
    ncclResult_t (*CommUserRank)(const ncclComm_t, int*) = nullptr;
    ncclResult_t (*CommCuDevice)(const ncclComm_t, int*) = nullptr;
};

const Api& api()
{
    static Api a;
    static std::once_flag once;
    static std::string err;
    std::call_once(once, [] {
        for (const char* name : {"librccl.so.1", "librccl.so", "/opt/rocm/lib/librccl.so.1"}) {
            a.lib = dlopen(name, RTLD_NOW | RTLD_GLOBAL);
            if (a.lib) break;
        }
        if (!a.lib) { err = std::string("cannot load librccl: ") + dlerror(); return; }
        auto sym = [&](const char* n) { void* p = dlsym(a.lib, n); if (!p) err = std::string("librccl lacks ") + n; return p; };
        a.GetUniqueId = reinterpret_cast<decltype(a.GetUniqueId)>(sym("ncclGetUniqueId"));
        a.CommInitRank = reinterpret_cast<decltype(a.CommInitRank)>(sym("ncclCommInitRank"));
        a.CommDestroy = reinterpret_cast<decltype(a.CommDestroy)>(sym("ncclCommDestroy"));
        a.AllReduce = reinterpret_cast<decltype(a.AllReduce)>(sym("ncclAllReduce"));
        a.AllGather = reinterpret_cast<decltype(a.AllGather)>(sym("ncclAllGather"));
        a.GetErrorString = reinterpret_cast<decltype(a.GetErrorString)>(sym("ncclGetErrorString"));
        a.CommCount = reinterpret_cast<decltype(a.CommCount)>(sym("ncclCommCount"));
        a.CommUserRank = reinterpret_cast<decltype(a.CommUserRank)>(sym("ncclCommUserRank"));
        a.CommCuDevice = reinterpret_cast<decltype(a.CommCuDevice)>(sym("ncclCommCuDevice"));
    });
    if (!err.empty()) throw std::runtime_error(err);
    return a;
}

void check(ncclResult_t r, const char* what)
{
    if (r != ncclSuccess) throw std::runtime_error(std::string(what) + ": " + api().GetErrorString(r));
}

}  // namespace

struct Comm {
    ncclComm_t c = nullptr;
    int device = 0;
};

void unique_id(unsigned char out[UNIQUE_ID_BYTES])
{
    static_assert(sizeof(ncclUniqueId) == UNIQUE_ID_BYTES, "ncclUniqueId size");
    ncclUniqueId id;
    check(api().GetUniqueId(&id), "ncclGetUniqueId");
    std::memcpy(out, &id, UNIQUE_ID_BYTES);
}

Comm* comm_create(const unsigned char idb[UNIQUE_ID_BYTES], int rank, int world, int device)
{
    ncclUniqueId id;
    std::memcpy(&id, idb, UNIQUE_ID_BYTES);
    PQ_HIP(hipSetDevice(device));
    Comm* c = new Comm;
    c->device = device;
    ncclResult_t r = api().CommInitRank(&c->c, world, id, rank);
    if (r != ncclSuccess) { delete c; check(r, "ncclCommInitRank"); }
    return c;
}

void comm_destroy(Comm* c)
{
    if (!c) return;
    if (c->c) { (void)hipSetDevice(c->device); (void)api().CommDestroy(c->c); }
    delete c;
}

void comm_info(Comm* c, int out[3])
{
    check(api().CommCount(c->c, &out[0]), "ncclCommCount");
    check(api().CommUserRank(c->c, &out[1]), "ncclCommUserRank");
    check(api().CommCuDevice(c->c, &out[2]), "ncclCommCuDevice");
}

void all_reduce_sum(Comm* c, double* buf, size_t count, hipStream_t s)
{
    check(api().AllReduce(buf, buf, count, ncclFloat64, ncclSum, c->c, s), "ncclAllReduce");
}

void all_reduce_max(Comm* c, double* buf, size_t count, hipStream_t s)
{
    check(api().AllReduce(buf, buf, count, ncclFloat64, ncclMax, c->c, s), "ncclAllReduce(max)");
}

void all_gather(Comm* c, double* buf, size_t count_per_rank, int rank, hipStream_t s)
{
    check(api().AllGather(buf + (size_t)rank * count_per_rank, buf, count_per_rank, ncclFloat64, c->c, s), "ncclAllGather");
}

}  // namespace rccl
}  // namespace pq
