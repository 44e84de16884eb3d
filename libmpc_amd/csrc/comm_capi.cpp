// extern "C" boundary of the one collective on the path (include/mpcx.h, mpcx_comm_* / mpcx_allgather_u): the all-gather
// of the optimal controls u* after a sharded solve.  The batch shards as contiguous slices, one process
// per GPU, with nothing exchanged during the solve; afterwards every rank contributes its [rows x nu] block of doubles
// and receives all of them -- one RCCL ncclAllGather over xGMI, issued on the stream the solve kernels were launched on,
// so that it starts as soon as the last kernel retires and needs no host synchronisation in between.
//
// RCCL is bound at run time (dlopen of librccl.so.1): libmpcx.so stays loadable on a host without RCCL, and inside a
// PyTorch process the soname resolves to the copy PyTorch already loaded instead of a second one.
#include <dlfcn.h>
#include <hip/hip_runtime.h>

#include <cstring>
#include <mutex>
#include <string>

#include "../../include/mpcx.h"

namespace mpcx {
int capi_fail(int code, const std::string &msg);
}

namespace {

// the five entry points used, with RCCL's own signatures (rccl.h: ncclUniqueId is 128 opaque bytes passed by value,
// ncclComm_t an opaque pointer, ncclFloat64 = 8, ncclSuccess = 0)
struct UniqueId { char internal[MPCX_COMM_ID_BYTES]; };
using comm_t = void *;
struct Rccl {
    void *lib = nullptr;
    int (*GetUniqueId)(UniqueId *) = nullptr;
    int (*CommInitRank)(comm_t *, int, UniqueId, int) = nullptr;
    int (*CommDestroy)(comm_t) = nullptr;
    int (*AllGather)(const void *, void *, size_t, int, comm_t, hipStream_t) = nullptr;
    const char *(*GetErrorString)(int) = nullptr;
    std::string error;
};

Rccl &rccl()
{
    static Rccl r;
    static std::once_flag once;
    std::call_once(once, [] {
        const char *names[] = {"librccl.so.1", "librccl.so", "/opt/rocm/lib/librccl.so.1"};
        for (const char *n : names) {
            r.lib = dlopen(n, RTLD_NOW | RTLD_LOCAL);
            if (r.lib) break;
        }
        if (!r.lib) { r.error = std::string("RCCL not found: ") + dlerror(); return; }
        auto sym = [&](const char *n) { void *p = dlsym(r.lib, n); if (!p) r.error = std::string("RCCL lacks ") + n; return p; };
        r.GetUniqueId = reinterpret_cast<decltype(r.GetUniqueId)>(sym("ncclGetUniqueId"));
        r.CommInitRank = reinterpret_cast<decltype(r.CommInitRank)>(sym("ncclCommInitRank"));
        r.CommDestroy = reinterpret_cast<decltype(r.CommDestroy)>(sym("ncclCommDestroy"));
        r.AllGather = reinterpret_cast<decltype(r.AllGather)>(sym("ncclAllGather"));
        r.GetErrorString = reinterpret_cast<decltype(r.GetErrorString)>(sym("ncclGetErrorString"));
    });
    return r;
}

int rccl_fail(const char *what, int rc)
{
    Rccl &r = rccl();
    return mpcx::capi_fail(MPCX_E_DEVICE, std::string(what) + ": " + (r.GetErrorString ? r.GetErrorString(rc) : "RCCL error"));
}

}  // namespace

struct mpcx_comm {
    comm_t comm = nullptr;
    int rank = 0, world = 1, device = 0;
};

extern "C" {

int mpcx_comm_get_unique_id(void *id_out)
{
    if (!id_out) return mpcx::capi_fail(MPCX_E_INVALID, "null id buffer");
    Rccl &r = rccl();
    if (!r.error.empty()) return mpcx::capi_fail(MPCX_E_DEVICE, r.error);
    UniqueId id;
    const int rc = r.GetUniqueId(&id);
    if (rc != 0) return rccl_fail("ncclGetUniqueId", rc);
    std::memcpy(id_out, &id, sizeof id);
    return MPCX_OK;
}

int mpcx_comm_create(int device, int rank, int world, const void *id, mpcx_comm_t *out)
{
    if (!out || !id) return mpcx::capi_fail(MPCX_E_INVALID, "null argument");
    if (world < 1 || rank < 0 || rank >= world) return mpcx::capi_fail(MPCX_E_INVALID, "need 0 <= rank < world");
    Rccl &r = rccl();
    if (!r.error.empty()) return mpcx::capi_fail(MPCX_E_DEVICE, r.error);
    if (hipSetDevice(device) != hipSuccess) return mpcx::capi_fail(MPCX_E_DEVICE, "hipSetDevice failed: no usable HIP device");
    UniqueId uid;
    std::memcpy(&uid, id, sizeof uid);
    comm_t c = nullptr;
    const int rc = r.CommInitRank(&c, world, uid, rank);
    if (rc != 0) return rccl_fail("ncclCommInitRank", rc);
    auto *h = new mpcx_comm;
    h->comm = c; h->rank = rank; h->world = world; h->device = device;
    *out = h;
    return MPCX_OK;
}

int mpcx_comm_destroy(mpcx_comm_t c)
{
    if (!c) return MPCX_OK;
    Rccl &r = rccl();
    (void)hipSetDevice(c->device);
    if (c->comm && r.CommDestroy) (void)r.CommDestroy(c->comm);
    delete c;
    return MPCX_OK;
}

int mpcx_comm_rank(mpcx_comm_t c) { return c ? c->rank : -1; }
int mpcx_comm_world(mpcx_comm_t c) { return c ? c->world : -1; }

int mpcx_allgather_u(mpcx_comm_t c, const double *u_local, int rows_per_rank, int nu, double *u_all, void *stream)
{
    if (!c) return mpcx::capi_fail(MPCX_E_INVALID, "null communicator");
    if (rows_per_rank < 0 || nu < 1) return mpcx::capi_fail(MPCX_E_INVALID, "bad block shape");
    if (rows_per_rank == 0) return MPCX_OK;
    if (!u_local || !u_all) return mpcx::capi_fail(MPCX_E_INVALID, "null buffer");
    Rccl &r = rccl();
    if (hipSetDevice(c->device) != hipSuccess) return mpcx::capi_fail(MPCX_E_DEVICE, "hipSetDevice failed");
    const int rc = r.AllGather(u_local, u_all, (size_t)rows_per_rank * nu, /*ncclFloat64*/ 8, c->comm,
                               reinterpret_cast<hipStream_t>(stream));
    if (rc != 0) return rccl_fail("ncclAllGather", rc);
    return MPCX_OK;
}

}  // extern "C"
