// Multi-GPU modular reduce behind the C ABI (include/sda_hip.h "Cross-GPU modular reduction"): one RCCL communicator
// per process/GPU, the exchange of comm_plan.hpp over ncclSend/ncclRecv (point-to-point over the xGMI mesh) and the
// exact modular sum on the device.  A host language needs nothing but these entry points and some way to hand the
// 128-byte unique id from rank 0 to the other ranks.
//
// RCCL is bound at run time (dlopen of librccl.so.1, reusing a copy the process has already loaded): single-GPU users
// of libsda_hip.so never map RCCL, and a host process that already carries an RCCL (e.g. PyTorch's) does not get a
// second one.
#include "../../include/sda_hip.h"

#include <dlfcn.h>
#include <hip/hip_runtime_api.h>
#include <stdlib.h>
#include <string.h>

#include <new>

#include "capi_internal.hpp"
#include "comm_plan.hpp"
#include "kernels.hpp"

using namespace sda;

namespace {

// the subset of rccl.h this file uses (RCCL 2.x ABI: ncclUniqueId is 128 opaque bytes passed by value, ncclInt64 = 4)
struct NcclUniqueId { char internal[SDA_COMM_ID_BYTES]; };
typedef void* NcclComm;
typedef int NcclResult;
enum { kNcclInt64 = 4 };

struct Rccl {
    void* handle = nullptr;
    NcclResult (*GetUniqueId)(NcclUniqueId*) = nullptr;
    NcclResult (*CommInitRank)(NcclComm*, int, NcclUniqueId, int) = nullptr;
    NcclResult (*CommDestroy)(NcclComm) = nullptr;
    NcclResult (*GroupStart)() = nullptr;
    NcclResult (*GroupEnd)() = nullptr;
    NcclResult (*Send)(const void*, size_t, int, int, NcclComm, hipStream_t) = nullptr;
    NcclResult (*Recv)(void*, size_t, int, int, NcclComm, hipStream_t) = nullptr;
    const char* (*GetErrorString)(NcclResult) = nullptr;
    NcclResult (*GetVersion)(int*) = nullptr;      // optional (diagnostics only)
};

Rccl g_rccl;

int load_rccl() {
    if (g_rccl.handle) return SDA_OK;
    static const char* names[] = {"librccl.so.1", "librccl.so", "/opt/rocm/lib/librccl.so.1"};
    void* h = nullptr;
    for (const char* n : names)
        if ((h = dlopen(n, RTLD_NOW | RTLD_NOLOAD))) break;          // a copy the process already has
    for (size_t i = 0; !h && i < sizeof names / sizeof *names; ++i) h = dlopen(names[i], RTLD_NOW | RTLD_LOCAL);
    if (!h) return capi_fail(SDA_ERR_COMM, "RCCL is not loadable: %s", dlerror());
    Rccl r;
    r.handle = h;
#define BIND(field, sym)                                                                                 \
    *reinterpret_cast<void**>(&r.field) = dlsym(h, sym);                                                 \
    if (!r.field) { dlclose(h); return capi_fail(SDA_ERR_COMM, "RCCL symbol %s is missing", sym); }
    BIND(GetUniqueId, "ncclGetUniqueId")
    BIND(CommInitRank, "ncclCommInitRank")
    BIND(CommDestroy, "ncclCommDestroy")
    BIND(GroupStart, "ncclGroupStart")
    BIND(GroupEnd, "ncclGroupEnd")
    BIND(Send, "ncclSend")
    BIND(Recv, "ncclRecv")
    BIND(GetErrorString, "ncclGetErrorString")
#undef BIND
    *reinterpret_cast<void**>(&r.GetVersion) = dlsym(h, "ncclGetVersion");
    g_rccl = r;
    return SDA_OK;
}

int nccl_fail(const char* what, NcclResult r) {
    return capi_fail(SDA_ERR_COMM, "%s failed: %s", what, g_rccl.GetErrorString ? g_rccl.GetErrorString(r) : "?");
}

struct RcclTransport : Transport {
    NcclComm comm;
    hipStream_t stream;
    int group_start() override { NcclResult r = g_rccl.GroupStart(); return r ? nccl_fail("ncclGroupStart", r) : 0; }
    int send(const int64_t* buf, size_t count, int peer) override {
        NcclResult r = g_rccl.Send(buf, count, kNcclInt64, peer, comm, stream);
        return r ? nccl_fail("ncclSend", r) : 0;
    }
    int recv(int64_t* buf, size_t count, int peer) override {
        NcclResult r = g_rccl.Recv(buf, count, kNcclInt64, peer, comm, stream);
        return r ? nccl_fail("ncclRecv", r) : 0;
    }
    int group_end() override { NcclResult r = g_rccl.GroupEnd(); return r ? nccl_fail("ncclGroupEnd", r) : 0; }
};

struct HipReducer : Reducer {
    ModParams mod;
    hipStream_t stream;
    int modsum(const int64_t* in, size_t parts, size_t stride, size_t len, int64_t* out) override {
        hipError_t e = launch_modsum_parts(in, parts, stride, len, mod, out, stream);
        return e == hipSuccess ? 0 : capi_fail(SDA_ERR_HIP, "modular sum kernel failed: %s", hipGetErrorString(e));
    }
};

}  // namespace

struct sda_comm {
    NcclComm comm = nullptr;
    int rank = 0, world = 1, device = 0;
    void* scratch = nullptr;       // (world + 1) * seg elements, grow-only
    size_t scratch_elems = 0;
};

extern "C" int sda_comm_unique_id(uint8_t id[SDA_COMM_ID_BYTES]) {
    if (!id) return capi_fail(SDA_ERR_INVALID_ARGUMENT, "id is NULL");
    if (sda_device_count() == 0) return capi_fail(SDA_ERR_NO_DEVICE, "%s", sda_strerror(SDA_ERR_NO_DEVICE));
    if (int st = load_rccl()) return st;
    NcclUniqueId u;
    if (NcclResult r = g_rccl.GetUniqueId(&u)) return nccl_fail("ncclGetUniqueId", r);
    memcpy(id, u.internal, SDA_COMM_ID_BYTES);
    return SDA_OK;
}

extern "C" int sda_comm_init(const uint8_t id[SDA_COMM_ID_BYTES], int rank, int world, sda_comm_t** out) {
    if (!out) return capi_fail(SDA_ERR_INVALID_ARGUMENT, "out is NULL");
    *out = nullptr;
    if (!id) return capi_fail(SDA_ERR_INVALID_ARGUMENT, "id is NULL");
    if (world < 1 || rank < 0 || rank >= world) return capi_fail(SDA_ERR_INVALID_ARGUMENT, "rank %d / world %d", rank, world);
    if (sda_device_count() == 0) return capi_fail(SDA_ERR_NO_DEVICE, "%s", sda_strerror(SDA_ERR_NO_DEVICE));
    if (int st = load_rccl()) return st;
    sda_comm* c = new (std::nothrow) sda_comm();
    if (!c) return capi_fail(SDA_ERR_ALLOC, "out of memory");
    c->rank = rank; c->world = world;
    if (hipGetDevice(&c->device) != hipSuccess) { delete c; return capi_fail(SDA_ERR_HIP, "hipGetDevice failed"); }
    NcclUniqueId u;
    memcpy(u.internal, id, SDA_COMM_ID_BYTES);
    if (NcclResult r = g_rccl.CommInitRank(&c->comm, world, u, rank)) { delete c; return nccl_fail("ncclCommInitRank", r); }
    *out = c;
    return SDA_OK;
}

extern "C" void sda_comm_free(sda_comm_t* c) {
    if (!c) return;
    (void)hipSetDevice(c->device);
    if (c->scratch) (void)hipFree(c->scratch);
    if (c->comm && g_rccl.CommDestroy) (void)g_rccl.CommDestroy(c->comm);
    delete c;
}

extern "C" int sda_comm_rccl_version(void) {
    int v = 0;
    if (!g_rccl.handle || !g_rccl.GetVersion || g_rccl.GetVersion(&v)) return 0;
    return v;
}

extern "C" int sda_comm_rank(const sda_comm_t* c) { return c ? c->rank : -1; }
extern "C" int sda_comm_world(const sda_comm_t* c) { return c ? c->world : 0; }
extern "C" int sda_comm_device(const sda_comm_t* c) { return c ? c->device : -1; }

extern "C" int sda_modular_allreduce_dev(sda_comm_t* c, int64_t modulus, const int64_t* d_partial, size_t len,
                                         int64_t* d_out, void* stream) {
    if (!c) return capi_fail(SDA_ERR_INVALID_ARGUMENT, "communicator is NULL");
    HipReducer red;
    if (int st = capi_make_mod(modulus, red.mod)) return st;
    if (len == 0) return SDA_OK;
    if (!d_partial || !d_out) return capi_fail(SDA_ERR_INVALID_ARGUMENT, "NULL device pointer");
    if (hipSetDevice(c->device) != hipSuccess) return capi_fail(SDA_ERR_HIP, "hipSetDevice failed");
    hipStream_t s = reinterpret_cast<hipStream_t>(stream);
    red.stream = s;
    // a single rank needs no exchange: canonicalise through the same kernel (SDA_FORCE_COLLECTIVES=1 sends the one
    // slice to itself through RCCL anyway - the 1-GPU test of the RCCL path)
    if (c->world == 1 && !sda::knob(sda::KNOB_FORCE_COLLECTIVES)) return red.modsum(d_partial, 1, len, len, d_out);
    const SlicePlan pl(c->world, len);
    const size_t need = ((size_t)c->world + 1) * pl.seg;
    if (need > c->scratch_elems) {
        if (c->scratch) { (void)hipFree(c->scratch); c->scratch = nullptr; c->scratch_elems = 0; }
        if (hipMalloc(&c->scratch, need * 8) != hipSuccess) return capi_fail(SDA_ERR_ALLOC, "hipMalloc(%zu) failed", need * 8);
        c->scratch_elems = need;
    }
    int64_t* recv = static_cast<int64_t*>(c->scratch);
    RcclTransport tr;
    tr.comm = c->comm; tr.stream = s;
    return modular_allreduce_plan(tr, red, c->rank, c->world, d_partial, len, recv, recv + (size_t)c->world * pl.seg, d_out);
}
