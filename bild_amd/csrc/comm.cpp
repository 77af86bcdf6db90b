// The one collective of a multi-GPU AMIS step behind the C ABI (include/bild_amd.h, "several GPUs"): an all-gather of
// the ranks' log-likelihood shards over RCCL, one process per GPU, on the caller's HIP stream.  The reference has
// nothing here (bild/amis.py:732-733 declines to parallelise); SURVEY section 8e asks for exactly this exchange.
//
// RCCL is resolved at run time (dlopen): the library itself keeps depending on nothing but the HIP runtime, and a
// process that already has an RCCL loaded (PyTorch ships one beside its HIP runtime) keeps using that one.
#include <dlfcn.h>
#include <hip/hip_runtime_api.h>

#include <cstdio>
#include <cstring>
#include <mutex>
#include <string>

#include "../../include/bild_amd.h"

namespace {

typedef struct ncclComm *ncclComm_t;
struct ncclUniqueId {
    char internal[128];
};
enum { kNcclSuccess = 0, kNcclFloat64 = 8 };

struct Rccl {
    void *handle = nullptr;
    int (*GetUniqueId)(ncclUniqueId *) = nullptr;
    int (*CommInitRank)(ncclComm_t *, int, ncclUniqueId, int) = nullptr;
    int (*AllGather)(const void *, void *, size_t, int, ncclComm_t, hipStream_t) = nullptr;
    int (*CommDestroy)(ncclComm_t) = nullptr;
    const char *(*GetErrorString)(int) = nullptr;
    std::string why;
};

std::mutex g_mu;
Rccl g_rccl;
std::string g_path;
thread_local std::string g_comm_err;

int fail(int code, const std::string &msg)
{
    g_comm_err = msg;
    bild_set_last_error(msg.c_str());
    return code;
}

bool load_rccl()
{
    std::lock_guard<std::mutex> lk(g_mu);
    if (g_rccl.handle) return true;
    // BILD_AMD_RCCL_ONLY=1: nothing but the library named by bild_comm_library / BILD_AMD_RCCL is tried (tests of the
    // path without RCCL; deployments that must not pick up whatever librccl the loader finds)
    const bool only_named = getenv("BILD_AMD_RCCL_ONLY") != nullptr;
    const char *candidates[] = {g_path.empty() ? nullptr : g_path.c_str(), getenv("BILD_AMD_RCCL"), only_named ? nullptr : "librccl.so.1",
                                only_named ? nullptr : "librccl.so", only_named ? nullptr : "/opt/rocm/lib/librccl.so.1"};
    void *h = nullptr;
    // an RCCL that is already in the process wins (two copies would each bring their own HIP runtime)
    if (!only_named)
        for (const char *name : {"librccl.so", "librccl.so.1"})
            if (!h) h = dlopen(name, RTLD_NOW | RTLD_NOLOAD);
    for (const char *c : candidates)
        if (!h && c) h = dlopen(c, RTLD_NOW | RTLD_GLOBAL);
    if (!h) {
        const char *e = dlerror(); // ONE call: dlerror() clears the state it returns
        g_rccl.why = std::string("librccl.so not found: ") + (e ? e : "?");
        return false;
    }
    g_rccl.GetUniqueId = (int (*)(ncclUniqueId *))dlsym(h, "ncclGetUniqueId");
    g_rccl.CommInitRank = (int (*)(ncclComm_t *, int, ncclUniqueId, int))dlsym(h, "ncclCommInitRank");
    g_rccl.AllGather = (int (*)(const void *, void *, size_t, int, ncclComm_t, hipStream_t))dlsym(h, "ncclAllGather");
    g_rccl.CommDestroy = (int (*)(ncclComm_t))dlsym(h, "ncclCommDestroy");
    g_rccl.GetErrorString = (const char *(*)(int))dlsym(h, "ncclGetErrorString");
    if (!g_rccl.GetUniqueId || !g_rccl.CommInitRank || !g_rccl.AllGather || !g_rccl.CommDestroy) {
        g_rccl.why = "librccl.so lacks ncclGetUniqueId / ncclCommInitRank / ncclAllGather / ncclCommDestroy";
        return false;
    }
    g_rccl.handle = h;
    return true;
}

std::string nccl_msg(const char *what, int rc)
{
    char buf[256];
    snprintf(buf, sizeof buf, "%s failed: %s (%d)", what, g_rccl.GetErrorString ? g_rccl.GetErrorString(rc) : "?", rc);
    return buf;
}

} // namespace

struct bild_comm {
    ncclComm_t comm = nullptr;
    int world = 1, rank = 0, device = -1;
};

extern "C" {

int bild_comm_library(const char *path)
{
    std::lock_guard<std::mutex> lk(g_mu);
    if (g_rccl.handle) return fail(BILD_ERR_INVALID, "RCCL is already loaded");
    g_path = path ? path : "";
    return BILD_OK;
}

int bild_comm_unique_id(char *id, int id_len)
{
    if (!id || id_len < BILD_COMM_ID_BYTES) return fail(BILD_ERR_INVALID, "id buffer must hold BILD_COMM_ID_BYTES bytes");
    if (!load_rccl()) return fail(BILD_ERR_UNSUPPORTED, g_rccl.why);
    ncclUniqueId uid;
    const int rc = g_rccl.GetUniqueId(&uid);
    if (rc != kNcclSuccess) return fail(BILD_ERR_HIP, nccl_msg("ncclGetUniqueId", rc));
    std::memcpy(id, uid.internal, sizeof uid.internal);
    return BILD_OK;
}

int bild_comm_create(const char *id, int world, int rank, bild_comm **out)
{
    if (!out) return fail(BILD_ERR_INVALID, "out is NULL");
    *out = nullptr;
    if (!id || world < 1 || rank < 0 || rank >= world) return fail(BILD_ERR_INVALID, "bad communicator arguments");
    if (!load_rccl()) return fail(BILD_ERR_UNSUPPORTED, g_rccl.why);
    int dev = -1;
    if (hipGetDevice(&dev) != hipSuccess) return fail(BILD_ERR_NO_DEVICE, "no current HIP device");
    ncclUniqueId uid;
    std::memcpy(uid.internal, id, sizeof uid.internal);
    bild_comm *c = new bild_comm;
    const int rc = g_rccl.CommInitRank(&c->comm, world, uid, rank);
    if (rc != kNcclSuccess) {
        delete c;
        return fail(BILD_ERR_HIP, nccl_msg("ncclCommInitRank", rc));
    }
    c->world = world;
    c->rank = rank;
    c->device = dev;
    *out = c;
    return BILD_OK;
}

int bild_comm_allgather(bild_comm *c, const double *d_send, double *d_recv, int64_t n_per_rank, void *hip_stream)
{
    if (!c || !d_send || !d_recv || n_per_rank < 0) return fail(BILD_ERR_INVALID, "bad all-gather arguments");
    if (n_per_rank == 0) return BILD_OK;
    const int rc = g_rccl.AllGather(d_send, d_recv, (size_t)n_per_rank, kNcclFloat64, c->comm, (hipStream_t)hip_stream);
    if (rc != kNcclSuccess) return fail(BILD_ERR_HIP, nccl_msg("ncclAllGather", rc));
    return BILD_OK;
}

// minimal device-buffer management, so that a host program without any GPU framework can hold the shard / gathered vectors
int bild_device_alloc(int64_t bytes, void **out)
{
    if (!out || bytes < 0) return fail(BILD_ERR_INVALID, "bad allocation arguments");
    *out = nullptr;
    if (bytes == 0) return BILD_OK;
    hipError_t e = hipMalloc(out, (size_t)bytes);
    if (e != hipSuccess) return fail(e == hipErrorNoDevice ? BILD_ERR_NO_DEVICE : BILD_ERR_NOMEM, std::string("hipMalloc failed: ") + hipGetErrorString(e));
    return BILD_OK;
}

int bild_device_free(void *ptr)
{
    if (ptr) (void)hipFree(ptr);
    return BILD_OK;
}

int bild_device_to_host(void *dst, const void *d_src, int64_t bytes, void *hip_stream)
{
    if (bytes == 0) return BILD_OK;
    if (!dst || !d_src || bytes < 0) return fail(BILD_ERR_INVALID, "bad copy arguments");
    hipError_t e = hipMemcpyAsync(dst, d_src, (size_t)bytes, hipMemcpyDeviceToHost, (hipStream_t)hip_stream);
    if (e == hipSuccess) e = hipStreamSynchronize((hipStream_t)hip_stream);
    if (e != hipSuccess) return fail(BILD_ERR_HIP, std::string("device-to-host copy failed: ") + hipGetErrorString(e));
    return BILD_OK;
}

int bild_comm_destroy(bild_comm *c)
{
    if (!c) return BILD_OK;
    if (c->comm && g_rccl.CommDestroy) (void)g_rccl.CommDestroy(c->comm);
    delete c;
    return BILD_OK;
}

} // extern "C"
