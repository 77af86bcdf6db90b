// Host side of the direct exchange (the protocol: exchange_kernel.hip): receive blocks, IPC handles, the launch.
#include <hip/hip_runtime_api.h>

#include <cstdint>
#include <cstring>
#include <string>
#include <vector>

#include "../../include/bild_amd.h"
#include "exchange.h"

namespace {

int fail(int code, const std::string &msg)
{
    bild_set_last_error(msg.c_str());
    return code;
}

#define EX_HIP(call)                                                                                          \
    do {                                                                                                      \
        hipError_t e_ = (call);                                                                               \
        if (e_ != hipSuccess) return fail(BILD_ERR_HIP, std::string(#call) + ": " + hipGetErrorString(e_));   \
    } while (0)

} // namespace

struct bild_exchange {
    int world = 1, rank = 0, device = -1;
    int64_t slot = 0;
    uint32_t step = 1;
    uint32_t arrivals[2] = {0, 0};    // arrivals every flag of a parity has seen so far (per peer), modulo 2^32
    char *block = nullptr;            // own receive block: data, then flags
    size_t data_bytes = 0, flag_bytes = 0;
    void *peer_base[bild::kExchangeMaxWorld] = {};  // mapped blocks of the peers (own block at [rank])
    bool connected = false;
    uint32_t *status = nullptr;       // pinned host memory the kernel writes to
    double timeout_s = 5.0;
};

extern "C" {

int bild_exchange_create(int world, int rank, int64_t slot_doubles, bild_exchange **out)
{
    if (!out) return fail(BILD_ERR_INVALID, "out is NULL");
    *out = nullptr;
    if (world < 1 || world > bild::kExchangeMaxWorld || rank < 0 || rank >= world || slot_doubles < 1)
        return fail(BILD_ERR_INVALID, "bad exchange arguments (world <= 16)");
    int dev = -1;
    if (hipGetDevice(&dev) != hipSuccess) {
        (void)hipGetLastError();
        return fail(BILD_ERR_NO_DEVICE, "no current HIP device");
    }
    bild_exchange *x = new bild_exchange;
    x->world = world;
    x->rank = rank;
    x->device = dev;
    x->slot = (slot_doubles + 1) & ~(int64_t)1; // slots start on 16-byte boundaries
    x->data_bytes = (size_t)2 * world * x->slot * sizeof(double);
    x->flag_bytes = ((size_t)2 * world * sizeof(uint32_t) + 127) & ~(size_t)127;
    hipError_t e = hipMalloc((void **)&x->block, x->data_bytes + x->flag_bytes);
    if (e == hipSuccess) e = hipMemset(x->block, 0, x->data_bytes + x->flag_bytes);
    if (e == hipSuccess) e = hipHostMalloc((void **)&x->status, 64, hipHostMallocDefault);
    if (e != hipSuccess) {
        if (x->block) (void)hipFree(x->block);
        delete x;
        return fail(BILD_ERR_NOMEM, std::string("exchange block: ") + hipGetErrorString(e));
    }
    std::memset(x->status, 0, 64);
    x->peer_base[rank] = x->block;
    x->connected = world == 1;
    *out = x;
    return BILD_OK;
}

int bild_exchange_handle(const bild_exchange *x, char *handle)
{
    if (!x || !handle) return fail(BILD_ERR_INVALID, "NULL argument");
    static_assert(sizeof(hipIpcMemHandle_t) == BILD_EXCHANGE_HANDLE_BYTES, "IPC handle size");
    hipIpcMemHandle_t h;
    EX_HIP(hipIpcGetMemHandle(&h, x->block));
    std::memcpy(handle, &h, sizeof h);
    return BILD_OK;
}

int bild_exchange_connect(bild_exchange *x, const char *handles)
{
    if (!x || !handles) return fail(BILD_ERR_INVALID, "NULL argument");
    if (x->connected) return BILD_OK;
    for (int r = 0; r < x->world; ++r) {
        if (r == x->rank) continue;
        hipIpcMemHandle_t h;
        std::memcpy(&h, handles + (size_t)r * BILD_EXCHANGE_HANDLE_BYTES, sizeof h);
        EX_HIP(hipIpcOpenMemHandle(&x->peer_base[r], h, hipIpcMemLazyEnablePeerAccess));
    }
    x->connected = true;
    return BILD_OK;
}

// tests: the step counter of the next exchange (the same call on EVERY rank, before any exchange or between two that all
// ranks have completed): own flags are set to step - 1, so that a counter about to wrap can be rehearsed
int bild_exchange_set_step(bild_exchange *x, uint32_t step)
{
    if (!x) return fail(BILD_ERR_INVALID, "NULL handle");
    // (the flags count arrivals, kExchangeParts per step of their parity: start them, and the expectation, at a value that
    // wraps where the step counter does)
    const uint32_t base = step * (uint32_t)bild::kExchangeParts;
    std::vector<uint32_t> f((size_t)2 * x->world, base);
    EX_HIP(hipDeviceSynchronize());
    EX_HIP(hipMemcpy(x->block + x->data_bytes, f.data(), f.size() * sizeof(uint32_t), hipMemcpyHostToDevice));
    x->step = step;
    x->arrivals[0] = x->arrivals[1] = base;
    return BILD_OK;
}

int bild_exchange_allgather(bild_exchange *x, const double *d_send, int64_t n, void *hip_stream, double *d_recv)
{
    if (!x || !d_send || !d_recv || n < 0) return fail(BILD_ERR_INVALID, "bad exchange arguments");
    if (!x->connected) return fail(BILD_ERR_INVALID, "bild_exchange_connect first");
    if (n > x->slot) return fail(BILD_ERR_INVALID, "shard longer than the exchange's slots");
    if (n == 0) return BILD_OK;
    bild::ExParams p{};
    p.world = x->world;
    p.rank = x->rank;
    p.n = n;
    p.slot = x->slot;
    p.step = x->step;
    // flags count arrivals (kExchangeParts per step and parity, every second step): steps s, s - 2, ... of this parity
    p.arrivals = x->arrivals[x->step & 1u] + (uint32_t)bild::kExchangeParts;
    for (int r = 0; r < x->world; ++r) {
        p.peer_data[r] = (double *)x->peer_base[r];
        p.peer_flags[r] = (uint32_t *)((char *)x->peer_base[r] + x->data_bytes);
    }
    p.send = d_send;
    p.recv = d_recv;
    p.status = x->status;
    p.timeout_ticks = (unsigned long long)(x->timeout_s * 1e8);
    if (int rc = bild::launch_exchange(p, hip_stream))
        return fail(BILD_ERR_HIP, std::string("exchange kernel launch failed: ") + hipGetErrorString((hipError_t)rc));
    x->arrivals[x->step & 1u] += (uint32_t)bild::kExchangeParts;
    x->step += 1u;
    return BILD_OK;
}

// after the stream has been waited for: BILD_ERR_HIP when a wait of an exchange gave up (*peer: the rank that did not
// deliver; may be NULL) -- the gathered vector of that step is then incomplete
int bild_exchange_status(bild_exchange *x, int *peer)
{
    if (!x) return fail(BILD_ERR_INVALID, "NULL handle");
    if (peer) *peer = -1;
    if (x->status[0] == 0) return BILD_OK;
    const int who = (int)x->status[1];
    x->status[0] = x->status[1] = 0;
    if (peer) *peer = who;
    return fail(BILD_ERR_HIP, "direct exchange: rank " + std::to_string(who) + " did not deliver its shard within the timeout");
}

int bild_exchange_set_timeout(bild_exchange *x, double seconds)
{
    if (!x || !(seconds > 0)) return fail(BILD_ERR_INVALID, "bad timeout");
    x->timeout_s = seconds;
    return BILD_OK;
}

int bild_exchange_destroy(bild_exchange *x)
{
    if (!x) return BILD_OK;
    (void)hipDeviceSynchronize();
    for (int r = 0; r < x->world; ++r)
        if (r != x->rank && x->peer_base[r]) (void)hipIpcCloseMemHandle(x->peer_base[r]);
    if (x->block) (void)hipFree(x->block);
    if (x->status) (void)hipHostFree(x->status);
    delete x;
    return BILD_OK;
}

} // extern "C"
