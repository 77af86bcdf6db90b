// The direct exchange: the one collective of a multi-GPU AMIS step (the vector of log-likelihoods every rank needs to
// form the importance weights, reference bild/amis.py:843-845) as ONE kernel per rank that stores the rank's shard
// straight into every peer's receive buffer over xGMI and waits for the peers' shards -- no ring, no proxy thread, no
// second launch.  SURVEY section 5 recommends exactly this for messages of this size (10-256 KB per rank: latency-bound;
// a ring all-gather over 8 GPUs is 7 dependent hops), include/bild_amd.h "several GPUs" describes the calls.
//
// Protocol.  Every rank owns a receive block in its own HBM, exported to the peers through an IPC handle:
//     data   [2 parities][world slots][slot doubles]      slot r of parity q: rank r's shard of a step with step % 2 == q
//     flags  [2 parities][world] uint32                   flags[q][r] = the step whose shard of rank r is complete in slot (q, r)
// A step (number `step`, counted per exchange object, the same on every rank) is one launch of `world` workgroups on the
// caller's stream; workgroup p
//     1. copies the local shard into slot (step % 2, rank) of PEER p's block (16-byte stores), then -- every storing wave
//        drained, a system-scope release -- stores `step` into peer p's flags[step % 2][rank];
//     2. polls its OWN flags[step % 2][p] until peer p's shard of this step has arrived (system-scope loads; the spin is
//        bounded by a wall-clock timeout that sets a status word), acquires, and copies slot (step % 2, p) into the caller's
//        gathered vector -- so the consumer of that vector reads bytes its own device wrote, whatever the caching of peer
//        writes.
// Two parities: a rank can be at most one step ahead of a peer (it cannot finish step s + 1 before the peer has posted its
// shard of s + 1, which the peer's stream does after it has consumed the results of step s), so the slots of step s are
// never overwritten while somebody still reads them.  Step numbers are compared as signed differences: the 32-bit counter
// may wrap (tested across the wrap with two processes sharing one GPU, tests/test_gpu_exchange.py).
#include <hip/hip_runtime.h>

#include "exchange.h"

namespace bild {
namespace {

constexpr int kThreads = 256;

__global__ void __launch_bounds__(kThreads) exchange_kernel(const ExParams p)
{
    const int peer = blockIdx.x, tid = threadIdx.x;
    const int par = (int)(p.step & 1u);
    // 1. the local shard into the peer's block
    {
        double *dst = p.peer_data[peer] + ((int64_t)par * p.world + p.rank) * p.slot;
        const int64_t n2 = p.n / 2;
        const double2 *s2 = reinterpret_cast<const double2 *>(p.send);
        double2 *d2 = reinterpret_cast<double2 *>(dst);
        for (int64_t i = tid; i < n2; i += kThreads) d2[i] = s2[i];
        if ((p.n & 1) && tid == 0) dst[p.n - 1] = p.send[p.n - 1];
        __builtin_amdgcn_fence(__ATOMIC_RELEASE, ""); // system scope: the stores of this wave are out before ...
        __syncthreads();                               // ... any lane of the workgroup signals for them
        if (tid == 0)
            __hip_atomic_store(p.peer_flags[peer] + par * p.world + p.rank, p.step, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_SYSTEM);
    }
    // 2. the peer's shard out of the own block
    {
        __shared__ int gave_up;
        if (tid == 0) {
            gave_up = 0;
            const uint32_t *flag = p.peer_flags[p.rank] + par * p.world + peer;
            const unsigned long long t0 = wall_clock64();
            for (;;) {
                const uint32_t seen = __hip_atomic_load(flag, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
                if ((int32_t)(seen - p.step) >= 0) break;
                if (wall_clock64() - t0 > p.timeout_ticks) {
                    gave_up = 1;
                    if (atomicCAS(p.status, 0u, 1u) == 0u) p.status[1] = (uint32_t)peer;
                    break;
                }
                __builtin_amdgcn_s_sleep(8);
            }
            __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "");
        }
        __syncthreads();
        __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, ""); // (every wave reads the slot: each drops what its CU's caches may hold of it)
        if (gave_up) return;
        const double *src = p.peer_data[p.rank] + ((int64_t)par * p.world + peer) * p.slot;
        double *dst = p.recv + (int64_t)peer * p.n;
        // (p.n may be odd: the gathered vector's shards are then not 16-byte aligned -- 8-byte copies)
        for (int64_t i = tid; i < p.n; i += kThreads) dst[i] = __hip_atomic_load(src + i, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
    }
}

} // namespace

int launch_exchange(const ExParams &p, void *stream)
{
    hipLaunchKernelGGL(exchange_kernel, dim3((unsigned)p.world), dim3(kThreads), 0, reinterpret_cast<hipStream_t>(stream), p);
    return (int)hipGetLastError();
}

} // namespace bild
