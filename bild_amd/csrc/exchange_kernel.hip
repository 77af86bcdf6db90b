// The direct exchange: the one collective of a multi-GPU AMIS step (the vector of log-likelihoods every rank needs to
// form the importance weights, reference bild/amis.py:843-845) as ONE kernel per rank that stores the rank's shard
// straight into every peer's receive buffer over xGMI and waits for the peers' shards -- no ring, no proxy thread, no
// second launch.  SURVEY section 5 recommends exactly this for messages of this size (10-256 KB per rank: latency-bound;
// a ring all-gather over 8 GPUs is 7 dependent hops), include/bild_amd.h "several GPUs" describes the calls.
//
// Protocol.  Every rank owns a receive block in its own HBM, exported to the peers through an IPC handle:
//     data   [2 parities][world slots][slot doubles]      slot r of parity q: rank r's shard of a step with step % 2 == q
//     flags  [2 parities][world] uint32                   flags[q][r] counts the arrivals of rank r's workgroups in slot (q, r)
// A step (number `step`, counted per exchange object, the same on every rank) is one launch of `world` workgroups on the
// caller's stream; workgroup p
//     1. copies its share of the local shard into slot (step % 2, rank) of PEER p's block with write-through stores, then --
//        every storing wave drained, the workgroup's barrier -- one lane adds 1 to peer p's flags[step % 2][rank] (four
//        workgroups per peer share a shard: four arrivals per step);
//     2. polls its OWN flags[step % 2][p] until all of peer p's arrivals of this step are in (system-scope loads; the spin
//        is bounded by a wall-clock timeout that sets a status word), acquires, and copies its share of slot (step % 2, p)
//        into the caller's gathered vector -- so the consumer of that vector reads bytes its own device wrote, whatever the
//        caching of peer writes.
// Two parities: a rank can be at most one step ahead of a peer (it cannot finish step s + 1 before the peer has posted its
// shard of s + 1, which the peer's stream does after it has consumed the results of step s), so the slots of step s are
// never overwritten while somebody still reads them.  Step numbers are compared as signed differences: the 32-bit counter
// may wrap (tested across the wrap with two processes sharing one GPU, tests/test_gpu_exchange.py).
#include <hip/hip_runtime.h>

#include "exchange.h"

namespace bild {
namespace {

constexpr int kThreads = 256;

// kParts workgroups per peer share the shard (the stores of one workgroup to one peer are a few microseconds of one CU's
// store queue; four CUs per peer hide that behind each other)
constexpr int kParts = kExchangeParts;

__global__ void __launch_bounds__(kThreads) exchange_kernel(const ExParams p)
{
    const int peer = blockIdx.x / kParts, part = blockIdx.x % kParts, tid = threadIdx.x;
    const int par = (int)(p.step & 1u);
    const int64_t lo = p.n * part / kParts, hi = p.n * (part + 1) / kParts; // this workgroup's stretch of the shard
    typedef __attribute__((address_space(1))) unsigned long long gu64;
    // 1. the local shard into the peer's block: WRITE-THROUGH stores (system scope: they leave this device's caches as they
    //    are issued), so no release fence -- which would write back every dirty line of the L2, microseconds -- is needed;
    //    every storing wave drains its stores, the workgroup meets, ONE lane signals
    {
        gu64 *dst = (gu64 *)(p.peer_data[peer] + ((int64_t)par * p.world + p.rank) * p.slot);
        const gu64 *src = (const gu64 *)p.send;
        for (int64_t i = lo + tid; i < hi; i += kThreads) __hip_atomic_store(dst + i, src[i], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        __syncthreads();
        if (tid == 0)
            __hip_atomic_fetch_add(p.peer_flags[peer] + par * p.world + p.rank, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
    }
    // 2. the peer's shard out of the own block: its kParts workgroups have each added 1 to the flag of (parity, peer) --
    //    kParts per step, and the flag counts on from step to step
    {
        __shared__ int gave_up;
        if (tid == 0) {
            gave_up = 0;
            const uint32_t *flag = p.peer_flags[p.rank] + par * p.world + peer;
            const uint32_t want = p.arrivals; // arrivals of this parity's flags after this step, modulo 2^32
            const unsigned long long t0 = wall_clock64();
            for (;;) {
                const uint32_t seen = __hip_atomic_load(flag, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
                if ((int32_t)(seen - want) >= 0) break;
                if (wall_clock64() - t0 > p.timeout_ticks) {
                    gave_up = 1;
                    if (atomicCAS(p.status, 0u, 1u) == 0u) p.status[1] = (uint32_t)peer;
                    break;
                }
                __builtin_amdgcn_s_sleep(4);
            }
            __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "");
            asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        }
        __syncthreads();
        if (gave_up) return;
        // (the slot lies in THIS device's memory, whose coherence point is this device's L2: behind the acquire above -- one
        // lane's, then the barrier -- plain loads read what the peer's write-through stores left there)
        const double *src = p.peer_data[p.rank] + ((int64_t)par * p.world + peer) * p.slot;
        double *dst = p.recv + (int64_t)peer * p.n;
        for (int64_t i = lo + tid; i < hi; i += kThreads) dst[i] = src[i];
    }
}

} // namespace

int launch_exchange(const ExParams &p, void *stream)
{
    hipLaunchKernelGGL(exchange_kernel, dim3((unsigned)(p.world * kParts)), dim3(kThreads), 0, reinterpret_cast<hipStream_t>(stream), p);
    return (int)hipGetLastError();
}

} // namespace bild
