// Launch parameters of the direct exchange (exchange_kernel.hip: the kernel; exchange.cpp: the host side behind the C ABI).
#pragma once
#include <stdint.h>

namespace bild {

constexpr int kExchangeMaxWorld = 16;
constexpr int kExchangeParts = 4; // workgroups per peer that share a shard (exchange_kernel.hip)

struct ExParams {
    int world, rank;
    int64_t n, slot;
    uint32_t step;
    uint32_t arrivals;                // what a flag of this step's parity reads once a peer's shard of this step is complete
    double *peer_data[kExchangeMaxWorld];    // data block of every rank (own block at [rank])
    uint32_t *peer_flags[kExchangeMaxWorld]; // flags block of every rank
    const double *send;
    double *recv;
    uint32_t *status;                 // [0] != 0: a wait gave up, [1]: the peer it waited for
    unsigned long long timeout_ticks; // of the 100 MHz wall clock
};

int launch_exchange(const ExParams &p, void *stream);

} // namespace bild
