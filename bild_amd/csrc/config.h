// The library's environment switches, read ONCE (at first use) into one struct -- a dozen getenv calls per launch were
// 1-2 us of a 17 us small-batch call, and nobody could see which experiment switches a process was running under.
// bild_config_string() (C ABI) prints the active set; bild_config_reload() re-reads the environment (tests and tools
// that flip a switch inside one process; not to be called while evaluations are running).
#pragma once
#include <stdint.h>
#include <string>

namespace bild {

struct Config {
    // tables of a trajectory set (decided at the set's first evaluation)
    bool no_prefix = false;      // BILD_NO_PREFIX        never build a prefix table
    bool no_transients = false;  // BILD_NO_TRANSIENTS    no transient table
    bool no_pairs = false;       // BILD_NO_PAIRS         no pair table
    bool no_states = false;      // BILD_NO_STATES        no transient state table
    int tail_tol_bits = 20;      // BILD_TAIL_TOL_BITS    first-order tail: mean columns within 2^-bits of the table's
    int tail_margin = 8;         // BILD_TAIL_MARGIN      ... and the next switch this many frames beyond the table's own transient
    int64_t table_cache_bytes = (int64_t)4 << 30; // BILD_TABLE_CACHE_BYTES  table memory kept for the next trajectory set (0: none)
    bool no_tail = false;        // BILD_NO_TAIL          no first-order tails: a transient runs until its means have converged too
    int64_t states_max_bytes = -1;                // BILD_STATES_MAX_BYTES  (-1: 4 GB, 64 GB for sets declared for >= 1e8 evaluations)
    int states_max_gap = 128;                     // BILD_STATES_MAX_GAP    largest gap the state table covers (<= 255; 64 until round 4)
    int states_stride = 3;                        // BILD_STATES_STRIDE     the state table keeps every n-th gap (1 ... 8)
    int pairs_max_gap = 128;                      // BILD_PAIRS_MAX_GAP
    int64_t pairs_max_tasks = (int64_t)40 << 20;  // BILD_PAIRS_MAX_TASKS
    int64_t tables_after = -1;                    // BILD_TABLES_AFTER     experiments: delay the tables (-1: the built-in thresholds)
    // launches
    bool no_jump = false;            // BILD_NO_JUMP            frame by frame behind the first switch
    bool no_split = false;           // BILD_NO_SPLIT           single launch
    bool no_walk_plan = false;       // BILD_NO_WALK_PLAN
    bool no_schedule = false;        // BILD_NO_SCHEDULE
    bool no_listed_geometry = false; // BILD_NO_LISTED_GEOMETRY
    bool dense_valu = false;         // BILD_DENSE_VALU         dense path on the vector pipe
    bool no_fused_launch = false;    // BILD_NO_FUSED_LAUNCH    walk and frame loop as two launches (A/B against the fused grid)
    int geom = -1;                   // BILD_GEOM               force a geometry id
    int work_blocks = 0;             // BILD_WORK_BLOCKS
    int wide_threads = 0;            // BILD_WIDE_THREADS       256 / 512 / 1024
    int walk_debug = 0;              // BILD_WALK_DEBUG
    std::string sched_mode;          // BILD_SCHED_MODE         "spread" / "sorted"
    // the host-buffer seam
    bool in_via_copy = false;  // BILD_IN_VIA_COPY
    bool out_via_copy = false; // BILD_OUT_VIA_COPY
    bool st_on_host = false;   // BILD_ST_ON_HOST
    bool trace_staged = false; // BILD_TRACE_STAGED
    // AMIS bookkeeping / the inference driver
    bool amis_trace = false; // BILD_AMIS_TRACE
    int amis_threads = 1;    // BILD_AMIS_THREADS      threads of one sampler's passes over a large pool
    int host_threads = 0;    // BILD_HOST_THREADS      worker threads of the inference driver (0: min(8, cores))
    // everything that differs from the defaults, as "NAME=value NAME=value" ("" when nothing is set)
    std::string active;
};

const Config &config();
void config_reload();

} // namespace bild
