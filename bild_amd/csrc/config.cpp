// See config.h: the environment switches of the library, read once.
#include "config.h"

#include <cstdlib>
#include <mutex>
#include <type_traits>

#include "../../include/bild_amd.h"

namespace bild {
namespace {

Config g_cfg;
std::once_flag g_once;
std::mutex g_mu;

void load(Config &c)
{
    c = Config();
    std::string act;
    auto note = [&](const char *name, const char *val) {
        if (!act.empty()) act += ' ';
        act += name;
        act += '=';
        act += val;
    };
    auto flag = [&](const char *name, bool &dst) {
        if (const char *e = getenv(name)) {
            dst = true;
            note(name, e[0] ? e : "1");
        }
    };
    auto num = [&](const char *name, auto &dst) {
        if (const char *e = getenv(name)) {
            dst = (typename std::remove_reference<decltype(dst)>::type)atoll(e);
            note(name, e);
        }
    };
    flag("BILD_NO_PREFIX", c.no_prefix);
    flag("BILD_NO_TRANSIENTS", c.no_transients);
    flag("BILD_NO_PAIRS", c.no_pairs);
    flag("BILD_NO_STATES", c.no_states);
    flag("BILD_NO_TAIL", c.no_tail);
    num("BILD_TABLE_CACHE_BYTES", c.table_cache_bytes);
    num("BILD_TAIL_TOL_BITS", c.tail_tol_bits);
    num("BILD_TAIL_MARGIN", c.tail_margin);
    num("BILD_STATES_MAX_BYTES", c.states_max_bytes);
    num("BILD_STATES_STRIDE", c.states_stride);
    num("BILD_STATES_MAX_GAP", c.states_max_gap);
    num("BILD_PAIRS_MAX_GAP", c.pairs_max_gap);
    if (c.pairs_max_gap < 2) c.pairs_max_gap = 2;
    num("BILD_PAIRS_MAX_TASKS", c.pairs_max_tasks);
    num("BILD_TABLES_AFTER", c.tables_after);
    flag("BILD_NO_JUMP", c.no_jump);
    flag("BILD_NO_SPLIT", c.no_split);
    flag("BILD_NO_WALK_PLAN", c.no_walk_plan);
    flag("BILD_NO_SCHEDULE", c.no_schedule);
    flag("BILD_NO_LISTED_GEOMETRY", c.no_listed_geometry);
    flag("BILD_DENSE_VALU", c.dense_valu);
    flag("BILD_NO_FUSED_LAUNCH", c.no_fused_launch);
    num("BILD_GEOM", c.geom);
    num("BILD_WORK_BLOCKS", c.work_blocks);
    num("BILD_WIDE_THREADS", c.wide_threads);
    num("BILD_WALK_DEBUG", c.walk_debug);
    if (const char *e = getenv("BILD_SCHED_MODE")) {
        c.sched_mode = e;
        note("BILD_SCHED_MODE", e);
    }
    flag("BILD_IN_VIA_COPY", c.in_via_copy);
    flag("BILD_OUT_VIA_COPY", c.out_via_copy);
    flag("BILD_ST_ON_HOST", c.st_on_host);
    flag("BILD_TRACE_STAGED", c.trace_staged);
    flag("BILD_AMIS_TRACE", c.amis_trace);
    num("BILD_AMIS_THREADS", c.amis_threads);
    num("BILD_HOST_THREADS", c.host_threads);
    c.active = act;
}

} // namespace

const Config &config()
{
    std::call_once(g_once, [] { load(g_cfg); });
    return g_cfg;
}

void config_reload()
{
    (void)config();
    std::lock_guard<std::mutex> lk(g_mu);
    load(g_cfg);
}

} // namespace bild

extern "C" {

const char *bild_config_string(void)
{
    // (a copy per thread: the pointer stays valid across a reload on another thread)
    static thread_local std::string s;
    s = bild::config().active;
    return s.c_str();
}

int bild_config_reload(void)
{
    bild::config_reload();
    return BILD_OK;
}

} // extern "C"
