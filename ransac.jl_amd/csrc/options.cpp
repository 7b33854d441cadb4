// options.cpp -- rh_set_option / rh_get_option (include/ransac_hip.h) and the library's one place that may look at the
// environment.
//
// The PRODUCT build (libransac_hip.so) never calls getenv: a library that sits under someone else's Julia / Python
// process must not change what it computes because of a variable in that process's environment.  What a caller may
// legitimately tune -- which score / refit path a cloud takes, the score kernel's row length, the segment width of the
// mask pass -- goes through rh_set_option, per cloud or process-wide.
//
// The DIAG build (-DRH_DIAG, libransac_hip_diag.so; tests marked `diag`, the fuzzers, the profiling tools) also knows
// the A/B switches of the rounds' experiments and the diagnostics, and reads every option's RH_* environment variable
// when neither the cloud nor the process has set it -- so the tools keep working by exporting a variable.
#include <stdlib.h>
#include <string.h>

#include "rh_internal.h"

namespace {

enum { T_INT = 0, T_FLAG = 1, T_ENUM_SCORE = 2, T_ENUM_REFIT = 3, T_ENUM_ORDER = 4 };

struct OptDef {
    int id;
    const char *key;
#ifdef RH_DIAG
    const char *env;
#endif
    int type;
};

#ifdef RH_DIAG
#define RH_O(id, key, env, type) { id, key, env, type }
#else
#define RH_O(id, key, env, type) { id, key, type }
#endif

const OptDef kDefs[] = {
    RH_O(RH_OPT_SCORE_PATH, "score_path", "RH_SCORE_PATH", T_ENUM_SCORE),
    RH_O(RH_OPT_S4_ROWS, "s4_rows", "RH_S4_R", T_INT),
    RH_O(RH_OPT_UNP_WORDS, "unp_words", "RH_UNP_WORDS", T_INT),
    RH_O(RH_OPT_ST_CULL, "st_cull", "RH_ST_CULL", T_INT),
    RH_O(RH_OPT_REFIT_PATH, "refit_path", "RH_REFIT_PATH", T_ENUM_REFIT),
    RH_O(RH_OPT_BATCHES_IN_FLIGHT, "batches_in_flight", "RH_BATCHES_IN_FLIGHT", T_INT),
#ifdef RH_DIAG
    RH_O(RH_OPT_CREATE_PROF, "create_prof", "RH_CREATE_PROF", T_FLAG),
    RH_O(RH_OPT_AABB_HOST, "aabb_host", "RH_AABB_HOST", T_FLAG),
    RH_O(RH_OPT_SUB_ORDER, "sub_order", "RH_SUB_ORDER", T_ENUM_ORDER),
    RH_O(RH_OPT_KD_HOST, "kd_host", "RH_KD_HOST", T_FLAG),
    RH_O(RH_OPT_NO_SPREAD, "no_spread", "RH_NO_SPREAD", T_FLAG),
    RH_O(RH_OPT_OCT_CHAIN_W, "oct_chain_w", "RH_OCT_CHAIN_W", T_INT),
    RH_O(RH_OPT_NO_MANAGED_STORE, "no_managed_store", "RH_NO_MANAGED_STORE", T_FLAG),
    RH_O(RH_OPT_OCT_ONE_WINDOW, "oct_one_window", "RH_OCT_ONE_WINDOW", T_FLAG),
    RH_O(RH_OPT_OCT_WINDOW_ITERS, "oct_window_iters", "RH_OCT_WINDOW_ITERS", T_INT),
    RH_O(RH_OPT_NO_FUSED_SCORE, "no_fused_score", "RH_NO_FUSED_SCORE", T_FLAG),
    RH_O(RH_OPT_NO_PIPELINE, "no_pipeline", "RH_NO_PIPELINE", T_FLAG),
    RH_O(RH_OPT_NO_OCT_CHAIN, "no_oct_chain", "RH_NO_OCT_CHAIN", T_FLAG),
    RH_O(RH_OPT_REFIT_BLOCKS, "refit_blocks", "RH_REFIT_BLOCKS", T_INT),
    RH_O(RH_OPT_NO_FUSED_SAMPLER, "no_fused_sampler", "RH_NO_FUSED_SAMPLER", T_FLAG),
    RH_O(RH_OPT_NO_CREC, "no_crec", "RH_NO_CREC", T_FLAG),
    RH_O(RH_OPT_LONG_WINDOW_SETS, "long_window_sets", "RH_LONG_WINDOW_SETS", T_INT),
    RH_O(RH_OPT_NO_OCT_TAB, "no_oct_tab", "RH_NO_OCT_TAB", T_FLAG),
    RH_O(RH_OPT_NO_DRIVER_CACHE, "no_driver_cache", "RH_NO_DRIVER_CACHE", T_FLAG),
    RH_O(RH_OPT_HOST_SAMPLER, "host_sampler", "RH_HOST_SAMPLER", T_FLAG),
    RH_O(RH_OPT_DRIVER_PROF, "driver_prof", "RH_DRIVER_PROF", T_FLAG),
    RH_O(RH_OPT_G2_DBG, "g2_dbg", "RH_G2_DBG", T_INT),
    RH_O(RH_OPT_KREFIT_DBG, "krefit_dbg", "RH_KREFIT_DBG", T_FLAG),
    RH_O(RH_OPT_NO_FAST_EXTRACT, "no_fast_extract", "RH_NO_FAST_EXTRACT", T_FLAG),
#endif
};
constexpr int kNDefs = (int)(sizeof(kDefs) / sizeof(kDefs[0]));

int64_t g_opt[RH_OPT_COUNT];
bool g_init = false;

void init_once()
{
    if (g_init) return;
    for (int i = 0; i < RH_OPT_COUNT; i++) g_opt[i] = RH_OPTION_UNSET;
    g_init = true;
}

const OptDef *find_key(const char *key)
{
    if (key == nullptr) return nullptr;
    for (int i = 0; i < kNDefs; i++)
        if (strcmp(kDefs[i].key, key) == 0) return &kDefs[i];
    return nullptr;
}

#ifdef RH_DIAG
int64_t parse_env(const OptDef &d, const char *e)
{
    switch (d.type) {
    case T_FLAG: return 1;   // presence switches it on, as the rounds' getenv() != nullptr tests did
    case T_ENUM_SCORE: return e[0] == 'b' ? RH_SCORE_PATH_BRUTE : (e[0] == 'g' ? RH_SCORE_PATH_GROUPS : RH_OPTION_UNSET);
    case T_ENUM_REFIT: return e[0] == 's' ? RH_REFIT_PATH_SCAN : (e[0] == 'c' ? RH_REFIT_PATH_CULLED : RH_OPTION_UNSET);
    case T_ENUM_ORDER: return e[0] == 'm' ? 1 : 0;
    default: return (int64_t)atoll(e);
    }
}
#endif

}  // namespace

// cloud -> process -> (diag build) environment; RH_OPTION_UNSET when nobody has said anything
int64_t rh_opt(const rh_cloud *c, int id)
{
    if (id < 0 || id >= RH_OPT_COUNT) return RH_OPTION_UNSET;
    init_once();
    if (c != nullptr && c->opt[id] != RH_OPTION_UNSET) return c->opt[id];
    if (g_opt[id] != RH_OPTION_UNSET) return g_opt[id];
#ifdef RH_DIAG
    for (int i = 0; i < kNDefs; i++)
        if (kDefs[i].id == id) {
            const char *e = getenv(kDefs[i].env);
            return e != nullptr ? parse_env(kDefs[i], e) : RH_OPTION_UNSET;
        }
#endif
    return RH_OPTION_UNSET;
}

#ifdef RH_DIAG
const char *rh_opt_env_string(const char *name) { return getenv(name); }
#endif

void rh_opt_init_cloud(rh_cloud *c)
{
    for (int i = 0; i < RH_OPT_COUNT; i++) c->opt[i] = RH_OPTION_UNSET;
}

extern "C" int rh_build_variant(void)
{
#ifdef RH_DIAG
    return 1;
#else
    return 0;
#endif
}

extern "C" int rh_set_option(rh_cloud *c, const char *key, int64_t value)
{
    init_once();
    const OptDef *d = find_key(key);
    if (d == nullptr) { rh_set_error("rh_set_option: unknown option '%s'", key ? key : "(null)"); return RH_E_INVALID; }
    if (value != RH_OPTION_UNSET) {
        bool ok = true;
        switch (d->id) {
        case RH_OPT_SCORE_PATH: ok = value >= 0 && value <= RH_SCORE_PATH_GROUPS; break;
        case RH_OPT_REFIT_PATH: ok = value >= 0 && value <= RH_REFIT_PATH_CULLED; break;
        case RH_OPT_S4_ROWS: ok = value == 0 || value == 1 || value == 2 || value == 4 || value == 8 || value == 12 || value == 16; break;
        case RH_OPT_UNP_WORDS: ok = value >= 0 && value <= 16384; break;
        case RH_OPT_ST_CULL: ok = value >= 0 && value <= 2; break;
        case RH_OPT_BATCHES_IN_FLIGHT: ok = value >= 0 && value <= RH_MAX_IN_FLIGHT; break;
        default: break;
        }
        if (!ok) { rh_set_error("rh_set_option: value %lld is not valid for '%s'", (long long)value, key); return RH_E_INVALID; }
    }
    if (d->id == RH_OPT_SCORE_PATH && c != nullptr) {
        rh_set_error("rh_set_option: 'score_path' is fixed when a cloud is created (its point order depends on it): set it process-wide (cloud = NULL) before rh_cloud_create");
        return RH_E_INVALID;
    }
    if (c != nullptr) {
        // (a cloud with batches in flight: the number of streams they take turns on must not change under them)
        if (d->id == RH_OPT_BATCHES_IN_FLIGHT) { const int rc = rh_cloud_join(c); if (rc != RH_OK) return rc; }
        c->opt[d->id] = value;
    } else {
        g_opt[d->id] = value;
    }
    return RH_OK;
}

extern "C" int rh_get_option(const rh_cloud *c, const char *key, int64_t *value_out, int32_t *is_set)
{
    const OptDef *d = find_key(key);
    if (d == nullptr || value_out == nullptr) { rh_set_error("rh_get_option: unknown option '%s'", key ? key : "(null)"); return RH_E_INVALID; }
    const int64_t v = rh_opt(c, d->id);
    *value_out = v == RH_OPTION_UNSET ? 0 : v;
    if (is_set != nullptr) *is_set = v != RH_OPTION_UNSET;
    return RH_OK;
}
