// knobs.cpp -- the one place that calls getenv (knobs.h holds the table).
#include "knobs.h"

#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <mutex>

namespace tolfg {

namespace {

Knobs g_knobs;
std::once_flag g_once;
#ifdef TOLFG_MEASURE
std::mutex g_mu;
#endif

#ifdef TOLFG_MEASURE
bool env_int(const char *name, int lo, int hi, int *out)
{
    const char *e = std::getenv(name);
    if (!e) return false;
    const int v = std::atoi(e);
    *out = v < lo ? lo : (v > hi ? hi : v);
    return true;
}
bool env_set(const char *name) { return std::getenv(name) != nullptr; }
#endif

void parse(Knobs &k, bool announce)
{
    k = Knobs();
    if (const char *e = std::getenv("TOLFG_RCCL_LIBRARY")) k.rccl_library = e;
    if (const char *e = std::getenv("TOLFG_TRACE")) k.trace = e[0] != '\0' && e[0] != '0';
    if (const char *e = std::getenv("TOLFG_MULTI_SHARED_DEVICES")) {
        // the seam needs a stand-in collective library beside it: alone it is ignored, loudly
        if (e[0] == '1' && !k.rccl_library.empty()) {
            k.multi_shared_devices = true;
            if (announce) std::fprintf(stderr, "tolfg: test seam active: tolfg_multi accepts a device more than once, collectives from %s\n",
                         k.rccl_library.c_str());
        } else if (e[0] == '1' && announce) {
            std::fprintf(stderr, "tolfg: the shared-devices test seam is ignored without an explicit collective library\n");
        }
    }
#ifdef TOLFG_MEASURE
    int v = 0;
    if (env_int("TOLFG_WAVES_PER_CU", 0, 32, &v)) k.waves_per_cu = v;
    if (env_int("TOLFG_TILE_NODES", 0, 128, &v)) k.tile_nodes = v;
    if (env_int("TOLFG_FUSED", 0, 1, &v)) k.fused = v;
    if (env_int("TOLFG_NT_STORES", 0, 1, &v)) k.nt_stores = v;
    if (env_int("TOLFG_XCD", 0, 1, &v)) k.xcd = v;
    if (env_int("TOLFG_STAGGER", 0, 1, &v)) k.stagger = v;
    if (env_int("TOLFG_SUB_NODES", 0, 32, &v)) k.sub_nodes = v >= 32 ? 32 : 0;
    if (const char *e = std::getenv("TOLFG_TAIL")) {        // "count:nt", count 0 = no tail
        k.tail_count = std::atoi(e) < 0 ? 0 : std::atoi(e);
        if (const char *c = std::strchr(e, ':')) k.tail_nt = std::atoi(c + 1);
    }
    k.no_single_launch = env_set("TOLFG_NO_SINGLE_LAUNCH");
    k.force_single_launch = env_set("TOLFG_FORCE_SINGLE_LAUNCH");
    k.x0_serial = env_set("TOLFG_X0_SERIAL");
    if (env_int("TOLFG_PLACE_CAP", 1, 64, &v)) k.place_cap = v;
    if (const char *e = std::getenv("TOLFG_PLACE_EARLY")) { const double r = std::atof(e); if (r >= 0.0 && r < 1.0) k.place_early = r; }
    if (const char *e = std::getenv("TOLFG_PLACED_CHUNK_KIB")) {
        const long c = std::atol(e);
        if (c >= 64 && c <= (1L << 20) && (c & (c - 1)) == 0) k.placed_chunk = (size_t)c << 10;
    }
    if (env_int("TOLFG_PLACE_FAIL_AT", 0, 64, &v)) k.place_fail_at = v;
    if (env_int("TOLFG_PLACE_SETTLE", 0, 1, &v)) k.place_settle = v;
    if (env_int("TOLFG_MULTI_GATHER_PRIORITY", 0, 1, &v)) k.multi_gather_priority = v;
    if (const char *e = std::getenv("TOLFG_MULTI_SLOT_WAIT")) k.multi_slot_wait_on_host = std::strcmp(e, "stream") != 0;
    if (const char *e = std::getenv("TOLFG_CALLBACK_STAGING")) k.callback_staging = e[0] == '1';
    if (const char *e = std::getenv("TOLFG_ZERO_COPY_LIMIT")) k.zero_copy_limit = std::atol(e);
    if (env_int("TOLFG_CHUNKS", 1, 6, &v)) k.chunks = v;
    k.multi_solo_comms = env_set("TOLFG_MULTI_SOLO_COMMS");
    k.no_register = env_set("TOLFG_NO_REGISTER");
    k.no_flag = env_set("TOLFG_NO_FLAG");
    k.callback_copy_x = env_set("TOLFG_CALLBACK_COPY_X");
#endif
}

}  // namespace

const Knobs &knobs()
{
    std::call_once(g_once, [] { parse(g_knobs, true); });
    return g_knobs;
}

void refresh_knobs()
{
#ifdef TOLFG_MEASURE
    (void)knobs();
    std::lock_guard<std::mutex> lk(g_mu);
    Knobs k;
    parse(k, false);
    // the collective library is chosen once per process (multi.cpp): what the first parse saw stays
    k.rccl_library = g_knobs.rccl_library;
    k.multi_shared_devices = g_knobs.multi_shared_devices;
    g_knobs = k;
#else
    (void)knobs();
#endif
}

bool measurement_build()
{
#ifdef TOLFG_MEASURE
    return true;
#else
    return false;
#endif
}

}  // namespace tolfg
