// Error plumbing and the tunables shared by every entry point of the C ABI.
#include <stdarg.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>

#include <mutex>

#include "tunables.h"

static thread_local char g_err[512] = "";

void pssr_set_error(const char* fmt, ...) {
    va_list ap;
    va_start(ap, fmt);
    vsnprintf(g_err, sizeof(g_err), fmt, ap);
    va_end(ap);
}

extern "C" const char* pssr_last_error(void) { return g_err; }
extern "C" int pssr_abi_version(void) { return 4; }

// ---- kernel-selection tunables: ONE process-wide table, filled once from the environment (PSSR_<NAME>) the first time any entry
// point asks, changed afterwards only through pssr_set_option().  No launch path reads the environment.
namespace {
struct Entry { const char* name; int PssrTunables::*field; int dflt; int lo, hi; };
const Entry kEntries[] = {
    {"IGEMM_FLAT", &PssrTunables::igemm_flat, 1, 0, 1},
    {"IGEMM_BIG", &PssrTunables::igemm_big, 1, 0, 2},
    {"IGEMM_V3", &PssrTunables::igemm_v3, 1, 0, 2},
    {"IGEMM_V3_64", &PssrTunables::igemm_v3_64, 1, 0, 1},
    {"V3_LDS_PAD", &PssrTunables::v3_lds_pad, 0, 0, 100},
    {"IGEMM_DBG", &PssrTunables::igemm_dbg, 0, 0, 255},
    {"IGEMM_N64", &PssrTunables::igemm_n64, 1, 0, 1},
    {"IGEMM_KSPLIT", &PssrTunables::igemm_ksplit, 384, 1, 1 << 20},
    {"CONV_EPI8", &PssrTunables::conv_epi8, 1, 0, 1},
    {"WGRAD_LEAN", &PssrTunables::wgrad_lean, 1, 0, 1},
    {"WGRAD_X2", &PssrTunables::wgrad_x2, 0, 0, 1},
    {"WGRAD_DMA", &PssrTunables::wgrad_dma, 1, 0, 1},
    {"WGRAD_BLOCKS", &PssrTunables::wgrad_blocks, 256, 1, 1 << 20},
    {"WGRAD_BLOCKS_1X1", &PssrTunables::wgrad_blocks_1x1, 384, 1, 1 << 20},   // (c3 step: 512 21.96, 384 21.81, 256 21.92, 128 22.49 ms; c2 neutral)
    {"DWCONV_TILE", &PssrTunables::dwconv_tile, 1, 0, 1},
    {"DWWG_BLOCKS", &PssrTunables::dwwg_blocks, 1024, 1, 1 << 20},
    {"LN_BWD_BLOCKS", &PssrTunables::ln_bwd_blocks, 256, 1, 1 << 20},
    {"LN_DBG", &PssrTunables::ln_dbg, 0, 0, 3},
};
PssrTunables g_tun;
std::once_flag g_tun_once;
void tun_init() {
    for (const Entry& e : kEntries) {
        char key[64];
        snprintf(key, sizeof(key), "PSSR_%s", e.name);
        const char* v = getenv(key);
        int x = v ? atoi(v) : e.dflt;
        if (x < e.lo || x > e.hi) x = e.dflt;
        g_tun.*(e.field) = x;
    }
}
}  // namespace

PssrTunables& pssr_tunables() {
    std::call_once(g_tun_once, tun_init);
    return g_tun;
}

extern "C" int pssr_set_option(const char* name, int value) {
    if (name == nullptr) { pssr_set_error("pssr_set_option: null name"); return PSSR_ERR_ARG; }
    PssrTunables& t = pssr_tunables();
    for (const Entry& e : kEntries)
        if (strcmp(e.name, name) == 0) {
            if (value < e.lo || value > e.hi) { pssr_set_error("pssr_set_option: %s=%d outside [%d, %d]", name, value, e.lo, e.hi); return PSSR_ERR_ARG; }
            const int old = t.*(e.field);
            t.*(e.field) = value;
            return old;
        }
    pssr_set_error("pssr_set_option: unknown option %s", name);
    return PSSR_ERR_ARG;
}

extern "C" int pssr_get_option(const char* name) {
    if (name == nullptr) { pssr_set_error("pssr_get_option: null name"); return PSSR_ERR_ARG; }
    PssrTunables& t = pssr_tunables();
    for (const Entry& e : kEntries)
        if (strcmp(e.name, name) == 0) return t.*(e.field);
    pssr_set_error("pssr_get_option: unknown option %s", name);
    return PSSR_ERR_ARG;
}
