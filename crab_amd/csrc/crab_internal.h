// Host-side internals shared by the C-ABI translation units.
#pragma once
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <string.h>
#include "../../include/crab_hip.h"

// the launch trace (crab_trace_begin / crab_trace_end, include/crab_hip.h): which kernels the entry points of this context launched and how
// often, by the name each launch site hands to crab_check_launch - a test can then hold a parity comparison to the kernel instantiation it
// claims to cover (attn_decode_kernel<128>, the 256 x 256 ring GEMM, ...).  Host-side bookkeeping only: off by default, no device work.
#define CRAB_TRACE_MAX 160
struct crab_trace_row { char name[64]; long count; };
struct crab_ctx {
    int device;
    char err[512];
    int trace_on;
    int trace_n;
    crab_trace_row trace[CRAB_TRACE_MAX];
};

static inline void crab_trace_note(crab_ctx* ctx, const char* what) {
    for (int i = 0; i < ctx->trace_n; ++i)
        if (strncmp(ctx->trace[i].name, what, sizeof(ctx->trace[i].name) - 1) == 0) { ++ctx->trace[i].count; return; }
    if (ctx->trace_n >= CRAB_TRACE_MAX) return;
    crab_trace_row* r = &ctx->trace[ctx->trace_n++];
    strncpy(r->name, what, sizeof(r->name) - 1);
    r->name[sizeof(r->name) - 1] = 0;
    r->count = 1;
}

static inline int crab_fail(crab_ctx* ctx, int code, const char* msg) {
    if (ctx) { strncpy(ctx->err, msg, sizeof(ctx->err) - 1); ctx->err[sizeof(ctx->err) - 1] = 0; }
    return code;
}

// Launch-configuration errors surface immediately; asynchronous faults surface at the next call or crab_sync
// (SURVEY.md 8b "Errors").  hipGetLastError is capture-safe.
static inline int crab_check_launch(crab_ctx* ctx, const char* what) {
    hipError_t e = hipGetLastError();
    if (e != hipSuccess) {
        if (ctx) snprintf(ctx->err, sizeof(ctx->err), "%s: %s", what, hipGetErrorString(e));
        return CRAB_E_HIP;
    }
    if (ctx && ctx->trace_on) crab_trace_note(ctx, what);
    return CRAB_OK;
}

#define CRAB_HIP_TRY(ctx, expr)                                                                  \
    do {                                                                                         \
        hipError_t _e = (expr);                                                                  \
        if (_e != hipSuccess) {                                                                  \
            if (ctx) snprintf((ctx)->err, sizeof((ctx)->err), "%s: %s", #expr, hipGetErrorString(_e)); \
            return CRAB_E_HIP;                                                                   \
        }                                                                                        \
    } while (0)

// the kernels' storage-flags word (common.h CF_C32 | CF_R32) of a GEMM descriptor
static inline int crab_cflags(const crab_gemm_desc* d) { return (d->c_fp32 ? 1 : 0) | ((d->r_fp32 && d->R) ? 2 : 0); }

// rowfin.hip: the M <= 16 layer tail
bool crab_rowfin_enabled();                                     // CRAB_ROWFIN != "0"
extern "C" int crab_rowfin_lora_ok(int nl, int r, int N);       // declared in include/crab_hip.h as well
