// Host-side internals shared by the C-ABI translation units.
#pragma once
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <string.h>
#include "../../include/crab_hip.h"

struct crab_ctx {
    int device;
    char err[512];
};

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
