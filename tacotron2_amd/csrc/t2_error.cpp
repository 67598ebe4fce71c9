// Thread-local last-error string for the C ABI (include/tacotron2_amd.h: t2_last_error).
#include <stdio.h>
#include "../../include/tacotron2_amd.h"
static thread_local char g_err[512] = "";
void t2_set_error(const char* msg, const char* file, int line) {
    snprintf(g_err, sizeof(g_err), "%s (%s:%d)", msg ? msg : "?", file ? file : "?", line);
}
extern "C" const char* t2_last_error(void) { return g_err; }
extern "C" int t2_version(void) { return 100; }

// sizeof() of every ABI struct, so bindings can verify their mirrored layouts (tests/test_abi.py).
#include <string.h>
extern "C" int t2_sizeof(const char* name) {
#define T2_SZ(T) if (strcmp(name, #T) == 0) return (int)sizeof(T)
    T2_SZ(T2Gemm); T2_SZ(T2Seg); T2_SZ(T2LstmStep); T2_SZ(T2LstmStride); T2_SZ(T2LstmBwdStep); T2_SZ(T2LstmBwdStride);
    T2_SZ(T2AttnStep); T2_SZ(T2AttnSeq); T2_SZ(T2AttnSeqBwd); T2_SZ(T2Bn); T2_SZ(T2Infer); T2_SZ(T2StopScan); T2_SZ(T2ZeroRegions);
#undef T2_SZ
    return -1;
}
