// thread-local last-error string + ABI version for libstllm_hip.so
#include <stdarg.h>
#include <stdio.h>

#include "../../include/stllm_hip.h"

static thread_local char g_err[512] = "";

void stllm_set_error(const char* fmt, ...) {
  va_list ap;
  va_start(ap, fmt);
  vsnprintf(g_err, sizeof(g_err), fmt, ap);
  va_end(ap);
}

extern "C" const char* stllm_last_error(void) { return g_err; }
extern "C" int stllm_abi_version(void) { return 2; }   // 2: stllm_gemm_args gained the trailing a_norm_* fields (round 2)

static thread_local const char* g_last_kernel = "";
void stllm_set_last_kernel(const char* name) { g_last_kernel = name; }
extern "C" const char* stllm_last_kernel(void) { return g_last_kernel; }
