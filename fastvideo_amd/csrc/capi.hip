// Error plumbing + introspection for libfvk_amd.so (host only).
#include <stdarg.h>
#include <stdio.h>
#include <string.h>

#include "gemm_common.h"

static thread_local char g_err[512] = "";

void fvk_set_error(const char* fmt, ...) {
    va_list ap;
    va_start(ap, fmt);
    vsnprintf(g_err, sizeof(g_err), fmt, ap);
    va_end(ap);
}

extern "C" const char* fvk_last_error(void) { return g_err; }
extern "C" int fvk_abi_version(void) { return 8; }  // 8: + fvk_attn_block_sparse_ws_bf16, fvk_attn_block_sparse_workspace_bytes (split last round of the 64-row list kernel); 7: + fvk_gemm_vt_bf16 (round 5's entry point, numbered in round 6), fvk_mfma_sustained_probe_bf16; 6: + fvk_qkv_norm_rope_pack2_bf16, dense kernel id 3 (attn_pp2 at any key length); 5: + fvk_attn_dense_kernel_bf16, fvk_attn_dense_split_bf16, fvk_qkvg_norm_rope_pack_bf16, fvk_is_probe_build
extern "C" int fvk_is_probe_build(void) { return FVK_VARIANTS; }

extern "C" int fvk_device_arch(char* buf, int len) {
    if (!buf || len <= 0) return FVK_ERR_ARG;
    int dev = 0;
    hipDeviceProp_t prop;
    if (hipGetDevice(&dev) != hipSuccess || hipGetDeviceProperties(&prop, dev) != hipSuccess) {
        fvk_set_error("fvk_device_arch: no HIP device");
        return FVK_ERR_LAUNCH;
    }
    strncpy(buf, prop.gcnArchName, len - 1);
    buf[len - 1] = 0;
    return FVK_OK;
}

// ---- tunables (A/B switches for measurements; defaults = shipped configuration) ---------------------------------------
static int g_tunables[fvk::TUNE_COUNT] = {0};
static const char* const g_tunable_names[fvk::TUNE_COUNT] = {"gemm_impl", "attn_impl", "vae_conv_impl", "vsa_impl", nullptr};

#if FVK_VARIANTS
int fvk::tunable(int id) { return (id >= 0 && id < fvk::TUNE_COUNT) ? g_tunables[id] : 0; }
#endif

extern "C" int fvk_set_tunable(const char* name, int value) {
    for (int i = 0; i < fvk::TUNE_COUNT; ++i)
        if (name && g_tunable_names[i] && strcmp(name, g_tunable_names[i]) == 0) {
#if !FVK_VARIANTS
            FVK_CHECK(value == 0, FVK_ERR_ARG, "fvk_set_tunable: '%s' = %d: the product library holds the shipped configuration only; measurement "
                      "variants live in scripts/probes/libfvk_probe.so (FVK_PROBE_LIB=1)", name, value);
#endif
            g_tunables[i] = value;
            return FVK_OK;
        }
    fvk_set_error("fvk_set_tunable: unknown tunable '%s'", name ? name : "(null)");
    return FVK_ERR_ARG;
}
