// Handle lifecycle and error reporting of libubd_hip.so.
#include <stdarg.h>
#include <stdlib.h>
#include <string.h>
#include <thread>
#include "common.h"

static thread_local char g_err[512] = "";

void ubd_set_error(const char *fmt, ...)
{
    va_list ap;
    va_start(ap, fmt);
    vsnprintf(g_err, sizeof(g_err), fmt, ap);
    va_end(ap);
}

extern "C" const char *ubd_last_error(void) { return g_err; }
extern "C" int ubd_abi_version(void) { return UBD_ABI_VERSION; }
#ifndef UBD_BUILD_ID
#define UBD_BUILD_ID "unknown"
#endif
// host staging helper (include/ubd.h): n bytes in `threads` contiguous pieces, one std::thread each (the caller's thread takes the first piece)
extern "C" int ubd_host_memcpy_mt(void *dst, const void *src, size_t n, int threads)
{
    if (threads < 1) threads = 1;
    if (threads > 16) threads = 16;
    if (n < ((size_t)1 << 20)) threads = 1;
    const size_t piece = ((n + threads - 1) / threads + 4095) & ~(size_t)4095;
    std::thread th[16];
    int started = 0;
    size_t done_to = n < piece ? n : piece;              // [0, done_to) is the caller's piece; pieces whose thread could not be started are copied here too
    for (int t = 1; t < threads; ++t) {
        const size_t a = (size_t)t * piece;
        if (a >= n) break;
        const size_t len = n - a < piece ? n - a : piece;
        try {
            th[started] = std::thread([=] { memcpy((char *)dst + a, (const char *)src + a, len); });
            ++started;
        } catch (...) {                                  // no more threads to be had (resource limits): the rest in this thread
            memcpy((char *)dst + a, (const char *)src + a, n - a);
            break;
        }
    }
    memcpy(dst, src, done_to);
    for (int t = 0; t < started; ++t) th[t].join();
    return 0;
}
extern "C" const char *ubd_build_id(void) { return UBD_BUILD_ID; }   // sha256 over the kernel sources at build time (build.sh); bench.py compares it with the committed profiles' fingerprint

extern "C" int ubd_create(const ubd_config *cfg, ubd_handle **out)
{
    UBD_REQUIRE(cfg && out, "ubd_create: null argument");
    UBD_REQUIRE(cfg->c_in == 1 || cfg->c_in == 3, "ubd_create: c_in must be 1 (grey) or 3, got %d", cfg->c_in);
    UBD_REQUIRE(cfg->n_classes >= 0 && cfg->n_classes <= UBD_MAX_CLASSES, "ubd_create: n_classes %d out of range [0,%d]", cfg->n_classes, UBD_MAX_CLASSES);
    UBD_REQUIRE(cfg->dtype == UBD_F32 || cfg->dtype == UBD_BF16 || cfg->dtype == UBD_F16, "ubd_create: bad dtype %d", cfg->dtype);
    int ndev = 0;
    hipError_t e = hipGetDeviceCount(&ndev);
    UBD_REQUIRE(e == hipSuccess && ndev > 0, "ubd_create: no HIP device visible (%s); this library has no CPU fallback", hipGetErrorString(e));
    int dev = 0;
    UBD_CHECK_HIP(hipGetDevice(&dev));
    hipDeviceProp_t prop;
    UBD_CHECK_HIP(hipGetDeviceProperties(&prop, dev));
    UBD_REQUIRE(strncmp(prop.gcnArchName, "gfx950", 6) == 0, "ubd_create: device arch %s is not gfx950; this library is built for MI355X only", prop.gcnArchName);

    ubd_handle *h = (ubd_handle *)calloc(1, sizeof(ubd_handle));
    UBD_REQUIRE(h, "ubd_create: out of host memory");
    h->cfg = *cfg;
    h->k_out = 1 + cfg->n_classes;
    h->num_cus = prop.multiProcessorCount;
    {
        const char *e = getenv("UBD_DILCONV");
        h->use_wino = !(e && strcmp(e, "direct") == 0);
        h->wino_x6 = h->use_wino && !(e && strcmp(e, "wino32") == 0);
        // Inference runs L2 -> L3 as ONE kernel with L2's output in LDS (stem23.h) when the model uses the fml padding (the
        // variant that inherits the 33rd L2 column from the tile to its left: 0.405 vs 0.417 ms per forward pass at
        // 32 x 512 x 512); with TF 'same' padding the fused kernel only ties the two separate kernels and they
        // stay the default.  UBD_STEM=fused / unfused overrides either way.  Training always runs the separate kernels.
        const char *s = getenv("UBD_STEM");
        h->fuse_stem = cfg->fml_compatible != 0 ? 2 : 0;        // 2: L1 -> L2 -> L3 in one kernel (stem123.h; fml padding only)
        if (s && strcmp(s, "fused") == 0) { h->fuse_stem = 1; h->fuse_force = 1; }      // 1: L1, then L2 -> L3 fused (stem23.h)
        if (s && strcmp(s, "fused123") == 0) { h->fuse_stem = 2; h->fuse_force = 1; }   // forced at any launch size (tests)
        if (s && strcmp(s, "cold123") == 0 && cfg->fml_compatible != 0) { h->fuse_stem = 3; h->fuse_force = 1; }   // one kernel, one cold-started tile per work unit, at any launch size (tests; default for small launches)
        if (s && strcmp(s, "unfused") == 0) h->fuse_stem = 0;
        // test hooks: pretend the device has fewer CUs, so that every persistent kernel walks many tiles per block even on
        // the small shapes the CPU oracle can check (tests/test_gpu_persistent.py; ubd_num_cus reports what was taken)
        { const char *c = getenv("UBD_TEST_NUM_CUS"); if (c && atoi(c) > 0) h->num_cus = atoi(c); }
        { const char *b = getenv("UBD_DILBWD"); h->split_dilbwd = (b && strcmp(b, "split") == 0) ? 1 : 0; h->no_pair_dilbwd = (b && strcmp(b, "pair8") == 0) ? 1 : 0; }
        { const char *b = getenv("UBD_SEPBWD"); h->split_sepbwd32 = (b && strcmp(b, "split") == 0) ? 1 : 0; }
        { const char *b = getenv("UBD_STEM16"); h->split_stem16 = (b && strcmp(b, "split") == 0) ? 1 : ((b && strcmp(b, "fused12") == 0) ? 2 : 0); }   // 0: L1 -> L2 -> L3 in one kernel, 2: L1 -> L2 fused + L3, 1: three kernels
        // postprocess test hooks (multi-launch front end at any map size / separate tail launches / LDS poisoning + forest integrity
        // check / one-lane box fit / the 512-thread block shape the job has inside the stem kernel)
        h->pp_global = getenv("UBD_PP_GLOBAL") != nullptr; h->pp_split = getenv("UBD_PP_SPLIT") != nullptr;
        h->pp_poison = getenv("UBD_PP_POISON") != nullptr; h->pp_serial_tail = getenv("UBD_PP_SERIAL_TAIL") != nullptr;
        h->pp_threads_512 = getenv("UBD_PP_THREADS_512") != nullptr;
        { const char *b = getenv("UBD_HEADBWD"); h->split_headbwd = (b && strcmp(b, "split") == 0) ? 1 : 0; }
        { const char *b = getenv("UBD_SEPB16_X"); h->sepb_x_regs = (b && strcmp(b, "regs") == 0) ? 1 : 0; }
        { const char *b = getenv("UBD_REDUCE"); h->chain_reduce = (b && strcmp(b, "batched") == 0) ? 0 : 1; }
        { const char *b = getenv("UBD_LOSS"); h->loss_chain = (b && strcmp(b, "chain") == 0) ? 1 : 0; }
        { const char *b = getenv("UBD_DILCONV16"); h->direct_dil16 = (b && strcmp(b, "direct") == 0) ? 1 : 0; }
    }
    // Keras model.get_weights() order
    size_t off = 0;
    int cin = cfg->c_in;
    for (int s = 0; s < 3; ++s) {
        h->off_sep_dw[s] = off; off += (size_t)9 * cin;
        h->off_sep_pw[s] = off; off += (size_t)cin * UBD_C;
        h->off_sep_b[s] = off;  off += UBD_C;
        cin = UBD_C;
    }
    for (int k = 0; k < UBD_NUM_DIL; ++k) {
        h->off_dil_k[k] = off; off += (size_t)9 * UBD_C * UBD_C;
        h->off_dil_b[k] = off; off += UBD_C;
    }
    h->off_head_k = off; off += (size_t)UBD_C * h->k_out;
    h->off_head_b = off; off += h->k_out;
    h->n_params = off;
    *out = h;
    return 0;
}

extern "C" void ubd_destroy(ubd_handle *h)
{
    if (h && h->comm) ubd_comm_destroy(h);
    free(h);
}

extern "C" size_t ubd_param_count(const ubd_handle *h) { return h ? h->n_params : 0; }
extern "C" int ubd_num_cus(const ubd_handle *h) { return h ? h->num_cus : 0; }
