// C-ABI entry points of liblidarreg.so: workspace management, the per-function operators that mirror
// the reference's matching.py / GC_RANSAC.py helpers, and the fused per-pair pipeline behind FR().
#include "lr_internal.h"
#include <stdarg.h>
#include <string.h>
#include <stdlib.h>
#include <new>

static thread_local char g_err[512] = "";

void lr_set_error(const char *fmt, ...)
{
    va_list ap;
    va_start(ap, fmt);
    vsnprintf(g_err, sizeof(g_err), fmt, ap);
    va_end(ap);
}

static_assert(sizeof(lr_ransac_params) == 72 && sizeof(lr_pair_params) == 112 && sizeof(lr_pair_result) == 496,
              "ABI structs changed: update include/lidarreg.h, _ext.py, INTEGRATION.md and tests/test_abi_cpu.py together");

extern "C" int lr_version(void) { return 103; }
extern "C" const char *lr_last_error(void) { return g_err; }

// Every entry point that writes arena state (scratch, counters, result temporaries) makes lr_icp_batch forget the last batched
// registration: its transforms / descriptor table would otherwise be read back stale (only lr_register_batch, on success, re-arms it)
static inline void forget_last_batch(lr_workspace *ws) { if (ws) { ws->last_batch = 0; ws->last_T_final = nullptr; } }

// ------------------------------------------------------------------ device
// The kernels are gfx950 code objects and the workspace is memory of ONE device: a workspace remembers the device it was created on and
// every entry point that takes one checks that this device is current (and that the stream belongs to it) before it launches anything --
// on another device the arena pointers would be dereferenced by kernels running elsewhere (a fault at best).  What the architecture
// and the number of compute units are is asked once per device.
static int g_fake_device = -1;          // test hook (lr_debug_fake_current_device): what check_device() takes for the current device
struct lr_device_info { int known, ok, cus; char arch[64]; };
static lr_device_info g_dev[64];

static int device_info(int dev, const lr_device_info **out)
{
    if (dev < 0 || dev >= 64) { lr_set_error("device index %d out of range", dev); return LR_EINVAL; }
    lr_device_info &d = g_dev[dev];
    if (!d.known) {
        hipDeviceProp_t prop;
        LR_HIP(hipGetDeviceProperties(&prop, dev));
        snprintf(d.arch, sizeof d.arch, "%s", prop.gcnArchName);
        d.ok = strncmp(prop.gcnArchName, "gfx950", 6) == 0;
        d.cus = prop.multiProcessorCount > 0 ? prop.multiProcessorCount : 256;
        d.known = 1;
    }
    *out = &d;
    return LR_OK;
}

static int check_device(const lr_workspace *ws, hipStream_t st, const char *who)
{
    if (!ws) { lr_set_error("%s: null workspace", who); return LR_EINVAL; }
    int cur = -1;
    LR_HIP(hipGetDevice(&cur));
    if (g_fake_device >= 0) cur = g_fake_device;
    if (cur != ws->device) {
        lr_set_error("%s: the workspace was created on device %d but device %d is current (hipSetDevice / torch.cuda.set_device before the call; one workspace per device)",
                     who, ws->device, cur);
        return LR_EINVAL;
    }
    if (st) {
        hipDevice_t sd = -1;
        if (hipStreamGetDevice(st, &sd) == hipSuccess) {
            if ((int)sd != ws->device) { lr_set_error("%s: the stream belongs to device %d, the workspace to device %d", who, (int)sd, ws->device); return LR_EINVAL; }
        } else (void)hipGetLastError();
    }
    return LR_OK;
}
#define LR_CHECK_DEVICE(ws, stream, who) do { int rc_d_ = check_device(ws, (hipStream_t)(stream), who); if (rc_d_ != LR_OK) return rc_d_; } while (0)

// Test hook (no reference counterpart): from now on the entry points take `device` for the current device (-1: ask HIP again).  Lets the
// one-GPU test box exercise the wrong-device refusal.
extern "C" int lr_debug_fake_current_device(int device) { g_fake_device = device; return LR_OK; }

// ------------------------------------------------------------------ workspace
namespace {
struct Carver {
    size_t off = 0;
    char *base = nullptr;
    template <class T> T *take(size_t count)
    {
        off = (off + 255) & ~size_t(255);
        T *p = base ? reinterpret_cast<T *>(base + off) : nullptr;
        off += sizeof(T) * count;
        return p;
    }
};

void carve(lr_workspace *ws, Carver &c)
{
    const size_t n0 = ws->max_n0, n1 = ws->max_n1, n = ws->max_n, it = ws->max_iters;
    ws->nrm0 = c.take<float>(n0); ws->nrm1 = c.take<float>(n1);
    ws->H0 = c.take<_Float16>(n0 * 32); ws->H1 = c.take<_Float16>(n1 * 32);
    if (ws->dim != LR_FEAT_DIM) { ws->P0 = c.take<float>(n0 * 32); ws->P1 = c.take<float>(n1 * 32); }
    ws->tau = c.take<float>(n);
    ws->yfin = c.take<float>(n * LR_NN_MAX_STRIPS);
    ws->yshare = c.take<uint32_t>(n);
    ws->cand_cnt = c.take<int32_t>(LR_NN16_CNT_INTS(n)); ws->cand = c.take<int32_t>(LR_NN16_SEG_INTS(n));
    ws->counters = c.take<int32_t>(LR_CNT_TOTAL);
    ws->bmax0 = c.take<float>(n0 / 32 + 2); ws->bmax1 = c.take<float>(n1 / 32 + 2);
    ws->bmin0 = c.take<float>(n0 / 32 + 2); ws->bmin1 = c.take<float>(n1 / 32 + 2);
    ws->nn_range = c.take<float>(8);
    ws->rev_seed = c.take<uint32_t>(n1); ws->rev_rows = c.take<int32_t>(n1);
    ws->rev_seed64 = c.take<unsigned long long>(n1);
    ws->rev_cols = c.take<int32_t>(n0); ws->rev_s1 = c.take<float>(n0); ws->rev_pos = c.take<int32_t>(n0);
    ws->rev_tmin = c.take<float>(n0 / 32 + 2); ws->rev_hist = c.take<int32_t>(2 * 4096);
    ws->Hs = c.take<_Float16>(n0 * 32); ws->nrms = c.take<float>(n0);
    ws->nn_idx1 = c.take<int32_t>(n0); ws->nn_idx2 = c.take<int32_t>(n0);
    ws->nn_s1 = c.take<float>(n0); ws->nn_s2 = c.take<float>(n0);
    ws->rev_idx1 = c.take<int32_t>(n1);
    ws->is_bb = c.take<uint8_t>(n0);
    ws->blk_cnt = c.take<int32_t>(n0 / 256 + 2);
    ws->corr_idx0 = c.take<int32_t>(n0); ws->corr_idx1 = c.take<int32_t>(n0); ws->corr_idx2 = c.take<int32_t>(n0);
    ws->corr_score = c.take<float>(n0);
    ws->ratio = c.take<float>(n0);
    ws->cell = c.take<int32_t>(n0); ws->cell_sorted = c.take<int32_t>(n0);
    ws->gpf_quota = c.take<double>(LR_GPF_MAX_CELLS);
    ws->gpf_cells = c.take<int32_t>(3 * (LR_GPF_MAX_CELLS + 8) > 2 * (LR_PR_BUCKETS + 8) + 8 ? 3 * (LR_GPF_MAX_CELLS + 8) : 2 * (LR_PR_BUCKETS + 8) + 8);
    ws->gpf_keep = c.take<uint8_t>(n0);
    ws->gpf_f = c.take<float>(8);
    ws->corr8 = c.take<float>((n0 + 2) * 8);
    ws->prosac_G = c.take<int32_t>(n0 + 2); ws->prosac_rank = c.take<int32_t>(n0);
    ws->models = c.take<float>(it * 12);
    ws->models64 = c.take<double>(it * 12);
    ws->model_h = c.take<int32_t>(it);
    ws->models2 = c.take<float>(it * 12); ws->models64_2 = c.take<double>(it * 12); ws->model_h2 = c.take<int32_t>(it);
    ws->score_cnt = c.take<uint32_t>(it);
    ws->score_ssq = c.take<unsigned long long>(it);
    ws->refit_part = c.take<double>((n0 / 256 + 2) * 16);
    ws->icp_ints = c.take<int32_t>(3 * (32768 + 8));
    ws->icp_bucket = c.take<int32_t>(n1); ws->icp_pts = c.take<float>(4 * n1);
    ws->icp_state = c.take<double>(32); ws->icp_part = c.take<double>((n0 / 256 + 2) * 18);
    ws->lo_list = c.take<int32_t>(n0);
    ws->lo_ctl = c.take<char>(LR_LO_CTL_BYTES);
    ws->sc_info = c.take<char>(LR_SC_INFO_BYTES);
    ws->corr8s = c.take<float>((n0 + 4) * 8);
    ws->sc_perm = c.take<int32_t>(it); ws->sc_glen = c.take<int32_t>(it / 64 + 2);
    ws->sc_mb = c.take<uint8_t>(it); ws->sc_cb = c.take<uint8_t>(n0);
    ws->models_s = c.take<float>(it * 12);
    ws->res_tmp = c.take<lr_ransac_result>(1);
    ws->T_tmp = c.take<double>(48);      // [0,16) RANSAC model, [16,32) refit, [32,48) ICP
}
}  // namespace

extern "C" int lr_workspace_create(lr_workspace **out, int max_n0, int max_n1, int dim, int max_iters)
{
    return lr_workspace_create_batch(out, 1, max_n0, max_n1, dim, max_iters);
}

extern "C" int lr_workspace_create_batch(lr_workspace **out, int max_pairs, int max_n0, int max_n1, int dim, int max_iters)
{
    LR_REQUIRE(out, LR_EINVAL, "lr_workspace_create: null output");
    LR_REQUIRE(max_pairs >= 1 && max_pairs <= LR_MAX_BATCH, LR_EINVAL, "lr_workspace_create_batch: max_pairs must be in [1, 64]");
    LR_REQUIRE(max_n0 > 0 && max_n1 > 0 && max_iters >= 0, LR_EINVAL, "lr_workspace_create: sizes must be positive");
    LR_REQUIRE(dim >= 1 && dim <= LR_FEAT_DIM, LR_EINVAL, "lr_workspace_create: descriptors of 1 to 32 dimensions are supported");
    LR_REQUIRE(max_n0 < (1 << 22) && max_n1 < (1 << 22), LR_ESIZE, "lr_workspace_create: clouds are limited to 2^22 points");
    int dev = -1;
    LR_HIP(hipGetDevice(&dev));
    const lr_device_info *di = nullptr;
    LR_TRY_HIP(device_info(dev, &di));
    if (!di->ok) {
        lr_set_error("lr_workspace_create: device %d is %s; this library holds gfx950 (MI355X) code objects only", dev, di->arch);
        return LR_EINVAL;
    }
    lr_workspace *ws = new (std::nothrow) lr_workspace();
    LR_REQUIRE(ws, LR_ENOMEM, "lr_workspace_create: host allocation failed");
    memset(ws, 0, sizeof(*ws));
    ws->device = dev; ws->n_cus = di->cus;
    ws->max_n0 = max_n0; ws->max_n1 = max_n1; ws->max_n = max_n0 > max_n1 ? max_n0 : max_n1;
    ws->dim = dim; ws->max_iters = max_iters > 0 ? max_iters : 1;
    ws->max_pairs = max_pairs; ws->zP = 1; ws->z = lr_zargs{ 0, nullptr };
    // tuning defaults (lr_workspace_option changes them; no environment variable is read anywhere in this library)
    ws->nn_blocks_target = 3 * ws->n_cus;      // 3 blocks per CU = every block of a single pair's filter pass resident at once (768 on an MI355X)
    ws->nn_blocks_batch = 12 * ws->n_cus;      // (3072)
    ws->nn_sample_stride = 0; ws->rev_strips = 0; ws->nn_second_auto = 0; ws->clock_probe = 0;
    Carver sizing;
    carve(ws, sizing);
    ws->stride = (sizing.off + 511) & ~size_t(255);          // one arena per pair, identical layout
    const size_t desc_off = ws->stride * (size_t)max_pairs;
    const size_t clk_off = (desc_off + sizeof(lr_pair_desc) * LR_MAX_BATCH + 255) & ~size_t(255);
    ws->bytes = clk_off + 256;
    hipError_t e = hipMalloc(&ws->base, ws->bytes);
    if (e != hipSuccess) {
        lr_set_error("lr_workspace_create: hipMalloc(%zu) -> %s", ws->bytes, hipGetErrorString(e));
        delete ws;
        return e == hipErrorOutOfMemory ? LR_ENOMEM : LR_EHIP;
    }
    Carver real;
    real.base = ws->base;
    carve(ws, real);
    ws->descs = reinterpret_cast<lr_pair_desc *>(ws->base + desc_off);
    ws->clk_dev = reinterpret_cast<unsigned long long *>(ws->base + clk_off);
    // (optional: without it every call launches both forms of the filter pass, as batched calls always do)
    if (hipHostMalloc(reinterpret_cast<void **>(&ws->form_host), 2 * sizeof(int32_t), hipHostMallocMapped) == hipSuccess) {
        ws->form_host[0] = ws->form_host[1] = 0;
        if (hipHostGetDevicePointer(reinterpret_cast<void **>(&ws->form_dev), ws->form_host, 0) != hipSuccess) { (void)hipHostFree(ws->form_host); ws->form_host = ws->form_dev = nullptr; }
    } else { (void)hipGetLastError(); ws->form_host = ws->form_dev = nullptr; }
    // The fill runs on the null stream; the caller's streams may be non-blocking ones (torch's are), which do not wait for it: without
    // the synchronisation the first call on the new workspace can overtake the fill, which then wipes what that call wrote (found by
    // tests/test_gpu_gc.py::test_lo_helper_protocol_under_contention, round 4: workspaces created while other streams keep the GPU busy)
    if (hipMemset(ws->base, 0, ws->bytes) != hipSuccess || hipStreamSynchronize(nullptr) != hipSuccess) {
        if (ws->form_host) (void)hipHostFree(ws->form_host);
        (void)hipFree(ws->base); delete ws; lr_set_error("lr_workspace_create: hipMemset failed"); return LR_EHIP;
    }
    for (int k = 0; k < LR_NEV; ++k)
        if (hipEventCreate(&ws->ev[k]) != hipSuccess) {
            // (everything created so far goes: the events before this one, the pinned hint, the arena)
            for (int q = 0; q < k; ++q) (void)hipEventDestroy(ws->ev[q]);
            if (ws->form_host) (void)hipHostFree(ws->form_host);
            (void)hipFree(ws->base); delete ws; lr_set_error("lr_workspace_create: hipEventCreate failed"); return LR_EHIP;
        }
    *out = ws;
    return LR_OK;
}

extern "C" int lr_workspace_destroy(lr_workspace *ws)
{
    if (!ws) return LR_OK;
    if (ws->form_host) (void)hipHostFree(ws->form_host);
    for (int k = 0; k < LR_NEV; ++k) (void)hipEventDestroy(ws->ev[k]);
    (void)hipFree(ws->base);
    delete ws;
    return LR_OK;
}

extern "C" size_t lr_workspace_bytes(const lr_workspace *ws) { return ws ? ws->bytes : 0; }

extern "C" int lr_workspace_option(lr_workspace *ws, int option, int value)
{
    LR_REQUIRE(ws, LR_EINVAL, "lr_workspace_option: null workspace");
    LR_REQUIRE(value >= 0, LR_EINVAL, "lr_workspace_option: value must be >= 0");
    switch (option) {
    case LR_OPT_NN_BLOCKS: ws->nn_blocks_target = value > 0 ? value : 3 * ws->n_cus; break;
    case LR_OPT_NN_BLOCKS_BATCH: ws->nn_blocks_batch = value > 0 ? value : 12 * ws->n_cus; break;
    case LR_OPT_NN_SAMPLE_STRIDE: ws->nn_sample_stride = value > 4096 ? 4096 : value; break;      // (lr_nn16_run clamps it to the strip length)
    case LR_OPT_REV_STRIPS: ws->rev_strips = value > 64 ? 64 : value; break;
    case LR_OPT_NN_SECOND_AUTO: ws->nn_second_auto = value ? 1 : 0; break;
    case LR_OPT_CLOCK_PROBE: ws->clock_probe = value ? 1 : 0; break;
    default: lr_set_error("lr_workspace_option: unknown option %d", option); return LR_EINVAL;
    }
    return LR_OK;
}

// test hook: overwrite the whole scratch arena with one byte value.  No entry point may depend on what an earlier call
// (or hipMalloc) left in the scratch; the parity tests poison it with different patterns and expect identical results.
extern "C" int lr_workspace_poison(lr_workspace *ws, int byte, void *stream)
{
    LR_CHECK_DEVICE(ws, stream, "lr_workspace_poison");
    forget_last_batch(ws);          // (the descriptor table and the result temporaries are overwritten)
    LR_HIP(hipMemsetAsync(ws->base, byte & 0xff, ws->bytes, (hipStream_t)stream));
    return LR_OK;
}

extern "C" int lr_workspace_lists(lr_workspace *ws, int n0, int32_t *nn_idx1, int32_t *nn_idx2,
                                  int32_t *corr_idx0, int32_t *corr_idx1, void *stream)
{
    return lr_workspace_lists_at(ws, 0, n0, nn_idx1, nn_idx2, corr_idx0, corr_idx1, stream);
}

extern "C" int lr_workspace_lists_at(lr_workspace *ws, int pair, int n0, int32_t *nn_idx1, int32_t *nn_idx2,
                                     int32_t *corr_idx0, int32_t *corr_idx1, void *stream)
{
    LR_CHECK_DEVICE(ws, stream, "lr_workspace_lists");
    LR_REQUIRE(pair >= 0 && pair < ws->max_pairs, LR_EINVAL, "lr_workspace_lists: pair outside the workspace");
    LR_REQUIRE(pair < ws->last_npairs, LR_EINVAL, "lr_workspace_lists: the last registration call on this workspace had fewer pairs");
    LR_REQUIRE(n0 > 0 && n0 <= ws->max_n0, LR_ESIZE, "lr_workspace_lists: n0 exceeds the workspace");
    hipStream_t st = (hipStream_t)stream;
    const size_t nb = sizeof(int32_t) * (size_t)n0, off = (size_t)pair * ws->stride;
    auto at = [&](const int32_t *p) { return reinterpret_cast<const char *>(p) + off; };
    if (nn_idx1) LR_HIP(hipMemcpyAsync(nn_idx1, at(ws->nn_idx1), nb, hipMemcpyDeviceToDevice, st));
    if (nn_idx2) LR_HIP(hipMemcpyAsync(nn_idx2, at(ws->nn_idx2), nb, hipMemcpyDeviceToDevice, st));
    if (corr_idx0) LR_HIP(hipMemcpyAsync(corr_idx0, at(ws->corr_idx0), nb, hipMemcpyDeviceToDevice, st));
    if (corr_idx1) LR_HIP(hipMemcpyAsync(corr_idx1, at(ws->corr_idx1), nb, hipMemcpyDeviceToDevice, st));
    return LR_OK;
}

// the same lists for the first `npairs` pairs of the last call at once: out[k * width + i], one strided copy per list
extern "C" int lr_workspace_lists_batch(lr_workspace *ws, int npairs, int width, int32_t *nn_idx1, int32_t *nn_idx2,
                                        int32_t *corr_idx0, int32_t *corr_idx1, void *stream)
{
    LR_CHECK_DEVICE(ws, stream, "lr_workspace_lists_batch");
    LR_REQUIRE(npairs >= 1 && npairs <= ws->last_npairs, LR_EINVAL, "lr_workspace_lists_batch: the last registration call on this workspace had fewer pairs");
    LR_REQUIRE(width > 0 && width <= ws->max_n0, LR_ESIZE, "lr_workspace_lists_batch: width exceeds the workspace");
    hipStream_t st = (hipStream_t)stream;
    const size_t wb = sizeof(int32_t) * (size_t)width;
    auto copy = [&](int32_t *dst, const int32_t *src) -> int {
        if (!dst) return LR_OK;
        LR_HIP(hipMemcpy2DAsync(dst, wb, src, ws->stride, wb, (size_t)npairs, hipMemcpyDeviceToDevice, st));
        return LR_OK;
    };
    LR_TRY_HIP(copy(nn_idx1, ws->nn_idx1)); LR_TRY_HIP(copy(nn_idx2, ws->nn_idx2));
    LR_TRY_HIP(copy(corr_idx0, ws->corr_idx0)); LR_TRY_HIP(copy(corr_idx1, ws->corr_idx1));
    return LR_OK;
}

// zero `bytes` (a multiple of 4) of scratch at p (an arena-0 pointer) in the arena of every pair of the call in flight.  A kernel, not
// hipMemset2DAsync: one launch whatever the number of pairs, and a plain kernel node when the caller captures the call in a HIP graph
// (the 2-D memset node faulted on replay with the arena stride as its pitch: tests/test_gpu_batch.py, round 4).
__global__ void __launch_bounds__(256) zero_scratch_kernel(uint32_t *__restrict__ p, size_t words, lr_zargs z)
{
    lr_z(p, z, blockIdx.z);
    for (size_t k = (size_t)blockIdx.x * 256 + threadIdx.x; k < words; k += (size_t)gridDim.x * 256) p[k] = 0u;
}
int lr_zero_scratch(lr_workspace *ws, void *p, size_t bytes, hipStream_t st)
{
    const size_t words = (bytes + 3) / 4;
    int blocks = (int)((words + 255) / 256);
    if (blocks > 1024) blocks = 1024;
    hipLaunchKernelGGL(zero_scratch_kernel, dim3(blocks < 1 ? 1 : blocks, 1, ws->zP), dim3(256), 0, st, reinterpret_cast<uint32_t *>(p), words, ws->z);
    LR_LAUNCH_CHECK();
    return LR_OK;
}

// Shader clock of the filter-pass blocks since the last reset (LR_OPT_CLOCK_PROBE): every block that walks adds its s_memtime cycles and its
// s_memrealtime ticks (100 MHz) to two words of the workspace; MHz = 100 * cycles / ticks, weighted by block lifetime.  The caller has
// synchronised the streams that used the workspace (blocking copies on the null stream).
extern "C" int lr_workspace_clock(lr_workspace *ws, double *mhz, unsigned long long *cycles, unsigned long long *ticks, int reset)
{
    LR_CHECK_DEVICE(ws, nullptr, "lr_workspace_clock");
    unsigned long long h[2] = { 0ull, 0ull };
    LR_HIP(hipMemcpy(h, ws->clk_dev, sizeof h, hipMemcpyDeviceToHost));
    if (cycles) *cycles = h[0];
    if (ticks) *ticks = h[1];
    if (mhz) *mhz = h[1] ? 100.0 * (double)h[0] / (double)h[1] : 0.0;
    if (reset) LR_HIP(hipMemset(ws->clk_dev, 0, sizeof h));
    return LR_OK;
}

extern "C" int lr_workspace_timing(lr_workspace *ws, int enable)
{
    LR_CHECK_DEVICE(ws, nullptr, "lr_workspace_timing");
    ws->timing = enable; ws->ev_pending = 0; ws->rev_recorded = 0; ws->nn_ms_acc = 0; ws->ransac_ms_acc = 0; ws->n_samples = 0;
    ws->call_ms_acc = 0; ws->fwd_ms_acc = 0; ws->fwd_filter_ms_acc = 0; ws->rev_filter_ms_acc = 0; ws->rev_ms_acc = 0; ws->rev_done_recorded = 0;
    return LR_OK;
}

// folds the events of the last timed call into the sums
static int lr_timing_collect(lr_workspace *ws)
{
    if (ws->ev_pending >= 1) {
        float ms = 0;
        LR_HIP(hipEventElapsedTime(&ms, ws->ev[0], ws->ev[1]));
        ws->nn_ms_acc += ms; ws->fwd_filter_ms_acc += ms;
        if (ws->rev_recorded) { LR_HIP(hipEventElapsedTime(&ms, ws->ev[4], ws->ev[5])); ws->nn_ms_acc += ms; ws->rev_filter_ms_acc += ms; ws->rev_recorded = 0; }
        if (ws->ev_pending >= 2) { LR_HIP(hipEventElapsedTime(&ms, ws->ev[2], ws->ev[3])); ws->ransac_ms_acc += ms; }
        if (ws->ev_pending >= 3) {
            LR_HIP(hipEventElapsedTime(&ms, ws->ev[6], ws->ev[8])); ws->call_ms_acc += ms;
            LR_HIP(hipEventElapsedTime(&ms, ws->ev[6], ws->ev[7])); ws->fwd_ms_acc += ms;
            if (ws->rev_done_recorded) { LR_HIP(hipEventElapsedTime(&ms, ws->ev[7], ws->ev[9])); ws->rev_ms_acc += ms; }
        }
        ws->rev_done_recorded = 0;
        ws->n_samples += 1;
        ws->ev_pending = 0;
    }
    return LR_OK;
}

// Reads the events of the last timed pair; the caller has synchronised the stream.
extern "C" int lr_workspace_timing_read(lr_workspace *ws, float *nn_ms, float *ransac_ms, int *n_samples)
{
    LR_CHECK_DEVICE(ws, nullptr, "lr_workspace_timing_read");
    LR_TRY_HIP(lr_timing_collect(ws));
    if (nn_ms) *nn_ms = ws->nn_ms_acc;
    if (ransac_ms) *ransac_ms = ws->ransac_ms_acc;
    if (n_samples) *n_samples = ws->n_samples;
    return LR_OK;
}

// Stage times of the timed pair-pipeline calls so far (sums in ms over *n_samples calls; the caller has synchronised the stream):
// out[0] the whole call, out[1] the forward NN (norms + f16 copies, filter pass, exact verification: find_nn of matching.py:22-65 with
// the second neighbour), out[2] / out[3] the forward / reverse filter-pass launch, out[4] hypothesis generation + scoring of the
// first RANSAC batch.  What FR.py:117 bills as registration time is out[0] - out[1] + (the second neighbour's share of out[1]).
extern "C" int lr_workspace_stage_times(lr_workspace *ws, float out[8], int *n_samples)
{
    LR_REQUIRE(ws && out, LR_EINVAL, "lr_workspace_stage_times: null pointer");
    LR_CHECK_DEVICE(ws, nullptr, "lr_workspace_stage_times");
    LR_TRY_HIP(lr_timing_collect(ws));
    out[0] = ws->call_ms_acc; out[1] = ws->fwd_ms_acc; out[2] = ws->fwd_filter_ms_acc; out[3] = ws->rev_filter_ms_acc; out[4] = ws->ransac_ms_acc;
    out[5] = ws->rev_ms_acc;
    out[6] = out[7] = 0.0f;
    if (n_samples) *n_samples = ws->n_samples;
    return LR_OK;
}

// ------------------------------------------------------------------ checks
static int check_nn_args(const lr_workspace *ws, const void *F0, int n0, const void *F1, int n1, int dim, const char *who)
{
    if (!ws || !F0 || !F1) { lr_set_error("%s: null pointer", who); return LR_EINVAL; }
    if (dim != ws->dim) { lr_set_error("%s: dim %d != workspace dim %d", who, dim, ws->dim); return LR_EINVAL; }
    if (n0 <= 0 || n1 <= 0) { lr_set_error("%s: empty cloud (n0=%d, n1=%d)", who, n0, n1); return LR_EINVAL; }
    if (n0 > ws->max_n0 || n1 > ws->max_n1) { lr_set_error("%s: (%d,%d) exceeds workspace (%d,%d)", who, n0, n1, ws->max_n0, ws->max_n1); return LR_ESIZE; }
    return LR_OK;
}

#define LR_TRY(x) do { int rc_ = (x); if (rc_ != LR_OK) return rc_; } while (0)


// The params structs start with their own size: a caller built against another version of include/lidarreg.h is turned away instead
// of having a shorter struct read past its end (lr_version 102)
static int check_ransac_params(const lr_ransac_params *p, const char *who)
{
    if (!p) { lr_set_error("%s: null params", who); return LR_EINVAL; }
    if (p->struct_size != sizeof(lr_ransac_params)) {
        lr_set_error("%s: lr_ransac_params.struct_size is %u, this library (lr_version %d) expects %zu -- set it to sizeof(lr_ransac_params) / rebuild against include/lidarreg.h",
                     who, p->struct_size, lr_version(), sizeof(lr_ransac_params));
        return LR_EINVAL;
    }
    return LR_OK;
}
static int check_pair_params(const lr_pair_params *p, const char *who)
{
    if (!p) { lr_set_error("%s: null params", who); return LR_EINVAL; }
    if (p->struct_size != sizeof(lr_pair_params)) {
        lr_set_error("%s: lr_pair_params.struct_size is %u, this library (lr_version %d) expects %zu -- set it to sizeof(lr_pair_params) / rebuild against include/lidarreg.h",
                     who, p->struct_size, lr_version(), sizeof(lr_pair_params));
        return LR_EINVAL;
    }
    return check_ransac_params(&p->ransac, who);
}

// Descriptors narrower than 32 (matching.py:22-65 is dimension-agnostic; FCGF_FAST/net/BBR_F.py:148-176 calls it with D = 3): zero-padded
// fp32 copies in the workspace, and everything downstream runs 32 wide on them.  A zero term changes neither the fma chains of the
// arithmetic contract (fma(0, 0, acc) == acc) nor the ratio's sum of squared differences: results are bit-identical to the dim-wide
// definitions (the oracle's).  One thread per element, rows of the input `dim` floats apart (no alignment assumed).
__global__ void __launch_bounds__(256)
pad_feats_kernel(const float *__restrict__ Fa, int na, float *__restrict__ Pa, const float *__restrict__ Fb, int nb, float *__restrict__ Pb,
                 int dim, int nblk_a, lr_zargs z)
{
    if (z.descs) { const lr_pair_desc d = z.descs[blockIdx.z]; Fa = d.F0; na = d.n0; Fb = d.F1; nb = d.n1; }
    lr_z(Pa, z, blockIdx.z); lr_z(Pb, z, blockIdx.z);
    const bool second = (int)blockIdx.x >= nblk_a;
    const float *__restrict__ F = second ? Fb : Fa;
    float *__restrict__ P = second ? Pb : Pa;
    const int n = second ? nb : na;
    const int row = (second ? blockIdx.x - nblk_a : blockIdx.x) * 8 + (threadIdx.x >> 5), k = threadIdx.x & 31;
    if (row < n) P[(size_t)row * 32 + k] = k < dim ? F[(size_t)row * dim + k] : 0.0f;
}
// (a batched call's descriptor table then names the padded copies: every later kernel reads its inputs through it)
__global__ void pad_patch_descs_kernel(lr_pair_desc *__restrict__ descs, int npairs, float *P0, float *P1, size_t stride)
{
    const int k = threadIdx.x;
    if (k < npairs) {
        descs[k].F0 = reinterpret_cast<const float *>(reinterpret_cast<const char *>(P0) + (size_t)k * stride);
        descs[k].F1 = reinterpret_cast<const float *>(reinterpret_cast<const char *>(P1) + (size_t)k * stride);
    }
}
static void pad_if_narrow(lr_workspace *ws, const float *&F0, int n0, const float *&F1, int n1, int &dim, hipStream_t st)
{
    if (dim == LR_FEAT_DIM) return;
    const int nblk_a = lr_cdiv(n0, 8);
    hipLaunchKernelGGL(pad_feats_kernel, dim3(nblk_a + lr_cdiv(n1, 8), 1, ws->zP), dim3(256), 0, st, F0, n0, ws->P0, F1, n1, ws->P1, dim, nblk_a, ws->z);
    if (ws->z.descs) hipLaunchKernelGGL(pad_patch_descs_kernel, dim3(1), dim3(64), 0, st, ws->descs, ws->zP, ws->P0, ws->P1, ws->stride);
    F0 = ws->P0; F1 = ws->P1; dim = LR_FEAT_DIM;
}

// norms + f16 operand copies of both clouds
static int prep_both(lr_workspace *ws, const float *F0, int n0, const float *F1, int n1, hipStream_t st, bool zero_counters = false)
{
    return lr_nn16_prep(ws, F0, n0, F1, n1, st, zero_counters);      // the prep kernel clears the counter block itself
}

// forward: rows of cloud 0 against cloud 1 (first + second NN)
static int nn_forward(lr_workspace *ws, const float *F0, int n0, const float *F1, int n1,
                      int32_t *idx1, int32_t *idx2, float *s1, float *s2, hipStream_t st, bool seed_reverse = false)
{
    return lr_nn16_run(ws, F0, ws->H0, ws->nrm0, n0, F1, ws->H1, ws->nrm1, ws->nn_range + 2, n1,
                       idx2 ? 2 : 1, idx1, idx2, s1, idx2 ? s2 : nullptr, st, seed_reverse);
}

// reverse: rows of cloud 1 against cloud 0 (first NN only).  The reference restricts it to the unique forward
// targets (matching.py:224-225); rows that are nobody's target never enter the intersection, so all rows is equivalent.
static int nn_reverse(lr_workspace *ws, const float *F0, int n0, const float *F1, int n1, const int32_t *fwd_idx1,
                      int32_t *rev, hipStream_t st, bool seeded = false)
{
    // seeded by the forward pairs: only columns some query points at are resolved (others get -1)
    return lr_nn16_reverse(ws, F0, ws->H0, ws->nrm0, ws->bmax0, n0, F1, ws->H1, ws->nrm1, n1, fwd_idx1, rev, st, seeded);
}

// ------------------------------------------------------------------ a1/a2
extern "C" int lr_nn_top2(lr_workspace *ws, const float *F0, int n0, const float *F1, int n1, int dim,
                          int32_t *idx1, int32_t *idx2, float *s1, float *s2, void *stream)
{
    LR_TRY(check_nn_args(ws, F0, n0, F1, n1, dim, "lr_nn_top2"));
    LR_CHECK_DEVICE(ws, stream, "lr_nn_top2");
    forget_last_batch(ws);
    LR_REQUIRE(idx1, LR_EINVAL, "lr_nn_top2: idx1 is required");
    hipStream_t st = (hipStream_t)stream;
    pad_if_narrow(ws, F0, n0, F1, n1, dim, st);
    LR_TRY(prep_both(ws, F0, n0, F1, n1, st));
    return nn_forward(ws, F0, n0, F1, n1, idx1, idx2, s1, s2, st);
}

// ------------------------------------------------------------------ a3-a5
extern "C" int lr_nn_to_mutual(lr_workspace *ws, const float *F0, int n0, const float *F1, int n1, int dim,
                               const int32_t *idx1, const int32_t *idx2, uint8_t *is_bb,
                               int32_t *o0, int32_t *o1, int32_t *o2, int32_t *n_out, void *stream)
{
    LR_TRY(check_nn_args(ws, F0, n0, F1, n1, dim, "lr_nn_to_mutual"));
    LR_CHECK_DEVICE(ws, stream, "lr_nn_to_mutual");
    forget_last_batch(ws);
    LR_REQUIRE(idx1, LR_EINVAL, "lr_nn_to_mutual: idx1 is required");
    hipStream_t st = (hipStream_t)stream;
    pad_if_narrow(ws, F0, n0, F1, n1, dim, st);
    LR_TRY(prep_both(ws, F0, n0, F1, n1, st));
    LR_TRY(nn_reverse(ws, F0, n0, F1, n1, idx1, ws->rev_idx1, st));
    return lr_mutual_run(ws, n0, idx1, idx2, ws->rev_idx1, is_bb, o0, o1, o2, n_out, st);
}

// ------------------------------------------------------------------ a7
extern "C" int lr_gpf(lr_workspace *ws, const float *F0, int n0, const float *F1, int n1, int dim,
                      const int32_t *idx1, const int32_t *idx2, const float *xyz0, int grid_wid, double factor,
                      int32_t *o0, int32_t *o1, int32_t *o2, float *oscore, int32_t *n_out, void *stream)
{
    LR_TRY(check_nn_args(ws, F0, n0, F1, n1, dim, "lr_gpf"));
    LR_CHECK_DEVICE(ws, stream, "lr_gpf");
    forget_last_batch(ws);
    LR_REQUIRE(idx1 && idx2 && xyz0 && o0 && o1, LR_EINVAL, "lr_gpf: null pointer");
    hipStream_t st = (hipStream_t)stream;
    pad_if_narrow(ws, F0, n0, F1, n1, dim, st);
    LR_TRY(prep_both(ws, F0, n0, F1, n1, st));
    LR_TRY(nn_reverse(ws, F0, n0, F1, n1, idx1, ws->rev_idx1, st));
    LR_TRY(lr_mutual_run(ws, n0, idx1, nullptr, ws->rev_idx1, ws->is_bb, nullptr, nullptr, nullptr, nullptr, st));
    return lr_gpf_run(ws, F0, n0, F1, dim, idx1, idx2, ws->is_bb, xyz0, grid_wid, factor, o0, o1, o2, oscore, n_out, st);
}

// a7, BB_first=True: mutual pairs first, then the grid filter over them with TOTAL_NUM = max_matches
extern "C" int lr_gpf_bb_first(lr_workspace *ws, const float *F0, int n0, const float *F1, int n1, int dim,
                               const int32_t *idx1, const int32_t *idx2, const float *xyz0, int grid_wid, double max_matches,
                               int32_t *o0, int32_t *o1, int32_t *o2, float *oscore, int32_t *n_out, int32_t *has_score,
                               void *stream)
{
    LR_TRY(check_nn_args(ws, F0, n0, F1, n1, dim, "lr_gpf_bb_first"));
    LR_CHECK_DEVICE(ws, stream, "lr_gpf_bb_first");
    forget_last_batch(ws);
    LR_REQUIRE(idx1 && idx2 && xyz0 && o0 && o1 && o2 && n_out && has_score, LR_EINVAL, "lr_gpf_bb_first: null pointer");
    hipStream_t st = (hipStream_t)stream;
    pad_if_narrow(ws, F0, n0, F1, n1, dim, st);
    int32_t *mb_dev = ws->counters + LR_CNT_NCORR;
    LR_TRY(prep_both(ws, F0, n0, F1, n1, st));
    LR_TRY(nn_reverse(ws, F0, n0, F1, n1, idx1, ws->rev_idx1, st));
    LR_TRY(lr_mutual_run(ws, n0, idx1, idx2, ws->rev_idx1, ws->is_bb, ws->corr_idx0, ws->corr_idx1, ws->corr_idx2, mb_dev, st));
    return lr_gpf_bb_run(ws, F0, n0, F1, dim, ws->corr_idx0, ws->corr_idx1, ws->corr_idx2, mb_dev, xyz0, grid_wid, max_matches,
                         o0, o1, o2, oscore, n_out, has_score, st);
}

// ------------------------------------------------------------------ a10/a12
extern "C" int lr_ransac(lr_workspace *ws, const float *src, const float *tgt, int m, const int32_t *m_dev,
                         const lr_ransac_params *p, double *T_out, lr_ransac_result *res, void *stream)
{
    LR_REQUIRE(ws && src && tgt && p && T_out && res, LR_EINVAL, "lr_ransac: null pointer");
    LR_TRY(check_ransac_params(p, "lr_ransac"));
    LR_CHECK_DEVICE(ws, stream, "lr_ransac");
    forget_last_batch(ws);
    LR_REQUIRE(m >= 0 && m <= ws->max_n0, LR_ESIZE, "lr_ransac: m exceeds the workspace");
    hipStream_t st = (hipStream_t)stream;
    LR_TRY(lr_pack_corr(ws, src, tgt, nullptr, nullptr, m, m_dev, ws->corr8, st));
    return lr_ransac_run(ws, ws->corr8, m, m_dev, p, T_out, res, st);
}

// the inlier mask pygcransac returns next to the pose (gcransac_python.cpp:594-603)
extern "C" int lr_inlier_mask(lr_workspace *ws, const float *src, const float *tgt, int m, const double *T, float thr2,
                              uint8_t *mask, int32_t *n_inliers, void *stream)
{
    LR_REQUIRE(src && tgt && mask && m >= 0, LR_EINVAL, "lr_inlier_mask: bad argument");
    LR_REQUIRE(T || ws, LR_EINVAL, "lr_inlier_mask: neither a model nor a workspace");
    LR_REQUIRE(thr2 > 0.0f, LR_EINVAL, "lr_inlier_mask: thr2 must be positive");
    if (!T) LR_CHECK_DEVICE(ws, stream, "lr_inlier_mask");
    return lr_inlier_mask_run(src, tgt, nullptr, nullptr, m, nullptr, T ? T : ws->T_tmp, thr2, mask, n_inliers, (hipStream_t)stream);
}

extern "C" int lr_workspace_mask_at(lr_workspace *ws, int pair, const float *xyz0, const float *xyz1, int n0, float thr2,
                                    uint8_t *mask, int32_t *n_inliers, void *stream)
{
    LR_REQUIRE(ws && xyz0 && xyz1 && mask, LR_EINVAL, "lr_workspace_mask_at: null pointer");
    LR_REQUIRE(pair >= 0 && pair < ws->max_pairs, LR_EINVAL, "lr_workspace_mask_at: pair outside the workspace");
    LR_REQUIRE(pair < ws->last_npairs, LR_EINVAL, "lr_workspace_mask_at: the last registration call on this workspace had fewer pairs");
    LR_REQUIRE(n0 > 0 && n0 <= ws->max_n0, LR_ESIZE, "lr_workspace_mask_at: n0 exceeds the workspace");
    LR_REQUIRE(thr2 > 0.0f, LR_EINVAL, "lr_workspace_mask_at: thr2 must be positive");
    LR_CHECK_DEVICE(ws, stream, "lr_workspace_mask_at");
    const size_t off = (size_t)pair * ws->stride;
    auto at = [&](auto *p) { return reinterpret_cast<decltype(p)>(reinterpret_cast<char *>(p) + off); };
    return lr_inlier_mask_run(xyz0, xyz1, at(ws->corr_idx0), at(ws->corr_idx1), n0, at(ws->counters) + LR_CNT_NCORR, at(ws->T_tmp), thr2,
                              mask, n_inliers, (hipStream_t)stream);
}

// ------------------------------------------------------------------ a11
extern "C" int lr_refit(lr_workspace *ws, const float *xyz0, int n0, const float *xyz1, const int32_t *idx1,
                        const double *T_in, double thr2, double *T_out, int32_t *n_inl, void *stream)
{
    LR_REQUIRE(ws && xyz0 && xyz1 && idx1 && T_in && T_out, LR_EINVAL, "lr_refit: null pointer");
    LR_REQUIRE(n0 > 0 && n0 <= ws->max_n0, LR_ESIZE, "lr_refit: n0 exceeds the workspace");
    LR_CHECK_DEVICE(ws, stream, "lr_refit");
    forget_last_batch(ws);          // (arena-0 counters and the refit scratch are overwritten)
    return lr_refit_run(ws, xyz0, n0, xyz1, idx1, T_in, thr2, T_out, n_inl, nullptr, (hipStream_t)stream);
}

// ------------------------------------------------------------------ f1: ICP
extern "C" int lr_icp(lr_workspace *ws, const float *xyz0, int n0, const float *xyz1, int n1, const double *T_init,
                      double max_dist, int max_iter, double rel_fitness, double rel_rmse, double *T_out, lr_icp_result *res, void *stream)
{
    LR_REQUIRE(ws && xyz0 && xyz1 && T_init && T_out, LR_EINVAL, "lr_icp: null pointer");
    LR_REQUIRE(n0 > 0 && n0 <= ws->max_n0 && n1 > 0 && n1 <= ws->max_n1, LR_ESIZE, "lr_icp: cloud exceeds the workspace");
    LR_CHECK_DEVICE(ws, stream, "lr_icp");
    forget_last_batch(ws);          // (the ICP scratch and the result temporaries are overwritten)
    return lr_icp_run(ws, xyz0, n0, xyz1, n1, T_init, nullptr, max_dist, max_iter, rel_fitness, rel_rmse, T_out, res, (hipStream_t)stream);
}

__global__ void pair_icp_kernel(const double *__restrict__ T_icp, const lr_icp_result *__restrict__ r, lr_pair_result *__restrict__ out, int have, lr_zargs z);

// The harness' ICP stage for every pair of the last lr_register_batch call (test.py:183-193 times it on its own, after the
// registration): starts from that call's final transforms, which are still in the arenas, over the clouds its descriptor table
// names; fills T_icp / icp of out[k].  One set of launches for all pairs, like every other stage.
extern "C" int lr_icp_batch(lr_workspace *ws, double max_dist, int max_iter, double rel_fitness, double rel_rmse, lr_pair_result *out, void *stream)
{
    LR_REQUIRE(ws && out, LR_EINVAL, "lr_icp_batch: null pointer");
    LR_CHECK_DEVICE(ws, stream, "lr_icp_batch");
    LR_REQUIRE(ws->last_batch && ws->last_npairs >= 1 && ws->last_T_final, LR_EINVAL,
               "lr_icp_batch: no successful lr_register_batch on this workspace since the last call that touched its scratch");
    hipStream_t st = (hipStream_t)stream;
    // the transforms it starts from are written by the registration call's stream: the ICP must be ordered behind it
    LR_REQUIRE(st == ws->last_stream, LR_EINVAL, "lr_icp_batch: must be enqueued on the stream of the lr_register_batch call it continues");
    ws->zP = ws->last_npairs; ws->z = lr_zargs{ ws->stride, ws->descs };
    lr_icp_result *icp_res = reinterpret_cast<lr_icp_result *>(ws->icp_state + 24);
    int rc = lr_icp_run(ws, nullptr, ws->last_mx0, nullptr, ws->last_mx1, ws->last_T_final, ws->res_tmp, max_dist, max_iter, rel_fitness, rel_rmse,
                        ws->T_tmp + 32, icp_res, st);
    if (rc == LR_OK) {
        hipLaunchKernelGGL(pair_icp_kernel, dim3(1, 1, ws->zP), dim3(64), 0, st, ws->T_tmp + 32, icp_res, out, 1, ws->z);
        if (hipGetLastError() != hipSuccess) { lr_set_error("lr_icp_batch: launch failed"); rc = LR_EHIP; }
    }
    ws->zP = 1; ws->z = lr_zargs{ 0, nullptr };
    return rc;
}

__global__ void pair_icp_kernel(const double *__restrict__ T_icp, const lr_icp_result *__restrict__ r, lr_pair_result *__restrict__ out, int have,
                                lr_zargs z)
{
    lr_z(T_icp, z, blockIdx.z); lr_z(r, z, blockIdx.z);
    out += blockIdx.z;
    const int k = threadIdx.x;
    if (k < 16) out->T_icp[k] = have ? T_icp[k] : out->T[k];
    if (k == 0) {
        if (have) out->icp = *r;
        else { out->icp.fitness = 0.0; out->icp.inlier_rmse = 0.0; out->icp.n_corr = 0; out->icp.iterations = 0; }
    }
}

// ------------------------------------------------------------------ a9: the whole pair
__global__ void pair_result_kernel(const double *__restrict__ T_ransac, const double *__restrict__ T_final,
                                   const lr_ransac_result *__restrict__ rr, const int32_t *__restrict__ counters,
                                   const int32_t *__restrict__ n_refit, lr_pair_result *__restrict__ out, lr_zargs z)
{
    lr_z(T_ransac, z, blockIdx.z); lr_z(T_final, z, blockIdx.z); lr_z(rr, z, blockIdx.z); lr_z(counters, z, blockIdx.z); lr_z(n_refit, z, blockIdx.z);
    out += blockIdx.z;
    const int k = threadIdx.x;
    if (k < 16) { out->T[k] = T_final[k]; out->T_ransac[k] = T_ransac[k]; }
    if (k == 0) {
        out->ransac = *rr;
        out->n_corr = counters[LR_CNT_NCORR];
        out->n_refit = n_refit ? *n_refit : 0;
        out->n_nn_fixed = counters[LR_CNT_FIX_TOTAL];
        out->status = rr->best_h < 0 ? 1 : 0;
        for (int q = 0; q < 8; ++q) out->reserved[q] = 0;
        out->reserved[1] = reinterpret_cast<const lr_ransac_state *>(counters + LR_CNT_COUNT)->lo_timeouts;
        out->reserved[2] = (counters[LR_CNT_FORM_MISS_F] ? 1 : 0) | (counters[LR_CNT_FORM_MISS_R] ? 2 : 0);
        out->icp.fitness = 0.0; out->icp.inlier_rmse = 0.0; out->icp.n_corr = 0; out->icp.iterations = 0;
    }
    if (k < 16) out->T_icp[k] = T_final[k];      // overwritten by pair_icp_kernel when the ICP stage runs
}

// The stages of FR() for the call in flight: one pair with its pointers as given (ws->zP == 1, ws->z.descs == nullptr), or
// ws->zP pairs described by ws->z.descs (n0 / n1 are then the largest cloud sizes of the batch: they size the grids).
static int register_stages(lr_workspace *ws, const float *xyz0, const float *xyz1, const float *F0, const float *F1,
                           int n0, int n1, int dim, const lr_pair_params *p, lr_pair_result *out, hipStream_t st)
{
    int32_t *m_dev = ws->counters + LR_CNT_NCORR;
    int32_t *n_refit = ws->counters + LR_CNT_COUNT - 2;
    pad_if_narrow(ws, F0, n0, F1, n1, dim, st);
    // 1. coarse correspondences (FR.py:38): first + second NN of every cloud-0 descriptor
    LR_TRY(prep_both(ws, F0, n0, F1, n1, st, true));
    const bool fuse_seed = p->mode != LR_MODE_NO_FILTER;   // the forward exact kernel seeds the reverse pass
    const bool timed = ws->timing && !ws->ev_pending;
    if (timed) LR_HIP(hipEventRecord(ws->ev[6], st));
    // The second neighbour (find_2nn, FR.py:38) feeds the feature-distance ratio only: GPF (matching.py:116) and the PROSAC
    // quality (FR.py:77).  By default it is computed as the reference does; with the option LR_OPT_NN_SECOND_AUTO it is left out
    // when no stage of this call reads it (plain mutual-NN / no filter with uniform sampling): the outputs are the same, the
    // candidate lists of the forward pass are half as long (and the reverse pass prunes less: not a gain on its own).
    const bool want2 = !(ws->nn_second_auto && p->mode != LR_MODE_GPF && p->ransac.sampler == 0);
    int32_t *idx2 = want2 ? ws->nn_idx2 : nullptr;
    LR_TRY(nn_forward(ws, F0, n0, F1, n1, ws->nn_idx1, idx2, ws->nn_s1, ws->nn_s2, st, fuse_seed));
    if (timed) LR_HIP(hipEventRecord(ws->ev[7], st));
    // 2. filter (FR.py:48-56)
    if (p->mode == LR_MODE_NO_FILTER) {
        LR_TRY(lr_identity_corr(ws, n0, ws->nn_idx1, idx2, ws->corr_idx0, ws->corr_idx1, idx2 ? ws->corr_idx2 : nullptr, m_dev, st));
    } else {
        LR_TRY(nn_reverse(ws, F0, n0, F1, n1, ws->nn_idx1, ws->rev_idx1, st, fuse_seed));
        if (timed) { LR_HIP(hipEventRecord(ws->ev[9], st)); ws->rev_done_recorded = 1; }
        if (p->mode == LR_MODE_MNN) {
            LR_TRY(lr_mutual_run(ws, n0, ws->nn_idx1, idx2, ws->rev_idx1, ws->is_bb, ws->corr_idx0, ws->corr_idx1,
                                 idx2 ? ws->corr_idx2 : nullptr, m_dev, st, xyz0, xyz1, ws->corr8));
        } else {
            LR_TRY(lr_mutual_run(ws, n0, ws->nn_idx1, nullptr, ws->rev_idx1, ws->is_bb, nullptr, nullptr, nullptr, nullptr, st));
            LR_TRY(lr_gpf_run(ws, F0, n0, F1, dim, ws->nn_idx1, ws->nn_idx2, ws->is_bb, xyz0, p->gpf_grid_wid, p->gpf_factor,
                              ws->corr_idx0, ws->corr_idx1, ws->corr_idx2, ws->corr_score, m_dev, st, xyz1, ws->corr8));
        }
    }
    // 3. RANSAC on the surviving pairs (FR.py:70-97); MNN / GPF pack the point pairs inside their compaction kernel
    if (p->ransac.sampler == 1) {
        // PROSAC (FR.py:73-80, GC_RANSAC.py:39-43): records re-packed best match quality first; quality = -feature-distance
        // ratio of the pair, or GPF's normalised feature distance
        LR_TRY(lr_prosac_order(ws, F0, F1, dim, p->mode == LR_MODE_GPF ? ws->corr_score : nullptr, n0, m_dev, st));
        LR_TRY(lr_pack_corr(ws, xyz0, xyz1, ws->corr_idx0, ws->corr_idx1, n0, m_dev, ws->corr8, st, ws->prosac_rank));
    } else if (p->mode == LR_MODE_NO_FILTER) LR_TRY(lr_pack_corr(ws, xyz0, xyz1, ws->corr_idx0, ws->corr_idx1, n0, m_dev, ws->corr8, st));
    LR_TRY(lr_ransac_run(ws, ws->corr8, n0, m_dev, &p->ransac, ws->T_tmp, ws->res_tmp, st));
    // 4. LS refit over the original NN pairs (FR.py:99-111)
    const double *T_final = ws->T_tmp;
    if (p->refit) {
        if (p->refit == 2)      // GC codebase: final least squares over the inliers among the FILTERED pairs RANSAC worked on
            LR_TRY(lr_refit_run(ws, xyz0, n0, xyz1, ws->corr_idx1, ws->T_tmp, p->refit_thr2, ws->T_tmp + 16, n_refit, ws->res_tmp, st, out,
                                ws->corr_idx0, m_dev));
        else if (p->refit == 3) // DGR register_FCGF: inverse-feature-distance weighted Procrustes over the original NN pairs
            LR_TRY(lr_refit_run(ws, xyz0, n0, xyz1, ws->nn_idx1, ws->T_tmp, p->refit_thr2, ws->T_tmp + 16, n_refit, ws->res_tmp, st, out,
                                nullptr, nullptr, F0, F1));
        else                    // open3D codebase: inliers over the ORIGINAL NN pairs (FR.py:99-111)
            LR_TRY(lr_refit_run(ws, xyz0, n0, xyz1, ws->nn_idx1, ws->T_tmp, p->refit_thr2,
                                ws->T_tmp + 16, n_refit, ws->res_tmp, st, out));
        T_final = ws->T_tmp + 16;
    } else
    hipLaunchKernelGGL(pair_result_kernel, dim3(1, 1, ws->zP), dim3(64), 0, st, ws->T_tmp, T_final, ws->res_tmp, ws->counters,
                       p->refit ? n_refit : (const int32_t *)nullptr, out, ws->z);
    ws->last_T_final = T_final; ws->last_mx0 = n0; ws->last_mx1 = n1;
    // 5. ICP refinement (test.py:183-189): max distance 2*voxel, Open3D's default criteria
    lr_icp_result *icp_res = reinterpret_cast<lr_icp_result *>(ws->icp_state + 24);
    if (p->icp)
        LR_TRY(lr_icp_run(ws, xyz0, n0, xyz1, n1, T_final, ws->res_tmp, 0.6, 30, 1e-6, 1e-6, ws->T_tmp + 32, icp_res, st));
    if (p->icp) hipLaunchKernelGGL(pair_icp_kernel, dim3(1, 1, ws->zP), dim3(64), 0, st, ws->T_tmp + 32, icp_res, out, 1, ws->z);
    if (timed && ws->ev_pending == 2) { LR_HIP(hipEventRecord(ws->ev[8], st)); ws->ev_pending = 3; }
    LR_LAUNCH_CHECK();
    return LR_OK;
}

extern "C" int lr_register_pair(lr_workspace *ws, const float *xyz0, const float *xyz1, const float *F0, const float *F1,
                                int n0, int n1, int dim, const lr_pair_params *p, lr_pair_result *out, void *stream)
{
    LR_REQUIRE(ws && xyz0 && xyz1 && p && out, LR_EINVAL, "lr_register_pair: null pointer");
    LR_TRY(check_pair_params(p, "lr_register_pair"));
    LR_TRY(check_nn_args(ws, F0, n0, F1, n1, dim, "lr_register_pair"));
    LR_CHECK_DEVICE(ws, stream, "lr_register_pair");
    LR_REQUIRE(p->mode == LR_MODE_NO_FILTER || p->mode == LR_MODE_MNN || p->mode == LR_MODE_GPF, LR_EINVAL,
               "lr_register_pair: unknown mode");
    ws->zP = 1; ws->z = lr_zargs{ 0, nullptr };
    ws->last_npairs = 1; ws->last_batch = 0;
    return register_stages(ws, xyz0, xyz1, F0, F1, n0, n1, dim, p, out, (hipStream_t)stream);
}

// descriptor table -> device memory (it travels as a kernel argument: no host staging buffer, no copy engine)
__global__ void batch_setup_kernel(lr_desc_table t, lr_pair_desc *__restrict__ descs, int npairs)
{
    const int k = threadIdx.x;
    if (k < npairs) descs[k] = t.d[k];
}

extern "C" int lr_register_batch(lr_workspace *ws, int npairs, const float *const *xyz0, const float *const *xyz1,
                                 const float *const *F0, const float *const *F1, const int32_t *n0, const int32_t *n1, int dim,
                                 const lr_pair_params *p, lr_pair_result *out, void *stream)
{
    LR_REQUIRE(ws && xyz0 && xyz1 && F0 && F1 && n0 && n1 && p && out, LR_EINVAL, "lr_register_batch: null pointer");
    LR_TRY(check_pair_params(p, "lr_register_batch"));
    LR_REQUIRE(npairs >= 1 && npairs <= ws->max_pairs, LR_ESIZE, "lr_register_batch: npairs exceeds the workspace (lr_workspace_create_batch)");
    LR_CHECK_DEVICE(ws, stream, "lr_register_batch");
    LR_REQUIRE(p->mode == LR_MODE_NO_FILTER || p->mode == LR_MODE_MNN || p->mode == LR_MODE_GPF, LR_EINVAL,
               "lr_register_batch: unknown mode");
    lr_desc_table t;
    int mx0 = 0, mx1 = 0;
    for (int k = 0; k < npairs; ++k) {
        LR_REQUIRE(xyz0[k] && xyz1[k], LR_EINVAL, "lr_register_batch: null pointer");
        LR_TRY(check_nn_args(ws, F0[k], n0[k], F1[k], n1[k], dim, "lr_register_batch"));
        t.d[k] = lr_pair_desc{ xyz0[k], xyz1[k], F0[k], F1[k], n0[k], n1[k] };
        mx0 = n0[k] > mx0 ? n0[k] : mx0; mx1 = n1[k] > mx1 ? n1[k] : mx1;
    }
    for (int k = npairs; k < LR_MAX_BATCH; ++k) t.d[k] = lr_pair_desc{ nullptr, nullptr, nullptr, nullptr, 0, 0 };
    hipStream_t st = (hipStream_t)stream;
    hipLaunchKernelGGL(batch_setup_kernel, dim3(1), dim3(64), 0, st, t, ws->descs, npairs);
    ws->zP = npairs; ws->z = lr_zargs{ ws->stride, ws->descs };
    ws->last_npairs = npairs; forget_last_batch(ws);
    const int rc = register_stages(ws, xyz0[0], xyz1[0], F0[0], F1[0], mx0, mx1, dim, p, out, st);
    ws->zP = 1; ws->z = lr_zargs{ 0, nullptr };
    if (rc == LR_OK) { ws->last_batch = 1; ws->last_stream = st; } else ws->last_T_final = nullptr;      // (lr_icp_batch: only after a call that went through)
    return rc;
}
