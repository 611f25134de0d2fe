// extern "C" surface of libicnv_hip.so (see include/icnv.h) + host-side
// orchestration: device workspace pool, descriptor uploads, reference rounds.
#include <algorithm>
#include <atomic>
#include <chrono>
#include <cmath>
#include <cstdlib>
#include <cstring>
#include <map>
#include <memory>
#include <mutex>
#include <numeric>
#include <string>
#include <vector>

#include "icnv_internal.h"
#include "emission_table.h"

namespace icnv {

// ------------------------------------------------------------------ errors
static thread_local std::string g_err;
void set_error(const std::string &msg) { g_err = msg; }
int hip_fail(hipError_t e, const char *what, const char *file, int line) {
    g_err = std::string("HIP error: ") + hipGetErrorString(e) + " in " + what + " (" + file + ":" +
            std::to_string(line) + ")";
    (void)hipGetLastError();
    return ICNV_ERR_HIP;
}

int num_cus() {
    static int cus = 0;
    if (cus == 0) {
        int dev = 0;
        hipDeviceProp_t prop;
        if (hipGetDevice(&dev) == hipSuccess && hipGetDeviceProperties(&prop, dev) == hipSuccess)
            cus = prop.multiProcessorCount;
        if (cus <= 0) cus = 256;
    }
    return cus;
}

int ensure_dynamic_lds(const void *kernel, int bytes, DeviceOnce &once) {
    static std::mutex mu;
    int dev = 0;
    ICNV_HIP(hipGetDevice(&dev));
    const unsigned long long bit = 1ull << (dev & 63);
    std::lock_guard<std::mutex> lk(mu);
    if (once.done & bit) return ICNV_OK;
    ICNV_HIP(hipFuncSetAttribute(kernel, hipFuncAttributeMaxDynamicSharedMemorySize, bytes));
    once.done |= bit;
    return ICNV_OK;
}

// ------------------------------------------------------------------ device memory pool
// Grow-only caching allocator so that steady-state calls never hipMalloc/hipFree.
namespace {
std::mutex g_pool_mu;
// Blocks are cached per DEVICE (a block is only ever handed back to the device it was allocated on).  Within a device
// the pool relies on the single-stream contract of include/icnv.h: a block released by one *_dev call may still be
// in use by work enqueued on that call's stream, and the next user enqueues on the same stream, i.e. behind it.
typedef std::pair<int, size_t> PoolKey;  // (device, size)
std::multimap<PoolKey, void *> g_free;   // (device, size) -> block
std::map<void *, PoolKey> g_live;        // block -> (device, size)

int current_device() {
    int dev = 0;
    if (hipGetDevice(&dev) != hipSuccess) { (void)hipGetLastError(); dev = 0; }
    return dev;
}
// Worker threads of the in-library multi-device path that share ONE physical device (ICNV_FAKE_DEVICES, the
// one-GPU test of that path) use one stream each: they get disjoint partitions of the device's pool.
thread_local int g_pool_part = 0;
}  // namespace
int pool_domain() {
    int dev = 0;
    if (hipGetDevice(&dev) != hipSuccess) { (void)hipGetLastError(); dev = 0; }
    return dev * 256 + g_pool_part;
}
namespace {

size_t round_size(size_t n) {
    if (n < 256) n = 256;
    size_t r = 256;
    while (r < n) r <<= 1;
    // above 64 MiB round to a multiple of 64 MiB instead of a power of two
    if (n > (64u << 20)) r = ((n + (64u << 20) - 1) / (64u << 20)) * (64u << 20);
    return r;
}
}  // namespace

// wall time spent in hipMalloc by the pool (a host-buffer call that finds no cached block pays for the allocation: what the
// host path's statistics report as alloc_ms)
static std::atomic<int64_t> g_malloc_us{0};
double pool_malloc_ms(bool reset) {
    const double v = (double)g_malloc_us.load() * 1e-3;
    if (reset) g_malloc_us.store(0);
    return v;
}
namespace {
struct MallocTimer {
    std::chrono::steady_clock::time_point t0 = std::chrono::steady_clock::now();
    ~MallocTimer() { g_malloc_us += std::chrono::duration_cast<std::chrono::microseconds>(std::chrono::steady_clock::now() - t0).count(); }
};
}  // namespace

int pool_alloc(void **p, size_t bytes) {
    const size_t sz = round_size(bytes);
    const int dev = pool_domain();
    {
        std::lock_guard<std::mutex> lk(g_pool_mu);
        auto it = g_free.lower_bound(PoolKey(dev, sz));
        if (it != g_free.end() && it->first.first == dev && it->first.second <= sz * 2) {
            *p = it->second;
            g_live[*p] = it->first;
            g_free.erase(it);
            return ICNV_OK;
        }
    }
    void *q = nullptr;
    MallocTimer malloc_timer;
    hipError_t e = hipMalloc(&q, sz);
    if (e != hipSuccess) {
        // release the idle resident matrices (icnv_residency) and this device's cached blocks, retry once
        (void)hipGetLastError();
        (void)residency_release_idle();
        {
            std::lock_guard<std::mutex> lk(g_pool_mu);
            for (auto it = g_free.begin(); it != g_free.end();) {
                if (it->first.first == dev) { (void)hipFree(it->second); it = g_free.erase(it); }
                else ++it;
            }
        }
        e = hipMalloc(&q, sz);
        if (e != hipSuccess) {
            (void)hipGetLastError();
            set_error("out of device memory allocating " + std::to_string(sz) + " bytes");
            return ICNV_ERR_NOMEM;
        }
    }
    std::lock_guard<std::mutex> lk(g_pool_mu);
    g_live[q] = PoolKey(dev, sz);
    *p = q;
    return ICNV_OK;
}

void set_pool_part(int part) { g_pool_part = part; }

void pool_free(void *p) {
    if (!p) return;
    std::lock_guard<std::mutex> lk(g_pool_mu);
    auto it = g_live.find(p);
    if (it == g_live.end()) return;
    g_free.emplace(it->second, p);
    g_live.erase(it);
}


// ------------------------------------------------------------------ kernel timing
namespace {
int g_timing = 0;   // 0 off, 1 every kernel family, 2 only the two hot launches of the bench step (chain_apply, viterbi)
struct TimedRec { std::string name; hipEvent_t e0, e1; int dev; };
std::vector<TimedRec> g_pending;
// recycled events, PER DEVICE (an event belongs to the device it was created on: with one worker thread per GPU -- icnv_set_devices --
// a timer on device A must never pop an event of device B); creating two per launch costs host time inside a timed region
std::map<int, std::vector<hipEvent_t>> g_event_pool;
std::map<std::string, std::pair<double, int64_t>> g_times;
std::mutex g_time_mu;
bool hot_kernel(const char *n) { return std::strcmp(n, "chain_apply") == 0 || std::strcmp(n, "viterbi") == 0; }
bool pooled_event(int dev, hipEvent_t *e) {
    {
        std::lock_guard<std::mutex> lk(g_time_mu);
        auto &pool = g_event_pool[dev];
        if (!pool.empty()) { *e = pool.back(); pool.pop_back(); return true; }
    }
    if (hipEventCreate(e) == hipSuccess) return true;
    (void)hipGetLastError();
    return false;
}
void unpool_events(int dev, hipEvent_t a, hipEvent_t b) {
    std::lock_guard<std::mutex> lk(g_time_mu);
    if (a) g_event_pool[dev].push_back(a);
    if (b) g_event_pool[dev].push_back(b);
}
}  // namespace

KernelTimer::KernelTimer(const char *n, hipStream_t s) : name(n), stream(s) {
    on = g_timing == 1 || (g_timing == 2 && hot_kernel(n));
    if (on) {
        dev = current_device();
        if (!pooled_event(dev, &e0)) { e0 = nullptr; on = false; return; }
        if (!pooled_event(dev, &e1)) { unpool_events(dev, e0, nullptr); e0 = e1 = nullptr; on = false; return; }
        if (hipEventRecord(e0, stream) != hipSuccess) {   // a failed record must not surface as the launch's error
            (void)hipGetLastError();
            unpool_events(dev, e0, e1);
            e0 = e1 = nullptr;
            on = false;
        }
    }
}
KernelTimer::~KernelTimer() {
    if (on) {
        if (hipEventRecord(e1, stream) != hipSuccess) {
            (void)hipGetLastError();
            unpool_events(dev, e0, e1);
            return;
        }
        std::lock_guard<std::mutex> lk(g_time_mu);
        g_pending.push_back({name, e0, e1, dev});
    }
}

static void drain_timers() {
    std::lock_guard<std::mutex> lk(g_time_mu);
    for (auto &r : g_pending) {
        float ms = 0.f;
        (void)hipEventSynchronize(r.e1);
        if (hipEventElapsedTime(&ms, r.e0, r.e1) == hipSuccess) {
            auto &acc = g_times[r.name];
            acc.first += ms;
            acc.second += 1;
        }
        g_event_pool[r.dev].push_back(r.e0);
        g_event_pool[r.dev].push_back(r.e1);
    }
    g_pending.clear();
}
static void destroy_event_pool() {   // icnv_shutdown: hipEventDestroy takes the handle whatever the current device is
    std::lock_guard<std::mutex> lk(g_time_mu);
    for (auto &kv : g_event_pool)
        for (hipEvent_t e : kv.second) (void)hipEventDestroy(e);
    g_event_pool.clear();
}

// ------------------------------------------------------------------ helpers
static int validate_chr(const int32_t *chr_start, int32_t n_chr, int64_t G) {
    if (!chr_start || n_chr < 1) ICNV_FAIL(ICNV_ERR_ARG, "chr_start/n_chr missing");
    if (chr_start[0] != 0 || chr_start[n_chr] != G) ICNV_FAIL(ICNV_ERR_ARG, "chr_start must run from 0 to G");
    for (int k = 0; k < n_chr; ++k)
        if (chr_start[k + 1] < chr_start[k]) ICNV_FAIL(ICNV_ERR_ARG, "chr_start must be non-decreasing");
    return ICNV_OK;
}

static int validate_groups(const int32_t *idx, const int32_t *off, int32_t n_grp, int64_t C, const char *what) {
    if (n_grp < 0 || (n_grp > 0 && (!off || off[0] != 0))) ICNV_FAIL(ICNV_ERR_ARG, std::string(what) + ": bad offsets");
    for (int q = 0; q < n_grp; ++q)
        if (off[q + 1] < off[q]) ICNV_FAIL(ICNV_ERR_ARG, std::string(what) + ": offsets must be non-decreasing");
    const int64_t n = n_grp > 0 ? off[n_grp] : 0;
    if (n > 0 && !idx) ICNV_FAIL(ICNV_ERR_ARG, std::string(what) + ": index vector missing");
    for (int64_t i = 0; i < n; ++i)
        if (idx[i] < 0 || idx[i] >= C) ICNV_FAIL(ICNV_ERR_ARG, std::string(what) + ": cell index out of range");
    return ICNV_OK;
}

template <typename T>
static int upload(DevBuf &b, const T *host, size_t n, hipStream_t s) {
    int rc = b.alloc(std::max<size_t>(n, 1) * sizeof(T));
    if (rc) return rc;
    if (n) ICNV_HIP(hipMemcpyAsync(b.p, host, n * sizeof(T), hipMemcpyHostToDevice, s));
    return ICNV_OK;
}

}  // namespace icnv

using namespace icnv;

// =================================================================== chain object
struct icnv_chain {
    icnv_chain_cfg cfg;
    std::vector<int32_t> chr_start, ref_idx, ref_off;
    std::vector<int> round_stage;  // ICNV_ST_* bit of each reference round
    int T = 0;
    uint32_t mask = 0;
    DevBuf d_chr, d_ref, d_b1, d_b2, d_den, d_partial, d_sums, d_cellstats, d_stats, d_inv, d_inv_codes, d_inv_dict;
    std::vector<double> inv_tab, inv_dict;
    std::vector<uint32_t> inv_codes;
    bool inv_coded = false;   // host copy of the smoothing normalisation table (kept alive for the async upload)
    bool uploaded = false;
    // Gene sets beyond the fused kernel's LDS-resident limit run the three-pass chain (chain_large.hip)
    bool large = false;
    int32_t max_chr_len = 0;
    DevBuf d_ref_off, d_large_tmp;
    // Round 6, pass 1 of the two-pass chain: the chromosomes in groups that fit the 1024 x 11 geometry of the fused kernel, which
    // runs steps 8 / 9 / 10 on each group as a strided view of the matrix (chain_w11s.hip).  Empty: a chromosome does not fit -- the
    // (2T + 1)-tap pass of chain_large.hip serves the whole matrix.
    struct ViewGroup {
        int32_t chr0, n_chr, g0, G;            // chromosomes [chr0, chr0 + n_chr), genes [g0, g0 + G)
        std::vector<int32_t> chr_start;        // relative to g0, n_chr + 1 entries
        std::vector<double> inv_tab, inv_dict;
        std::vector<uint32_t> inv_codes;
        bool inv_coded = false;
        DevBuf d_chr, d_inv, d_inv_codes, d_inv_dict;
    };
    std::vector<std::unique_ptr<ViewGroup>> views;
    // Reference-cell cache: the round that first runs the expensive stages (smoothing, centring) on the reference
    // cells keeps its output (one column per position of ref_idx), so that the later rounds and the apply pass
    // continue from it instead of smoothing the same cells again (three times per chain otherwise).
    bool na_aware = false;            // ICNV_ST_NA_AWARE: cells that hold a NaN are recomputed by chain_na.hip
    DevBuf d_naflags, d_nanbound;     // d_nanbound: [1] set when a no-bounds mean is NaN (every cell then takes the NA pass)
    bool cache_enabled = false;
    DevBuf d_cache, d_nonref, d_iota;   // d_iota: 0 .. n_ref - 1 (the cache holds one column per position of ref_idx)
    std::vector<int32_t> nonref;      // cells that are in no reference group
    const double *cache_in = nullptr; // matrix the cache was computed from (nullptr: invalid)
    uint32_t cache_mask = 0;          // stages already applied to the cached columns
};

static uint32_t stages_before(uint32_t mask, uint32_t stage_bit) { return mask & (stage_bit - 1u); }

extern "C" {

int icnv_version(void) { return 100; }
const char *icnv_last_error(void) { return g_err.c_str(); }

int icnv_init(int device) {
    int n = 0;
    ICNV_HIP(hipGetDeviceCount(&n));
    if (n <= 0) ICNV_FAIL(ICNV_ERR_HIP, "no HIP device visible");
    if (device >= 0) {
        if (device >= n) ICNV_FAIL(ICNV_ERR_ARG, "device ordinal out of range");
        ICNV_HIP(hipSetDevice(device));
    }
    int dev = 0;
    ICNV_HIP(hipGetDevice(&dev));
    hipDeviceProp_t prop;
    ICNV_HIP(hipGetDeviceProperties(&prop, dev));
    if (std::string(prop.gcnArchName).rfind("gfx950", 0) != 0)
        ICNV_FAIL(ICNV_ERR_UNSUPPORTED, std::string("libicnv_hip is built for gfx950 only, found ") + prop.gcnArchName);
    return ICNV_OK;
}

void icnv_shutdown(void) {
    drain_timers();
    destroy_event_pool();
    icnv_residency_drop();
    viterbi_release_contexts();
    std::lock_guard<std::mutex> lk(g_pool_mu);
    for (auto &kv : g_free) (void)hipFree(kv.second);
    g_free.clear();
}

void icnv_timing_enable(int on) { g_timing = on == 2 ? 2 : (on != 0 ? 1 : 0); }
void icnv_timing_reset(void) {
    drain_timers();
    std::lock_guard<std::mutex> lk(g_time_mu);
    g_times.clear();
}
int icnv_timing_get(const char *kernel, double *total_ms, int64_t *launches) {
    drain_timers();
    std::lock_guard<std::mutex> lk(g_time_mu);
    auto it = g_times.find(kernel ? kernel : "");
    if (it == g_times.end()) {
        if (total_ms) *total_ms = 0.0;
        if (launches) *launches = 0;
        return ICNV_OK;
    }
    if (total_ms) *total_ms = it->second.first;
    if (launches) *launches = it->second.second;
    return ICNV_OK;
}

// ------------------------------------------------------------------ chain
int icnv_chain_begin(icnv_chain_t **out, const icnv_chain_cfg *cfg) {
    if (!out || !cfg) ICNV_FAIL(ICNV_ERR_ARG, "null argument");
    *out = nullptr;
    if (cfg->G < 1 || cfg->C < 0 || cfg->G > 0x7fffffff || cfg->C > 0x7fffffff)
        ICNV_FAIL(ICNV_ERR_ARG, "bad matrix dimensions");
    int rc = validate_chr(cfg->chr_start, cfg->n_chr, cfg->G);
    if (rc) return rc;
    uint32_t mask = cfg->stage_mask & (ICNV_ST_ALL | ICNV_ST_CENTER_MEAN);
    if ((mask & ICNV_ST_MAX_THRESH) && std::isnan(cfg->max_thresh)) mask &= ~ICNV_ST_MAX_THRESH;
    if ((mask & ICNV_ST_SMOOTH) && cfg->window_length < 2) mask &= ~ICNV_ST_SMOOTH;  // R/inferCNV_ops.R:2444
    if ((mask & ICNV_ST_SMOOTH) && (cfg->window_length % 2 == 0))
        ICNV_FAIL(ICNV_ERR_ARG, "window_length must be odd (the reference is undefined for even windows)");
    if ((mask & ICNV_ST_DENOISE) && !std::isnan(cfg->noise_filter) && cfg->noise_filter == 0.0)
        mask &= ~ICNV_ST_DENOISE;  // clear_noise(threshold = 0) is a no-op, R/inferCNV_ops.R:2236
    if (cfg->inv_log && mask != ICNV_ST_SUBTRACT_REF_1)
        ICNV_FAIL(ICNV_ERR_UNSUPPORTED, "inv_log is the stand-alone subtract_ref_expr_from_obs(inv_log = TRUE): stage_mask must be "
                                        "ICNV_ST_SUBTRACT_REF_1 alone (run() never sets it, R/inferCNV_ops.R:771,952)");
    const bool needs_ref = mask & (ICNV_ST_SUBTRACT_REF_1 | ICNV_ST_SUBTRACT_REF_2 | ICNV_ST_DENOISE);
    if (needs_ref) {
        if (cfg->n_ref_grp < 1) ICNV_FAIL(ICNV_ERR_ARG, "reference groups required for steps 8/12/22");
        rc = validate_groups(cfg->ref_idx, cfg->ref_off, cfg->n_ref_grp, cfg->C, "ref groups");
        if (rc) return rc;
    }
    icnv_chain *ch = new icnv_chain();
    ch->cfg = *cfg;
    ch->mask = mask;
    ch->na_aware = (cfg->stage_mask & ICNV_ST_NA_AWARE) != 0;
    ch->T = (mask & ICNV_ST_SMOOTH) ? (cfg->window_length - 1) / 2 : 0;
    for (int k = 0; k < cfg->n_chr; ++k) ch->max_chr_len = std::max(ch->max_chr_len, cfg->chr_start[k + 1] - cfg->chr_start[k]);
    {
        const char *force = std::getenv("ICNV_CHAIN_LARGE");   // developer switch: the three-pass chain for any size
        ch->large = (force && force[0] == '1') || cfg->n_chr > 510 || !chain_fused_fits(cfg->G, cfg->n_chr, ch->T);
        if (ch->large && cfg->inv_log) {
            delete ch;
            ICNV_FAIL(ICNV_ERR_UNSUPPORTED, "inv_log is not available for gene sets that need the three-pass chain");
        }
        if (ch->na_aware && (cfg->inv_log || cfg->noise_logistic)) {
            // (explicit, not silent: the stand-alone subtract with inv_log = TRUE sums 2^x - 1 over the reference cells and
            // noise_logistic rewrites every value -- neither has an NA pass here)
            delete ch;
            ICNV_FAIL(ICNV_ERR_UNSUPPORTED, "ICNV_ST_NA_AWARE is not available with inv_log or noise_logistic");
        }
        if (ch->large && chain_large_lds_bytes(ch->max_chr_len, ch->T) > 152 * 1024) {
            delete ch;
            ICNV_FAIL(ICNV_ERR_UNSUPPORTED, "smoothing chain: a single chromosome (plus the window) exceeds the 160 KiB LDS");
        }
    }
    ch->chr_start.assign(cfg->chr_start, cfg->chr_start + cfg->n_chr + 1);
    if (needs_ref) {
        ch->ref_off.assign(cfg->ref_off, cfg->ref_off + cfg->n_ref_grp + 1);
        ch->ref_idx.assign(cfg->ref_idx, cfg->ref_idx + cfg->ref_off[cfg->n_ref_grp]);
    } else {
        ch->cfg.n_ref_grp = 0;
    }
    ch->cfg.chr_start = ch->chr_start.data();
    ch->cfg.ref_idx = ch->ref_idx.data();
    ch->cfg.ref_off = ch->ref_off.data();
    for (uint32_t bit : {ICNV_ST_SUBTRACT_REF_1, ICNV_ST_SUBTRACT_REF_2, ICNV_ST_DENOISE})
        if (mask & bit) ch->round_stage.push_back((int)bit);
    *out = ch;
    return ICNV_OK;
}

int icnv_chain_num_rounds(const icnv_chain_t *ch) { return ch ? (int)ch->round_stage.size() : 0; }

static int chain_upload(icnv_chain *ch, hipStream_t s) {
    if (ch->uploaded) return ICNV_OK;
    const int64_t G = ch->cfg.G;
    const int ng = std::max(ch->cfg.n_ref_grp, 1);
    int rc;
    if ((rc = upload(ch->d_chr, ch->chr_start.data(), ch->chr_start.size(), s))) return rc;
    if ((rc = upload(ch->d_ref, ch->ref_idx.data(), ch->ref_idx.size(), s))) return rc;
    if ((rc = ch->d_b1.alloc(2 * G * sizeof(double)))) return rc;
    if ((rc = ch->d_b2.alloc(2 * G * sizeof(double)))) return rc;
    if ((rc = ch->d_den.alloc(2 * sizeof(double)))) return rc;
    if ((rc = ch->d_partial.alloc((size_t)256 * G * sizeof(double)))) return rc;
    if ((rc = ch->d_sums.alloc(((size_t)G * ng + ng) * sizeof(double)))) return rc;
    if ((rc = ch->d_cellstats.alloc(std::max<size_t>(ch->ref_idx.size(), 1) * 2 * sizeof(double)))) return rc;
    if ((rc = ch->d_stats.alloc(4 * sizeof(double)))) return rc;
    {
        const size_t nref = ch->ref_idx.size();
        const size_t bytes = nref * (size_t)G * sizeof(double);
        const char *off = std::getenv("ICNV_REF_CACHE");   // developer switch: ICNV_REF_CACHE=0 recomputes the reference cells
        if (!ch->large && nref > 0 && bytes <= ((size_t)16 << 30) && !(off && off[0] == '0') &&
            (ch->mask & (ICNV_ST_SMOOTH | ICNV_ST_CENTER))) {
            std::vector<char> is_ref((size_t)ch->cfg.C, 0);
            for (int32_t c : ch->ref_idx) is_ref[c] = 1;
            for (int64_t c = 0; c < ch->cfg.C; ++c)
                if (!is_ref[c]) ch->nonref.push_back((int32_t)c);
            if ((rc = ch->d_cache.alloc(bytes))) return rc;
            if ((rc = upload(ch->d_nonref, ch->nonref.data(), ch->nonref.size(), s))) return rc;
            std::vector<int32_t> iota(nref);
            for (size_t i = 0; i < nref; ++i) iota[i] = (int32_t)i;
            if ((rc = upload(ch->d_iota, iota.data(), iota.size(), s))) return rc;
            ch->cache_enabled = true;
        }
    }
    if ((rc = upload(ch->d_ref_off, ch->ref_off.data(), ch->ref_off.size(), s))) return rc;
    if (ch->na_aware) {
        if ((rc = ch->d_naflags.alloc((size_t)std::max<int64_t>(ch->cfg.C, 1)))) return rc;
        if ((rc = ch->d_nanbound.alloc(sizeof(int32_t)))) return rc;
        ICNV_HIP(hipMemsetAsync(ch->d_nanbound.p, 0, sizeof(int32_t), s));
    }
    if (ch->large) {
        ch->cache_enabled = false;
        if (!ch->ref_idx.empty() && (rc = ch->d_large_tmp.alloc(ch->ref_idx.size() * (size_t)G * sizeof(double)))) return rc;
        // pass 1 as strided views of the fused kernel: whole chromosomes, greedily, while the group fits the 1024 x 11 geometry
        const char *old_s = std::getenv("ICNV_CHAIN_LARGE_TAPS");   // developer / test switch: 1 = the (2T + 1)-tap pass of round 1-5
        if (ch->T >= 1 && !(old_s && old_s[0] == '1')) {
            bool ok = true;
            int k = 0;
            while (k < ch->cfg.n_chr && ok) {
                int k1 = k;
                // (the upper limits only while the group grows; a view also needs four genes -- checked below on the finished groups)
                while (k1 < ch->cfg.n_chr && chain_view_fits(std::max(ch->chr_start[k1 + 1] - ch->chr_start[k], 4), k1 + 1 - k, ch->T)) ++k1;
                if (k1 == k) { ok = false; break; }   // one chromosome alone is too long for the geometry
                auto v = std::make_unique<icnv_chain::ViewGroup>();
                v->chr0 = k; v->n_chr = k1 - k; v->g0 = ch->chr_start[k]; v->G = ch->chr_start[k1] - ch->chr_start[k];
                for (int c = k; c <= k1; ++c) v->chr_start.push_back(ch->chr_start[c] - v->g0);
                ch->views.push_back(std::move(v));
                k = k1;
            }
            for (auto &v : ch->views) ok = ok && chain_view_fits(v->G, v->n_chr, ch->T);
            if (!ok) ch->views.clear();
            for (auto &v : ch->views) {
                if ((rc = chain_build_inv_table(v->chr_start.data(), v->n_chr, v->G, ch->T, v->inv_tab, v->inv_codes, v->inv_dict, v->inv_coded, true))) return rc;
                if ((rc = upload(v->d_chr, v->chr_start.data(), v->chr_start.size(), s))) return rc;
                if ((rc = upload(v->d_inv, v->inv_tab.data(), v->inv_tab.size(), s))) return rc;
                if ((rc = upload(v->d_inv_codes, v->inv_codes.data(), v->inv_codes.size(), s))) return rc;
                if ((rc = upload(v->d_inv_dict, v->inv_dict.data(), v->inv_dict.size(), s))) return rc;
            }
        }
        ch->uploaded = true;
        return ICNV_OK;
    }
    if (ch->T >= 1) {
        if ((rc = chain_build_inv_table(ch->chr_start.data(), ch->cfg.n_chr, (int32_t)G, ch->T, ch->inv_tab, ch->inv_codes,
                                        ch->inv_dict, ch->inv_coded)))
            return rc;
        if ((rc = upload(ch->d_inv, ch->inv_tab.data(), ch->inv_tab.size(), s))) return rc;
        if ((rc = upload(ch->d_inv_codes, ch->inv_codes.data(), ch->inv_codes.size(), s))) return rc;
        if ((rc = upload(ch->d_inv_dict, ch->inv_dict.data(), ch->inv_dict.size(), s))) return rc;
    }
    ch->uploaded = true;
    return ICNV_OK;
}

static ChainArgs chain_args(icnv_chain *ch, const double *in) {
    ChainArgs a;
    std::memset(&a, 0, sizeof(a));
    a.in = in;
    a.G = (int32_t)ch->cfg.G;
    a.chr_start = ch->d_chr.as<int32_t>();
    a.n_chr = ch->cfg.n_chr;
    a.T = ch->T;
    a.use_bounds = ch->cfg.use_bounds;
    a.inv_log = ch->cfg.inv_log;
    a.max_thresh = ch->cfg.max_thresh;
    a.b1 = ch->d_b1.as<double>();
    a.b2 = ch->d_b2.as<double>();
    a.denoise = ch->d_den.as<double>();
    a.inv_pos = ch->T >= 1 ? ch->d_inv.as<double>() : nullptr;
    a.inv_codes = ch->T >= 1 ? ch->d_inv_codes.as<uint32_t>() : nullptr;
    a.inv_dict = ch->T >= 1 ? ch->d_inv_dict.as<double>() : nullptr;
    a.inv_coded = ch->inv_coded ? 1 : 0;
    a.partial = ch->d_partial.as<double>();
    a.cell_stats = ch->d_cellstats.as<double>();
    a.pre_ld = (int32_t)ch->cfg.G;
    return a;
}

// ---- three-pass chain (chain_large.hip): the stages `m` of the chain on `n_rows` rows ----
static LargeChainArgs large_args(icnv_chain *ch) {
    LargeChainArgs a;
    std::memset(&a, 0, sizeof(a));
    a.G = (int32_t)ch->cfg.G;
    a.chr_start = ch->d_chr.as<int32_t>();
    a.n_chr = ch->cfg.n_chr;
    a.T = ch->T;
    a.max_thresh = ch->cfg.max_thresh;
    a.b1 = ch->d_b1.as<double>();
    a.b2 = ch->d_b2.as<double>();
    a.denoise = ch->d_den.as<double>();
    return a;
}
// stages S and M of mask m: in (rows in_rows) -> out (rows out_rows); returns the bits of m still to do (E)
static int large_smooth_center(icnv_chain *ch, uint32_t m, const double *in, const int32_t *in_rows, double *out,
                               const int32_t *out_rows, int32_t n_rows, hipStream_t s) {
    LargeChainArgs a = large_args(ch);
    a.in = in; a.in_rows = in_rows; a.out = out; a.out_rows = out_rows; a.n_rows = n_rows;
    a.mask = m & (ICNV_ST_SUBTRACT_REF_1 | ICNV_ST_MAX_THRESH | ICNV_ST_SMOOTH);
    int rc;
    if (!ch->views.empty() && (a.mask & ICNV_ST_SMOOTH) && !(in_rows && out_rows)) {
        // Round 6: steps 8 / 9 / 10 by the fused kernel, one launch per group of chromosomes (a strided view of the matrix): the
        // O(1) sliding pyramid of chain_kernel.inc instead of 2T + 1 taps per gene, sixteen wavefronts per CU instead of four
        for (auto &v : ch->views) {
            ChainArgs c;
            std::memset(&c, 0, sizeof(c));
            c.in = in + v->g0;
            c.out = out + v->g0;
            c.G = v->G;
            c.ld = (int32_t)ch->cfg.G;
            c.pre_ld = (int32_t)ch->cfg.G;
            c.cells = in_rows ? in_rows : out_rows;       // (rows read by list entry and written by position, or the other way round)
            c.in_by_pos = in_rows ? 0 : 1;
            c.out_by_pos = out_rows ? 0 : 1;
            c.n_cells = n_rows;
            c.chr_start = v->d_chr.as<int32_t>();
            c.n_chr = v->n_chr;
            c.T = ch->T;
            c.mask = a.mask;
            c.use_bounds = ch->cfg.use_bounds;
            c.max_thresh = ch->cfg.max_thresh;
            c.b1 = ch->d_b1.as<double>() + v->g0;
            c.inv_pos = v->d_inv.as<double>();
            c.inv_codes = v->d_inv_codes.as<uint32_t>();
            c.inv_dict = v->d_inv_dict.as<double>();
            c.inv_coded = v->inv_coded ? 1 : 0;
            if ((rc = launch_chain_strided(c, s))) return rc;
        }
    } else if ((rc = launch_chain_large_smooth(a, ch->max_chr_len, s))) {
        return rc;
    }
    if (m & ICNV_ST_CENTER) {
        a.mask = m & (ICNV_ST_CENTER | ICNV_ST_CENTER_MEAN);
        if ((rc = launch_chain_large_center(a, s))) return rc;
    }
    return ICNV_OK;
}

static int large_round_partial(icnv_chain *ch, uint32_t bit, uint32_t m, const double *expr_in, hipStream_t s) {
    const int64_t G = ch->cfg.G;
    const int ng = ch->cfg.n_ref_grp;
    const int32_t nref = (int32_t)ch->ref_idx.size();
    int rc;
    double *tmp = ch->d_large_tmp.as<double>();   // one row per position of the reference list
    const uint32_t sm_bits = ICNV_ST_SUBTRACT_REF_1 | ICNV_ST_MAX_THRESH | ICNV_ST_SMOOTH | ICNV_ST_CENTER | ICNV_ST_CENTER_MEAN;
    const bool any_sm = m & (sm_bits & ~(uint32_t)ICNV_ST_CENTER_MEAN);
    // (round 6) the staged rows of the round in front serve this round when they hold the same stages of the same matrix: the
    // denoise round continues from the step-12 round's rows instead of smoothing and centring the reference cells again
    const bool staged = any_sm && ch->cache_in == expr_in && ch->cache_mask == (m & sm_bits);
    if (any_sm && nref > 0 && !staged) {
        ch->cache_in = nullptr;
        if ((rc = large_smooth_center(ch, m, expr_in, ch->d_ref.as<int32_t>(), tmp, nullptr, nref, s))) return rc;
        if (ch->na_aware) {
            // (round 6) reference cells that hold a NaN: their staged rows redone with the reference's NA semantics (chain_na.hip works
            // on any gene count: one chromosome at a time through the LDS)
            ChainArgs na = chain_args(ch, expr_in);
            na.mask = m & sm_bits;
            na.cells = ch->d_ref.as<int32_t>();
            na.n_cells = nref;
            na.out = tmp;
            na.out_by_pos = 1;
            if ((rc = launch_chain_na_fixup(na, ch->max_chr_len, ch->d_naflags.as<uint8_t>(), ch->d_nanbound.as<int32_t>(), s))) return rc;
        }
        ch->cache_in = expr_in;
        ch->cache_mask = m & sm_bits;
    }
    if (bit == ICNV_ST_DENOISE) {
        if (nref > 0) {
            LargeChainArgs a = large_args(ch);
            a.in = any_sm ? tmp : expr_in;
            a.in_rows = any_sm ? nullptr : ch->d_ref.as<int32_t>();
            a.n_rows = nref;
            a.mask = m & (ICNV_ST_SUBTRACT_REF_2 | ICNV_ST_INVERT_LOG2);
            a.cell_stats = ch->d_cellstats.as<double>();
            if ((rc = launch_chain_large_finish(a, s))) return rc;
        }
        return launch_reduce_cell_stats(ch->d_cellstats.as<double>(), nref, (int32_t)G, ch->d_stats.as<double>(), s);
    }
    // per-gene sums of the reference groups: over the staged rows (positions) or straight over the input's cells
    return launch_chain_large_group_sums(any_sm ? tmp : expr_in, (int32_t)G, any_sm ? nullptr : ch->d_ref.as<int32_t>(),
                                         ch->d_ref_off.as<int32_t>(), ng, ch->d_sums.as<double>(), s);
}

static int large_apply(icnv_chain *ch, const double *expr_in, double *expr_out, double *pre_denoise, hipStream_t s) {
    const int32_t C = (int32_t)ch->cfg.C;
    int rc;
    if (!ch->views.empty() && (ch->mask & ICNV_ST_SMOOTH) && chain_large_center_finish_covers((int32_t)ch->cfg.G)) {
        // the two-pass form (round 6): pass 1 = steps 8 - 10 into expr_out, pass 2 = the centre and steps 12 - 22 from registers
        if ((rc = large_smooth_center(ch, ch->mask & ~(uint32_t)(ICNV_ST_CENTER | ICNV_ST_CENTER_MEAN), expr_in, nullptr, expr_out, nullptr, C, s))) return rc;
        LargeChainArgs a = large_args(ch);
        a.in = expr_out; a.out = expr_out; a.pre = pre_denoise; a.n_rows = C;
        a.mask = ch->mask & (ICNV_ST_CENTER | ICNV_ST_CENTER_MEAN | ICNV_ST_SUBTRACT_REF_2 | ICNV_ST_INVERT_LOG2 | ICNV_ST_DENOISE);
        return launch_chain_large_center_finish(a, s);
    }
    if ((rc = large_smooth_center(ch, ch->mask, expr_in, nullptr, expr_out, nullptr, C, s))) return rc;
    LargeChainArgs a = large_args(ch);
    a.in = expr_out; a.out = expr_out; a.pre = pre_denoise; a.n_rows = C;
    a.mask = ch->mask & (ICNV_ST_SUBTRACT_REF_2 | ICNV_ST_INVERT_LOG2 | ICNV_ST_DENOISE);
    return launch_chain_large_finish(a, s);
}

int icnv_chain_round_partial_dev(icnv_chain_t *ch, int round, const double *expr_in, double **partial_dev,
                                 int64_t *n, void *stream) {
    if (!ch || !expr_in || round < 0 || round >= (int)ch->round_stage.size()) ICNV_FAIL(ICNV_ERR_ARG, "bad round");
    hipStream_t s = (hipStream_t)stream;
    int rc = chain_upload(ch, s);
    if (rc) return rc;
    const uint32_t bit = (uint32_t)ch->round_stage[round];
    const int64_t G = ch->cfg.G;
    const int ng = ch->cfg.n_ref_grp;
    if (round == 0) ch->cache_in = nullptr;   // a new run: whatever the rounds of an earlier run staged is not this matrix's (same address or not)
    ChainArgs a = chain_args(ch, expr_in);
    a.mask = stages_before(ch->mask, bit) | (ch->mask & ICNV_ST_CENTER_MEAN);
    if (ch->large) {
        if ((rc = large_round_partial(ch, bit, a.mask, expr_in, s))) return rc;
        if (bit == ICNV_ST_DENOISE) {
            if (partial_dev) *partial_dev = ch->d_stats.as<double>();
            if (n) *n = 4;
        } else {
            if (partial_dev) *partial_dev = ch->d_sums.as<double>();
            if (n) *n = G * ng + ng;
        }
        return ICNV_OK;
    }
    if (bit == ICNV_ST_DENOISE) {
        const int nref = (int)ch->ref_idx.size();
        a.cells = ch->d_ref.as<int32_t>();
        a.n_cells = nref;
        if (ch->cache_in == expr_in && (ch->cache_mask & ~a.mask) == 0) {   // continue from the cached columns
            a.in = ch->d_cache.as<double>();
            a.in_by_pos = 1;
            a.mask &= ~ch->cache_mask;
        }
        if (nref > 0) {
            // from the cache only the elementwise steps 12 / 14 are left: one streaming pass instead of the chain geometry
            const char *force_chain = std::getenv("ICNV_CELL_STATS_CHAIN");   // developer / test switch: 1 = the chain geometry for this round too
            const bool elementwise = a.in_by_pos && cache_cell_stats_covers((int32_t)G) && !(force_chain && force_chain[0] == '1') &&
                                     (a.mask & ~(uint32_t)(ICNV_ST_SUBTRACT_REF_2 | ICNV_ST_INVERT_LOG2 | ICNV_ST_CENTER_MEAN)) == 0;
            if (elementwise) rc = launch_cache_cell_stats(a.in, (int32_t)G, nref, a.mask, a.b2, a.cell_stats, s);
            else rc = launch_chain(a, MODE_CELL_STATS, s);
            if (rc) return rc;
        }
        if ((rc = launch_reduce_cell_stats(ch->d_cellstats.as<double>(), nref, (int32_t)G, ch->d_stats.as<double>(), s)))
            return rc;
        if (partial_dev) *partial_dev = ch->d_stats.as<double>();
        if (n) *n = 4;
        return ICNV_OK;
    }
    double *sums = ch->d_sums.as<double>();
    if (a.mask == 0 && !ch->cfg.inv_log && ng <= 256) {
        // no stage in front of this round (run()'s first one): the raw sums of every group in one streaming launch
        if ((rc = launch_group_gene_sums(expr_in, (int32_t)G, ch->d_ref.as<int32_t>(), ch->d_ref_off.as<int32_t>(), ng,
                                         ch->d_partial.as<double>(), 256, sums, s)))
            return rc;
        if (partial_dev) *partial_dev = sums;
        if (n) *n = G * ng + ng;
        return ICNV_OK;
    }
    const bool fill_cache = ch->cache_enabled && (a.mask & (ICNV_ST_SMOOTH | ICNV_ST_CENTER));
    if (fill_cache) ch->cache_in = nullptr;
    if (fill_cache && !ch->cfg.inv_log && ng <= 256) {
        // One launch runs the stages in front of this round on the reference cells of EVERY group into their cache, one streaming
        // launch sums the cached columns per gene and group: two launches whatever the number of groups, where a
        // statistics launch + a reduction per group made this round the most expensive of the three.
        const int nref = (int)ch->ref_idx.size();
        a.cells = ch->d_ref.as<int32_t>();
        a.n_cells = nref;
        a.out = ch->d_cache.as<double>();
        a.out_by_pos = 1;
        const int rc2 = launch_chain(a, MODE_APPLY, s);
        if (rc2) return rc2;
        // (reference cells that hold a NaN: their cached columns redone with the reference's NA semantics before they are summed)
        if (ch->na_aware && (rc = launch_chain_na_fixup(a, ch->max_chr_len, ch->d_naflags.as<uint8_t>(), ch->d_nanbound.as<int32_t>(), s))) return rc;
        if ((rc = launch_group_gene_sums(ch->d_cache.as<double>(), (int32_t)G, ch->d_iota.as<int32_t>(), ch->d_ref_off.as<int32_t>(), ng,
                                         ch->d_partial.as<double>(), 256, sums, s)))
            return rc;
        ch->cache_in = expr_in;
        ch->cache_mask = a.mask;
        if (partial_dev) *partial_dev = sums;
        if (n) *n = G * ng + ng;
        return ICNV_OK;
    }
    for (int q = 0; q < ng; ++q) {
        const int cnt = ch->ref_off[q + 1] - ch->ref_off[q];
        double *dst = sums + (size_t)q * G;
        double *cdst = sums + (size_t)ng * G + q;
        if (cnt == 0) {
            ICNV_HIP(hipMemsetAsync(dst, 0, G * sizeof(double), s));
            ICNV_HIP(hipMemsetAsync(cdst, 0, sizeof(double), s));
            continue;
        }
        a.cells = ch->d_ref.as<int32_t>() + ch->ref_off[q];
        a.n_cells = cnt;
        a.cache_out = fill_cache ? ch->d_cache.as<double>() + (size_t)ch->ref_off[q] * G : nullptr;
        if ((rc = launch_chain(a, MODE_GENE_SUMS, s))) return rc;
        const int nblk = std::min(cnt, 256);
        if ((rc = launch_reduce_partials(a.partial, nblk, (int32_t)G, dst, (double)cnt, cdst, s))) return rc;
    }
    if (fill_cache) {
        ch->cache_in = expr_in;
        ch->cache_mask = a.mask;
    }
    if (partial_dev) *partial_dev = sums;
    if (n) *n = G * ng + ng;
    return ICNV_OK;
}

int icnv_chain_round_finish_dev(icnv_chain_t *ch, int round, void *stream) {
    if (!ch || round < 0 || round >= (int)ch->round_stage.size()) ICNV_FAIL(ICNV_ERR_ARG, "bad round");
    hipStream_t s = (hipStream_t)stream;
    const uint32_t bit = (uint32_t)ch->round_stage[round];
    if (bit == ICNV_ST_DENOISE)
        return launch_denoise_from_stats(ch->d_stats.as<double>(), ch->cfg.sd_amplifier, ch->cfg.noise_filter,
                                         ch->d_den.as<double>(), s);
    double *bounds = (bit == ICNV_ST_SUBTRACT_REF_1) ? ch->d_b1.as<double>() : ch->d_b2.as<double>();
    int32_t *nan_flag = (ch->na_aware && !ch->cfg.use_bounds) ? ch->d_nanbound.as<int32_t>() : nullptr;
    if (nan_flag && round == 0) ICNV_HIP(hipMemsetAsync(nan_flag, 0, sizeof(int32_t), s));   // (a chain object may be run again)
    return launch_bounds_from_sums(ch->d_sums.as<double>(), (int32_t)ch->cfg.G, ch->cfg.n_ref_grp, ch->cfg.use_bounds,
                                   ch->cfg.inv_log, bounds, nan_flag, s);
}

// the apply pass with the stage mask `amask` (the chain's own, or -- noise_logistic -- the chain's without step 22)
static int chain_apply_masked(icnv_chain_t *ch, uint32_t amask, const double *expr_in, double *expr_out, double *pre_denoise,
                              hipStream_t s, int64_t ld_pre = 0) {
    int rc;
    if (ld_pre > 0 && ld_pre != ch->cfg.G) {
        // a padded HMM input (icnv_chain_apply_ld_dev): written by the fused pass itself; the side paths that copy or patch whole
        // matrices (three- / two-pass chain, NA pass, a chain without step 22) keep the contiguous layout
        if (ld_pre < ch->cfg.G || ld_pre > 0x7fffffff) ICNV_FAIL(ICNV_ERR_ARG, "ld_pre below the number of genes");
        if (ch->large || ch->na_aware || !pre_denoise || !(amask & ICNV_ST_DENOISE))
            ICNV_FAIL(ICNV_ERR_UNSUPPORTED, "a padded HMM input needs the fused chain with step 22 (no NA pass, no three-pass chain)");
    }
    if (ch->large) {
        ch->cache_in = nullptr;   // (the rounds' staged rows are not used by the apply)
        if (ch->na_aware && (expr_in == expr_out || expr_in == pre_denoise))
            ICNV_FAIL(ICNV_ERR_UNSUPPORTED, "ICNV_ST_NA_AWARE with the two- / three-pass chain cannot run in place (the NA pass reads the input again)");
        const uint32_t keep = ch->mask;
        ch->mask = amask;
        rc = large_apply(ch, expr_in, expr_out, (amask & ICNV_ST_DENOISE) ? pre_denoise : nullptr, s);
        ch->mask = keep;
        if (!rc && ch->na_aware) {   // the cells that hold a NaN again, stage by stage, the reference's way
            ChainArgs na = chain_args(ch, expr_in);
            na.mask = amask;
            na.out = expr_out;
            na.pre_out = (amask & ICNV_ST_DENOISE) ? pre_denoise : nullptr;
            na.cells = nullptr;
            na.n_cells = (int32_t)ch->cfg.C;
            rc = launch_chain_na_fixup(na, ch->max_chr_len, ch->d_naflags.as<uint8_t>(), ch->d_nanbound.as<int32_t>(), s);
        }
        if (!rc && pre_denoise && !(amask & ICNV_ST_DENOISE))
            ICNV_HIP(hipMemcpyAsync(pre_denoise, expr_out, (size_t)ch->cfg.G * ch->cfg.C * sizeof(double), hipMemcpyDeviceToDevice, s));
        return rc;
    }
    ChainArgs a = chain_args(ch, expr_in);
    a.mask = amask;
    a.out = expr_out;
    a.pre_out = pre_denoise;
    if (ld_pre > 0) a.pre_ld = (int32_t)ld_pre;
    a.cells = nullptr;
    a.n_cells = (int32_t)ch->cfg.C;
    // The reference cells continue from the cache the rounds left (same matrix, stages a prefix of this chain's);
    // all other cells run the whole chain.  The cache is consumed: an apply without fresh rounds recomputes.
    const bool from_cache = ch->cache_in == expr_in && ch->cache_mask != 0 && (ch->cache_mask & ~amask) == 0;
    const uint32_t cached = ch->cache_mask;
    ch->cache_in = nullptr;
    // the fused pass, then -- ICNV_ST_NA_AWARE -- the cells that hold a NaN again (`host_cells`: the host copy of args.cells)
    auto launch_apply = [&](const ChainArgs &args, const std::vector<int32_t> *host_cells) -> int {
        if (!ch->na_aware) return launch_chain(args, MODE_APPLY, s);
        uint8_t *flags = ch->d_naflags.as<uint8_t>();
        const int32_t *all_flag = ch->d_nanbound.as<int32_t>();
        const bool in_place = args.in == args.out || (args.pre_out && args.in == args.pre_out);
        if (!in_place) {
            int r = launch_chain(args, MODE_APPLY, s);
            if (!r) r = launch_chain_na_fixup(args, ch->max_chr_len, flags, all_flag, s);
            return r;
        }
        // The chain runs IN PLACE (expr_out aliases expr_in: include/icnv.h allows it): the fused pass overwrites the input the NA
        // pass reads, so the cells that hold a NaN are found FIRST and their input columns kept aside.  The flags cross to the host
        // (this path waits for the stream: a matrix with NAs is not run()'s case), the flagged columns -- a handful -- are gathered
        // into a stash, the fused pass runs, and the NA pass recomputes the flagged cells from the stash.
        int r = launch_nan_flags(args, flags, all_flag, s);
        if (r) return r;
        std::vector<uint8_t> hflags((size_t)args.n_cells);
        ICNV_HIP(hipMemcpyAsync(hflags.data(), flags, hflags.size(), hipMemcpyDeviceToHost, s));
        ICNV_HIP(hipStreamSynchronize(s));
        std::vector<int32_t> ids;
        for (int32_t i = 0; i < args.n_cells; ++i)
            if (hflags[(size_t)i]) ids.push_back(host_cells ? (*host_cells)[(size_t)i] : i);
        if (ids.empty()) return launch_chain(args, MODE_APPLY, s);
        const int32_t nf = (int32_t)ids.size();
        DevBuf d_ids, d_stash, d_ones;
        if ((r = upload(d_ids, ids.data(), ids.size(), s))) return r;
        if ((r = d_stash.alloc((size_t)nf * (size_t)args.G * sizeof(double)))) return r;
        if ((r = d_ones.alloc((size_t)nf))) return r;
        ICNV_HIP(hipMemsetAsync(d_ones.p, 1, (size_t)nf, s));
        if ((r = launch_gather_columns(args.in, args.G, d_ids.as<int32_t>(), nf, d_stash.as<double>(), s))) return r;
        if ((r = launch_chain(args, MODE_APPLY, s))) return r;
        ChainArgs b = args;            // the flagged cells: input = their stashed columns (by position), output = their own columns
        b.in = d_stash.as<double>();
        b.in_by_pos = 1;
        b.out_by_pos = 0;
        b.cells = d_ids.as<int32_t>();
        b.n_cells = nf;
        r = launch_chain_na_cells(b, ch->max_chr_len, d_ones.as<uint8_t>(), s);
        ICNV_HIP(hipStreamSynchronize(s));   // `ids` (pageable, uploaded asynchronously) and the three buffers are done with when this returns
        return r;
    };
    auto run = [&](ChainArgs args) -> int {
        if (!from_cache) return launch_apply(args, nullptr);
        int r = ICNV_OK;
        if (!ch->nonref.empty()) {
            args.cells = ch->d_nonref.as<int32_t>();
            args.n_cells = (int32_t)ch->nonref.size();
            if ((r = launch_apply(args, &ch->nonref))) return r;
        }
        args.in = ch->d_cache.as<double>();
        args.in_by_pos = 1;
        args.cells = ch->d_ref.as<int32_t>();
        args.n_cells = (int32_t)ch->ref_idx.size();
        args.mask = amask & ~cached;
        // (NA-aware: a cached column may hold a NaN the stages in front left in place -- e.g. a chain without step 8 -- and
        // steps 12 / 22 treat it the reference's way only in the NA pass)
        return launch_apply(args, &ch->ref_idx);
    };
    if (pre_denoise && !(amask & ICNV_ST_DENOISE)) {
        // no denoise stage: the "pre-denoise" matrix is the output itself
        a.pre_out = nullptr;
        if ((rc = run(a))) return rc;
        ICNV_HIP(hipMemcpyAsync(pre_denoise, expr_out, (size_t)ch->cfg.G * ch->cfg.C * sizeof(double),
                                hipMemcpyDeviceToDevice, s));
        return ICNV_OK;
    }
    return run(a);
}

int icnv_chain_apply_dev(icnv_chain_t *ch, const double *expr_in, double *expr_out, double *pre_denoise,
                         void *stream) {
    return icnv_chain_apply_ld_dev(ch, expr_in, expr_out, pre_denoise, 0, stream);
}

int icnv_chain_apply_ld_dev(icnv_chain_t *ch, const double *expr_in, double *expr_out, double *pre_denoise, int64_t ld_pre,
                            void *stream) {
    if (!ch || !expr_in || !expr_out) ICNV_FAIL(ICNV_ERR_ARG, "null argument");
    hipStream_t s = (hipStream_t)stream;
    int rc = chain_upload(ch, s);
    if (rc) return rc;
    if (ld_pre > 0 && ld_pre != ch->cfg.G) {
        if ((ch->mask & ICNV_ST_DENOISE) && ch->cfg.noise_logistic) ICNV_FAIL(ICNV_ERR_UNSUPPORTED, "a padded HMM input is not available with noise_logistic");
        return chain_apply_masked(ch, ch->mask, expr_in, expr_out, pre_denoise, s, ld_pre);
    }
    if ((ch->mask & ICNV_ST_DENOISE) && ch->cfg.noise_logistic) {
        // noise_logistic = TRUE (R/inferCNV_ops.R:2249-2252, 2326-2330): the rounds have produced the centre and half width
        // of step 22 as for the select; the matrix before step 22 comes out of the pass, the logistic adjustment
        // (.apply_logistic_val_adj, R/inferCNV_heatmap.R:2791-2810) runs over it in place
        if ((rc = chain_apply_masked(ch, ch->mask & ~(uint32_t)ICNV_ST_DENOISE, expr_in, expr_out, pre_denoise, s))) return rc;
        return launch_logistic_denoise(expr_out, ch->cfg.G * ch->cfg.C, ch->d_den.as<double>(), s);
    }
    return chain_apply_masked(ch, ch->mask, expr_in, expr_out, pre_denoise, s);
}

}  // extern "C"
namespace icnv {
int chain_apply_columns(icnv_chain_t *ch, const double *expr_in, double *expr_out, double *pre_denoise, int64_t c0, int64_t c1,
                        hipStream_t s) {
    if (!ch || !expr_in || !expr_out) ICNV_FAIL(ICNV_ERR_ARG, "null argument");
    if (ch->large || ch->na_aware || ((ch->mask & ICNV_ST_DENOISE) && ch->cfg.noise_logistic) || (pre_denoise && !(ch->mask & ICNV_ST_DENOISE))) return -1000;
    if (c1 <= c0) return ICNV_OK;
    int rc = chain_upload(ch, s);
    if (rc) return rc;
    // every cell of the block runs the whole chain (the reference cells too: the cache the rounds left is not used here --
    // the same arithmetic either way, tests/test_gpu_parity.py compares the two)
    ChainArgs a = chain_args(ch, expr_in + c0 * ch->cfg.G);
    a.mask = ch->mask;
    a.out = expr_out + c0 * ch->cfg.G;
    a.pre_out = pre_denoise ? pre_denoise + c0 * ch->cfg.G : nullptr;
    a.cells = nullptr;
    a.n_cells = (int32_t)(c1 - c0);
    return launch_chain(a, MODE_APPLY, s);
}
}  // namespace icnv
extern "C" {

int icnv_chain_get_denoise(icnv_chain_t *ch, double *mu_s, void *stream) {
    if (!ch || !mu_s) ICNV_FAIL(ICNV_ERR_ARG, "null argument");
    if (!(ch->mask & ICNV_ST_DENOISE) || !ch->uploaded) ICNV_FAIL(ICNV_ERR_ARG, "no denoise stage in this chain");
    hipStream_t s = (hipStream_t)stream;
    ICNV_HIP(hipMemcpyAsync(mu_s, ch->d_den.p, 2 * sizeof(double), hipMemcpyDeviceToHost, s));
    ICNV_HIP(hipStreamSynchronize(s));
    return ICNV_OK;
}

void icnv_chain_end(icnv_chain_t *ch) { delete ch; }

int icnv_smooth_chain_dev(const double *expr_in, double *expr_out, double *pre_denoise, const icnv_chain_cfg *cfg,
                          void *stream) {
    icnv_chain_t *ch = nullptr;
    int rc = icnv_chain_begin(&ch, cfg);
    if (rc) return rc;
    for (int q = 0; q < ch->cfg.n_ref_grp && !ch->round_stage.empty(); ++q)
        if (ch->ref_off[q + 1] == ch->ref_off[q]) {
            icnv_chain_end(ch);
            ICNV_FAIL(ICNV_ERR_ARG, "empty reference group");
        }
    for (int r = 0; r < icnv_chain_num_rounds(ch) && !rc; ++r) {
        rc = icnv_chain_round_partial_dev(ch, r, expr_in, nullptr, nullptr, stream);
        if (!rc) rc = icnv_chain_round_finish_dev(ch, r, stream);
    }
    if (!rc) rc = icnv_chain_apply_dev(ch, expr_in, expr_out, pre_denoise, stream);
    // device buffers of the chain go back to the pool; work already enqueued on
    // `stream` keeps using them, and later pool users enqueue on streams that the
    // caller orders after it (single-stream usage) -- see DESIGN.md "workspace".
    icnv_chain_end(ch);
    return rc;
}

int icnv_average_bounds_dev(const double *expr, int64_t G, int64_t C, double *out2_host, void *stream) {
    if (!expr || !out2_host || G < 1 || C < 1) ICNV_FAIL(ICNV_ERR_ARG, "bad argument");
    hipStream_t s = (hipStream_t)stream;
    DevBuf d;
    int rc = d.alloc((size_t)C * 2 * sizeof(double));
    if (rc) return rc;
    if ((rc = launch_minmax_cells(expr, (int32_t)G, C, d.as<double>(), s))) return rc;
    std::vector<double> h((size_t)C * 2);
    ICNV_HIP(hipMemcpyAsync(h.data(), d.p, h.size() * sizeof(double), hipMemcpyDeviceToHost, s));
    ICNV_HIP(hipStreamSynchronize(s));
    // mean over cells, like R's mean(): long double accumulation + refinement
    for (int k = 0; k < 2; ++k) {
        long double acc = 0;
        for (int64_t c = 0; c < C; ++c) acc += h[2 * c + k];
        acc /= (long double)C;
        long double t = 0;
        for (int64_t c = 0; c < C; ++c) t += (h[2 * c + k] - acc);
        acc += t / (long double)C;
        out2_host[k] = (double)acc;
    }
    return ICNV_OK;
}

int icnv_average_bounds(const double *expr, int64_t G, int64_t C, double *out2) {
    if (!expr || !out2) ICNV_FAIL(ICNV_ERR_ARG, "null argument");
    MatrixLease in;
    int rc = acquire_input(expr, G * C, nullptr, in);
    if (rc) return rc;
    return icnv_average_bounds_dev(in.dev, G, C, out2, nullptr);
}

// ------------------------------------------------------------------ step 5: scale_infercnv_expr (R/inferCNV_ops.R:3174-3185)
int icnv_scale_genes_dev(const double *expr_in, double *expr_out, int64_t G, int64_t C, void *stream) {
    if (!expr_in || !expr_out || G < 1 || C < 1 || G > 0x7fffffff) ICNV_FAIL(ICNV_ERR_ARG, "bad argument");
    hipStream_t s = (hipStream_t)stream;
    const int tiles = (int)((G + 255) / 256);
    int64_t ns = (4096 + tiles - 1) / tiles;
    ns = std::max<int64_t>(1, std::min<int64_t>(std::min<int64_t>(ns, C), 1024));
    DevBuf part, ms;
    int rc;
    if ((rc = part.alloc((size_t)ns * G * sizeof(double))) || (rc = ms.alloc((size_t)2 * G * sizeof(double)))) return rc;
    if ((rc = launch_scale_genes(expr_in, expr_out, (int32_t)G, C, (int)ns, part.as<double>(), ms.as<double>(), s))) return rc;
    ICNV_HIP(hipStreamSynchronize(s));   // the workspace goes back to the pool
    return ICNV_OK;
}

int icnv_scale_genes(const double *expr_in, double *expr_out, int64_t G, int64_t C) {
    if (!expr_in || !expr_out || G < 1 || C < 1) ICNV_FAIL(ICNV_ERR_ARG, "bad argument");
    MatrixLease in;
    DevBuf dout;
    int rc;
    const size_t bytes = (size_t)G * (size_t)C * sizeof(double);
    if ((rc = acquire_input(expr_in, G * C, nullptr, in)) || (rc = dout.alloc(bytes))) return rc;
    if ((rc = icnv_scale_genes_dev(in.dev, dout.as<double>(), G, C, nullptr))) return rc;
    ICNV_HIP(hipMemcpy(expr_out, dout.p, bytes, hipMemcpyDeviceToHost));
    publish_output(expr_out, G * C, std::move(dout));
    return ICNV_OK;
}

// ------------------------------------------------------------------ step 16: remove_outliers_norm (R/inferCNV_ops.R:1969-2054)
// Hard thresholds when both bounds are given (:2017-2022), else out_method = "average_bound": .get_average_bounds of the
// input (:2029-2033).  bounds_used2 (nullable, host) receives {lower, upper}.
int icnv_remove_outliers_dev(const double *expr_in, double *expr_out, int64_t G, int64_t C, double lower_bound, double upper_bound,
                             double *bounds_used2, void *stream) {
    if (!expr_in || !expr_out || G < 1 || C < 1) ICNV_FAIL(ICNV_ERR_ARG, "Error, something is wrong with the data, either null or no rows or columns");
    double b[2] = {lower_bound, upper_bound};
    if (std::isnan(lower_bound) || std::isnan(upper_bound)) {
        int rc = icnv_average_bounds_dev(expr_in, G, C, b, stream);
        if (rc) return rc;
    }
    if (bounds_used2) { bounds_used2[0] = b[0]; bounds_used2[1] = b[1]; }
    return launch_clamp_bounds(expr_in, expr_out, G * C, b[0], b[1], (hipStream_t)stream);
}

int icnv_remove_outliers(const double *expr_in, double *expr_out, int64_t G, int64_t C, double lower_bound, double upper_bound,
                         double *bounds_used2) {
    if (!expr_in || !expr_out || G < 1 || C < 1) ICNV_FAIL(ICNV_ERR_ARG, "Error, something is wrong with the data, either null or no rows or columns");
    MatrixLease in;
    DevBuf dout;
    int rc;
    const size_t bytes = (size_t)G * (size_t)C * sizeof(double);
    if ((rc = acquire_input(expr_in, G * C, nullptr, in)) || (rc = dout.alloc(bytes))) return rc;
    if ((rc = icnv_remove_outliers_dev(in.dev, dout.as<double>(), G, C, lower_bound, upper_bound, bounds_used2, nullptr))) return rc;
    ICNV_HIP(hipMemcpy(expr_out, dout.p, bytes, hipMemcpyDeviceToHost));
    publish_output(expr_out, G * C, std::move(dout));
    return ICNV_OK;
}

// ------------------------------------------------------------------ ingest (steps 3-4)
int icnv_col_sums_dev(const double *expr, int64_t G, int64_t C, double *sums_dev, void *stream) {
    if (!expr || !sums_dev || G < 1 || C < 0 || G > 0x7fffffff) ICNV_FAIL(ICNV_ERR_ARG, "bad argument");
    return launch_col_sums(expr, (int32_t)G, C, sums_dev, (hipStream_t)stream);
}

int icnv_normalize_log2_dev(const double *expr_in, double *expr_out, int64_t G, int64_t C, const double *col_sums_dev,
                            double normalize_factor, int32_t do_normalize, int32_t do_log2, void *stream) {
    if (!expr_in || !expr_out || G < 1 || C < 0 || G > 0x7fffffff) ICNV_FAIL(ICNV_ERR_ARG, "bad argument");
    if (do_normalize && (!col_sums_dev || std::isnan(normalize_factor)))
        ICNV_FAIL(ICNV_ERR_ARG, "normalisation needs the column sums and a factor");
    return launch_normalize_log2(expr_in, expr_out, (int32_t)G, C, col_sums_dev, normalize_factor, do_normalize, do_log2,
                                 (hipStream_t)stream);
}

static double host_median(std::vector<double> v) {
    const size_t n = v.size(), h = n / 2;
    std::nth_element(v.begin(), v.begin() + h, v.end());
    const double hi = v[h];
    if (n & 1) return hi;
    const double lo = *std::max_element(v.begin(), v.begin() + h);
    return (lo + hi) * 0.5;
}

int icnv_normalize_log2(const double *expr_in, double *expr_out, int64_t G, int64_t C, double normalize_factor,
                        int32_t do_normalize, int32_t do_log2, double *factor_used) {
    if (!expr_in || !expr_out || G < 1 || C < 1) ICNV_FAIL(ICNV_ERR_ARG, "bad argument");
    MatrixLease in;
    DevBuf dout, dsum;
    int rc;
    const size_t bytes = (size_t)G * (size_t)C * sizeof(double);
    if ((rc = acquire_input(expr_in, G * C, nullptr, in)) || (rc = dout.alloc(bytes)) || (rc = dsum.alloc((size_t)C * sizeof(double)))) return rc;
    double factor = normalize_factor;
    if (do_normalize) {
        if ((rc = icnv_col_sums_dev(in.dev, G, C, dsum.as<double>(), nullptr))) return rc;
        if (std::isnan(factor)) {   // median(colSums), R/inferCNV_ops.R:3096
            std::vector<double> cs((size_t)C);
            ICNV_HIP(hipMemcpy(cs.data(), dsum.p, (size_t)C * sizeof(double), hipMemcpyDeviceToHost));
            factor = host_median(std::move(cs));
        }
        if (std::isnan(factor)) ICNV_FAIL(ICNV_ERR_ARG, "normalize factor not estimated");   // :3105
    }
    if (factor_used) *factor_used = factor;
    if ((rc = icnv_normalize_log2_dev(in.dev, dout.as<double>(), G, C, dsum.as<double>(), factor, do_normalize,
                                      do_log2, nullptr)))
        return rc;
    ICNV_HIP(hipMemcpy(expr_out, dout.p, bytes, hipMemcpyDeviceToHost));
    publish_output(expr_out, G * C, std::move(dout));
    return ICNV_OK;
}

// ------------------------------------------------------------------ gene filters / block statistics
int icnv_gene_stats_dev(const double *expr, int64_t G, int64_t C, double *gene_sums, int32_t *gene_nnz, void *stream) {
    if (!expr || !gene_sums || !gene_nnz || G < 1 || C < 0 || G > 0x7fffffff) ICNV_FAIL(ICNV_ERR_ARG, "bad argument");
    hipStream_t s = (hipStream_t)stream;
    if (C == 0) {
        ICNV_HIP(hipMemsetAsync(gene_sums, 0, (size_t)G * sizeof(double), s));
        ICNV_HIP(hipMemsetAsync(gene_nnz, 0, (size_t)G * sizeof(int32_t), s));
        return ICNV_OK;
    }
    const int ns = gene_stats_nsplit((int32_t)G, C);
    DevBuf ps, pn;
    int rc;
    if ((rc = ps.alloc((size_t)ns * G * sizeof(double))) || (rc = pn.alloc((size_t)ns * G * sizeof(int32_t)))) return rc;
    if ((rc = launch_gene_stats(expr, (int32_t)G, C, ns, ps.as<double>(), pn.as<int32_t>(), gene_sums, gene_nnz, s))) return rc;
    ICNV_HIP(hipStreamSynchronize(s));   // the partial buffers go back to the pool
    return ICNV_OK;
}

int icnv_gene_stats(const double *expr, int64_t G, int64_t C, double *gene_sums, int32_t *gene_nnz) {
    if (!expr || !gene_sums || !gene_nnz || G < 1 || C < 1) ICNV_FAIL(ICNV_ERR_ARG, "bad argument");
    MatrixLease in;
    DevBuf ds, dn;
    int rc;
    if ((rc = acquire_input(expr, G * C, nullptr, in)) || (rc = ds.alloc((size_t)G * sizeof(double))) || (rc = dn.alloc((size_t)G * sizeof(int32_t))))
        return rc;
    if ((rc = icnv_gene_stats_dev(in.dev, G, C, ds.as<double>(), dn.as<int32_t>(), nullptr))) return rc;
    ICNV_HIP(hipMemcpy(gene_sums, ds.p, (size_t)G * sizeof(double), hipMemcpyDeviceToHost));
    ICNV_HIP(hipMemcpy(gene_nnz, dn.p, (size_t)G * sizeof(int32_t), hipMemcpyDeviceToHost));
    return ICNV_OK;
}

static int validate_index_list(const int32_t *idx, int64_t n, int64_t limit, const char *what) {
    if (n < 0 || (n > 0 && !idx)) ICNV_FAIL(ICNV_ERR_ARG, std::string("bad ") + what);
    for (int64_t i = 0; i < n; ++i)
        if (idx[i] < 0 || idx[i] >= limit) ICNV_FAIL(ICNV_ERR_ARG, std::string(what) + " index out of range");
    return ICNV_OK;
}

int icnv_select_genes_dev(const double *expr_in, int64_t G_in, int64_t C, const int32_t *keep_idx, int64_t G_out,
                          double *expr_out, void *stream) {
    if (!expr_in || !expr_out || G_in < 1 || C < 0 || G_out < 0 || G_in > 0x7fffffff) ICNV_FAIL(ICNV_ERR_ARG, "bad argument");
    int rc = validate_index_list(keep_idx, G_out, G_in, "gene");
    if (rc) return rc;
    if (G_out == 0 || C == 0) return ICNV_OK;
    hipStream_t s = (hipStream_t)stream;
    DevBuf dk;
    if ((rc = upload(dk, keep_idx, (size_t)G_out, s))) return rc;
    if ((rc = launch_select_genes(expr_in, (int32_t)G_in, C, dk.as<int32_t>(), (int32_t)G_out, expr_out, s))) return rc;
    ICNV_HIP(hipStreamSynchronize(s));
    return ICNV_OK;
}

int icnv_select_genes(const double *expr_in, int64_t G_in, int64_t C, const int32_t *keep_idx, int64_t G_out,
                      double *expr_out) {
    if (!expr_in || !expr_out || G_in < 1 || C < 1 || G_out < 1) ICNV_FAIL(ICNV_ERR_ARG, "bad argument");
    MatrixLease in;
    DevBuf dout;
    int rc;
    if ((rc = acquire_input(expr_in, G_in * C, nullptr, in)) || (rc = dout.alloc((size_t)G_out * C * sizeof(double)))) return rc;
    if ((rc = icnv_select_genes_dev(in.dev, G_in, C, keep_idx, G_out, dout.as<double>(), nullptr))) return rc;
    ICNV_HIP(hipMemcpy(expr_out, dout.p, (size_t)G_out * C * sizeof(double), hipMemcpyDeviceToHost));
    publish_output(expr_out, G_out * C, std::move(dout));
    return ICNV_OK;
}

// values of the matrix at arbitrary element offsets (g + G c): the draws of the spike-in resampling fit
int icnv_gather_values_dev(const double *expr, int64_t n_elements, const int64_t *offsets_host, int64_t n, double *out_host, void *stream) {
    if (!expr || n < 0 || (n > 0 && (!offsets_host || !out_host))) ICNV_FAIL(ICNV_ERR_ARG, "bad argument");
    for (int64_t i = 0; i < n; ++i)
        if (offsets_host[i] < 0 || offsets_host[i] >= n_elements) ICNV_FAIL(ICNV_ERR_ARG, "element offset out of range");
    if (n == 0) return ICNV_OK;
    hipStream_t s = (hipStream_t)stream;
    DevBuf doff, dout;
    int rc;
    if ((rc = upload(doff, offsets_host, (size_t)n, s)) || (rc = dout.alloc((size_t)n * sizeof(double)))) return rc;
    if ((rc = launch_gather_values(expr, doff.as<int64_t>(), n, dout.as<double>(), s))) return rc;
    ICNV_HIP(hipMemcpyAsync(out_host, dout.p, (size_t)n * sizeof(double), hipMemcpyDeviceToHost, s));
    ICNV_HIP(hipStreamSynchronize(s));
    return ICNV_OK;
}
int icnv_gather_values(const double *expr, int64_t G, int64_t C, const int64_t *offsets, int64_t n, double *out) {
    if (!expr || G < 1 || C < 1) ICNV_FAIL(ICNV_ERR_ARG, "bad argument");
    MatrixLease in;
    int rc;
    if ((rc = acquire_input(expr, G * C, nullptr, in))) return rc;
    return icnv_gather_values_dev(in.dev, G * C, offsets, n, out, nullptr);
}

int icnv_block_mean_sd_dev(const double *expr, int64_t G, int64_t C, const int32_t *gene_idx, int64_t n_genes,
                           const int32_t *cell_idx, int64_t n_cells, double *out2_host, void *stream) {
    if (!expr || !out2_host || G < 1 || G > 0x7fffffff || n_cells < 1 || n_cells > 0x7fffffff || !cell_idx)
        ICNV_FAIL(ICNV_ERR_ARG, "bad argument");
    if (!gene_idx) n_genes = G;   // all genes
    if (n_genes < 1) ICNV_FAIL(ICNV_ERR_ARG, "empty gene list");
    int rc = validate_index_list(cell_idx, n_cells, C, "cell");
    if (rc) return rc;
    if (gene_idx && (rc = validate_index_list(gene_idx, n_genes, G, "gene"))) return rc;
    hipStream_t s = (hipStream_t)stream;
    DevBuf dg, dc, dp;
    if (gene_idx && (rc = upload(dg, gene_idx, (size_t)n_genes, s))) return rc;
    if ((rc = upload(dc, cell_idx, (size_t)n_cells, s))) return rc;
    if ((rc = dp.alloc((size_t)n_cells * sizeof(double)))) return rc;
    std::vector<double> part((size_t)n_cells);
    const long double N = (long double)n_cells * (long double)n_genes;
    long double acc[2] = {0, 0};
    double mean = 0.0;
    for (int pass = 0; pass < 2; ++pass) {   // R's mean() then sd(): two passes over the block
        if ((rc = launch_block_cell_reduce(pass, expr, (int32_t)G, gene_idx ? dg.as<int32_t>() : nullptr, (int32_t)n_genes,
                                           dc.as<int32_t>(), (int32_t)n_cells, mean, dp.as<double>(), s)))
            return rc;
        ICNV_HIP(hipMemcpyAsync(part.data(), dp.p, part.size() * sizeof(double), hipMemcpyDeviceToHost, s));
        ICNV_HIP(hipStreamSynchronize(s));
        for (double v : part) acc[pass] += v;
        if (pass == 0) mean = (double)(acc[0] / N);
    }
    out2_host[0] = mean;
    out2_host[1] = N > 1 ? std::sqrt((double)(acc[1] / (N - 1))) : NAN;
    return ICNV_OK;
}

int icnv_block_mean_sd(const double *expr, int64_t G, int64_t C, const int32_t *gene_idx, int64_t n_genes,
                       const int32_t *cell_idx, int64_t n_cells, double *out2) {
    if (!expr || G < 1 || C < 1) ICNV_FAIL(ICNV_ERR_ARG, "bad argument");
    MatrixLease in;
    int rc;
    if ((rc = acquire_input(expr, G * C, nullptr, in))) return rc;
    return icnv_block_mean_sd_dev(in.dev, G, C, gene_idx, n_genes, cell_idx, n_cells, out2, nullptr);
}

// ------------------------------------------------------------------ HMM
static int fill_hmm(HmmParams &p, int32_t K, const double *mean, const double *logPi, const double *logDelta) {
    if (K != 3 && K != 6) ICNV_FAIL(ICNV_ERR_UNSUPPORTED, "HMM kernels are built for K = 6 (i6) and K = 3 (i3)");
    if (!mean || !logPi || !logDelta) ICNV_FAIL(ICNV_ERR_ARG, "null HMM parameter");
    std::memset(&p, 0, sizeof(p));
    p.K = K;
    for (int k = 0; k < K; ++k) { p.mean[k] = mean[k]; p.logDelta[k] = logDelta[k]; }
    for (int i = 0; i < K * K; ++i) p.logPi[i] = logPi[i];
    return ICNV_OK;
}

static void chr_order_longest_first(const int32_t *chr_start, int32_t n_chr, std::vector<int32_t> &order) {
    order.resize(n_chr);
    std::iota(order.begin(), order.end(), 0);
    std::stable_sort(order.begin(), order.end(), [&](int a, int b) {
        return (chr_start[a + 1] - chr_start[a]) > (chr_start[b + 1] - chr_start[b]);
    });
}

// ---- certified fast path (viterbi_fast.hip) ----
// eps_spec: distance of the exact kernel's emission arithmetic (Cody's pnorm, table-driven log, correctly rounded
// divisions: <= ~25 roundings on quantities of relative condition <= 4) from the mathematically exact scores;
// measured 8.9e-16 (tests/test_viterbi_fast_host.py::test_emission_spec_vs_exact), budgeted three orders of
// magnitude higher: the band only grows from 9.4e-9 to 1.4e-8 for a 1 072-gene chromosome.
static constexpr double EPS_SPEC = 1e-12;
static std::atomic<int> g_viterbi_mode{0};   // 0 = auto (fast when eligible), 1 = exact kernel only, 2 = auto without the staged fast kernel (process-wide switch)

// A column batch of the certified fast path whose flagged sequences exceed this share is recomputed as a whole by the
// lane-per-sequence exact kernel instead of sequence by sequence on the wave-per-sequence redo kernel (which wins only
// while the list is short).  The decision is taken ON THE DEVICE from the batch's own flag count (both kernels are
// launched, one of them returns at once): no state survives a call, so which kernels serve a call depends on that
// call's data alone.
static constexpr double REDO_MAX_SHARE = 0.02;
static constexpr int STAGED_MIN_TAIL = 64;   // grid intervals (4 sd) the staged fast Viterbi's short table must reach beyond the outer state means

namespace {
// State of the per-cell Viterbi of ONE device: the emission table of the last (K, mean, sd), the task / flag counters,
// the pinned word that receives the flag count of the last column batch and the statistics of the last call.  Guarded
// by `mu` (a Viterbi call holds it from its first launch to its last enqueue), one instance per device ordinal.
struct ViterbiCtx {
    std::mutex mu;
    bool valid = false, eligible = false;
    int K = 0;
    double mean[8] = {0}, sd = 0;
    EmisTable tab;
    void *dev = nullptr;           // device image of the table, owned (hipMalloc)
    size_t dev_bytes = 0;
    // the short table of the staged kernel (observations through LDS, viterbi_fast.hip) behind the full one in `dev`
    EmisTable tab_s;
    bool staged = false;           // tab_s exists and its tails are long enough to try it first
    size_t dev_s_off = 0;          // its offset in `dev` (bytes)
    int32_t *counters = nullptr;   // device: [0], [1] task counters of the first / second attempt of a batch, [2], [3] their flag counts
    int32_t *host_flag = nullptr;  // pinned [2], receives the flag counts of the last column batch: [0] the staged attempt's own
                                   // (0 without one), [1] the one the redo / exact kernels acted on
    hipEvent_t flag_ev = nullptr;
    int64_t stats[4] = {0, 0, 0, 0};   // last call: path (0 exact / 1 fast), sequences, flagged (-1: pending), table intervals
    int64_t flag_limit = 0;            // of the last column batch: more flagged sequences than this -> exact kernel
    bool staged_last = false;          // the last call tried the staged kernel first
    // chromosome layout of the last call on the device ([n_chr + 1] starts, then [n_chr] chromosomes longest first): a
    // pipeline calls with one layout over and over, and two pageable uploads per call are two stalls of the stream
    std::map<std::vector<int32_t>, int32_t *> layouts;
    int32_t *d_layout = nullptr;   // the current call's
};
std::mutex g_vctx_mu;
std::map<int, ViterbiCtx *> g_vctx;
ViterbiCtx *viterbi_ctx_ptr() {
    // per device AND pool partition: the workers of the one-GPU test of the multi-device path (ICNV_FAKE_DEVICES) run
    // concurrently on one device and must not share task / flag counters any more than workers on different GPUs do
    const int dev = pool_domain();
    std::lock_guard<std::mutex> lk(g_vctx_mu);
    ViterbiCtx *&c = g_vctx[dev];
    if (!c) c = new ViterbiCtx();
    return c;
}

// a = off-diagonal, b = diagonal log transition probability when logPi has that shape (.get_HMM / .i3HMM_get_HMM)
bool structured_pi(const HmmParams &p, double &a, double &b) {
    const int K = p.K;
    a = p.logPi[1];
    b = p.logPi[0];
    for (int j = 0; j < K; ++j)
        for (int k = 0; k < K; ++k) {
            const double v = p.logPi[j + K * k];
            if (memcmp(&v, (j == k) ? &b : &a, sizeof(double)) != 0) return false;
        }
    return std::isfinite(a) && std::isfinite(b) && b <= 0.0 && a < 0.0 && (b - a) > 1e-3;
}
}  // namespace

extern "C++" {
namespace icnv {
void viterbi_release_contexts() {
    int home = 0;
    if (hipGetDevice(&home) != hipSuccess) { (void)hipGetLastError(); return; }
    std::lock_guard<std::mutex> lk(g_vctx_mu);
    for (auto &kv : g_vctx) {
        ViterbiCtx &c = *kv.second;
        std::lock_guard<std::mutex> lk2(c.mu);
        if (hipSetDevice(kv.first / 256) != hipSuccess) { (void)hipGetLastError(); continue; }
        if (c.dev) (void)hipFree(c.dev);
        if (c.counters) (void)hipFree(c.counters);
        if (c.host_flag) (void)hipHostFree(c.host_flag);
        if (c.flag_ev) (void)hipEventDestroy(c.flag_ev);
        for (auto &kv2 : c.layouts) (void)hipFree(kv2.second);
        c.layouts.clear();
        c.d_layout = nullptr;
        c.dev = nullptr; c.dev_bytes = 0; c.counters = nullptr; c.host_flag = nullptr; c.flag_ev = nullptr;
        c.valid = false;
    }
    (void)hipSetDevice(home);
}
}  // namespace icnv
}  // extern "C++"

// caller holds c.mu
static int fast_table_for(ViterbiCtx &c, const HmmParams &p, double sd, hipStream_t s, bool &eligible) {
    eligible = false;
    if (!c.counters) {
        ICNV_HIP(hipMalloc((void **)&c.counters, 4 * sizeof(int32_t)));
        ICNV_HIP(hipHostMalloc((void **)&c.host_flag, 2 * sizeof(int32_t)));
        c.host_flag[0] = c.host_flag[1] = 0;
        ICNV_HIP(hipEventCreateWithFlags(&c.flag_ev, hipEventDisableTiming));
    }
    if (!(c.valid && c.K == p.K && c.sd == sd && memcmp(c.mean, p.mean, sizeof(double) * p.K) == 0)) {
        c.valid = false;   // set again only once the table is built AND its device image is in place
        const char *why = nullptr;
        const bool ok = build_emission_table(p.K, p.mean, sd, viterbi_fast_max_intervals(p.K, false), c.tab, &why) == 0;
        c.staged = false;
        if (ok) {
            std::vector<double> img;
            viterbi_fast_table_image(c.tab, img);
            // the staged kernel's table: the same grid, what fits next to its LDS buffers -- tried first when its tails reach
            // STAGED_MIN_TAIL grid intervals (sd / 16 each) beyond the outer means (data that leave them flag their sequences;
            // a batch with too many of those is redone with the full table)
            const char *why_s = nullptr;
            if (build_emission_table(p.K, p.mean, sd, viterbi_fast_max_intervals(p.K, true), c.tab_s, &why_s) == 0) {
                const int n_tail = (c.tab_s.n_grid - (int)(std::ceil((p.mean[p.K - 1] - p.mean[0]) * c.tab_s.inv_w) + 1.0)) / 2;
                if (n_tail >= STAGED_MIN_TAIL && c.tab_s.n_int < c.tab.n_int) {
                    std::vector<double> img_s;
                    viterbi_fast_table_image(c.tab_s, img_s);
                    if (img.size() & 1) img.push_back(0.0);
                    c.dev_s_off = img.size() * sizeof(double);
                    img.insert(img.end(), img_s.begin(), img_s.end());
                    c.staged = true;
                }
            }
            const size_t bytes = img.size() * sizeof(double);
            if (bytes > c.dev_bytes) {
                if (c.dev) (void)hipFree(c.dev);
                c.dev = nullptr;
                c.dev_bytes = 0;
                ICNV_HIP(hipMalloc(&c.dev, bytes));
                c.dev_bytes = bytes;
            }
            // the previous table may still be read by work queued on the stream: order the upload behind it
            ICNV_HIP(hipStreamSynchronize(s));
            ICNV_HIP(hipMemcpy(c.dev, img.data(), bytes, hipMemcpyHostToDevice));
        }
        c.K = p.K;
        c.sd = sd;
        memcpy(c.mean, p.mean, sizeof(c.mean));
        c.eligible = ok;
        c.valid = true;
    }
    eligible = c.eligible;
    return ICNV_OK;
}

// Viterbi over the columns of x (G x ncols), batched so that the back-pointer
// scratch stays below ~4 GiB.
static int viterbi_columns(const double *x, uint8_t *states, int64_t G, int64_t ncols, const int32_t *chr_start,
                           int32_t n_chr, const HmmParams &p, const double *sd_per_col_dev, double sd_shared,
                           int32_t *n_underflow_dev, hipStream_t s, int64_t ld_x = 0, int64_t ld_st = 0) {
    if (ld_x <= 0) ld_x = G;      // elements between the columns of x / of states (contiguous columns by default)
    if (ld_st <= 0) ld_st = G;
    DevBuf d_bp, d_list, d_redo;
    int rc;
    std::vector<int32_t> order;
    chr_order_longest_first(chr_start, n_chr, order);
    int32_t max_len = 0;
    for (int k = 0; k < n_chr; ++k) max_len = std::max(max_len, chr_start[k + 1] - chr_start[k]);

    ViterbiCtx &vc = *viterbi_ctx_ptr();
    std::lock_guard<std::mutex> vlk(vc.mu);
    {
        // One device buffer per chromosome layout seen (a few hundred bytes each), never overwritten: work queued on any
        // stream may still read an older layout, and a *_dev call must not stall the device (nor break a stream capture)
        // because the layout changed.  A new layout costs one small allocation and one synchronous copy into memory nobody
        // reads yet; a pipeline calls with one layout over and over and finds it here.
        std::vector<int32_t> key(chr_start, chr_start + n_chr + 1);
        key.insert(key.end(), order.begin(), order.end());
        auto it = vc.layouts.find(key);
        if (it == vc.layouts.end()) {
            if (vc.layouts.size() >= 256) {   // a caller that never repeats a layout: start over (the only place that waits)
                ICNV_HIP(hipDeviceSynchronize());
                for (auto &kv : vc.layouts) (void)hipFree(kv.second);
                vc.layouts.clear();
            }
            int32_t *d = nullptr;
            ICNV_HIP(hipMalloc((void **)&d, key.size() * sizeof(int32_t)));
            if (hipMemcpy(d, key.data(), key.size() * sizeof(int32_t), hipMemcpyHostToDevice) != hipSuccess) {
                (void)hipFree(d);
                ICNV_FAIL(ICNV_ERR_HIP, "upload of the chromosome layout failed");
            }
            it = vc.layouts.emplace(std::move(key), d).first;
        }
        vc.d_layout = it->second;
    }
    const int32_t *const dev_chr = vc.d_layout, *const dev_ord = vc.d_layout + n_chr + 1;
    // the certified fast path needs a shared sd, the .get_HMM transition structure and a table that met its accuracy target
    bool fast = false;
    double a = 0, b = 0;
    // ... and initial log probabilities the recurrence can start from: no NaN / +Inf, at least one finite (a state with
    // probability 0 -- -Inf -- is fine: it takes the off-diagonal candidate at the first step)
    bool delta_ok = false;
    for (int k = 0; k < p.K; ++k) delta_ok = delta_ok || std::isfinite(p.logDelta[k]);
    for (int k = 0; k < p.K; ++k) delta_ok = delta_ok && !(std::isnan(p.logDelta[k]) || p.logDelta[k] == INFINITY);
    if (g_viterbi_mode.load() != 1 && !sd_per_col_dev && ncols >= 64 && delta_ok && structured_pi(p, a, b)) {
        if ((rc = fast_table_for(vc, p, sd_shared, s, fast))) return rc;
    }
    // observations through LDS (the staged kernel, its short table first, the full table for a batch that leaves it)?
    const bool staged = fast && vc.staged && g_viterbi_mode.load() == 0 && ld_x * 64 < ((int64_t)1 << 32);
    vc.stats[0] = fast ? 1 : 0;
    vc.stats[1] = ncols * n_chr;
    vc.stats[2] = fast ? -1 : 0;
    vc.stats[3] = fast ? (staged ? vc.tab_s.n_int : vc.tab.n_int) : 0;
    vc.staged_last = staged;

    // Few sequences (the group modes: one column per subcluster / sample): the lane-per-sequence kernel would take
    // as long as its longest chromosome on one lane (~2 ms); the wave-per-sequence kernel -- the redo kernel run
    // over the list of ALL (chromosome, column) pairs, longest chromosomes first -- scores 64 genes at a time.
    if (!fast && ncols * n_chr <= 8192) {
        std::vector<int32_t> list;
        list.reserve((size_t)2 * ncols * n_chr + 1);
        for (int32_t c : order)
            for (int64_t col = 0; col < ncols; ++col) { list.push_back(c); list.push_back((int32_t)col); }
        const int32_t count = (int32_t)(list.size() / 2);
        list.push_back(count);   // the count rides behind the list
        DevBuf d_all, d_scr;
        if ((rc = upload(d_all, list.data(), list.size(), s))) return rc;
        if ((rc = d_scr.alloc(viterbi_redo_scratch_bytes(max_len)))) return rc;
        return launch_viterbi_redo(x, states, (int32_t)G, dev_chr, p, sd_per_col_dev, sd_shared,
                                   d_all.as<int32_t>() + 2 * (size_t)count, d_all.as<int32_t>(), 0x7fffffff, d_scr.as<uint32_t>(),
                                   n_underflow_dev, max_len, "viterbi", s, ld_x, ld_st);
    }
    // the scratch holds the exact kernel's 4-byte back-pointer words; the fast kernel's 2-byte words and its 16-gene block
    // summaries (viterbi_fast_scratch_bytes: G + G / 16 + 3 n_chr + 1 rows of 2 bytes per column) use its first half -- the
    // larger of the two sizes is allocated (the exact kernel only runs after the fast kernel of the same batch has finished)
    // back-pointer scratch per column batch: 16 GiB = 429 000 cells at 10 000 genes in ONE launch (a second, small launch
    // balances its (chromosome, 64 columns) tasks badly over the 4 096 wavefronts: 125 000 cells as 107 000 + 18 000 took
    // 5.9 ms where one launch takes 5.3); HBM is 288 GB
    int64_t scratch_budget = (int64_t)16 << 30;
    if (const char *e = std::getenv("ICNV_VITERBI_SCRATCH_MB")) {   // developer switch: small batches for the tests
        const long v = std::atol(e);
        if (v > 0) scratch_budget = (int64_t)v << 20;
    }
    int64_t batch = scratch_budget / ((int64_t)G * 4);
    batch = std::max<int64_t>(64, (batch / 64) * 64);
    batch = std::min(batch, ((ncols + 63) / 64) * 64);
    // (a smaller GPU, or one whose memory is held by resident matrices: halve the column batch until the scratch fits)
    for (;;) {
        rc = d_bp.alloc(std::max((size_t)G * (size_t)batch * 4, fast ? viterbi_fast_scratch_bytes((int32_t)G, n_chr, batch) : (size_t)0));
        if (!rc || batch <= 64) break;
        batch = std::max<int64_t>(64, (batch / 2 / 64) * 64);
    }
    if (rc) return rc;
    if (fast) {
        if ((rc = d_list.alloc((size_t)2 * n_chr * batch * sizeof(int32_t)))) return rc;
        if ((rc = d_redo.alloc(viterbi_redo_scratch_bytes(max_len)))) return rc;
    }
    for (int64_t c0 = 0; c0 < ncols; c0 += batch) {
        const int64_t nc = std::min(batch, ncols - c0);
        if (!fast) {
            rc = launch_viterbi(x + c0 * ld_x, states + c0 * ld_st, (int32_t)G, nc, dev_chr, dev_ord, n_chr,
                                0, p, sd_per_col_dev ? sd_per_col_dev + c0 : nullptr, sd_shared, d_bp.as<uint32_t>(),
                                n_underflow_dev, nullptr, 0, s, ld_x, ld_st);
            if (rc) return rc;
            continue;
        }
        FastViterbiArgs fa;
        std::memset(&fa, 0, sizeof(fa));
        fa.x = x + c0 * ld_x;
        fa.states = states + c0 * ld_st;
        fa.ld_x = ld_x;
        fa.ld_st = ld_st;
        fa.G = (int32_t)G;
        fa.ncols = nc;
        fa.chr_start = dev_chr;
        fa.chr_order = dev_ord;
        fa.n_chr = n_chr;
        double dmax = 0.0;
        for (int k = 0; k < p.K; ++k) {
            fa.mean[k] = p.mean[k];
            fa.logDelta[k] = p.logDelta[k];
            if (std::isfinite(p.logDelta[k])) dmax = std::max(dmax, std::fabs(p.logDelta[k]));
        }
        fa.a = a;
        fa.b = b;
        fa.b0 = dmax + std::fabs(a);
        fa.bp = d_bp.as<uint16_t>();
        fa.flag_list = d_list.as<int32_t>();
        auto use_table = [&](const EmisTable &t, size_t off) {
            fa.table = (const double *)((const char *)vc.dev + off);
            fa.n_int = t.n_int;
            fa.n_grid = t.n_grid;
            fa.x_lo = t.x_lo;
            fa.x_hi = t.x_hi;
            fa.inv_w = t.inv_w;
            fa.eps = t.eps_tab + 2.0 * EPS_SPEC;   // table: s_k - s_1; the exact kernel's difference carries two of its errors
            fa.s_step = t.s_max + std::fabs(b);
        };
        // flagged sequences: a short list goes to the wave-per-sequence redo kernel; a batch with more than
        // REDO_MAX_SHARE of its sequences flagged (data the table cannot score, or riddled with exact ties) is
        // recomputed as a whole by the lane-per-sequence exact kernel.  Both are launched, the flag count -- on the
        // device -- decides which of them does the work.
        const int32_t limit = (int32_t)std::min<double>(REDO_MAX_SHARE * (double)(nc * n_chr), 2e9);
        ICNV_HIP(hipMemsetAsync(vc.counters, 0, 4 * sizeof(int32_t), s));
        const bool staged_here = staged && nc >= 64;
        if (staged_here) {
            // first attempt: the staged kernel with its short table ...
            use_table(vc.tab_s, vc.dev_s_off);
            fa.task_counter = vc.counters;
            fa.flag_count = vc.counters + 2;
            if ((rc = launch_viterbi_fast(fa, p.K, true, s))) return rc;
            // ... and, only if it flagged more than the redo kernel takes (observations beyond the short table's tails), the
            // register kernel with the full table over the whole batch; otherwise that launch hands the first count on
            use_table(vc.tab, 0);
            fa.gate_count = vc.counters + 2;
            fa.gate_limit = limit;
            fa.task_counter = vc.counters + 1;
            fa.flag_count = vc.counters + 3;
            if ((rc = launch_viterbi_fast(fa, p.K, false, s))) return rc;
        } else {
            use_table(vc.tab, 0);
            fa.task_counter = vc.counters;
            fa.flag_count = vc.counters + 3;
            if ((rc = launch_viterbi_fast(fa, p.K, false, s))) return rc;
        }
        if ((rc = launch_viterbi_redo(fa.x, fa.states, (int32_t)G, dev_chr, p, nullptr, sd_shared, fa.flag_count,
                                      fa.flag_list, limit, d_redo.as<uint32_t>(), n_underflow_dev, max_len, "viterbi_redo", s, ld_x, ld_st)))
            return rc;
        if ((rc = launch_viterbi(fa.x, fa.states, (int32_t)G, nc, dev_chr, dev_ord, n_chr, 0, p, nullptr,
                                 sd_shared, d_bp.as<uint32_t>(), n_underflow_dev, fa.flag_count, limit, s, ld_x, ld_st)))
            return rc;
        // one copy for both counts: [2] the staged attempt's (0 when there was none), [3] the one acted on
        ICNV_HIP(hipMemcpyAsync(vc.host_flag, vc.counters + 2, 2 * sizeof(int32_t), hipMemcpyDeviceToHost, s));
        ICNV_HIP(hipEventRecord(vc.flag_ev, s));
        vc.flag_limit = limit;
    }
    return ICNV_OK;
}

int icnv_viterbi_set_mode(int mode) {
    if (mode != 0 && mode != 1 && mode != 2)
        ICNV_FAIL(ICNV_ERR_ARG, "mode must be 0 (auto), 1 (exact kernel only) or 2 (auto without the staged fast kernel)");
    g_viterbi_mode.store(mode);
    return ICNV_OK;
}

int icnv_viterbi_last_stats(int64_t *out4) {
    if (!out4) ICNV_FAIL(ICNV_ERR_ARG, "null argument");
    ViterbiCtx &vc = *viterbi_ctx_ptr();
    std::lock_guard<std::mutex> vlk(vc.mu);
    if (vc.stats[0] >= 1 && vc.flag_ev) {
        ICNV_HIP(hipEventSynchronize(vc.flag_ev));
        vc.stats[2] = vc.host_flag[1];   // of the last column batch
        // 2: that batch was recomputed by the exact kernel; 3: the staged kernel did it; 4: the staged kernel flagged too many
        // sequences (data beyond its short table) and the register kernel redid the batch with the full table
        if (vc.stats[2] > vc.flag_limit) vc.stats[0] = 2;
        else if (vc.staged_last) vc.stats[0] = (vc.host_flag[0] > vc.flag_limit) ? 4 : 3;
        else vc.stats[0] = 1;
    }
    for (int i = 0; i < 4; ++i) out4[i] = vc.stats[i];
    return ICNV_OK;
}

// Host-only views of the emission table (no GPU needed): what the CPU tests check the certified bound with.
int icnv_hmm_emission_table(int32_t K, const double *mean, double sd, double *meta8, double *seg_out, double *coef_out,
                            int64_t coef_cap) {
    if (!mean || !meta8) ICNV_FAIL(ICNV_ERR_ARG, "null argument");
    if (K != 3 && K != 6) ICNV_FAIL(ICNV_ERR_UNSUPPORTED, "K must be 3 or 6");
    EmisTable t;
    const char *why = "";
    if (build_emission_table(K, mean, sd, viterbi_fast_max_intervals(K, false), t, &why) != 0)
        ICNV_FAIL(ICNV_ERR_UNSUPPORTED, std::string("parameters not eligible for the fast Viterbi path: ") + why);
    meta8[0] = t.n_int; meta8[1] = t.x_lo; meta8[2] = t.x_hi; meta8[3] = t.eps_tab;
    meta8[4] = t.s_max; meta8[5] = EMIS_DEG; meta8[6] = 1; meta8[7] = EPS_SPEC;
    if (seg_out) {   // the uniform grid as one "segment": origin, 1 / width, first record, grid intervals - 1
        seg_out[0] = t.x_lo; seg_out[1] = t.inv_w; seg_out[2] = 0; seg_out[3] = t.n_grid - 1;
    }
    if (coef_out) {
        if ((int64_t)t.coef.size() > coef_cap) ICNV_FAIL(ICNV_ERR_ARG, "coefficient buffer too small");
        std::copy(t.coef.begin(), t.coef.end(), coef_out);
    }
    return ICNV_OK;
}

int icnv_hmm_emission_scores(int32_t K, const double *mean, double sd, const double *x, int64_t n, int32_t which,
                             double *out, uint8_t *ok_out) {
    if (!mean || !x || !out || n < 0) ICNV_FAIL(ICNV_ERR_ARG, "null argument");
    if (K != 3 && K != 6) ICNV_FAIL(ICNV_ERR_UNSUPPORTED, "K must be 3 or 6");
    if (which == 0) {   // exact functions, 80-bit arithmetic, rounded to double
        for (int64_t i = 0; i < n; ++i) emission_scores_exact(K, mean, sd, x[i], out + i * K);
        return ICNV_OK;
    }
    EmisTable t;
    const char *why = "";
    if (build_emission_table(K, mean, sd, viterbi_fast_max_intervals(K, false), t, &why) != 0)
        ICNV_FAIL(ICNV_ERR_UNSUPPORTED, std::string("parameters not eligible for the fast Viterbi path: ") + why);
    for (int64_t i = 0; i < n; ++i) {
        const bool ok = emission_table_eval(t, mean, x[i], out + i * K);
        if (ok_out) ok_out[i] = ok ? 1 : 0;
        if (!ok)
            for (int k = 0; k < K; ++k) out[i * K + k] = NAN;
    }
    return ICNV_OK;
}

int icnv_viterbi_cells_ld_dev(const double *expr, int64_t ld_expr, uint8_t *states, int64_t ld_states, int64_t G, int64_t C,
                              const int32_t *chr_start, int32_t n_chr, int32_t K, const double *mean, double sd_shared,
                              const double *logPi, const double *logDelta, int32_t *n_underflow_dev, void *stream) {
    if (!expr || !states || G < 1 || C < 0 || G > 0x7fffffff) ICNV_FAIL(ICNV_ERR_ARG, "bad argument");
    if (ld_expr < G || ld_states < G) ICNV_FAIL(ICNV_ERR_ARG, "leading dimension below the number of genes");
    int rc = validate_chr(chr_start, n_chr, G);
    if (rc) return rc;
    if (!(sd_shared > 0.0)) ICNV_FAIL(ICNV_ERR_ARG, "sd_shared must be positive");
    HmmParams p;
    if ((rc = fill_hmm(p, K, mean, logPi, logDelta))) return rc;
    return viterbi_columns(expr, states, G, C, chr_start, n_chr, p, nullptr, sd_shared, n_underflow_dev,
                           (hipStream_t)stream, ld_expr, ld_states);
}

int icnv_viterbi_cells_dev(const double *expr, uint8_t *states, int64_t G, int64_t C, const int32_t *chr_start,
                           int32_t n_chr, int32_t K, const double *mean, double sd_shared, const double *logPi,
                           const double *logDelta, int32_t *n_underflow_dev, void *stream) {
    return icnv_viterbi_cells_ld_dev(expr, G, states, G, G, C, chr_start, n_chr, K, mean, sd_shared, logPi, logDelta, n_underflow_dev,
                                     stream);
}

int icnv_group_means_dev(const double *expr, int64_t G, int64_t C, const int32_t *grp_idx, const int32_t *grp_off,
                         int32_t n_grp, double *out, void *stream) {
    if (!expr || !out || G < 1) ICNV_FAIL(ICNV_ERR_ARG, "bad argument");
    int rc = validate_groups(grp_idx, grp_off, n_grp, C, "groups");
    if (rc) return rc;
    if (n_grp == 0) return ICNV_OK;
    for (int q = 0; q < n_grp; ++q)
        if (grp_off[q + 1] == grp_off[q]) ICNV_FAIL(ICNV_ERR_ARG, "empty group");
    hipStream_t s = (hipStream_t)stream;
    DevBuf d_idx, d_off, d_part;
    if ((rc = upload(d_idx, grp_idx, (size_t)grp_off[n_grp], s))) return rc;
    if ((rc = upload(d_off, grp_off, (size_t)n_grp + 1, s))) return rc;
    // The partial-sum workspace is sized per CHUNK of groups (<= 1 GiB), not for all groups at once: with
    // cluster_by_groups = FALSE the reference makes every observation cell its own "sample" (R/inferCNV_HMM.R:528-533), i.e.
    // n_grp ~ C, and a workspace of n_grp * nsplit * 3 * G doubles would be three times the matrix.
    const size_t ws_budget = (size_t)1 << 30;
    const int ns_all = group_means_nsplit((int32_t)G, n_grp);
    int64_t chunk = (int64_t)(ws_budget / ((size_t)ns_all * 3 * (size_t)G * sizeof(double)));
    chunk = std::max<int64_t>(1, std::min<int64_t>(chunk, n_grp));
    const int ns = group_means_nsplit((int32_t)G, (int32_t)chunk);
    if ((rc = d_part.alloc((size_t)chunk * ns * 3 * G * sizeof(double)))) return rc;
    for (int64_t q0 = 0; q0 < n_grp; q0 += chunk) {
        const int32_t nq = (int32_t)std::min<int64_t>(chunk, n_grp - q0);
        if ((rc = launch_group_means_ws(expr, (int32_t)G, d_idx.as<int32_t>(), d_off.as<int32_t>() + q0, nq, ns, d_part.as<double>(),
                                        out + q0 * G, s)))
            return rc;
    }
    return ICNV_OK;
}

// parallelDist(t(expr[, cells])) (Euclidean) as the reference computes it before hclust
// (R/inferCNV_tumor_subclusters.R:191, R/inferCNV_ops.R:1930, 3242): full symmetric n x n matrix.
int icnv_cell_distances_dev(const double *expr, int64_t G, int64_t C, const int32_t *cell_idx, int64_t n, double *dist_out,
                            void *stream) {
    if (!expr || !dist_out || !cell_idx || G < 1 || G > 0x7fffffff || n < 1 || n > 0x7fffffff)
        ICNV_FAIL(ICNV_ERR_ARG, "bad argument");
    int rc = validate_index_list(cell_idx, n, C, "cell");
    if (rc) return rc;
    hipStream_t s = (hipStream_t)stream;
    DevBuf d_idx, d_off, d_part, d_mean, d_diag;
    const int32_t off[2] = {0, (int32_t)n};
    if ((rc = upload(d_idx, cell_idx, (size_t)n, s))) return rc;
    if ((rc = upload(d_off, off, 2, s))) return rc;
    const int ns = group_means_nsplit((int32_t)G, 1);
    if ((rc = d_part.alloc((size_t)ns * 3 * G * sizeof(double))) || (rc = d_mean.alloc((size_t)G * sizeof(double))) ||
        (rc = d_diag.alloc((size_t)n * sizeof(double))))
        return rc;
    if ((rc = launch_group_means_ws(expr, (int32_t)G, d_idx.as<int32_t>(), d_off.as<int32_t>(), 1, ns, d_part.as<double>(),
                                    d_mean.as<double>(), s)))
        return rc;
    return launch_cell_distances(expr, (int32_t)G, d_idx.as<int32_t>(), (int32_t)n, d_mean.as<double>(), d_diag.as<double>(),
                                 dist_out, s);
}

int icnv_cell_distances(const double *expr, int64_t G, int64_t C, const int32_t *cell_idx, int64_t n, double *dist_out) {
    if (!expr || !dist_out || G < 1 || C < 1 || n < 1) ICNV_FAIL(ICNV_ERR_ARG, "bad argument");
    MatrixLease in;
    DevBuf dout;
    int rc;
    if ((rc = acquire_input(expr, G * C, nullptr, in)) || (rc = dout.alloc((size_t)n * n * sizeof(double)))) return rc;
    if ((rc = icnv_cell_distances_dev(in.dev, G, C, cell_idx, n, dout.as<double>(), nullptr))) return rc;
    ICNV_HIP(hipMemcpy(dist_out, dout.p, (size_t)n * n * sizeof(double), hipMemcpyDeviceToHost));
    return ICNV_OK;
}

int icnv_viterbi_groups_dev(const double *expr, uint8_t *states, int64_t G, int64_t C, const int32_t *chr_start,
                            int32_t n_chr, const int32_t *grp_idx, const int32_t *grp_off, int32_t n_grp, int32_t K,
                            const double *mean, const double *sd_shared_per_grp, const double *logPi,
                            const double *logDelta, int32_t *n_underflow_dev, void *stream) {
    if (!expr || !states || G < 1 || C < 0 || G > 0x7fffffff) ICNV_FAIL(ICNV_ERR_ARG, "bad argument");
    int rc = validate_chr(chr_start, n_chr, G);
    if (rc) return rc;
    if ((rc = validate_groups(grp_idx, grp_off, n_grp, C, "groups"))) return rc;
    if (n_grp > 0 && !sd_shared_per_grp) ICNV_FAIL(ICNV_ERR_ARG, "sd_shared_per_grp missing");
    for (int q = 0; q < n_grp; ++q)
        if (!(sd_shared_per_grp[q] > 0.0)) ICNV_FAIL(ICNV_ERR_ARG, "group sd must be positive");
    HmmParams p;
    if ((rc = fill_hmm(p, K, mean, logPi, logDelta))) return rc;
    hipStream_t s = (hipStream_t)stream;
    DevBuf d_gm, d_gs, d_sd, d_map;
    std::vector<int32_t> cell_to_grp((size_t)std::max<int64_t>(C, 1), -1);
    for (int q = 0; q < n_grp; ++q)
        for (int i = grp_off[q]; i < grp_off[q + 1]; ++i) cell_to_grp[grp_idx[i]] = q;  // later groups win, as in R
    if ((rc = upload(d_map, cell_to_grp.data(), (size_t)C, s))) return rc;
    if (n_grp > 0) {
        if ((rc = d_gm.alloc((size_t)G * n_grp * sizeof(double)))) return rc;
        if ((rc = d_gs.alloc((size_t)G * n_grp))) return rc;
        if ((rc = upload(d_sd, sd_shared_per_grp, (size_t)n_grp, s))) return rc;
        if ((rc = icnv_group_means_dev(expr, G, C, grp_idx, grp_off, n_grp, d_gm.as<double>(), stream))) return rc;
        if ((rc = viterbi_columns(d_gm.as<double>(), d_gs.as<uint8_t>(), G, n_grp, chr_start, n_chr, p,
                                  d_sd.as<double>(), 0.0, n_underflow_dev, s)))
            return rc;
    }
    return launch_broadcast_states(d_gs.as<uint8_t>(), (int32_t)G, C, d_map.as<int32_t>(), states, s);
}

// ------------------------------------------------------------------ i3 HMM at group level, device-resident parameters (round 6)
// predict via i3HMM_predict_CNV_via_HMM_on_tumor_subclusters / _whole_tumor_samples (R/inferCNV_i3HMM.R:249-389) as a PLAN: the
// group structure goes to the device once; a step is  group means + the reference cells' moments (ONE pass over the matrix)
// -> [all-reduce of 3 doubles in a cell-sharded run] -> mu, sigma, delta on the device -> Viterbi per group with its parameters
// read from device memory -> broadcast.  No upload, no download and no stream synchronisation inside a step (round 5's step
// spent 0.29 of its 1.43 ms on them: two host round trips for the moments, five uploads of the group structure).
struct icnv_group_hmm {
    int64_t G = 0, C = 0;
    int32_t n_chr = 0, n_grp = 0, max_len = 0, ns = 1;
    int64_t n_ref = 0;
    std::vector<int32_t> chr_start;
    DevBuf d_chr, d_idx, d_off, d_map, d_ref, d_kind, d_list, d_gm, d_gs, d_part, d_mom, d_m3, d_params, d_scr, d_bad;
    int32_t count = 0;
};

int icnv_group_hmm_begin(icnv_group_hmm_t **out, int64_t G, int64_t C, const int32_t *chr_start, int32_t n_chr,
                         const int32_t *grp_idx, const int32_t *grp_off, int32_t n_grp, const int32_t *ref_idx, int64_t n_ref) {
    if (!out) ICNV_FAIL(ICNV_ERR_ARG, "null argument");
    *out = nullptr;
    if (G < 1 || C < 0 || G > 0x7fffffff || C > 0x7fffffff || n_grp < 0 || n_ref < 0) ICNV_FAIL(ICNV_ERR_ARG, "bad dimensions");
    int rc = validate_chr(chr_start, n_chr, G);
    if (rc) return rc;
    if ((rc = validate_groups(grp_idx, grp_off, n_grp, C, "groups"))) return rc;
    if (n_ref > 0 && (rc = validate_index_list(ref_idx, n_ref, C, "reference cell"))) return rc;
    for (int q = 0; q < n_grp; ++q)
        if (grp_off[q + 1] == grp_off[q]) ICNV_FAIL(ICNV_ERR_ARG, "empty group");
    if ((int64_t)n_grp * n_chr > 8192)
        ICNV_FAIL(ICNV_ERR_UNSUPPORTED, "group HMM plan: more than 8192 (group, chromosome) sequences -- use icnv_viterbi_groups_dev");
    // every cell in at most one group, every reference cell in exactly one: the moments are gathered while the groups' cells stream by
    std::vector<int32_t> cell_to_grp((size_t)std::max<int64_t>(C, 1), -1);
    for (int q = 0; q < n_grp; ++q)
        for (int i = grp_off[q]; i < grp_off[q + 1]; ++i) {
            if (cell_to_grp[grp_idx[i]] >= 0) ICNV_FAIL(ICNV_ERR_UNSUPPORTED, "group HMM plan: a cell in two groups -- use icnv_viterbi_groups_dev");
            cell_to_grp[grp_idx[i]] = q;
        }
    std::vector<uint8_t> flag((size_t)std::max<int64_t>(C, 1), 0);
    for (int64_t i = 0; i < n_ref; ++i) {
        if (cell_to_grp[ref_idx[i]] < 0) ICNV_FAIL(ICNV_ERR_UNSUPPORTED, "group HMM plan: a reference cell in no group");
        if (flag[ref_idx[i]]) ICNV_FAIL(ICNV_ERR_ARG, "reference cell listed twice");
        flag[ref_idx[i]] = 1;
    }
    auto *h = new icnv_group_hmm();
    h->G = G; h->C = C; h->n_chr = n_chr; h->n_grp = n_grp; h->n_ref = n_ref;
    h->chr_start.assign(chr_start, chr_start + n_chr + 1);
    for (int k = 0; k < n_chr; ++k) h->max_len = std::max(h->max_len, chr_start[k + 1] - chr_start[k]);
    std::vector<int32_t> order;
    chr_order_longest_first(chr_start, n_chr, order);
    std::vector<int32_t> list;
    for (int32_t c : order)
        for (int32_t q = 0; q < n_grp; ++q) { list.push_back(c); list.push_back(q); }
    h->count = (int32_t)(list.size() / 2);
    list.push_back(h->count);   // the count rides behind the list
    h->ns = group_means_nsplit((int32_t)G, std::max(n_grp, 1));
    hipStream_t s = nullptr;
    auto fail = [&](int code) { delete h; return code; };
    if ((rc = upload(h->d_chr, h->chr_start.data(), h->chr_start.size(), s))) return fail(rc);
    if ((rc = upload(h->d_idx, grp_idx, n_grp ? (size_t)grp_off[n_grp] : 0, s))) return fail(rc);
    if ((rc = upload(h->d_off, grp_off, (size_t)n_grp + 1, s))) return fail(rc);
    if ((rc = upload(h->d_map, cell_to_grp.data(), cell_to_grp.size(), s))) return fail(rc);
    if ((rc = upload(h->d_ref, flag.data(), flag.size(), s))) return fail(rc);
    std::vector<uint8_t> kind((size_t)std::max(n_grp, 1), 0);   // per group: 0 no reference cell, 1 reference cells only, 2 mixed
    for (int q = 0; q < n_grp; ++q) {
        int nr = 0;
        for (int i = grp_off[q]; i < grp_off[q + 1]; ++i) nr += flag[grp_idx[i]];
        kind[(size_t)q] = nr == 0 ? 0 : (nr == grp_off[q + 1] - grp_off[q] ? 1 : 2);
    }
    if ((rc = upload(h->d_kind, kind.data(), kind.size(), s))) return fail(rc);
    if ((rc = upload(h->d_list, list.data(), list.size(), s))) return fail(rc);
    const size_t ng = (size_t)std::max(n_grp, 1);
    if ((rc = h->d_gm.alloc((size_t)G * ng * sizeof(double))) || (rc = h->d_gs.alloc((size_t)G * ng)) ||
        (rc = h->d_part.alloc(ng * h->ns * 3 * (size_t)G * sizeof(double))) ||
        (rc = h->d_mom.alloc((size_t)std::max<int64_t>(group_means_moment_blocks((int32_t)G, (int32_t)ng, h->ns), 1) * 2 * sizeof(double))) ||
        (rc = h->d_m3.alloc(4 * sizeof(double))) || (rc = h->d_params.alloc(8 * sizeof(double))) ||
        (rc = h->d_scr.alloc(viterbi_redo_scratch_bytes(h->max_len))) || (rc = h->d_bad.alloc(sizeof(int32_t))))
        return fail(rc);
    if (hipStreamSynchronize(s) != hipSuccess) { delete h; ICNV_FAIL(ICNV_ERR_HIP, "upload of the group structure failed"); }   // (the host vectors go out of scope)
    *out = h;
    return ICNV_OK;
}

int icnv_group_hmm_i3_partial_dev(icnv_group_hmm_t *h, const double *expr, double **moments_dev, void *stream) {
    if (!h || (!expr && h->C > 0)) ICNV_FAIL(ICNV_ERR_ARG, "null argument");
    hipStream_t s = (hipStream_t)stream;
    int rc;
    const int64_t nb = group_means_moment_blocks((int32_t)h->G, h->n_grp, h->ns);
    if (h->n_grp > 0) {
        if ((rc = launch_group_means_ws(expr, (int32_t)h->G, h->d_idx.as<int32_t>(), h->d_off.as<int32_t>(), h->n_grp, h->ns, h->d_part.as<double>(),
                                        h->d_gm.as<double>(), s, h->d_ref.as<uint8_t>(), h->d_mom.as<double>(), h->d_kind.as<uint8_t>())))
            return rc;
    }
    // {S1, S2, n}: this rank's share (zeros from a rank without groups); all-reduce(sum) them in a cell-sharded run
    if ((rc = launch_reduce_moments(h->d_mom.as<double>(), h->n_grp > 0 ? nb : 0, (double)h->n_ref * (double)h->G, h->d_m3.as<double>(), s))) return rc;
    if (moments_dev) *moments_dev = h->d_m3.as<double>();
    return ICNV_OK;
}

int icnv_group_hmm_i3_finish_dev(icnv_group_hmm_t *h, uint8_t *states, const double *logPi, const double *logDelta, double z_abs,
                                 double delta_abs, int32_t *n_underflow_dev, void *stream) {
    if (!h || (!states && h->C > 0)) ICNV_FAIL(ICNV_ERR_ARG, "null argument");
    if (!(delta_abs == delta_abs) && !(z_abs > 0.0)) ICNV_FAIL(ICNV_ERR_ARG, "z_abs must be positive when no delta is given");
    hipStream_t s = (hipStream_t)stream;
    int rc;
    if ((rc = launch_i3_params(h->d_m3.as<double>(), z_abs, delta_abs, h->d_params.as<double>(), s))) return rc;
    // logPi / logDelta: log of .i3HMM_get_HMM's matrices (R/inferCNV_i3HMM.R:99-156), prepared by the caller as for icnv_viterbi_groups_dev
    const double mean0[3] = {0.0, 0.0, 0.0};   // (the means come from the device)
    HmmParams p;
    if ((rc = fill_hmm(p, 3, mean0, logPi, logDelta))) return rc;
    int32_t *bad = n_underflow_dev ? n_underflow_dev : h->d_bad.as<int32_t>();
    if (h->n_grp > 0) {
        if ((rc = launch_viterbi_redo(h->d_gm.as<double>(), h->d_gs.as<uint8_t>(), (int32_t)h->G, h->d_chr.as<int32_t>(), p, nullptr, 1.0,
                                      h->d_list.as<int32_t>() + 2 * (size_t)h->count, h->d_list.as<int32_t>(), 0x7fffffff, h->d_scr.as<uint32_t>(),
                                      bad, h->max_len, "viterbi", s, 0, 0, h->d_params.as<double>())))
            return rc;
    }
    return launch_broadcast_states(h->d_gs.as<uint8_t>(), (int32_t)h->G, h->C, h->d_map.as<int32_t>(), states, s);
}

int icnv_group_hmm_get_i3_params(icnv_group_hmm_t *h, double *mu_sigma_delta, void *stream) {
    if (!h || !mu_sigma_delta) ICNV_FAIL(ICNV_ERR_ARG, "null argument");
    hipStream_t s = (hipStream_t)stream;
    double buf[6];
    ICNV_HIP(hipMemcpyAsync(buf, h->d_params.p, sizeof(buf), hipMemcpyDeviceToHost, s));
    ICNV_HIP(hipStreamSynchronize(s));
    mu_sigma_delta[0] = buf[4]; mu_sigma_delta[1] = buf[3]; mu_sigma_delta[2] = buf[5];
    return ICNV_OK;
}

void icnv_group_hmm_end(icnv_group_hmm_t *h) { delete h; }

}  // extern "C"
// single-device forms of the host-buffer group entry points (the extern "C" ones live in host_path.hip: they deal whole
// groups / tiles to the devices of icnv_set_devices)
namespace icnv {
int viterbi_groups_host_one(const double *expr, uint8_t *states, int64_t G, int64_t C, const int32_t *chr_start,
                            int32_t n_chr, const int32_t *grp_idx, const int32_t *grp_off, int32_t n_grp, int32_t K,
                            const double *mean, const double *sd_shared_per_grp, const double *logPi,
                            const double *logDelta) {
    if (!expr || !states) ICNV_FAIL(ICNV_ERR_ARG, "null argument");
    MatrixLease in;
    DevBuf ds, dn;
    int rc;
    const size_t n = (size_t)G * (size_t)C;
    if ((rc = acquire_input(expr, G * C, nullptr, in)) || (rc = ds.alloc(std::max<size_t>(n, 1))) || (rc = dn.alloc(sizeof(int32_t))))
        return rc;
    ICNV_HIP(hipMemset(dn.p, 0, sizeof(int32_t)));
    rc = icnv_viterbi_groups_dev(in.dev, ds.as<uint8_t>(), G, C, chr_start, n_chr, grp_idx, grp_off, n_grp, K,
                                 mean, sd_shared_per_grp, logPi, logDelta, dn.as<int32_t>(), nullptr);
    if (rc) return rc;
    int32_t bad = 0;
    ICNV_HIP(hipMemcpy(states, ds.p, n, hipMemcpyDeviceToHost));
    ICNV_HIP(hipMemcpy(&bad, dn.p, sizeof(int32_t), hipMemcpyDeviceToHost));
    if (bad) ICNV_FAIL(ICNV_ERR_UNDERFLOW, "Problems With Underflow in " + std::to_string(bad) + " sequences");
    return ICNV_OK;
}
}  // namespace icnv
extern "C" {

int icnv_state_consensus_dev(const uint8_t *states, int64_t G, int64_t C, const int32_t *grp_idx,
                             const int32_t *grp_off, int32_t n_grp, uint8_t *consensus, uint8_t *states_out,
                             void *stream) {
    if (!states || G < 1 || C < 0 || G > 0x7fffffff || (!consensus && !states_out)) ICNV_FAIL(ICNV_ERR_ARG, "bad argument");
    int rc = validate_groups(grp_idx, grp_off, n_grp, C, "groups");
    if (rc) return rc;
    if (n_grp > 65535) ICNV_FAIL(ICNV_ERR_UNSUPPORTED, "more than 65535 groups");
    hipStream_t s = (hipStream_t)stream;
    DevBuf d_idx, d_off, d_cons, d_map;
    if ((rc = upload(d_idx, grp_idx, n_grp ? (size_t)grp_off[n_grp] : 0, s))) return rc;
    if ((rc = upload(d_off, grp_off, (size_t)n_grp + 1, s))) return rc;
    uint8_t *cons = consensus;
    if (!cons) {
        if ((rc = d_cons.alloc((size_t)G * std::max(n_grp, 1)))) return rc;
        cons = d_cons.as<uint8_t>();
    }
    if ((rc = launch_state_consensus(states, (int32_t)G, d_idx.as<int32_t>(), d_off.as<int32_t>(), n_grp, cons, s))) return rc;
    if (states_out) {   // every member cell gets its group's consensus; cells in no group keep their states
        std::vector<int32_t> cell_to_grp((size_t)std::max<int64_t>(C, 1), -1);
        for (int q = 0; q < n_grp; ++q)
            for (int i = grp_off[q]; i < grp_off[q + 1]; ++i) cell_to_grp[grp_idx[i]] = q;
        if ((rc = upload(d_map, cell_to_grp.data(), (size_t)C, s))) return rc;
        if (states_out != states)
            ICNV_HIP(hipMemcpyAsync(states_out, states, (size_t)G * C, hipMemcpyDeviceToDevice, s));
        if ((rc = launch_broadcast_states_keep(cons, (int32_t)G, C, d_map.as<int32_t>(), states_out, s))) return rc;
    }
    return ICNV_OK;
}

int icnv_state_consensus(const uint8_t *states, int64_t G, int64_t C, const int32_t *grp_idx, const int32_t *grp_off,
                         int32_t n_grp, uint8_t *consensus, uint8_t *states_out) {
    if (!states) ICNV_FAIL(ICNV_ERR_ARG, "null argument");
    DevBuf ds, dc, dout;
    int rc;
    const size_t n = (size_t)G * (size_t)C;
    if ((rc = ds.alloc(std::max<size_t>(n, 1)))) return rc;
    if (consensus && (rc = dc.alloc((size_t)G * std::max(n_grp, 1)))) return rc;
    if (states_out && (rc = dout.alloc(std::max<size_t>(n, 1)))) return rc;
    ICNV_HIP(hipMemcpy(ds.p, states, n, hipMemcpyHostToDevice));
    rc = icnv_state_consensus_dev(ds.as<uint8_t>(), G, C, grp_idx, grp_off, n_grp, consensus ? dc.as<uint8_t>() : nullptr,
                                  states_out ? dout.as<uint8_t>() : nullptr, nullptr);
    if (rc) return rc;
    if (consensus) ICNV_HIP(hipMemcpy(consensus, dc.p, (size_t)G * n_grp, hipMemcpyDeviceToHost));
    if (states_out) ICNV_HIP(hipMemcpy(states_out, dout.p, n, hipMemcpyDeviceToHost));
    return ICNV_OK;
}

int icnv_states_to_proxy_dev(const uint8_t *states, double *out, int64_t n, int32_t K, void *stream) {
    if (!states || !out || n < 0 || (K != 3 && K != 6)) ICNV_FAIL(ICNV_ERR_ARG, "bad argument");
    return launch_states_to_proxy(states, out, n, K, (hipStream_t)stream);
}

int icnv_states_to_proxy(const uint8_t *states, double *out, int64_t n, int32_t K) {
    if (!states || !out) ICNV_FAIL(ICNV_ERR_ARG, "null argument");
    DevBuf ds, dout;
    int rc;
    if ((rc = ds.alloc(std::max<int64_t>(n, 1))) || (rc = dout.alloc(std::max<int64_t>(n, 1) * sizeof(double)))) return rc;
    ICNV_HIP(hipMemcpy(ds.p, states, n, hipMemcpyHostToDevice));
    if ((rc = icnv_states_to_proxy_dev(ds.as<uint8_t>(), dout.as<double>(), n, K, nullptr))) return rc;
    ICNV_HIP(hipMemcpy(out, dout.p, n * sizeof(double), hipMemcpyDeviceToHost));
    return ICNV_OK;
}

// Sum over the listed cells of: the cell's values (pass 0) or their squared deviations from `mean` (pass 1) -- one
// streaming launch (a workgroup per cell, fixed-order tree), the per-cell partials added in long double on the host.
static int cells_pass_sum(int pass, const double *expr, int64_t G, const int32_t *cells_dev, int64_t n_cells, double mean,
                          DevBuf &dp, std::vector<double> &part, long double &acc, hipStream_t s) {
    int rc = launch_block_cell_reduce(pass, expr, (int32_t)G, nullptr, (int32_t)G, cells_dev, (int32_t)n_cells, mean, dp.as<double>(), s);
    if (rc) return rc;
    ICNV_HIP(hipMemcpyAsync(part.data(), dp.p, part.size() * sizeof(double), hipMemcpyDeviceToHost, s));
    ICNV_HIP(hipStreamSynchronize(s));
    acc = 0;
    for (double v : part) acc += v;
    return ICNV_OK;
}

// Split-phase mean / sd over ALL values of the listed cells (i3 emission parameters, R/inferCNV_i3HMM.R:17-80) for a
// cell-sharded caller: the statistic is two dependent sums,
//   phase 0: out3 = {sum of the values, number of values, 0}                       -> all-reduce -> mean = sum / n
//   phase 1: out3 = {sum of (x - mean)^2 over the values, number of values, 0}     -> all-reduce -> sd = sqrt(ss / (n - 1))
// (R's sd() is this two-pass form: each phase is one pass over the listed cells at HBM speed).
// A rank without listed cells passes n_cells = 0 and contributes zeros.
int icnv_cells_moments_partial_dev(const double *expr, int64_t G, int64_t C, const int32_t *cell_idx, int64_t n_cells,
                                   int32_t phase, double mean, double *out3_host, void *stream) {
    if (!out3_host || G < 1 || G > 0x7fffffff || n_cells < 0 || n_cells > 0x7fffffff || (n_cells > 0 && !expr) || (phase != 0 && phase != 1))
        ICNV_FAIL(ICNV_ERR_ARG, "bad argument");
    int rc = validate_index_list(cell_idx, n_cells, C, "cell");
    if (rc) return rc;
    long double acc = 0;
    if (n_cells > 0) {
        hipStream_t s = (hipStream_t)stream;
        DevBuf dc, dp;
        if ((rc = upload(dc, cell_idx, (size_t)n_cells, s))) return rc;
        if ((rc = dp.alloc((size_t)n_cells * sizeof(double)))) return rc;
        std::vector<double> part((size_t)n_cells);
        if ((rc = cells_pass_sum(phase, expr, G, dc.as<int32_t>(), n_cells, mean, dp, part, acc, s))) return rc;
    }
    out3_host[0] = (double)acc;
    out3_host[1] = (double)n_cells * (double)G;
    out3_host[2] = 0.0;
    return ICNV_OK;
}

int icnv_cells_mean_sd_dev(const double *expr, int64_t G, int64_t C, const int32_t *cell_idx, int64_t n_cells,
                           double *out2_host, void *stream) {
    if (!expr || !out2_host || G < 1 || n_cells < 1) ICNV_FAIL(ICNV_ERR_ARG, "bad argument");
    // one rank: mean() then sd() over the block of all genes x the listed cells
    return icnv_block_mean_sd_dev(expr, G, C, nullptr, 0, cell_idx, n_cells, out2_host, stream);
}

int icnv_cells_mean_sd(const double *expr, int64_t G, int64_t C, const int32_t *cell_idx, int64_t n_cells, double *out2) {
    if (!expr || !out2 || G < 1 || C < 1) ICNV_FAIL(ICNV_ERR_ARG, "bad argument");
    int rc = validate_index_list(cell_idx, n_cells, C, "cell");
    if (rc) return rc;
    MatrixLease in;
    if ((rc = acquire_input(expr, G * C, nullptr, in))) return rc;
    return icnv_cells_mean_sd_dev(in.dev, G, C, cell_idx, n_cells, out2, nullptr);
}

// ------------------------------------------------------------------ median filter
int icnv_median_filter_dev(const double *expr_in, double *expr_out, int64_t G, int64_t C, const int32_t *chr_start,
                           int32_t n_chr, const int32_t *tile_idx, const int32_t *tile_off, int32_t n_tiles,
                           int32_t window_size, void *stream) {
    if (!expr_in || !expr_out || G < 1 || C < 0 || G > 0x7fffffff) ICNV_FAIL(ICNV_ERR_ARG, "bad argument");
    if (expr_in == expr_out) ICNV_FAIL(ICNV_ERR_ARG, "median filter cannot run in place");
    if (window_size < 1) ICNV_FAIL(ICNV_ERR_ARG, "window_size must be >= 1");
    int rc = validate_chr(chr_start, n_chr, G);
    if (rc) return rc;
    if ((rc = validate_groups(tile_idx, tile_off, n_tiles, C, "tiles"))) return rc;
    hipStream_t s = (hipStream_t)stream;
    // cells that belong to no tile keep their values (R/noise_reduction.R:60-86 only assigns the tiles' blocks); when the
    // tiles cover every cell -- the usual call: all subclusters and all reference groups -- every element is
    // written by the filter and the copy (2 x 8 bytes per element) is skipped
    bool covered = false;
    if (n_tiles > 0 && (int64_t)tile_off[n_tiles] >= C) {
        std::vector<bool> seen((size_t)C, false);
        int64_t n_seen = 0;
        for (int64_t i = 0; i < tile_off[n_tiles]; ++i)
            if (!seen[(size_t)tile_idx[i]]) { seen[(size_t)tile_idx[i]] = true; ++n_seen; }
        covered = n_seen == C;
    }
    if (!covered) ICNV_HIP(hipMemcpyAsync(expr_out, expr_in, (size_t)G * C * sizeof(double), hipMemcpyDeviceToDevice, s));
    if (n_tiles == 0) return ICNV_OK;
    std::vector<int32_t> blk_off((size_t)n_tiles + 1, 0);
    for (int t = 0; t < n_tiles; ++t)
        blk_off[t + 1] = blk_off[t] + (tile_off[t + 1] - tile_off[t] + MEDIAN_CELLS_PER_PATCH - 1) / MEDIAN_CELLS_PER_PATCH;
    // 9 x 9 windows (median_kernels.hip: a classification pass, a dense pass over the tiles that need one, single outputs).
    // Kernel 1 cuts every (cell tile, chromosome) block into tiles of 56 genes x 32 cells from its first gene / cell, borders
    // included; the dense pass has its own grid of 32 genes x 16 cells over the same blocks (a tile of kernel 1 covers two of
    // its cell blocks exactly and two or three of its gene blocks).
    std::vector<int32_t> gdesc, cdesc, g1desc, c1desc, sdesc, segdesc, sw1desc, sw2desc;
    if (median_is_9x9(window_size)) {
        for (int k = 0; k < n_chr; ++k) {
            const int32_t cs = chr_start[k], xdim = chr_start[k + 1] - chr_start[k];
            const int32_t kb2 = (int32_t)(gdesc.size() / 4);   // the chromosome's first dense-pass gene block
            for (int g0 = 0; g0 < xdim; g0 += MEDIAN9_STRIP_GENES) {   // strip kernel: 64 output genes = two dense-pass gene blocks
                const int32_t r[4] = {cs, xdim, g0, kb2 + g0 / MEDIAN_GENES_PER_PATCH};
                sdesc.insert(sdesc.end(), r, r + 4);
            }
            for (int g0 = 0; g0 < xdim; g0 += MEDIAN_GENES_PER_PATCH) {   // dense pass: {cs, xdim, first gene, end of its interior outputs}
                const int32_t r[4] = {cs, xdim, g0, std::min(g0 + MEDIAN_GENES_PER_PATCH, xdim - 4)};
                gdesc.insert(gdesc.end(), r, r + 4);
            }
            const int32_t kb1 = (int32_t)(g1desc.size() / 4);   // the chromosome's first gene block of kernel 1
            for (int g0 = 0; g0 < xdim; g0 += MEDIAN9_K1_GENES) {
                const int32_t r1[4] = {cs, xdim, g0, kb2};
                g1desc.insert(g1desc.end(), r1, r1 + 4);
            }
            for (int nh = 1; nh <= 2; ++nh) {                   // the sweep's gene blocks: 56 genes (one 64-gene row per wavefront) and 120 (two)
                std::vector<int32_t> &sw = nh == 1 ? sw1desc : sw2desc;
                for (int g0 = 0; g0 < xdim; g0 += 64 * nh - 8) {
                    const int32_t r[8] = {cs, xdim, g0, kb2, kb1, 0, 0, 0};
                    sw.insert(sw.end(), r, r + 8);
                }
            }
        }
        for (int t = 0; t < n_tiles; ++t) {
            const int32_t ydim = tile_off[t + 1] - tile_off[t];
            for (int c0 = 0; c0 < ydim; c0 += MEDIAN9_SEG_BLOCKS * MEDIAN9_CELLS_PER_PATCH) {   // strip kernel: segments of eight dense-pass cell blocks
                const int32_t r[4] = {tile_off[t], ydim, c0, (int32_t)(cdesc.size() / 4) + c0 / MEDIAN9_CELLS_PER_PATCH};
                segdesc.insert(segdesc.end(), r, r + 4);
            }
            for (int c0 = 0; c0 < ydim; c0 += MEDIAN9_K1_CELLS) {
                const int32_t r1[4] = {tile_off[t], ydim, c0, (int32_t)(cdesc.size() / 4)};
                c1desc.insert(c1desc.end(), r1, r1 + 4);
                for (int half = 0; half < 2; ++half) {
                    const int32_t r[4] = {tile_off[t], ydim, c0 + half * MEDIAN9_CELLS_PER_PATCH, 0};
                    cdesc.insert(cdesc.end(), r, r + 4);
                }
            }
        }
    }
    while (!c1desc.empty() && (c1desc.size() / 4) % MEDIAN9_K1_RUN != 0) {   // kernel 1 walks runs of MEDIAN9_K1_RUN cell blocks: pad with empty ones
        const int32_t r1[4] = {0, 0, 0, 0};
        c1desc.insert(c1desc.end(), r1, r1 + 4);
    }
    DevBuf d_chr, d_idx, d_off, d_blk, d_gd, d_cd, d_g1, d_c1, d_sd, d_seg, d_sw1, d_sw2;
    if (!sw1desc.empty() && (rc = upload(d_sw1, sw1desc.data(), sw1desc.size(), s))) return rc;
    if (!sw2desc.empty() && (rc = upload(d_sw2, sw2desc.data(), sw2desc.size(), s))) return rc;
    if (!sdesc.empty() && (rc = upload(d_sd, sdesc.data(), sdesc.size(), s))) return rc;
    if (!segdesc.empty() && (rc = upload(d_seg, segdesc.data(), segdesc.size(), s))) return rc;
    if (!gdesc.empty() && (rc = upload(d_gd, gdesc.data(), gdesc.size(), s))) return rc;
    if (!cdesc.empty() && (rc = upload(d_cd, cdesc.data(), cdesc.size(), s))) return rc;
    if (!g1desc.empty() && (rc = upload(d_g1, g1desc.data(), g1desc.size(), s))) return rc;
    if (!c1desc.empty() && (rc = upload(d_c1, c1desc.data(), c1desc.size(), s))) return rc;
    if ((rc = upload(d_chr, chr_start, (size_t)n_chr + 1, s))) return rc;
    if ((rc = upload(d_idx, tile_idx, (size_t)tile_off[n_tiles], s))) return rc;
    if ((rc = upload(d_off, tile_off, (size_t)n_tiles + 1, s))) return rc;
    if ((rc = upload(d_blk, blk_off.data(), blk_off.size(), s))) return rc;
    Median9Plan plan9;
    DevBuf d_queue;
    plan9.queue = &d_queue;
    plan9.gene_block_desc = d_gd.as<int32_t>();
    plan9.cell_patch_desc = d_cd.as<int32_t>();
    plan9.n_gene_blocks = (int32_t)(gdesc.size() / 4);
    plan9.n_cell_patches = (int32_t)(cdesc.size() / 4);
    plan9.gene1_desc = d_g1.as<int32_t>();
    plan9.cell1_desc = d_c1.as<int32_t>();
    plan9.n_gene_blocks1 = (int32_t)(g1desc.size() / 4);
    plan9.n_cell_patches1 = (int32_t)(c1desc.size() / 4);
    plan9.strip_desc = d_sd.as<int32_t>();
    plan9.seg_desc = d_seg.as<int32_t>();
    plan9.n_strips = (int32_t)(sdesc.size() / 4);
    plan9.n_segs = (int32_t)(segdesc.size() / 4);
    plan9.n_list = tile_off[n_tiles];
    plan9.sweep_desc1 = d_sw1.as<int32_t>();
    plan9.sweep_desc2 = d_sw2.as<int32_t>();
    plan9.n_sweep_blocks1 = (int32_t)(sw1desc.size() / 8);
    plan9.n_sweep_blocks2 = (int32_t)(sw2desc.size() / 8);
    return launch_median_filter(expr_in, expr_out, (int32_t)G, C, d_chr.as<int32_t>(), n_chr, d_idx.as<int32_t>(),
                                d_off.as<int32_t>(), n_tiles, d_blk.as<int32_t>(), chr_start, blk_off[n_tiles],
                                window_size, plan9, s);
}

}  // extern "C"
namespace icnv {
int median_filter_host_one(const double *expr_in, double *expr_out, int64_t G, int64_t C, const int32_t *chr_start,
                           int32_t n_chr, const int32_t *tile_idx, const int32_t *tile_off, int32_t n_tiles,
                           int32_t window_size) {
    if (!expr_in || !expr_out) ICNV_FAIL(ICNV_ERR_ARG, "null argument");
    MatrixLease in;
    DevBuf dout;
    int rc;
    const size_t bytes = (size_t)G * (size_t)C * sizeof(double);
    if ((rc = acquire_input(expr_in, G * C, nullptr, in)) || (rc = dout.alloc(bytes))) return rc;
    rc = icnv_median_filter_dev(in.dev, dout.as<double>(), G, C, chr_start, n_chr, tile_idx, tile_off, n_tiles,
                                window_size, nullptr);
    if (rc) return rc;
    ICNV_HIP(hipMemcpy(expr_out, dout.p, bytes, hipMemcpyDeviceToHost));
    publish_output(expr_out, G * C, std::move(dout));
    return ICNV_OK;
}
}  // namespace icnv
