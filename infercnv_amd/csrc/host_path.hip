// Host-buffer side of the C ABI (include/icnv.h): what an R process reaches through the .Call shim.
//
//  * Residency.  run() calls the step functions back to back (R/inferCNV_ops.R:771, 817, 865, 911, 952, 1031,
//    1237-1309), each one handing over the matrix the previous one returned.  With icnv_residency(1) the library keeps
//    the matrices it uploaded or produced on the device(s) and recognises them when they come back -- by host address,
//    dimensions and a fingerprint of a strided sample of the values -- so that a step skips its upload.
//  * Several GPUs from one process.  icnv_set_devices(n) makes the host-buffer smoothing chain and per-cell Viterbi
//    split the cells into one contiguous block per device; one host thread per device uploads its block, runs the
//    *_dev path on its own stream and downloads.  The chain's reference statistics (per-gene sums of the
//    reference groups for steps 8 and 12, four scalars for step 22) meet on the host: every worker downloads its
//    partial (160 KB), all of them add the partials in device order -- the same sum everywhere, deterministic -- and
//    upload the total.  No data-path exchange between devices.
#include <algorithm>
#include <atomic>
#include <condition_variable>
#include <cstdlib>
#include <cstring>
#include <list>
#include <map>
#include <mutex>
#include <string>
#include <thread>
#include <vector>

#include "icnv_internal.h"

namespace icnv {

int pool_domain();   // api.hip: device ordinal * 256 + pool partition of the calling thread

// ------------------------------------------------------------------ residency
namespace {
struct ResidentEntry {
    const void *host;
    int64_t n;         // doubles
    uint64_t fp;
    DevBuf buf;
    uint64_t stamp;
    int busy;
};
struct ResidentDomain {
    std::mutex mu;
    std::list<ResidentEntry> entries;
    size_t bytes = 0;
};
std::mutex g_res_mu;
std::map<int, ResidentDomain *> g_res;
std::atomic<int> g_res_on{0};
std::atomic<int64_t> g_res_hits{0}, g_res_misses{0};
std::atomic<uint64_t> g_res_clock{1};

ResidentDomain &res_domain() {
    const int d = pool_domain();
    std::lock_guard<std::mutex> lk(g_res_mu);
    ResidentDomain *&p = g_res[d];
    if (!p) p = new ResidentDomain();
    return *p;
}
size_t res_budget_bytes() {
    static size_t b = 0;
    if (!b) {
        double gb = 64.0;
        if (const char *e = std::getenv("ICNV_RESIDENT_MAX_GB")) gb = std::atof(e);
        if (!(gb > 0.0)) gb = 64.0;
        b = (size_t)(gb * 1073741824.0);
    }
    return b;
}
// FNV-1a over ~16 k words spread over the block, its length and both ends
uint64_t fingerprint(const double *x, int64_t n) {
    uint64_t h = 1469598103934665603ull;
    auto mix = [&](uint64_t v) {
        for (int i = 0; i < 8; ++i) { h ^= (v >> (8 * i)) & 0xffu; h *= 1099511628211ull; }
    };
    mix((uint64_t)n);
    if (n <= 0) return h;
    const int64_t step = std::max<int64_t>(1, n / 16384);
    const uint64_t *w = reinterpret_cast<const uint64_t *>(x);
    for (int64_t i = 0; i < n; i += step) mix(w[i]);
    mix(w[n - 1]);
    return h;
}
void res_evict(ResidentDomain &d, size_t budget, size_t max_entries) {   // caller holds d.mu
    while (d.bytes > budget || d.entries.size() > max_entries) {
        auto victim = d.entries.end();
        for (auto it = d.entries.begin(); it != d.entries.end(); ++it)
            if (it->busy == 0 && (victim == d.entries.end() || it->stamp < victim->stamp)) victim = it;
        if (victim == d.entries.end()) return;
        d.bytes -= (size_t)victim->n * sizeof(double);
        d.entries.erase(victim);
    }
}
}  // namespace

MatrixLease::~MatrixLease() {
    if (entry) {
        ResidentDomain &d = res_domain();
        std::lock_guard<std::mutex> lk(d.mu);
        --static_cast<ResidentEntry *>(entry)->busy;
    }
}

int acquire_input(const double *host, int64_t n, hipStream_t s, MatrixLease &lease) {
    if (g_res_on.load()) {
        const uint64_t fp = fingerprint(host, n);
        ResidentDomain &d = res_domain();
        {
            std::lock_guard<std::mutex> lk(d.mu);
            for (auto &e : d.entries)
                if (e.host == host && e.n == n && e.fp == fp) {
                    ++e.busy;
                    e.stamp = g_res_clock++;
                    lease.entry = &e;
                    lease.dev = e.buf.as<double>();
                    ++g_res_hits;
                    return ICNV_OK;
                }
        }
        ++g_res_misses;
        DevBuf b;
        int rc = b.alloc((size_t)std::max<int64_t>(n, 1) * sizeof(double));
        if (rc) return rc;
        if (n) ICNV_HIP(hipMemcpyAsync(b.p, host, (size_t)n * sizeof(double), hipMemcpyHostToDevice, s));
        // the uploaded matrix stays resident too (the HMM input is read by the Viterbi and again by the median filter)
        std::lock_guard<std::mutex> lk(d.mu);
        for (auto it = d.entries.begin(); it != d.entries.end();)   // the address now holds other data
            if (it->host == host && it->busy == 0) { d.bytes -= (size_t)it->n * sizeof(double); it = d.entries.erase(it); }
            else ++it;
        d.entries.push_back(ResidentEntry{host, n, fp, std::move(b), g_res_clock++, 1});
        d.bytes += (size_t)n * sizeof(double);
        lease.entry = &d.entries.back();
        lease.dev = d.entries.back().buf.as<double>();
        res_evict(d, res_budget_bytes(), 6);
        return ICNV_OK;
    }
    int rc = lease.own.alloc((size_t)std::max<int64_t>(n, 1) * sizeof(double));
    if (rc) return rc;
    if (n) ICNV_HIP(hipMemcpyAsync(lease.own.p, host, (size_t)n * sizeof(double), hipMemcpyHostToDevice, s));
    lease.dev = lease.own.as<double>();
    return ICNV_OK;
}

// `host` has just received a copy of `buf` (download complete): remember the pair
void publish_output(const double *host, int64_t n, DevBuf &&buf) {
    if (!g_res_on.load() || n <= 0) return;   // (buf returns to the pool with the caller's DevBuf)
    const uint64_t fp = fingerprint(host, n);
    ResidentDomain &d = res_domain();
    std::lock_guard<std::mutex> lk(d.mu);
    for (auto it = d.entries.begin(); it != d.entries.end();)
        if (it->host == host && it->busy == 0) { d.bytes -= (size_t)it->n * sizeof(double); it = d.entries.erase(it); }
        else ++it;
    d.entries.push_back(ResidentEntry{host, n, fp, std::move(buf), g_res_clock++, 0});
    d.bytes += (size_t)n * sizeof(double);
    res_evict(d, res_budget_bytes(), 6);
}

// ------------------------------------------------------------------ devices of the host-buffer path
namespace {
std::atomic<int> g_ndev{1};
int fake_devices() {
    static int f = -1;
    if (f < 0) {
        const char *e = std::getenv("ICNV_FAKE_DEVICES");   // developer / test switch: n logical devices on the current GPU
        f = e ? std::max(0, std::atoi(e)) : 0;
    }
    return f;
}

struct Rendezvous {   // barrier for the worker threads; a failed worker keeps arriving so that nobody waits for ever
    explicit Rendezvous(int n) : n(n) {}
    void wait() {
        std::unique_lock<std::mutex> lk(mu);
        const int gen = generation;
        if (++count == n) { count = 0; ++generation; cv.notify_all(); }
        else cv.wait(lk, [&] { return gen != generation; });
    }
    std::mutex mu;
    std::condition_variable cv;
    int n, count = 0, generation = 0;
};

void shard(int64_t C, int nd, int w, int64_t &c0, int64_t &c1) {   // contiguous blocks, the first C % nd one cell longer
    const int64_t base = C / nd, rem = C % nd;
    c0 = w * base + std::min<int64_t>(w, rem);
    c1 = c0 + base + (w < rem ? 1 : 0);
}

// runs body(w, stream) on nd worker threads, worker w bound to its device; returns the first error
template <class Body>
int on_devices(int nd, Body body) {
    int home = 0;
    ICNV_HIP(hipGetDevice(&home));
    const bool fake = fake_devices() > 0;
    std::vector<int> rcs((size_t)nd, ICNV_OK);
    std::vector<std::string> msgs((size_t)nd);
    std::vector<std::thread> th;
    for (int w = 0; w < nd; ++w)
        th.emplace_back([&, w] {
            int rc = ICNV_OK;
            hipStream_t s = nullptr;
            if (hipSetDevice(fake ? home : w) != hipSuccess) rc = ICNV_ERR_HIP;
            set_pool_part(fake ? w + 1 : 0);
            if (!rc && hipStreamCreateWithFlags(&s, hipStreamNonBlocking) != hipSuccess) rc = ICNV_ERR_HIP;
            if (rc) set_error("cannot bind worker " + std::to_string(w) + " to its device");
            rc = body(w, s, rc);
            if (s) { (void)hipStreamSynchronize(s); (void)hipStreamDestroy(s); }
            rcs[(size_t)w] = rc;
            if (rc) msgs[(size_t)w] = icnv_last_error();
        });
    for (auto &t : th) t.join();
    for (int w = 0; w < nd; ++w)
        if (rcs[(size_t)w]) { set_error("device " + std::to_string(w) + ": " + msgs[(size_t)w]); return rcs[(size_t)w]; }
    return ICNV_OK;
}
}  // namespace

}  // namespace icnv

using namespace icnv;

extern "C" {

int icnv_residency(int on) {
    g_res_on.store(on ? 1 : 0);
    if (!on) icnv_residency_drop();
    return ICNV_OK;
}

void icnv_residency_drop(void) {
    std::lock_guard<std::mutex> lk(g_res_mu);
    for (auto &kv : g_res) {
        std::lock_guard<std::mutex> lk2(kv.second->mu);
        for (auto it = kv.second->entries.begin(); it != kv.second->entries.end();)
            if (it->busy == 0) { kv.second->bytes -= (size_t)it->n * sizeof(double); it = kv.second->entries.erase(it); }
            else ++it;
    }
}

int icnv_residency_stats(int64_t *out4) {
    if (!out4) ICNV_FAIL(ICNV_ERR_ARG, "null argument");
    int64_t bytes = 0, entries = 0;
    {
        std::lock_guard<std::mutex> lk(g_res_mu);
        for (auto &kv : g_res) {
            std::lock_guard<std::mutex> lk2(kv.second->mu);
            bytes += (int64_t)kv.second->bytes;
            entries += (int64_t)kv.second->entries.size();
        }
    }
    out4[0] = g_res_hits.load();
    out4[1] = g_res_misses.load();
    out4[2] = bytes;
    out4[3] = entries;
    return ICNV_OK;
}

int icnv_set_devices(int n_devices) {
    int n = 0;
    ICNV_HIP(hipGetDeviceCount(&n));
    if (n <= 0) ICNV_FAIL(ICNV_ERR_HIP, "no HIP device visible");
    const int avail = fake_devices() > 0 ? fake_devices() : n;
    if (n_devices < 0) ICNV_FAIL(ICNV_ERR_ARG, "n_devices must be >= 0 (0 = every visible device)");
    if (n_devices == 0) n_devices = avail;
    if (n_devices > avail) ICNV_FAIL(ICNV_ERR_ARG, "more devices requested than are visible");
    if (fake_devices() == 0)
        for (int d = 0; d < n_devices; ++d) {
            hipDeviceProp_t prop;
            ICNV_HIP(hipGetDeviceProperties(&prop, d));
            if (std::string(prop.gcnArchName).rfind("gfx950", 0) != 0)
                ICNV_FAIL(ICNV_ERR_UNSUPPORTED, std::string("libicnv_hip is built for gfx950 only, device ") + std::to_string(d) +
                                                    " is " + prop.gcnArchName);
        }
    g_ndev.store(n_devices);
    return ICNV_OK;
}

int icnv_get_devices(void) { return g_ndev.load(); }

// ------------------------------------------------------------------ smoothing chain, host buffers
int icnv_smooth_chain(const double *expr_in, double *expr_out, double *pre_denoise, const icnv_chain_cfg *cfg) {
    if (!expr_in || !expr_out || !cfg) ICNV_FAIL(ICNV_ERR_ARG, "null argument");
    if (cfg->G < 1 || cfg->C < 0) ICNV_FAIL(ICNV_ERR_ARG, "bad matrix dimensions");
    const int64_t G = cfg->G, C = cfg->C;
    const int nd = (int)std::max<int64_t>(1, std::min<int64_t>(g_ndev.load(), C));
    if (nd == 1) {
        const int64_t n = G * C;
        MatrixLease in;
        DevBuf dout, dpre;
        int rc;
        if ((rc = acquire_input(expr_in, n, nullptr, in))) return rc;
        if ((rc = dout.alloc((size_t)std::max<int64_t>(n, 1) * sizeof(double)))) return rc;
        if (pre_denoise && (rc = dpre.alloc((size_t)std::max<int64_t>(n, 1) * sizeof(double)))) return rc;
        rc = icnv_smooth_chain_dev(in.dev, dout.as<double>(), pre_denoise ? dpre.as<double>() : nullptr, cfg, nullptr);
        if (rc) return rc;
        ICNV_HIP(hipMemcpy(expr_out, dout.p, (size_t)n * sizeof(double), hipMemcpyDeviceToHost));
        if (pre_denoise) ICNV_HIP(hipMemcpy(pre_denoise, dpre.p, (size_t)n * sizeof(double), hipMemcpyDeviceToHost));
        publish_output(expr_out, n, std::move(dout));
        if (pre_denoise) publish_output(pre_denoise, n, std::move(dpre));
        return ICNV_OK;
    }
    // ---- one contiguous block of cells per device
    {   // argument checks once, with the caller's error reporting (a plan on the whole matrix, nothing is launched)
        icnv_chain_t *probe = nullptr;
        int rc = icnv_chain_begin(&probe, cfg);
        if (rc) return rc;
        const int rounds = icnv_chain_num_rounds(probe);
        icnv_chain_end(probe);
        if (rounds > 0)
            for (int q = 0; q < cfg->n_ref_grp; ++q)
                if (cfg->ref_off[q + 1] == cfg->ref_off[q]) ICNV_FAIL(ICNV_ERR_ARG, "empty reference group");
    }
    std::vector<std::vector<std::vector<double>>> part(4, std::vector<std::vector<double>>((size_t)nd));   // [round][worker]
    Rendezvous meet(nd);
    return on_devices(nd, [&](int w, hipStream_t s, int rc0) -> int {
        int rc = rc0;
        int64_t c0, c1;
        shard(C, nd, w, c0, c1);
        const int64_t n = G * (c1 - c0);
        // this block's reference cells, local indices, order kept
        std::vector<int32_t> ridx, roff(1, 0);
        for (int q = 0; q < cfg->n_ref_grp; ++q) {
            for (int32_t i = cfg->ref_off[q]; i < cfg->ref_off[q + 1]; ++i)
                if (cfg->ref_idx[i] >= c0 && cfg->ref_idx[i] < c1) ridx.push_back((int32_t)(cfg->ref_idx[i] - c0));
            roff.push_back((int32_t)ridx.size());
        }
        icnv_chain_cfg lc = *cfg;
        lc.C = c1 - c0;
        lc.ref_idx = ridx.data();
        lc.ref_off = roff.data();
        icnv_chain_t *ch = nullptr;
        MatrixLease in;
        DevBuf dout, dpre;
        if (!rc) rc = icnv_chain_begin(&ch, &lc);
        if (!rc) rc = acquire_input(expr_in + c0 * G, n, s, in);
        if (!rc) rc = dout.alloc((size_t)std::max<int64_t>(n, 1) * sizeof(double));
        if (!rc && pre_denoise) rc = dpre.alloc((size_t)std::max<int64_t>(n, 1) * sizeof(double));
        const int rounds = ch ? icnv_chain_num_rounds(ch) : 0;
        // every worker walks through the same number of meeting points, failed or not
        int total_rounds = 0;
        {
            icnv_chain_cfg tmp = *cfg;
            const uint32_t m = tmp.stage_mask;
            total_rounds = ((m & ICNV_ST_SUBTRACT_REF_1) ? 1 : 0) + ((m & ICNV_ST_SUBTRACT_REF_2) ? 1 : 0) + ((m & ICNV_ST_DENOISE) ? 1 : 0);
            if ((m & ICNV_ST_DENOISE) && tmp.noise_filter == 0.0) --total_rounds;   // clear_noise(threshold = 0): no stage (icnv_chain_begin)
        }
        (void)rounds;
        for (int r = 0; r < total_rounds; ++r) {
            double *pd = nullptr;
            int64_t pn = 0;
            if (!rc) rc = icnv_chain_round_partial_dev(ch, r, in.dev, &pd, &pn, s);
            if (!rc) {
                part[(size_t)r][(size_t)w].resize((size_t)pn);
                if (hipMemcpyAsync(part[(size_t)r][(size_t)w].data(), pd, (size_t)pn * sizeof(double), hipMemcpyDeviceToHost, s) != hipSuccess ||
                    hipStreamSynchronize(s) != hipSuccess) {
                    set_error("copy of the reference statistics failed");
                    rc = ICNV_ERR_HIP;
                }
            }
            if (rc) part[(size_t)r][(size_t)w].clear();
            meet.wait();
            if (!rc) {
                std::vector<double> tot((size_t)pn, 0.0);
                for (int v = 0; v < nd && !rc; ++v) {   // device order: the same sum on every device
                    const std::vector<double> &p = part[(size_t)r][(size_t)v];
                    if ((int64_t)p.size() != pn) { set_error("another device failed"); rc = ICNV_ERR_HIP; break; }
                    for (int64_t i = 0; i < pn; ++i) tot[(size_t)i] += p[(size_t)i];
                }
                if (!rc && (hipMemcpyAsync(pd, tot.data(), (size_t)pn * sizeof(double), hipMemcpyHostToDevice, s) != hipSuccess ||
                            hipStreamSynchronize(s) != hipSuccess)) {
                    set_error("upload of the reference statistics failed");
                    rc = ICNV_ERR_HIP;
                }
                if (!rc) rc = icnv_chain_round_finish_dev(ch, r, s);
            }
        }
        if (!rc) rc = icnv_chain_apply_dev(ch, in.dev, dout.as<double>(), pre_denoise ? dpre.as<double>() : nullptr, s);
        if (!rc && n > 0) {
            hipError_t e = hipMemcpyAsync(expr_out + c0 * G, dout.p, (size_t)n * sizeof(double), hipMemcpyDeviceToHost, s);
            if (e == hipSuccess && pre_denoise)
                e = hipMemcpyAsync(pre_denoise + c0 * G, dpre.p, (size_t)n * sizeof(double), hipMemcpyDeviceToHost, s);
            if (e == hipSuccess) e = hipStreamSynchronize(s);
            if (e != hipSuccess) rc = hip_fail(e, "download", __FILE__, __LINE__);
        }
        if (ch) { (void)hipStreamSynchronize(s); icnv_chain_end(ch); }
        if (!rc) {
            publish_output(expr_out + c0 * G, n, std::move(dout));
            if (pre_denoise) publish_output(pre_denoise + c0 * G, n, std::move(dpre));
        }
        return rc;
    });
}

// ------------------------------------------------------------------ per-cell Viterbi, host buffers
int icnv_viterbi_cells(const double *expr, uint8_t *states, int64_t G, int64_t C, const int32_t *chr_start, int32_t n_chr,
                       int32_t K, const double *mean, double sd_shared, const double *logPi, const double *logDelta) {
    if (!expr || !states || G < 1 || C < 0) ICNV_FAIL(ICNV_ERR_ARG, "bad argument");
    const int nd = (int)std::max<int64_t>(1, std::min<int64_t>(g_ndev.load(), C));
    std::atomic<int64_t> bad_total{0};
    auto block = [&](int64_t c0, int64_t c1, hipStream_t s) -> int {
        const int64_t n = G * (c1 - c0);
        MatrixLease in;
        DevBuf ds, dn;
        int rc;
        if ((rc = acquire_input(expr + c0 * G, n, s, in))) return rc;
        if ((rc = ds.alloc((size_t)std::max<int64_t>(n, 1))) || (rc = dn.alloc(sizeof(int32_t)))) return rc;
        ICNV_HIP(hipMemsetAsync(dn.p, 0, sizeof(int32_t), s));
        rc = icnv_viterbi_cells_dev(in.dev, ds.as<uint8_t>(), G, c1 - c0, chr_start, n_chr, K, mean, sd_shared, logPi, logDelta,
                                    dn.as<int32_t>(), s);
        if (rc) return rc;
        int32_t bad = 0;
        if (n) ICNV_HIP(hipMemcpyAsync(states + c0 * G, ds.p, (size_t)n, hipMemcpyDeviceToHost, s));
        ICNV_HIP(hipMemcpyAsync(&bad, dn.p, sizeof(int32_t), hipMemcpyDeviceToHost, s));
        ICNV_HIP(hipStreamSynchronize(s));
        bad_total += bad;
        return ICNV_OK;
    };
    int rc;
    if (nd == 1) rc = block(0, C, nullptr);
    else
        rc = on_devices(nd, [&](int w, hipStream_t s, int rc0) -> int {
            if (rc0) return rc0;
            int64_t c0, c1;
            shard(C, nd, w, c0, c1);
            return block(c0, c1, s);
        });
    if (rc) return rc;
    if (bad_total.load())
        ICNV_FAIL(ICNV_ERR_UNDERFLOW, "Problems With Underflow in " + std::to_string(bad_total.load()) + " sequences");
    return ICNV_OK;
}

}  // extern "C"
