// Host-buffer side of the C ABI (include/icnv.h): what an R process reaches through the .Call shim.
//
//  * Residency.  run() calls the step functions back to back (R/inferCNV_ops.R:771, 817, 865, 911, 952, 1031,
//    1237-1309), each one handing over the matrix the previous one returned.  With icnv_residency(1) the library keeps
//    the matrices it uploaded or produced on the device(s) and recognises them when they come back -- by CONTENT: the
//    length, a strided sample as the quick reject, and then a 64-bit hash of EVERY value (computed by the host's cores
//    for the incoming matrix, by a device reduction for a matrix the library produced) -- so that a step skips its
//    upload.  Addresses play no part: R's allocator reuses them, and an edit of a single element changes the hash.
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

#include <chrono>

namespace icnv {

int pool_domain();   // api.hip: device ordinal * 256 + pool partition of the calling thread

// ------------------------------------------------------------------ where a host-buffer call spends its time
namespace {
struct HostPathStats {
    std::mutex mu;
    double v[11] = {0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0};
    void add(int i, double x) { std::lock_guard<std::mutex> lk(mu); v[i] += x; }
    void set(int i, double x) { std::lock_guard<std::mutex> lk(mu); v[i] = x; }
};
HostPathStats g_hp;
enum { HP_CALLS = 0, HP_FP_MS, HP_HASH_MS, HP_HASH_THREADS, HP_H2D_MS, HP_H2D_BYTES, HP_D2H_MS, HP_D2H_BYTES, HP_DEV_MS, HP_PIPELINED, HP_WALL_MS };
struct WallTimer {   // adds the elapsed wall time to one slot when it goes out of scope
    int slot;
    std::chrono::steady_clock::time_point t0 = std::chrono::steady_clock::now();
    explicit WallTimer(int s) : slot(s) {}
    double ms() const { return std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now() - t0).count(); }
    ~WallTimer() { if (slot >= 0) g_hp.add(slot, ms()); }
};
// synchronous copies with their time and bytes on the books
int timed_h2d(void *dst, const void *src, size_t bytes, hipStream_t s) {
    if (!bytes) return ICNV_OK;
    WallTimer t(HP_H2D_MS);
    ICNV_HIP(hipMemcpyAsync(dst, src, bytes, hipMemcpyHostToDevice, s));
    ICNV_HIP(hipStreamSynchronize(s));
    g_hp.add(HP_H2D_BYTES, (double)bytes);
    return ICNV_OK;
}
int timed_d2h(void *dst, const void *src, size_t bytes, hipStream_t s) {
    if (!bytes) return ICNV_OK;
    WallTimer t(HP_D2H_MS);
    ICNV_HIP(hipMemcpyAsync(dst, src, bytes, hipMemcpyDeviceToHost, s));
    ICNV_HIP(hipStreamSynchronize(s));
    g_hp.add(HP_D2H_BYTES, (double)bytes);
    return ICNV_OK;
}
}  // namespace

// ------------------------------------------------------------------ residency
namespace {
struct ResidentEntry {
    const void *host;  // where the content was last seen (statistics / debugging only: identity is by content)
    int64_t n;         // doubles
    uint64_t fp;       // strided sample: quick reject
    DevBuf buf;
    uint64_t stamp;
    int busy;
    uint64_t hash = 0; // content hash of all n values
    bool hash_valid = false;
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
// ---- content hash over the 64-bit words w_i of the matrix, in blocks of eight words (one 64-byte line):
//        b_q = sum_j (w_{8q+j} + j PHI) K_j  (mod 2^64, K_j odd),   H = fmix64( n ^ sum_q fmix64(b_q + (q + 1) PHI) ).
// An edit of one word moves its block's b_q by delta * K_j != 0 (K_j is odd: multiplication by it is a bijection), hence the
// block's mix, hence H; the outer sum is commutative, so any partition of the BLOCKS (host threads, device threads) yields
// the same value.  One multiply-add per word and one fmix64 per line: round 3's hash mixed every word (two multiplies and
// three xor-shifts each) and cost the host more time than the PCIe upload it saves on a 16-core quota.
inline __host__ __device__ uint64_t fmix64(uint64_t k) {
    k ^= k >> 33;
    k *= 0xff51afd7ed558ccdull;
    k ^= k >> 33;
    k *= 0xc4ceb9fe1a85ec53ull;
    k ^= k >> 33;
    return k;
}
constexpr uint64_t HASH_PHI = 0x9E3779B97F4A7C15ull;
#define ICNV_HASH_K0 0x9E3779B97F4A7C15ull
#define ICNV_HASH_K1 0xBF58476D1CE4E5B9ull
#define ICNV_HASH_K2 0x94D049BB133111EBull
#define ICNV_HASH_K3 0xD6E8FEB86659FD93ull
#define ICNV_HASH_K4 0xFF51AFD7ED558CCDull
#define ICNV_HASH_K5 0xC4CEB9FE1A85EC53ull
#define ICNV_HASH_K6 0xD1B54A32D192ED03ull
#define ICNV_HASH_K7 0x8CB92BA72F3D8DD7ull
inline __host__ __device__ uint64_t hash_block8(const uint64_t *w, uint64_t q) {   // a whole block: words 8q .. 8q+7
    uint64_t b = (w[0] + 0 * HASH_PHI) * ICNV_HASH_K0;
    b += (w[1] + 1 * HASH_PHI) * ICNV_HASH_K1;
    b += (w[2] + 2 * HASH_PHI) * ICNV_HASH_K2;
    b += (w[3] + 3 * HASH_PHI) * ICNV_HASH_K3;
    b += (w[4] + 4 * HASH_PHI) * ICNV_HASH_K4;
    b += (w[5] + 5 * HASH_PHI) * ICNV_HASH_K5;
    b += (w[6] + 6 * HASH_PHI) * ICNV_HASH_K6;
    b += (w[7] + 7 * HASH_PHI) * ICNV_HASH_K7;
    return fmix64(b + (q + 1) * HASH_PHI);
}
inline __host__ __device__ uint64_t hash_block_tail(const uint64_t *w, int m, uint64_t q) {   // the last block's m < 8 words
    const uint64_t K[8] = {ICNV_HASH_K0, ICNV_HASH_K1, ICNV_HASH_K2, ICNV_HASH_K3, ICNV_HASH_K4, ICNV_HASH_K5, ICNV_HASH_K6, ICNV_HASH_K7};
    uint64_t b = 0;
    for (int j = 0; j < m; ++j) b += (w[j] + (uint64_t)j * HASH_PHI) * K[j];
    return fmix64(b + (q + 1) * HASH_PHI);
}
__global__ void __launch_bounds__(256) content_hash_kernel(const uint64_t *__restrict__ w, int64_t n, unsigned long long *out) {
    uint64_t acc = 0;
    const int64_t nb = n >> 3;
    for (int64_t q = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; q < nb; q += (int64_t)gridDim.x * blockDim.x) {
        uint64_t v[8];
        const ulonglong2 *p = reinterpret_cast<const ulonglong2 *>(w + 8 * q);   // the matrix is 16-byte aligned (pool allocation)
#pragma unroll
        for (int j = 0; j < 4; ++j) { const ulonglong2 t = p[j]; v[2 * j] = t.x; v[2 * j + 1] = t.y; }
        acc += hash_block8(v, (uint64_t)q);
    }
    if (blockIdx.x == 0 && threadIdx.x == 0 && (n & 7)) acc += hash_block_tail(w + 8 * nb, (int)(n & 7), (uint64_t)nb);
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) acc += __shfl_xor((unsigned long long)acc, o, 64);
    if ((threadIdx.x & 63) == 0) atomicAdd(out, (unsigned long long)acc);
}
// threads the host may keep busy: the hardware's, capped by the container's CPU quota (cgroup v2 cpu.max, v1 cfs quota) --
// more runnable threads than the quota allows are throttled, not run
static int host_hash_threads() {
    if (const char *e = std::getenv("ICNV_HASH_THREADS")) return std::max(1, std::atoi(e));
    int nt = (int)std::max(1u, std::thread::hardware_concurrency());
    double quota = 0.0;
    if (FILE *f = std::fopen("/sys/fs/cgroup/cpu.max", "r")) {
        char q[64];
        double per = 0.0;
        if (std::fscanf(f, "%63s %lf", q, &per) == 2 && std::strcmp(q, "max") != 0 && per > 0.0) quota = std::atof(q) / per;
        std::fclose(f);
    } else {
        double q = 0.0, per = 0.0;
        if (FILE *g = std::fopen("/sys/fs/cgroup/cpu/cpu.cfs_quota_us", "r")) { if (std::fscanf(g, "%lf", &q) != 1) q = 0.0; std::fclose(g); }
        if (FILE *g = std::fopen("/sys/fs/cgroup/cpu/cpu.cfs_period_us", "r")) { if (std::fscanf(g, "%lf", &per) != 1) per = 0.0; std::fclose(g); }
        if (q > 0.0 && per > 0.0) quota = q / per;
    }
    if (quota >= 1.0) nt = std::min(nt, (int)(quota + 0.5));
    return std::min(nt, 32);
}
uint64_t content_hash_host(const double *x, int64_t n) {
    if (n <= 0) return fmix64(0);
    const uint64_t *w = reinterpret_cast<const uint64_t *>(x);
    const int64_t nb = n >> 3;
    static const int nt_max = host_hash_threads();
    int nt = (int)std::max<int64_t>(1, std::min<int64_t>(nt_max, (nb + (1 << 17) - 1) >> 17));   // at least 8 MiB per thread
    std::vector<uint64_t> part((size_t)nt, 0);
    g_hp.set(HP_HASH_THREADS, (double)nt);
    auto work = [&](int t) {
        const int64_t b = nb * t / nt, e = nb * (t + 1) / nt;
        uint64_t a0 = 0, a1 = 0;
        int64_t q = b;
        for (; q + 2 <= e; q += 2) {
            a0 += hash_block8(w + 8 * q, (uint64_t)q);
            a1 += hash_block8(w + 8 * q + 8, (uint64_t)q + 1);
        }
        for (; q < e; ++q) a0 += hash_block8(w + 8 * q, (uint64_t)q);
        part[(size_t)t] = a0 + a1;
    };
    std::vector<std::thread> th;
    for (int t = 1; t < nt; ++t) th.emplace_back(work, t);
    work(0);
    for (auto &t : th) t.join();
    uint64_t sum = 0;
    for (uint64_t v : part) sum += v;
    if (n & 7) sum += hash_block_tail(w + 8 * nb, (int)(n & 7), (uint64_t)nb);
    return fmix64(sum ^ (uint64_t)n);
}
// the same value from the device copy (synchronous; the entry's content is complete: every host-buffer call ends with a
// synchronisation of the stream it used)
int content_hash_device(const double *dev, int64_t n, uint64_t *out) {
    if (n <= 0) { *out = fmix64(0); return ICNV_OK; }
    DevBuf acc;
    int rc = acc.alloc(sizeof(unsigned long long));
    if (rc) return rc;
    ICNV_HIP(hipMemsetAsync(acc.p, 0, sizeof(unsigned long long), nullptr));
    const int grid = (int)std::max<int64_t>(1, std::min<int64_t>(((n >> 3) + 255) / 256, (int64_t)num_cus() * 8));
    hipLaunchKernelGGL(content_hash_kernel, dim3(grid), dim3(256), 0, nullptr, reinterpret_cast<const uint64_t *>(dev), n,
                       acc.as<unsigned long long>());
    ICNV_HIP(hipGetLastError());
    unsigned long long sum = 0;
    ICNV_HIP(hipMemcpy(&sum, acc.p, sizeof(sum), hipMemcpyDeviceToHost));
    *out = fmix64((uint64_t)sum ^ (uint64_t)n);
    return ICNV_OK;
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
        uint64_t fp;
        { WallTimer t(HP_FP_MS); fp = fingerprint(host, n); }
        ResidentDomain &d = res_domain();
        uint64_t h = 0;
        bool have_h = false;
        // Candidates (same length, same strided sample) are pinned under the lock; the hashes -- a pass over the host
        // matrix, and for a resident matrix that has none yet a device kernel whose 8-byte accumulator comes from the pool --
        // are computed WITHOUT the lock: pool_alloc may call residency_release_idle() under memory pressure, which takes it.
        std::vector<ResidentEntry *> cands;
        {
            std::lock_guard<std::mutex> lk(d.mu);
            for (auto &e : d.entries)
                if (e.n == n && e.fp == fp) { ++e.busy; cands.push_back(&e); }
        }
        if (!cands.empty()) {
            // the sample agrees with a resident matrix: identity is decided by the hash of EVERY value
            { WallTimer t(HP_HASH_MS); h = content_hash_host(host, n); }
            have_h = true;
            int rc_hash = ICNV_OK;
            std::vector<uint64_t> dev_hash(cands.size(), 0);
            std::vector<char> fresh(cands.size(), 0);
            for (size_t i = 0; i < cands.size() && !rc_hash; ++i)
                if (!cands[i]->hash_valid) {   // (a pinned entry's buffer cannot go away; hash_valid only ever turns true)
                    rc_hash = content_hash_device(cands[i]->buf.as<double>(), cands[i]->n, &dev_hash[i]);
                    fresh[i] = 1;
                }
            std::lock_guard<std::mutex> lk(d.mu);
            ResidentEntry *hit = nullptr;
            for (size_t i = 0; i < cands.size(); ++i) {
                ResidentEntry &e = *cands[i];
                if (fresh[i] && !rc_hash) { e.hash = dev_hash[i]; e.hash_valid = true; }
                if (!hit && !rc_hash && e.hash_valid && e.hash == h) hit = &e;   // keeps its pin: the lease's
                else --e.busy;
            }
            if (rc_hash) return rc_hash;
            if (hit) {
                hit->stamp = g_res_clock++;
                hit->host = host;
                lease.entry = hit;
                lease.dev = hit->buf.as<double>();
                ++g_res_hits;
                return ICNV_OK;
            }
        }
        ++g_res_misses;
        DevBuf b;
        int rc = b.alloc((size_t)std::max<int64_t>(n, 1) * sizeof(double));
        if (rc) return rc;
        if (n) { WallTimer t(HP_H2D_MS); ICNV_HIP(hipMemcpyAsync(b.p, host, (size_t)n * sizeof(double), hipMemcpyHostToDevice, s)); g_hp.add(HP_H2D_BYTES, (double)n * 8.0); }
        // (a copy from pageable memory returns when the data is staged: its wall time is the upload's)
        // the uploaded matrix stays resident too (the HMM input is read by the Viterbi and again by the median filter);
        // its hash is taken from the device copy the first time a look-alike arrives
        std::lock_guard<std::mutex> lk(d.mu);
        d.entries.push_back(ResidentEntry{host, n, fp, std::move(b), g_res_clock++, 1, h, have_h});
        d.bytes += (size_t)n * sizeof(double);
        lease.entry = &d.entries.back();
        lease.dev = d.entries.back().buf.as<double>();
        res_evict(d, res_budget_bytes(), 6);
        return ICNV_OK;
    }
    int rc = lease.own.alloc((size_t)std::max<int64_t>(n, 1) * sizeof(double));
    if (rc) return rc;
    if (n) { WallTimer t(HP_H2D_MS); ICNV_HIP(hipMemcpyAsync(lease.own.p, host, (size_t)n * sizeof(double), hipMemcpyHostToDevice, s)); g_hp.add(HP_H2D_BYTES, (double)n * 8.0); }
    lease.dev = lease.own.as<double>();
    return ICNV_OK;
}

// `host` has just received a copy of `buf` (download complete): remember the pair
void publish_output(const double *host, int64_t n, DevBuf &&buf) {
    if (!g_res_on.load() || n <= 0) return;   // (buf returns to the pool with the caller's DevBuf)
    const uint64_t fp = fingerprint(host, n);
    ResidentDomain &d = res_domain();
    std::lock_guard<std::mutex> lk(d.mu);
    d.entries.push_back(ResidentEntry{host, n, fp, std::move(buf), g_res_clock++, 0, 0, false});
    d.bytes += (size_t)n * sizeof(double);
    res_evict(d, res_budget_bytes(), 6);
}

// Under memory pressure (pool_alloc failed): give the idle resident matrices of the calling thread's pool domain back
bool residency_release_idle() {
    if (!g_res_on.load()) return false;
    ResidentDomain &d = res_domain();
    std::lock_guard<std::mutex> lk(d.mu);
    const size_t before = d.entries.size();
    res_evict(d, 0, 0);
    return d.entries.size() != before;
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


// ------------------------------------------------------------------ upload | kernels | download over column blocks
// A copy from or to pageable memory (what R hands over) occupies the host thread that issues it, so the three stages of
// a host-buffer call run on three threads and three streams: the uploader copies block after block and records an event per
// block; the caller's thread waits for the blocks it needs, enqueues the kernels of a block behind its upload event and records
// an event per finished block; the downloader copies a block back as soon as its event is recorded.  PCIe is full duplex:
// the downloads of the early blocks overlap the uploads of the late ones.
namespace {
bool host_pipeline_enabled() {
    static int v = -1;
    if (v < 0) {
        const char *e = std::getenv("ICNV_HOST_PIPELINE");
        v = (e && e[0] == '0') ? 0 : 1;
    }
    return v != 0;
}

struct HostPipeline {
    int nb = 0;
    hipStream_t s_up = nullptr, s_c = nullptr, s_dn = nullptr;
    std::vector<hipEvent_t> ev_up, ev_c;
    std::mutex mu;
    std::condition_variable cv;
    int up_done = 0, c_done = 0;
    int rc = ICNV_OK;
    std::string msg;
    int home = 0;

    int open(int n_blocks) {
        nb = n_blocks;
        ICNV_HIP(hipGetDevice(&home));
        ICNV_HIP(hipStreamCreateWithFlags(&s_up, hipStreamNonBlocking));
        ICNV_HIP(hipStreamCreateWithFlags(&s_c, hipStreamNonBlocking));
        ICNV_HIP(hipStreamCreateWithFlags(&s_dn, hipStreamNonBlocking));
        ev_up.assign((size_t)nb, nullptr);
        ev_c.assign((size_t)nb, nullptr);
        for (int i = 0; i < nb; ++i) {
            ICNV_HIP(hipEventCreateWithFlags(&ev_up[(size_t)i], hipEventDisableTiming));
            ICNV_HIP(hipEventCreateWithFlags(&ev_c[(size_t)i], hipEventDisableTiming));
        }
        return ICNV_OK;
    }
    ~HostPipeline() {
        for (hipEvent_t e : ev_up) if (e) (void)hipEventDestroy(e);
        for (hipEvent_t e : ev_c) if (e) (void)hipEventDestroy(e);
        if (s_up) { (void)hipStreamSynchronize(s_up); (void)hipStreamDestroy(s_up); }
        if (s_c) { (void)hipStreamSynchronize(s_c); (void)hipStreamDestroy(s_c); }
        if (s_dn) { (void)hipStreamSynchronize(s_dn); (void)hipStreamDestroy(s_dn); }
    }
    void fail(int code, const std::string &m) {   // first error wins; everybody is released
        std::lock_guard<std::mutex> lk(mu);
        if (rc == ICNV_OK) { rc = code; msg = m; }
        up_done = nb;
        c_done = nb;
        cv.notify_all();
    }
    bool failed() { std::lock_guard<std::mutex> lk(mu); return rc != ICNV_OK; }

    // up(i, stream) / down(i, stream): the copies of block i (hipError_t); first(stream): once, behind the uploads of the blocks
    // [0, n_first); compute(i, stream): the kernels of block i (library return code)
    template <class Up, class First, class Compute, class Down>
    int run(int n_first, Up up, First first, Compute compute, Down down) {
        std::thread uploader([&] {
            if (hipSetDevice(home) != hipSuccess) { fail(ICNV_ERR_HIP, "uploader: hipSetDevice failed"); return; }
            WallTimer t(HP_H2D_MS);
            for (int i = 0; i < nb && !failed(); ++i) {
                hipError_t e = up(i, s_up);
                if (e == hipSuccess) e = hipEventRecord(ev_up[(size_t)i], s_up);
                if (e != hipSuccess) { (void)hipGetLastError(); fail(ICNV_ERR_HIP, std::string("upload failed: ") + hipGetErrorString(e)); return; }
                std::lock_guard<std::mutex> lk(mu);
                up_done = std::max(up_done, i + 1);
                cv.notify_all();
            }
            (void)hipStreamSynchronize(s_up);
        });
        std::thread downloader([&] {
            if (hipSetDevice(home) != hipSuccess) { fail(ICNV_ERR_HIP, "downloader: hipSetDevice failed"); return; }
            double busy = 0.0;
            for (int i = 0; i < nb; ++i) {
                {
                    std::unique_lock<std::mutex> lk(mu);
                    cv.wait(lk, [&] { return c_done > i || rc != ICNV_OK; });
                    if (rc != ICNV_OK) break;
                }
                WallTimer t(-1);
                hipError_t e = hipStreamWaitEvent(s_dn, ev_c[(size_t)i], 0);
                if (e == hipSuccess) e = down(i, s_dn);
                if (e != hipSuccess) { (void)hipGetLastError(); fail(ICNV_ERR_HIP, std::string("download failed: ") + hipGetErrorString(e)); break; }
                busy += t.ms();
            }
            { WallTimer t(-1); (void)hipStreamSynchronize(s_dn); busy += t.ms(); }
            g_hp.add(HP_D2H_MS, busy);
        });
        auto wait_up = [&](int i) -> bool {   // block i uploaded (its event recorded)?  false: the pipeline failed
            std::unique_lock<std::mutex> lk(mu);
            cv.wait(lk, [&] { return up_done > i || rc != ICNV_OK; });
            return rc == ICNV_OK;
        };
        bool ok = true;
        for (int i = 0; i < n_first && ok; ++i) {
            ok = wait_up(i);
            if (ok && hipStreamWaitEvent(s_c, ev_up[(size_t)i], 0) != hipSuccess) { fail(ICNV_ERR_HIP, "hipStreamWaitEvent failed"); ok = false; }
        }
        if (ok) {
            const int r = first(s_c);
            if (r) { fail(r, icnv_last_error()); ok = false; }
        }
        for (int i = 0; i < nb && ok; ++i) {
            ok = wait_up(i);
            if (!ok) break;
            if (hipStreamWaitEvent(s_c, ev_up[(size_t)i], 0) != hipSuccess) { fail(ICNV_ERR_HIP, "hipStreamWaitEvent failed"); break; }
            const int r = compute(i, s_c);
            if (r) { fail(r, icnv_last_error()); break; }
            if (hipEventRecord(ev_c[(size_t)i], s_c) != hipSuccess) { fail(ICNV_ERR_HIP, "hipEventRecord failed"); break; }
            std::lock_guard<std::mutex> lk(mu);
            c_done = std::max(c_done, i + 1);
            cv.notify_all();
        }
        uploader.join();
        downloader.join();
        (void)hipStreamSynchronize(s_c);
        if (rc != ICNV_OK) { set_error(msg); return rc; }
        g_hp.add(HP_PIPELINED, 1.0);
        return ICNV_OK;
    }
};

// column blocks of ~128 MB (at least 2, at most 64); `first` = blocks that hold a marked cell come first, in order
void pipeline_blocks(int64_t G, int64_t C, const std::vector<char> *marked, std::vector<int64_t> &b0, std::vector<int64_t> &b1, int &n_first) {
    const int64_t bytes = G * C * 8;
    int64_t nb = std::max<int64_t>(2, std::min<int64_t>(64, (bytes + (128ll << 20) - 1) / (128ll << 20)));
    nb = std::min<int64_t>(nb, std::max<int64_t>(1, C / 64));
    nb = std::max<int64_t>(nb, 1);
    const int64_t per = (C + nb - 1) / nb;
    std::vector<int64_t> lo, hi;
    for (int64_t c = 0; c < C; c += per) { lo.push_back(c); hi.push_back(std::min(C, c + per)); }
    std::vector<char> has((size_t)lo.size(), 0);
    if (marked)
        for (size_t k = 0; k < lo.size(); ++k)
            for (int64_t c = lo[k]; c < hi[k] && !has[k]; ++c) has[k] = (*marked)[(size_t)c];
    b0.clear(); b1.clear();
    n_first = 0;
    for (size_t k = 0; k < lo.size(); ++k) if (has[k]) { b0.push_back(lo[k]); b1.push_back(hi[k]); ++n_first; }
    for (size_t k = 0; k < lo.size(); ++k) if (!has[k]) { b0.push_back(lo[k]); b1.push_back(hi[k]); }
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

int icnv_host_path_stats(double *out, int32_t n) {
    if (!out || n < 1) ICNV_FAIL(ICNV_ERR_ARG, "bad argument");
    std::lock_guard<std::mutex> lk(g_hp.mu);
    for (int i = 0; i < n && i < 11; ++i) out[i] = g_hp.v[i];
    if (n > 11) out[11] = pool_malloc_ms(false);
    return ICNV_OK;
}
void icnv_host_path_stats_reset(void) {
    std::lock_guard<std::mutex> lk(g_hp.mu);
    for (double &x : g_hp.v) x = 0.0;
    (void)pool_malloc_ms(true);
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
    WallTimer wall(HP_WALL_MS);
    g_hp.add(HP_CALLS, 1.0);
    if (nd == 1) {
        const int64_t n = G * C;
        // ---- one device, no residency: upload | kernels | download pipelined over column blocks
        if (!g_res_on.load() && host_pipeline_enabled() && n * 8 >= (64ll << 20) && C >= 128) {
            icnv_chain_t *ch = nullptr;
            int rc = icnv_chain_begin(&ch, cfg);
            if (rc) return rc;
            struct ChainGuard { icnv_chain_t *c; ~ChainGuard() { if (c) icnv_chain_end(c); } } guard{ch};
            const int rounds = icnv_chain_num_rounds(ch);
            for (int q = 0; q < cfg->n_ref_grp && rounds > 0; ++q)
                if (cfg->ref_off[q + 1] == cfg->ref_off[q]) ICNV_FAIL(ICNV_ERR_ARG, "empty reference group");
            if (chain_apply_columns(ch, expr_in, expr_out, pre_denoise, 0, 0, nullptr) != -1000) {
                std::vector<char> is_ref((size_t)C, 0);
                if (rounds > 0)
                    for (int32_t i = 0; i < cfg->ref_off[cfg->n_ref_grp]; ++i) is_ref[(size_t)cfg->ref_idx[i]] = 1;
                std::vector<int64_t> b0, b1;
                int n_first = 0;
                pipeline_blocks(G, C, rounds > 0 ? &is_ref : nullptr, b0, b1, n_first);
                DevBuf din, dout, dpre;
                const size_t bytes = (size_t)n * sizeof(double);
                if ((rc = din.alloc(bytes)) || (rc = dout.alloc(bytes)) || (pre_denoise && (rc = dpre.alloc(bytes)))) return rc;
                HostPipeline pipe;
                if ((rc = pipe.open((int)b0.size()))) return rc;
                const double *x = din.as<double>();
                double *o = dout.as<double>(), *p = pre_denoise ? dpre.as<double>() : nullptr;
                rc = pipe.run(
                    n_first,
                    [&](int i, hipStream_t s) {
                        const size_t off = (size_t)(b0[(size_t)i] * G), cnt = (size_t)((b1[(size_t)i] - b0[(size_t)i]) * G);
                        g_hp.add(HP_H2D_BYTES, (double)cnt * 8.0);
                        return hipMemcpyAsync(din.as<double>() + off, expr_in + off, cnt * sizeof(double), hipMemcpyHostToDevice, s);
                    },
                    [&](hipStream_t s) -> int {   // the reference rounds: every reference cell is on the device
                        int r = ICNV_OK;
                        for (int k = 0; k < rounds && !r; ++k) {
                            r = icnv_chain_round_partial_dev(ch, k, x, nullptr, nullptr, s);
                            if (!r) r = icnv_chain_round_finish_dev(ch, k, s);
                        }
                        return r;
                    },
                    [&](int i, hipStream_t s) -> int { return chain_apply_columns(ch, x, o, p, b0[(size_t)i], b1[(size_t)i], s); },
                    [&](int i, hipStream_t s) {
                        const size_t off = (size_t)(b0[(size_t)i] * G), cnt = (size_t)((b1[(size_t)i] - b0[(size_t)i]) * G);
                        g_hp.add(HP_D2H_BYTES, (double)cnt * 8.0 * (pre_denoise ? 2.0 : 1.0));
                        hipError_t e = hipMemcpyAsync(expr_out + off, o + off, cnt * sizeof(double), hipMemcpyDeviceToHost, s);
                        if (e == hipSuccess && pre_denoise) e = hipMemcpyAsync(pre_denoise + off, p + off, cnt * sizeof(double), hipMemcpyDeviceToHost, s);
                        return e;
                    });
                return rc;
            }
        }
        MatrixLease in;
        DevBuf dout, dpre;
        int rc;
        if ((rc = acquire_input(expr_in, n, nullptr, in))) return rc;
        if ((rc = dout.alloc((size_t)std::max<int64_t>(n, 1) * sizeof(double)))) return rc;
        if (pre_denoise && (rc = dpre.alloc((size_t)std::max<int64_t>(n, 1) * sizeof(double)))) return rc;
        {
            WallTimer t(HP_DEV_MS);
            rc = icnv_smooth_chain_dev(in.dev, dout.as<double>(), pre_denoise ? dpre.as<double>() : nullptr, cfg, nullptr);
            if (rc) return rc;
            ICNV_HIP(hipStreamSynchronize(nullptr));
        }
        if ((rc = timed_d2h(expr_out, dout.p, (size_t)n * sizeof(double), nullptr))) return rc;
        if (pre_denoise && (rc = timed_d2h(pre_denoise, dpre.p, (size_t)n * sizeof(double), nullptr))) return rc;
        publish_output(expr_out, n, std::move(dout));
        if (pre_denoise) publish_output(pre_denoise, n, std::move(dpre));
        return ICNV_OK;
    }
    // ---- one contiguous block of cells per device
    int total_rounds = 0;
    {   // argument checks once, with the caller's error reporting (a plan on the whole matrix, nothing is launched)
        icnv_chain_t *probe = nullptr;
        int rc = icnv_chain_begin(&probe, cfg);
        if (rc) return rc;
        total_rounds = icnv_chain_num_rounds(probe);   // the meeting points of every worker: the probe's count, not a re-derivation
        icnv_chain_end(probe);
        if (total_rounds > 0)
            for (int q = 0; q < cfg->n_ref_grp; ++q)
                if (cfg->ref_off[q + 1] == cfg->ref_off[q]) ICNV_FAIL(ICNV_ERR_ARG, "empty reference group");
    }
    const bool needs_ref = total_rounds > 0;   // only then are cfg->ref_idx / ref_off read (they may be NULL otherwise)
    std::vector<std::vector<std::vector<double>>> part(4, std::vector<std::vector<double>>((size_t)nd));   // [round][worker]
    Rendezvous meet(nd);
    return on_devices(nd, [&](int w, hipStream_t s, int rc0) -> int {
        int rc = rc0;
        int64_t c0, c1;
        shard(C, nd, w, c0, c1);
        const int64_t n = G * (c1 - c0);
        // this block's reference cells, local indices, order kept
        std::vector<int32_t> ridx, roff(1, 0);
        for (int q = 0; needs_ref && q < cfg->n_ref_grp; ++q) {
            for (int32_t i = cfg->ref_off[q]; i < cfg->ref_off[q + 1]; ++i)
                if (cfg->ref_idx[i] >= c0 && cfg->ref_idx[i] < c1) ridx.push_back((int32_t)(cfg->ref_idx[i] - c0));
            roff.push_back((int32_t)ridx.size());
        }
        icnv_chain_cfg lc = *cfg;
        lc.C = c1 - c0;
        if (needs_ref) {
            lc.ref_idx = ridx.data();
            lc.ref_off = roff.data();
        }
        icnv_chain_t *ch = nullptr;
        MatrixLease in;
        DevBuf dout, dpre;
        if (!rc) rc = icnv_chain_begin(&ch, &lc);
        if (!rc) rc = acquire_input(expr_in + c0 * G, n, s, in);
        if (!rc) rc = dout.alloc((size_t)std::max<int64_t>(n, 1) * sizeof(double));
        if (!rc && pre_denoise) rc = dpre.alloc((size_t)std::max<int64_t>(n, 1) * sizeof(double));
        // every worker walks through the same number of meeting points (the probe chain's), failed or not
        if (!rc && icnv_chain_num_rounds(ch) != total_rounds) { set_error("worker chain disagrees with the probe about the reference rounds"); rc = ICNV_ERR_ARG; }
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
    WallTimer wall(HP_WALL_MS);
    g_hp.add(HP_CALLS, 1.0);
    int rc;
    if (nd == 1 && !g_res_on.load() && host_pipeline_enabled() && G * C * 8 >= (64ll << 20) && C >= 1024) {
        // one device, no residency: upload | Viterbi | download of the states pipelined over column blocks
        std::vector<int64_t> b0, b1;
        int n_first = 0;
        pipeline_blocks(G, C, nullptr, b0, b1, n_first);
        DevBuf din, ds, dn;
        const size_t n = (size_t)G * (size_t)C;
        if ((rc = din.alloc(n * sizeof(double))) || (rc = ds.alloc(n)) || (rc = dn.alloc(sizeof(int32_t)))) return rc;
        HostPipeline pipe;
        if ((rc = pipe.open((int)b0.size()))) return rc;
        int32_t bad = 0;
        rc = pipe.run(
            0,
            [&](int i, hipStream_t s) {
                const size_t off = (size_t)(b0[(size_t)i] * G), cnt = (size_t)((b1[(size_t)i] - b0[(size_t)i]) * G);
                g_hp.add(HP_H2D_BYTES, (double)cnt * 8.0);
                return hipMemcpyAsync(din.as<double>() + off, expr + off, cnt * sizeof(double), hipMemcpyHostToDevice, s);
            },
            [&](hipStream_t s) -> int {
                ICNV_HIP(hipMemsetAsync(dn.p, 0, sizeof(int32_t), s));
                return ICNV_OK;
            },
            [&](int i, hipStream_t s) -> int {
                const int64_t c0 = b0[(size_t)i], c1 = b1[(size_t)i];
                return icnv_viterbi_cells_dev(din.as<double>() + c0 * G, ds.as<uint8_t>() + c0 * G, G, c1 - c0, chr_start, n_chr, K, mean, sd_shared,
                                              logPi, logDelta, dn.as<int32_t>(), s);
            },
            [&](int i, hipStream_t s) {
                const size_t off = (size_t)(b0[(size_t)i] * G), cnt = (size_t)((b1[(size_t)i] - b0[(size_t)i]) * G);
                g_hp.add(HP_D2H_BYTES, (double)cnt);
                return hipMemcpyAsync(states + off, ds.as<uint8_t>() + off, cnt, hipMemcpyDeviceToHost, s);
            });
        if (rc) return rc;
        ICNV_HIP(hipMemcpy(&bad, dn.p, sizeof(int32_t), hipMemcpyDeviceToHost));
        bad_total += bad;
    } else if (nd == 1) rc = block(0, C, nullptr);
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

// ------------------------------------------------------------------ group HMM and median filter, host buffers
// Groups (HMM subclusters / samples) and median-filter tiles are independent (R/inferCNV_HMM.R:371, 529-533;
// R/noise_reduction.R:57-86): with icnv_set_devices(n) whole groups / tiles are dealt to the devices -- longest first onto
// the least loaded device --, every worker packs the columns of ITS groups' cells from the host matrix, runs the *_dev
// entry point on its own stream and hands back the columns of those cells.  No exchange between devices.
namespace {
struct GroupDeal {
    std::vector<int32_t> cells;       // global cell ids this device holds, first appearance in its groups' order
    std::vector<int32_t> idx, off;    // its groups as LOCAL index lists (packed)
    std::vector<int32_t> gids;        // global ids of its groups, ascending
};
void deal_groups(const int32_t *grp_idx, const int32_t *grp_off, int32_t n_grp, int nd, int64_t C, std::vector<GroupDeal> &deal) {
    deal.assign((size_t)nd, GroupDeal());
    std::vector<int32_t> order((size_t)n_grp);
    for (int32_t q = 0; q < n_grp; ++q) order[(size_t)q] = q;
    std::stable_sort(order.begin(), order.end(), [&](int32_t a, int32_t b) { return grp_off[a + 1] - grp_off[a] > grp_off[b + 1] - grp_off[b]; });
    std::vector<int64_t> load((size_t)nd, 0);
    for (int32_t q : order) {
        int best = 0;
        for (int w = 1; w < nd; ++w)
            if (load[(size_t)w] < load[(size_t)best]) best = w;
        deal[(size_t)best].gids.push_back(q);
        load[(size_t)best] += grp_off[q + 1] - grp_off[q];
    }
    std::vector<int32_t> pos((size_t)std::max<int64_t>(C, 1));
    for (auto &d : deal) {
        std::sort(d.gids.begin(), d.gids.end());
        std::fill(pos.begin(), pos.end(), -1);
        d.off.push_back(0);
        for (int32_t q : d.gids) {
            for (int32_t i = grp_off[q]; i < grp_off[q + 1]; ++i) {
                const int32_t c = grp_idx[i];
                if (pos[(size_t)c] < 0) { pos[(size_t)c] = (int32_t)d.cells.size(); d.cells.push_back(c); }
                d.idx.push_back(pos[(size_t)c]);
            }
            d.off.push_back((int32_t)d.idx.size());
        }
    }
}
int check_groups_host(const int32_t *idx, const int32_t *off, int32_t n, int64_t C, const char *what) {
    if (n < 0 || (n > 0 && (!idx || !off))) ICNV_FAIL(ICNV_ERR_ARG, std::string("bad ") + what);
    for (int32_t q = 0; q < n; ++q) {
        if (off[q + 1] < off[q]) ICNV_FAIL(ICNV_ERR_ARG, std::string(what) + " offsets must not decrease");
        for (int32_t i = off[q]; i < off[q + 1]; ++i)
            if (idx[i] < 0 || idx[i] >= C) ICNV_FAIL(ICNV_ERR_ARG, std::string(what) + " index out of range");
    }
    return ICNV_OK;
}
}  // namespace

int icnv_viterbi_groups(const double *expr, uint8_t *states, int64_t G, int64_t C, const int32_t *chr_start, int32_t n_chr,
                        const int32_t *grp_idx, const int32_t *grp_off, int32_t n_grp, int32_t K, const double *mean,
                        const double *sd_shared_per_grp, const double *logPi, const double *logDelta) {
    if (!expr || !states || G < 1 || C < 0) ICNV_FAIL(ICNV_ERR_ARG, "bad argument");
    const int nd = (int)std::max<int64_t>(1, std::min<int64_t>(g_ndev.load(), n_grp));
    if (nd == 1) return viterbi_groups_host_one(expr, states, G, C, chr_start, n_chr, grp_idx, grp_off, n_grp, K, mean, sd_shared_per_grp, logPi, logDelta);
    int rc = check_groups_host(grp_idx, grp_off, n_grp, C, "groups");
    if (rc) return rc;
    if (!sd_shared_per_grp) ICNV_FAIL(ICNV_ERR_ARG, "sd_shared_per_grp missing");
    std::vector<GroupDeal> deal;
    deal_groups(grp_idx, grp_off, n_grp, nd, C, deal);
    // a cell listed by several groups takes the states of the LAST of them (as the one-device path and R's loop do)
    std::vector<int32_t> winner((size_t)std::max<int64_t>(C, 1), -1), dev_of((size_t)n_grp, 0);
    for (int32_t q = 0; q < n_grp; ++q)
        for (int32_t i = grp_off[q]; i < grp_off[q + 1]; ++i) winner[(size_t)grp_idx[i]] = q;
    for (int w = 0; w < nd; ++w)
        for (int32_t q : deal[(size_t)w].gids) dev_of[(size_t)q] = w;
    std::memset(states, 0xFF, (size_t)G * (size_t)C);
    std::atomic<int64_t> bad_total{0};
    rc = on_devices(nd, [&](int w, hipStream_t s, int rc0) -> int {
        if (rc0) return rc0;
        const GroupDeal &d = deal[(size_t)w];
        const int64_t nc = (int64_t)d.cells.size();
        if (nc == 0) return ICNV_OK;
        std::vector<double> xin((size_t)G * (size_t)nc);
        for (int64_t j = 0; j < nc; ++j) std::memcpy(&xin[(size_t)(j * G)], expr + (int64_t)d.cells[(size_t)j] * G, (size_t)G * sizeof(double));
        std::vector<double> sd;
        for (int32_t q : d.gids) sd.push_back(sd_shared_per_grp[q]);
        DevBuf dx, ds, dn;
        int r;
        if ((r = dx.alloc(xin.size() * sizeof(double))) || (r = ds.alloc((size_t)G * (size_t)nc)) || (r = dn.alloc(sizeof(int32_t)))) return r;
        ICNV_HIP(hipMemcpyAsync(dx.p, xin.data(), xin.size() * sizeof(double), hipMemcpyHostToDevice, s));
        ICNV_HIP(hipMemsetAsync(dn.p, 0, sizeof(int32_t), s));
        if ((r = icnv_viterbi_groups_dev(dx.as<double>(), ds.as<uint8_t>(), G, nc, chr_start, n_chr, d.idx.data(), d.off.data(),
                                         (int32_t)d.gids.size(), K, mean, sd.data(), logPi, logDelta, dn.as<int32_t>(), s)))
            return r;
        std::vector<uint8_t> st((size_t)G * (size_t)nc);
        int32_t bad = 0;
        ICNV_HIP(hipMemcpyAsync(st.data(), ds.p, st.size(), hipMemcpyDeviceToHost, s));
        ICNV_HIP(hipMemcpyAsync(&bad, dn.p, sizeof(int32_t), hipMemcpyDeviceToHost, s));
        ICNV_HIP(hipStreamSynchronize(s));
        bad_total += bad;
        for (int64_t j = 0; j < nc; ++j) {
            const int32_t c = d.cells[(size_t)j];
            if (dev_of[(size_t)winner[(size_t)c]] == w)   // (the winning group is this device's last group holding the cell)
                std::memcpy(states + (int64_t)c * G, &st[(size_t)(j * G)], (size_t)G);
        }
        return ICNV_OK;
    });
    if (rc) return rc;
    if (bad_total.load())
        ICNV_FAIL(ICNV_ERR_UNDERFLOW, "Problems With Underflow in " + std::to_string(bad_total.load()) + " sequences");
    return ICNV_OK;
}

int icnv_median_filter(const double *expr_in, double *expr_out, int64_t G, int64_t C, const int32_t *chr_start, int32_t n_chr,
                       const int32_t *tile_idx, const int32_t *tile_off, int32_t n_tiles, int32_t window_size) {
    if (!expr_in || !expr_out || G < 1 || C < 0) ICNV_FAIL(ICNV_ERR_ARG, "bad argument");
    if (expr_in == expr_out) ICNV_FAIL(ICNV_ERR_ARG, "median filter cannot run in place");   // (as the one-device path)
    const int nd = (int)std::max<int64_t>(1, std::min<int64_t>(g_ndev.load(), n_tiles));
    bool disjoint = true;   // tiles that share a cell keep the one-device path (its in-place order of the tiles)
    if (nd > 1) {
        if (int rc = check_groups_host(tile_idx, tile_off, n_tiles, C, "tiles")) return rc;
        std::vector<char> seen((size_t)std::max<int64_t>(C, 1), 0);
        for (int32_t i = 0; i < tile_off[n_tiles] && disjoint; ++i) {
            if (seen[(size_t)tile_idx[i]]) disjoint = false;
            seen[(size_t)tile_idx[i]] = 1;
        }
    }
    if (nd == 1 || !disjoint) return median_filter_host_one(expr_in, expr_out, G, C, chr_start, n_chr, tile_idx, tile_off, n_tiles, window_size);
    std::vector<GroupDeal> deal;
    deal_groups(tile_idx, tile_off, n_tiles, nd, C, deal);
    std::vector<char> covered((size_t)std::max<int64_t>(C, 1), 0);
    for (int32_t i = 0; i < tile_off[n_tiles]; ++i) covered[(size_t)tile_idx[i]] = 1;
    for (int64_t c = 0; c < C; ++c)   // cells in no tile pass through (R/noise_reduction.R:57-86 touches the tiles only)
        if (!covered[(size_t)c]) std::memcpy(expr_out + c * G, expr_in + c * G, (size_t)G * sizeof(double));
    return on_devices(nd, [&](int w, hipStream_t s, int rc0) -> int {
        if (rc0) return rc0;
        const GroupDeal &d = deal[(size_t)w];
        const int64_t nc = (int64_t)d.cells.size();
        if (nc == 0) return ICNV_OK;
        std::vector<double> xin((size_t)G * (size_t)nc);
        for (int64_t j = 0; j < nc; ++j) std::memcpy(&xin[(size_t)(j * G)], expr_in + (int64_t)d.cells[(size_t)j] * G, (size_t)G * sizeof(double));
        DevBuf dx, dout;
        int r;
        if ((r = dx.alloc(xin.size() * sizeof(double))) || (r = dout.alloc(xin.size() * sizeof(double)))) return r;
        ICNV_HIP(hipMemcpyAsync(dx.p, xin.data(), xin.size() * sizeof(double), hipMemcpyHostToDevice, s));
        if ((r = icnv_median_filter_dev(dx.as<double>(), dout.as<double>(), G, nc, chr_start, n_chr, d.idx.data(), d.off.data(),
                                        (int32_t)d.gids.size(), window_size, s)))
            return r;
        ICNV_HIP(hipMemcpyAsync(xin.data(), dout.p, xin.size() * sizeof(double), hipMemcpyDeviceToHost, s));
        ICNV_HIP(hipStreamSynchronize(s));
        for (int64_t j = 0; j < nc; ++j) std::memcpy(expr_out + (int64_t)d.cells[(size_t)j] * G, &xin[(size_t)(j * G)], (size_t)G * sizeof(double));
        return ICNV_OK;
    });
}

}  // extern "C"
