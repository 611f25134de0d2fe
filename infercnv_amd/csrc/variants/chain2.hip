// Fused smoothing chain, second generation: two resident cells per CU, wave-private smoothing.
//
// chain_kernel.inc keeps ONE cell per CU in a 90 KB LDS buffer and walks it through ~10 workgroup barriers; every
// wavefront waits for the same L2 round trip, the same LDS round trip, the same barrier at the same time (DESIGN.md
// section 4).  This kernel keeps the cell in REGISTERS (24 values per lane, 512 threads) and streams it through small
// wave-private LDS windows for the one stage that needs a different layout, so that
//   * two workgroups (two cells) are resident per CU (< 80 KB of LDS, <= 128 VGPRs): one cell's memory phases overlap the
//     other's arithmetic;
//   * steps 8, 9, 10 need NO workgroup barrier: a wavefront smooths its own two sub-blocks of the cell; the half window
//     either side of a sub-block (the halo) is loaded a second time from L2 instead of being exchanged between waves;
//   * only the median (step 11) synchronises the workgroup (histogram select, as in chain_kernel.inc).
//
// Geometry (compile time): 8 wavefronts x 2 sub-blocks; a sub-block is a WINDOW of 64 lanes x 13 padded positions =
// [HALO | core <= 728 | HALO] with HALO = PAD = even(T + 2) zeros between chromosomes, exactly as in the LDS layout of
// chain_kernel.inc (the pyramid needs no edge tests).  The host plans the sub-blocks (chain2_build_plan): core gene
// ranges with even bounds (aligned 16-byte pairs never straddle two sub-blocks), the window position of every loaded
// pair, the zero runs, one byte per position into the dictionary of 1/denominator values.  Layouts the plan cannot
// cover (more than 16 sub-blocks, odd G, other windows, too many distinct denominators) run chain_kernel.inc.
//
// Stages and arithmetic are those of chain_kernel.inc (same sliding pyramid started from chunk sums and first moments,
// same exact median select, same lean exp2): R/inferCNV_ops.R:1742-1786 (steps 8, 12), :2970-2983 (9), :2406-2532,
// :2640-2661 (10), :2074-2109 (11), :2814-2826 (14), :2302-2346 (22).
#include "chain_kernel.inc"

#include <algorithm>
#include <cmath>
#include <cstdlib>
#include <map>

namespace icnv {

namespace {

constexpr int C2_NW = 8;                  // wavefronts per workgroup (one cell)
constexpr int C2_NT = 64 * C2_NW;         // 512 threads
constexpr int C2_NSB = 2;                 // sub-blocks per wavefront
constexpr int C2_NSBT = C2_NW * C2_NSB;   // 16 sub-blocks per cell
constexpr int C2_L = 13;                  // chunk: padded positions per lane (odd: conflict-free 8-byte LDS access)
constexpr int C2_WIN = 64 * C2_L;         // 832 positions per window
constexpr int C2_NCS = 6;                 // core slots (gene pairs per lane) of a sub-block: 768 genes >= core
constexpr int C2_NS = C2_NSB * C2_NCS;    // 12 gene pairs per lane = 24 values
constexpr int C2_GL = 4, C2_GR = 4;       // guard entries either side of a wavefront's chunk-sum array
constexpr int C2_CSLEN = 64 + C2_GL + C2_GR;
constexpr int C2_MAXZR = 12;              // zero runs per sub-block the plan may hold
constexpr int C2_HDR = 4 + 2 * C2_MAXZR;  // ints per sub-block header: cg, cg_next, n_zero, -, (start, len) x MAXZR
// plan image (uint32 words)
constexpr int C2_P_HDR = 0;
// per (sub-block, lane) eight words, read as two 16-byte loads at the start of the sub-block (L2-resident; held in
// registers for the whole launch they are what the median and the last phase spill):
//   [0..2] six 16-bit core slot codes, [3] halo slot code, [4..7] thirteen bytes: index into the 1/denominator dictionary
constexpr int C2_P_LANE = C2_P_HDR + C2_NSBT * C2_HDR;          // [16][64][8]
constexpr int C2_P_VM = C2_P_LANE + C2_NSBT * 64 * 8;           // [8][64]: bit s = pair of slot s exists
constexpr int C2_P_END = C2_P_VM + C2_NW * 64;
// slot code: bits 0..9 window index of element 0, 10 element 0 is written, 11 element 1 sits PAD + 1 further (the pair
// straddles a chromosome boundary), 12 element 1 is written
constexpr uint32_t C2_V0 = 0x400u, C2_STRAD = 0x800u, C2_V1 = 0x1000u;

struct Chain2Args {
    const double *in;
    double *out, *pre_out;
    int32_t G;
    const int32_t *cells;
    int32_t n_cells, in_by_pos, out_by_pos;   // *_by_pos: the matrix holds one column per LIST POSITION (the reference-cell cache)
    double max_thresh;
    const double *b1, *b2, *denoise;
    const uint32_t *plan;
    const double *inv_dict;
};

__device__ inline void wave_mem_fence() { asm volatile("" ::: "memory"); }   // LDS is in order per wavefront: a compiler fence suffices

template <int MODE, int CMASK, int TC>
__global__ void __launch_bounds__(C2_NT, 4) chain2_kernel(const Chain2Args a) {
    constexpr int T = TC, PAD = (TC + 3) & ~1, HALO = PAD, L = C2_L;
    constexpr uint32_t mask = (uint32_t)CMASK;
    static_assert(L - 1 <= T + 1 && 2 * HALO < C2_WIN, "window geometry");
    static_assert(MODE == MODE_APPLY, "the statistics rounds read the reference-cell cache this kernel fills (mask 0x0F)");
    static_assert((mask & (ICNV_ST_SMOOTH | ICNV_ST_CENTER)) == (ICNV_ST_SMOOTH | ICNV_ST_CENTER) && !(mask & ICNV_ST_CENTER_MEAN),
                  "chain2 serves the passes that smooth and centre by the median");
    static_assert((mask & (ICNV_ST_SUBTRACT_REF_1 | ICNV_ST_MAX_THRESH)) == (ICNV_ST_SUBTRACT_REF_1 | ICNV_ST_MAX_THRESH),
                  "steps 8 + 9 keep non-finite values out of the LDS windows");
    const int t = threadIdx.x, lane = t & 63;
    const int w = __builtin_amdgcn_readfirstlane(t >> 6);
    const uint32_t G = (uint32_t)a.G;

    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    constexpr int WINS_LEN = HALO + C2_NW * C2_WIN + HALO + 2;   // the neighbouring windows (finite data) are each other's guards
    double *wins = reinterpret_cast<double *>(smem);
    double *win = wins + HALO + w * C2_WIN;
    double2 *cs_all = reinterpret_cast<double2 *>(wins + WINS_LEN);   // [NW][CSLEN] chunk sums and first moments (S, M)
    double2 *cs = cs_all + w * C2_CSLEN + C2_GL;
    double *cand = reinterpret_cast<double *>(cs_all);               // median candidates alias the chunk sums (disjoint in time)
    static_assert(CAND_CAP * 8 <= C2_NW * C2_CSLEN * 16, "candidate buffer does not fit its alias");
    uint32_t *hist = reinterpret_cast<uint32_t *>(cs_all + C2_NW * C2_CSLEN);   // [NB_HIST], kept zero between uses
    double *red = reinterpret_cast<double *>(hist + NB_HIST);       // [64]
    uint32_t *redu = reinterpret_cast<uint32_t *>(red + 64);        // [32]
    int32_t *sel = reinterpret_cast<int32_t *>(redu + 32);          // [8]
    double *seld = reinterpret_cast<double *>(sel + 8);             // [2]
    unsigned long long *skey = reinterpret_cast<unsigned long long *>(seld + 2);   // [4]
    double *s_dict = reinterpret_cast<double *>(skey + 4);          // [256]

    for (int i = t; i < WINS_LEN; i += C2_NT) wins[i] = 0.0;
    for (int i = t; i < C2_NW * C2_CSLEN; i += C2_NT) cs_all[i] = make_double2(0.0, 0.0);
    for (int i = t; i < NB_HIST; i += C2_NT) hist[i] = 0u;
    for (int i = t; i < 256; i += C2_NT) s_dict[i] = a.inv_dict[i];
    if (t == 0) { skey[0] = ~0ull; skey[1] = 0ull; skey[2] = ~0ull; }
    __syncthreads();

    const uint32_t vm = a.plan[C2_P_VM + w * 64 + lane];
    const int n_empty = C2_NT * C2_NS * 2 - (int)G;

    double mu = 0.0, lo_d = 0.0, hi_d = 0.0;   // step 22 bounds (R/inferCNV_ops.R:2335), fixed for the whole launch
    if (mask & ICNV_ST_DENOISE) {
        mu = a.denoise[0];
        const double sdv = a.denoise[1];
        lo_d = sgpr_f64(mu - sdv);
        hi_d = sgpr_f64(mu + sdv);
    }
    bool range_known = false;   // level-0 range of the median select: [min, max] of the last cell this workgroup measured
    double range_lo = 0.0, range_hi = 0.0;
    const double thr_hi = a.max_thresh, thr_lo = -a.max_thresh;
    const int32_t *hdr_all = reinterpret_cast<const int32_t *>(a.plan) + C2_P_HDR;

    for (int cell = (int)blockIdx.x; cell < a.n_cells; cell += (int)gridDim.x) {
        PROF_DECL;
        const int64_t col = (int64_t)((a.cells && !a.in_by_pos) ? sload_i32(a.cells + cell) : cell);
        const char *src = reinterpret_cast<const char *>(a.in + col * (int64_t)G);
        const char *b1lo = reinterpret_cast<const char *>(a.b1), *b1hi = reinterpret_cast<const char *>(a.b1 + G);
        double wv[C2_NCS][2];   // the smoothed core slots of sub-block 0 (S layout); sub-block 1's stay in the window
        uint32_t cc1[3];       // core slot codes of sub-block 1 (window positions of its pairs)
        uint32_t lane_u = (uint32_t)lane;
        asm volatile("" : "+v"(lane_u));   // per-slot address arithmetic stays inside the cell loop (hoisted, it is spilled)
        // the same for what derives from the validity word (twelve masks): an opaque copy per cell
        uint32_t vmc = vm;
        asm volatile("" : "+v"(vmc));
        auto slot_valid = [&](int s) -> bool { return (vmc >> s) & 1u; };

        // ---------------- steps 8, 9, 10: wave-private, sub-block by sub-block ----------------
        // Sub-block 0's values (HBM) are requested here, sub-block 1's as soon as sub-block 0's are in its window, so that
        // they arrive while sub-block 0 is smoothed (LDS work only); the step-8 bound vectors (L2) follow per group of
        // four slots.  The first wait of the cell also drains the previous cell's stores (vmcnt counts in order).
        double xin[C2_NCS + 1][2];
        auto slot_offsets = [&](int sb, uint32_t lu, uint32_t (&go)[C2_NCS + 1]) {
            const int32_t *hdr = hdr_all + (C2_NSB * w + sb) * C2_HDR;
            const int cg = hdr[0], cgn = hdr[1];
            // six core slots and the halo slot (lanes 0..31 the pairs below the core, 32..63 above it)
#pragma unroll
            for (int s = 0; s < C2_NCS; ++s) go[s] = 8u * min((uint32_t)cg + 2u * (64u * s + lu), G - 2u);
            const int hp = (lu < 32u) ? (cg >> 1) - 32 + (int)lu : (cgn >> 1) + (int)lu - 32;
            go[C2_NCS] = 16u * (uint32_t)min(max(hp, 0), (int)(G >> 1) - 1);
        };
        auto request_cell = [&](int sb) {
            uint32_t go[C2_NCS + 1];
            slot_offsets(sb, lane_u, go);
#pragma unroll
            for (int s = 0; s <= C2_NCS; ++s) load_vec_stream<2>(reinterpret_cast<const double *>(src + go[s]), xin[s]);
        };
        request_cell(0);
#pragma unroll
        for (int sb = 0; sb < C2_NSB; ++sb) {
            const int32_t *hdr = hdr_all + (C2_NSB * w + sb) * C2_HDR;
            const int nz = hdr[2];
            uint32_t go[C2_NCS + 1];
            {
                uint32_t lu = lane_u;
                asm volatile("" : "+v"(lu));   // (recomputed per sub-block: carried, the offsets are spilled)
                slot_offsets(sb, lu, go);
            }
            // this lane's plan words of the sub-block (position-only, L2-resident)
            uint32_t cc[3], hcw, ic[4];
            {
                const uint4 *pl = reinterpret_cast<const uint4 *>(a.plan + C2_P_LANE) + ((C2_NSB * w + sb) * 64 + (int)lane_u) * 2;
                const uint4 p0 = pl[0], p1 = pl[1];
                cc[0] = p0.x; cc[1] = p0.y; cc[2] = p0.z; hcw = p0.w;
                ic[0] = p1.x; ic[1] = p1.y; ic[2] = p1.z; ic[3] = p1.w;
            }
            // padding inside the window's needed range: zero runs (position-only, from the plan)
            for (int z = 0; z < nz; ++z) {
                const int zs = hdr[4 + 2 * z], zl = hdr[5 + 2 * z];
                if ((int)lane_u < zl) win[zs + (int)lane_u] = 0.0;
            }
            // steps 8, 9 in two groups of slots (4 + 3): the step-8 bound vectors (L2) of a group are requested together,
            // one exposed L2 latency per group (all seven pairs of bounds at once would hold 56 registers)
            constexpr int AGS = 4;
#pragma unroll
            for (int grp = 0; grp * AGS <= C2_NCS; ++grp) {
                double lo1[AGS][2], hi1[AGS][2];
#pragma unroll
                for (int j = 0; j < AGS; ++j) {
                    const int s = grp * AGS + j;
                    if (s > C2_NCS) continue;
                    if (mask & ICNV_ST_SUBTRACT_REF_1) {
                        load_vec<2>(reinterpret_cast<const double *>(b1lo + go[s]), lo1[j]);
                        load_vec<2>(reinterpret_cast<const double *>(b1hi + go[s]), hi1[j]);
                    }
                }
#pragma unroll
                for (int j = 0; j < AGS; ++j) {
                    const int s = grp * AGS + j;
                    if (s > C2_NCS) continue;
                    double y[2];
#pragma unroll
                    for (int v = 0; v < 2; ++v) {
                        double x = xin[s][v];
                        if (mask & ICNV_ST_SUBTRACT_REF_1) x = subtract_ref(x, lo1[j][v], hi1[j][v], 1);
                        if (mask & ICNV_ST_MAX_THRESH) x = max_raw(min_raw(x, thr_hi), thr_lo);   // R/inferCNV_ops.R:2974-2975
                        y[v] = x;
                    }
                    const uint32_t code = (s < C2_NCS) ? ((cc[s >> 1] >> (16 * (s & 1))) & 0xFFFFu) : hcw;
                    const uint32_t wi0 = code & 0x3FFu;
                    if (s < C2_NCS && __builtin_expect(__builtin_amdgcn_ballot_w64((code & C2_STRAD) != 0) == 0, 1)) {
                        if (code & C2_V0) *reinterpret_cast<double2 *>(win + wi0) = make_double2(y[0], y[1]);   // both elements or none
                    } else {
                        if (code & C2_V0) win[wi0] = y[0];
                        if (code & C2_V1) win[wi0 + 1 + ((code & C2_STRAD) ? PAD : 0)] = y[1];
                    }
                }
                __builtin_amdgcn_sched_barrier(0);
            }
            PROF(sb == 0 ? 0 : 2);
            if (sb + 1 < C2_NSB) request_cell(sb + 1);   // arrives during this sub-block's smoothing
            wave_mem_fence();
            // ---- chunk layout: this lane owns window positions [L lane, L lane + L) ----
            const int p0 = L * (int)lane_u;
            double S = 0.0, M = 0.0, x0 = 0.0;
#pragma unroll
            for (int q = 0; q < L; ++q) {
                const double x = win[p0 + q];
                if (q == 0) x0 = x;
                S += x;
                M = __builtin_fma((double)q, x, M);
            }
            cs[(int)lane_u] = make_double2(S, M);
            wave_mem_fence();
            // window state at the chunk start: A = sum_{|d| <= T} (T + 1 - |d|) x[p0 + d], Lb = sum x[p0 - T .. p0],
            // Rb = sum x[p0 + 1 .. p0 + T + 1]; whole chunks through their (S, M), see chain_kernel.inc
            double A = __builtin_fma((double)(T + 1), S, -M);
            double Lb = x0;
            double Rb = S - x0;
            constexpr int NL = (T + L - 1) / L;   // chunks to the left with an element inside the half window
            constexpr int NR = (T + 1) / L;       // ... to the right (d = T + 1 counts for Rb)
            static_assert(NL <= C2_GL && NR <= C2_GR, "chunk-sum guards");
#pragma unroll
            for (int i = 1; i <= NL; ++i) {   // chunk i to the left: element q at distance d = i L - q, weight T + 1 - i L + q
                const double2 sm = cs[(int)lane_u - i];
                A += __builtin_fma((double)(T + 1 - i * L), sm.x, sm.y);
                Lb += sm.x;
#pragma unroll
                for (int q = 0; q < L; ++q) {
                    const int d = i * L - q;
                    if (d > T) {   // outside the half window: take its share out again
                        const double x = win[p0 - i * L + q];
                        Lb -= x;
                        if (d > T + 1) A = __builtin_fma((double)(d - T - 1), x, A);
                    }
                }
            }
#pragma unroll
            for (int i = 1; i <= NR; ++i) {   // chunk i to the right: d = i L + q, weight T + 1 - i L - q
                const double2 sm = cs[(int)lane_u + i];
                A += __builtin_fma((double)(T + 1 - i * L), sm.x, -sm.y);
                Rb += sm.x;
#pragma unroll
                for (int q = 0; q < L; ++q) {
                    const int d = i * L + q;
                    if (d > T + 1) {
                        const double x = win[p0 + i * L + q];
                        Rb -= x;
                        A = __builtin_fma((double)(d - T - 1), x, A);
                    }
                }
            }
            static_assert((NR + 1) * L > T + 1 || true, "");
            // elements beyond the last whole chunk to the right (none for T = 50, L = 13: 4 chunks reach d = 51)
            if constexpr ((NR + 1) * L - 1 < T + 1) {
#pragma unroll
                for (int d = (NR + 1) * L; d <= T + 1; ++d) {
                    const double x = win[p0 + d];
                    Rb += x;
                    A = __builtin_fma((double)(T + 1 - d), x, A);
                }
            }
            double r[L];
#pragma unroll
            for (int q = 0; q < L; ++q) {
                const int p = p0 + q;
                r[q] = A * s_dict[(ic[q >> 2] >> (8 * (q & 3))) & 255u];   // 1 / denominator of this position (0: padding, halo)
                const double x1 = win[p + 1], xr = win[p + T + 2], xl = win[p - T];   // (LDS reads are cheaper than 26 registers)
                A += Rb - Lb;
                Rb += xr - x1;
                Lb += x1 - xl;
                if (q % 4 == 3) __builtin_amdgcn_sched_barrier(0);   // at most 12 LDS reads in flight (all 39 at once: 78 registers)
            }
            wave_mem_fence();   // every halo read of this wavefront precedes the results
#pragma unroll
            for (int q = 0; q < L; ++q) win[p0 + q] = r[q];
            wave_mem_fence();
            // back to the S layout.  Sub-block 0's core slots return to registers; the LAST sub-block's stay in the window
            // (free until the next cell) and are read from there by the median's passes and the last phase: 24 registers
            // less through the workgroup-wide part (held, they are what the kernel spills)
            if (sb == 0) {
#pragma unroll
                for (int s = 0; s < C2_NCS; ++s) {
                    const uint32_t code = (cc[s >> 1] >> (16 * (s & 1))) & 0xFFFFu;
                    const uint32_t wi0 = code & 0x3FFu;
                    if (__builtin_expect(__builtin_amdgcn_ballot_w64((code & C2_STRAD) != 0) == 0, 1)) {
                        const double2 d2 = *reinterpret_cast<const double2 *>(win + wi0);
                        wv[s][0] = d2.x;
                        wv[s][1] = d2.y;
                    } else {
                        wv[s][0] = win[wi0];
                        wv[s][1] = win[wi0 + 1 + ((code & C2_STRAD) ? PAD : 0)];
                    }
                }
            } else {
#pragma unroll
                for (int j = 0; j < 3; ++j) cc1[j] = cc[j];
            }
            PROF(sb == 0 ? 1 : 3);
            wave_mem_fence();
        }
        // the pair of S-layout slot s: registers (sub-block 0) or this wavefront's window (sub-block 1)
        auto getv = [&](int s, double (&v)[2]) {
            if (s < C2_NCS) {
                v[0] = wv[s < C2_NCS ? s : 0][0];
                v[1] = wv[s < C2_NCS ? s : 0][1];
            } else {
                const int j = s - C2_NCS;
                const uint32_t code = (cc1[j >> 1] >> (16 * (j & 1))) & 0xFFFFu;
                const uint32_t wi0 = code & 0x3FFu;
                if (__builtin_expect(__builtin_amdgcn_ballot_w64((code & C2_STRAD) != 0) == 0, 1)) {
                    const double2 d2 = *reinterpret_cast<const double2 *>(win + wi0);
                    v[0] = d2.x;
                    v[1] = d2.y;
                } else {
                    v[0] = win[wi0];
                    v[1] = win[wi0 + 1 + ((code & C2_STRAD) ? PAD : 0)];
                }
            }
        };

        // ---------------- step 11: exact median over the G values (R/inferCNV_ops.R:2098), workgroup-wide ----------------
        // Histogram select of chain_kernel.inc: level 0 bins a borrowed value range (any range gives an exact selection),
        // the bin holding the middle rank is ranked directly or refined.  The NT * 24 - G pairs that do not exist count as
        // phantom values below everything: they are added to bin 0 at level 0 and to the wanted rank, nothing else sees them.
        double center = 0.0;
        {
            constexpr uint32_t OUT = 0xFFFFu;
            uint32_t pb[C2_NS];
            bool guessed = range_known;
            double lo, hi;
          measure_range:
            if (!guessed) {
                lo = __builtin_inf();
                hi = -__builtin_inf();
#pragma unroll
                for (int s = 0; s < C2_NS; ++s) {
                    double x[2];
                    getv(s, x);
                    if (slot_valid(s)) {
#pragma unroll
                        for (int v = 0; v < 2; ++v) { lo = min_raw(lo, x[v]); hi = max_raw(hi, x[v]); }
                    }
                }
                lo = wave_min(lo);
                hi = wave_max(hi);
                if (lane == 0) { atomicMin(&skey[0], f64_key(lo)); atomicMax(&skey[1], f64_key(hi)); }
                __syncthreads();
                lo = key_f64(skey[0]);
                hi = key_f64(skey[1]);
                if (lo < hi) {
                    range_lo = sgpr_f64(lo);
                    range_hi = sgpr_f64(hi);
                    range_known = true;
                }
            } else {
                lo = range_lo;
                hi = range_hi;
            }
            if (!(lo < hi)) {
                center = lo;   // every value of the cell equals lo
                __syncthreads();   // (rare path) every thread has read the keys; reset them behind a second barrier
                if (t == 0) { skey[0] = fresh_u64(~0ull); skey[1] = fresh_u64(0ull); }
                __syncthreads();
            } else {
                const double glo = sgpr_f64(lo), ghi = sgpr_f64(hi);
                double mid_lo = 0.0, mid_hi = 0.0;
                int npass = 1;
                for (int pass = 0; pass < npass; ++pass) {
                    const int target = (((int)G - 1) >> 1) + n_empty + pass;   // 0-based rank of the lower (upper) middle, phantoms included
                    int base = 0;
                    if (pass) {
                        lo = glo;
                        hi = ghi;
                        __syncthreads();
                        if (t == 0) { skey[0] = fresh_u64(~0ull); skey[1] = fresh_u64(0ull); }
                    }
                    for (int level = 0; level < MAX_LEVELS; ++level) {
                        const double scale = fmin((double)NB_HIST / (hi - lo), 0x1p1000);
                        auto hist_pass = [&](auto first_tag) {
                            constexpr bool FIRST = decltype(first_tag)::value;
                            if (FIRST && t == 0 && n_empty) atomicAdd(&hist[0], (uint32_t)n_empty);   // the phantoms
#pragma unroll
                            for (int s = 0; s < C2_NS; ++s) {
                                double x[2];
                                getv(s, x);
                                uint32_t b[2];
                                const bool valid = slot_valid(s);
#pragma unroll
                                for (int v = 0; v < 2; ++v) {
                                    const double tt = (x[v] - lo) * scale;
                                    // v_cvt_u32_f64 saturates (negative and NaN -> 0), then one integer min
                                    b[v] = min((uint32_t)__double2uint_rz(tt), (uint32_t)(NB_HIST - 1));
                                    if (!FIRST) b[v] = (x[v] >= lo && x[v] <= hi) ? b[v] : OUT;
                                    b[v] = valid ? b[v] : OUT;
                                }
                                if (b[0] == b[1]) {
                                    if (b[0] != OUT) atomicAdd(&hist[b[0]], 2u);   // neighbouring genes of a smoothed profile usually share a bin
                                } else {
#pragma unroll
                                    for (int v = 0; v < 2; ++v)
                                        if (b[v] != OUT) atomicAdd(&hist[b[v]], 1u);
                                }
                                pb[s] = b[0] | (b[1] << 16);
                            }
                        };
                        if (level == 0) hist_pass(std::true_type{}); else hist_pass(std::false_type{});
                        __syncthreads();
                        PROF(4);
                        if (t < 64) {   // one wavefront scans the histogram: lane l owns bins [32 l, 32 l + 32)
                            constexpr int BPL = NB_HIST / 64;
                            uint32_t mine = 0;
#pragma unroll 4
                            for (int b = 0; b < BPL; b += 4) {
                                const uint4 h4 = *reinterpret_cast<const uint4 *>(hist + t * BPL + b);
                                mine += (h4.x + h4.y) + (h4.z + h4.w);
                            }
                            const uint32_t inc = wave_incl_scan_u32(mine);
                            const uint32_t before = inc - mine;
                            const uint32_t rel = (uint32_t)(target - base);
                            const unsigned long long hit = __ballot(rel >= before && rel < before + mine);
                            const int owner = __ffsll((long long)hit) - 1;
                            const uint32_t obefore = __shfl(before, owner, 64);
                            const uint32_t hb = (t < BPL) ? hist[owner * BPL + t] : 0u;
                            const uint32_t inc2 = wave_incl_scan_u32(hb);
                            const uint32_t b2 = obefore + inc2 - hb;
                            if (t < BPL && rel >= b2 && rel < b2 + hb) {
                                sel[0] = owner * BPL + t;
                                sel[1] = (int)b2;
                                sel[2] = (int)hb;
                            }
                            if (t == 0) { sel[3] = (int)fresh_u64(0ull); skey[0] = fresh_u64(~0ull); skey[1] = fresh_u64(0ull); }
                        }
                        __syncthreads();
                        PROF(5);
                        const int sbin = sel[0];
                        int sbefore = sel[1], scnt = sel[2];
                        for (int b = t; b < NB_HIST; b += C2_NT) hist[b] = 0u;   // ready for the next use
                        if (__builtin_expect(guessed && sbin == 0, 0)) {
                            // the middle rank lies at or below the borrowed range's lower end (or among the phantoms): measure
                            // this cell's own range and bin again (the barrier of the measurement orders the zeroed histogram)
                            guessed = false;
                            goto measure_range;
                        }
                        if (level == 0 && sbin == 0) {   // the phantoms sit in front of bin 0's real members
                            sbefore += n_empty;
                            scnt -= n_empty;
                        }
                        auto in_bin = [&](int s, int v) -> bool { return ((pb[s] >> (16 * v)) & 0xFFFFu) == (uint32_t)sbin; };
                        if (scnt <= CAND_CAP) {
                            // collect the bin's members: one compare per value and a scalar branch on its ballot; a wavefront
                            // that has members takes one LDS atomic for all of them
#pragma unroll
                            for (int s = 0; s < C2_NS; ++s) {
                                if (__builtin_amdgcn_ballot_w64(in_bin(s, 0) || in_bin(s, 1)) == 0) continue;
                                double x[2];
                                getv(s, x);
#pragma unroll
                                for (int v = 0; v < 2; ++v) {
                                    const bool hit = in_bin(s, v);
                                    const unsigned long long bal = __ballot(hit);
                                    if (bal) {
                                        int wbase = 0;
                                        if (lane == 0) wbase = atomicAdd(&sel[3], (int)__popcll(bal));
                                        wbase = __builtin_amdgcn_readfirstlane(wbase);
                                        const uint32_t below = __builtin_amdgcn_mbcnt_hi((uint32_t)(bal >> 32),
                                                               __builtin_amdgcn_mbcnt_lo((uint32_t)bal, 0u));
                                        const int pos = wbase + (int)below;
                                        if (hit && pos < CAND_CAP) cand[pos] = x[v];
                                    }
                                }
                            }
                            __syncthreads();
                            PROF(6);
                            const int want = target - base - sbefore;
                            {   // rank the candidates with the whole workgroup: candidate x slice, partial counts meet in cnt[]
                                uint32_t *cnt = hist;
                                int lg = 0;
                                while (lg < 6 && (scnt << (lg + 1)) <= C2_NT) ++lg;
                                const int P = 1 << lg;
                                const int chunk = (scnt + P - 1) >> lg;
                                for (int it = t; it < (scnt << lg); it += C2_NT) {
                                    const int ci = it >> lg, part = it & (P - 1);
                                    const double cv = cand[ci];
                                    const int j0 = part * chunk, j1 = min(scnt, j0 + chunk);
                                    uint32_t less = 0;
                                    for (int cj = j0; cj < j1; ++cj) {
                                        const double o = cand[cj];
                                        less += (o < cv || (o == cv && cj < ci)) ? 1u : 0u;
                                    }
                                    if (less) atomicAdd(&cnt[ci], less);
                                }
                                __syncthreads();
                                for (int ci = t; ci < scnt; ci += C2_NT) {
                                    const int rk = (int)cnt[ci];
                                    cnt[ci] = 0u;   // the histogram is kept zero between uses
                                    if (rk == want) seld[0] = cand[ci];
                                    if (rk == want + 1) seld[1] = cand[ci];
                                }
                            }
                            __syncthreads();
                            const double vlo = seld[0];
                            if (pass) {
                                mid_hi = vlo;
                            } else {
                                mid_lo = vlo;
                                if (want + 1 < scnt) mid_hi = seld[1];
                                else if (!(G & 1)) npass = 2;
                            }
                            break;
                        }
                        // refine inside the selected bin: its members are exactly the values in [min, max] of the bin
                        base += sbefore;
                        double nlo = __builtin_inf(), nhi = -__builtin_inf();
#pragma unroll
                        for (int s = 0; s < C2_NS; ++s) {
                            double x[2];
                            getv(s, x);
#pragma unroll
                            for (int v = 0; v < 2; ++v)
                                if (in_bin(s, v)) { nlo = min_raw(nlo, x[v]); nhi = max_raw(nhi, x[v]); }
                        }
                        nlo = wave_min(nlo);
                        nhi = wave_max(nhi);
                        if (lane == 0) { atomicMin(&skey[0], f64_key(nlo)); atomicMax(&skey[1], f64_key(nhi)); }
                        __syncthreads();
                        lo = key_f64(skey[0]);
                        hi = key_f64(skey[1]);
                        if (!(lo < hi)) {
                            // every remaining member equals lo; the upper middle is lo too unless the members end exactly at
                            // rank `target`
                            double above = __builtin_inf();
                            uint32_t le = 0;
#pragma unroll
                            for (int s = 0; s < C2_NS; ++s) {
                                double x[2];
                                getv(s, x);
                                if (slot_valid(s)) {
#pragma unroll
                                    for (int v = 0; v < 2; ++v) {
                                        le += (x[v] <= lo) ? 1u : 0u;
                                        if (x[v] > lo) above = fmin(above, x[v]);
                                    }
                                }
                            }
                            const uint32_t cle = block_sum_u32<C2_NT>(le, redu) + (uint32_t)n_empty;   // (the phantoms lie below)
                            double dummy = -__builtin_inf();
                            block_minmax<C2_NT>(above, dummy, red);
                            if (t == 0) { skey[0] = fresh_u64(~0ull); skey[1] = fresh_u64(0ull); }   // (ordered by the barrier that ends the select)
                            if (pass) {
                                mid_hi = lo;
                            } else {
                                mid_lo = lo;
                                mid_hi = (cle > (uint32_t)(target + 1)) ? lo : above;
                            }
                            break;
                        }
                        __syncthreads();   // keys were read by everyone before the next level resets them
                        if (t == 0) { skey[0] = fresh_u64(~0ull); skey[1] = fresh_u64(0ull); }
                    }
                }
                center = (G & 1) ? mid_lo : (mid_lo + mid_hi) * 0.5;
                // candidates (alias of the chunk sums) and selection words are free for the next cell; the keys were reset by
                // the last scan (or on the rare exits above), behind which no atomic touched them
                __syncthreads();
            }
        }

        PROF(7);
        // ---------------- steps 11 (subtract), 12, 14, 22 and the stores: S layout ----------------
        const int64_t ocol = (int64_t)((a.cells && !a.out_by_pos) ? sload_i32(a.cells + cell) : cell);
        char *dst = (MODE == MODE_APPLY) ? reinterpret_cast<char *>(a.out + ocol * (int64_t)G) : nullptr;
        char *dpre = (MODE == MODE_APPLY && a.pre_out) ? reinterpret_cast<char *>(a.pre_out + ocol * (int64_t)G) : nullptr;
        const char *b2lo = reinterpret_cast<const char *>(a.b2), *b2hi = reinterpret_cast<const char *>(a.b2 + G);
        // Every step-12 bound vector (L2) is requested before the cell's first store -- a load issued behind a store waits
        // for that store to complete (vmcnt counts in order): sub-block 0's bounds, its final values formed in place
        // (registers), then sub-block 1's bounds, and only then the stores of both.
        auto final_values = [&](int sb, int s, const double (&lo2)[2], const double (&hi2)[2], double (&y)[2]) {
            getv(sb * C2_NCS + s, y);
#pragma unroll
            for (int v = 0; v < 2; ++v) {
                double x = y[v] - center;   // step 11
                if (mask & ICNV_ST_SUBTRACT_REF_2) x = subtract_ref(x, lo2[v], hi2[v], 1);
                y[v] = x;
            }
            if (mask & ICNV_ST_INVERT_LOG2) {   // R/inferCNV_ops.R:2818
                const bool wide = !(__builtin_fabs(y[0]) < 1022.0) || !(__builtin_fabs(y[1]) < 1022.0);
                if (__builtin_expect(__builtin_amdgcn_ballot_w64(wide) != 0, 0)) {
                    asm volatile("; exp2 outside the lean range" ::: "memory");
                    y[0] = exp2(y[0]);
                    y[1] = exp2(y[1]);
                } else {
                    y[0] = exp2_lean(y[0]);
                    y[1] = exp2_lean(y[1]);
                }
            }
        };
        auto store_pair = [&](int sl, uint32_t gofs, const double (&y)[2]) {
            if (slot_valid(sl)) {
                if (dpre) store_vec_stream<2>(reinterpret_cast<double *>(dpre + gofs), y);
                double o[2];
#pragma unroll
                for (int v = 0; v < 2; ++v) {
                    o[v] = y[v];
                    if (mask & ICNV_ST_DENOISE)   // strict bounds, R/inferCNV_ops.R:2335
                        if (o[v] > lo_d && o[v] < hi_d) o[v] = mu;
                }
                store_vec_stream<2>(reinterpret_cast<double *>(dst + gofs), o);
            }
        };
        {
            uint32_t lane_e = lane_u;
            asm volatile("" : "+v"(lane_e));   // the slots' byte offsets are formed again here (carried from the loads, they are spilled)
            uint32_t go0[C2_NCS + 1], go1[C2_NCS + 1];
            slot_offsets(0, lane_e, go0);
            double lo2[C2_NCS][2], hi2[C2_NCS][2];
#pragma unroll
            for (int s = 0; s < C2_NCS; ++s) {
                if (mask & ICNV_ST_SUBTRACT_REF_2) {
                    load_vec<2>(reinterpret_cast<const double *>(b2lo + go0[s]), lo2[s]);
                    load_vec<2>(reinterpret_cast<const double *>(b2hi + go0[s]), hi2[s]);
                }
            }
#pragma unroll
            for (int s = 0; s < C2_NCS; ++s) final_values(0, s, lo2[s], hi2[s], wv[s]);
            __builtin_amdgcn_sched_barrier(0);
            slot_offsets(1, lane_e, go1);
#pragma unroll
            for (int s = 0; s < C2_NCS; ++s) {
                if (mask & ICNV_ST_SUBTRACT_REF_2) {
                    load_vec<2>(reinterpret_cast<const double *>(b2lo + go1[s]), lo2[s]);
                    load_vec<2>(reinterpret_cast<const double *>(b2hi + go1[s]), hi2[s]);
                }
            }
            __builtin_amdgcn_sched_barrier(0);
            uint32_t lane_s = lane_u;
            asm volatile("" : "+v"(lane_s));   // (and once more for the stores: twelve offsets held through the arithmetic are spilled)
            uint32_t gs[C2_NCS + 1];
            slot_offsets(0, lane_s, gs);
#pragma unroll
            for (int s = 0; s < C2_NCS; ++s) store_pair(s, gs[s], wv[s]);
            slot_offsets(1, lane_s, gs);
#pragma unroll
            for (int s = 0; s < C2_NCS; ++s) {
                double y[2];
                final_values(1, s, lo2[s], hi2[s], y);
                store_pair(C2_NCS + s, gs[s], y);
            }
        }
        PROF(8);
    }
}

size_t chain2_lds_bytes(int T) {
    const int HALO = (T + 3) & ~1;
    return (size_t)(HALO + C2_NW * C2_WIN + HALO + 2) * 8 + (size_t)C2_NW * C2_CSLEN * 16 + NB_HIST * 4 + 64 * 8 + 32 * 4 + 8 * 4 + 2 * 8 +
           4 * 8 + 256 * 8;
}

template <int MODE, int CMASK, int TC>
int launch_chain2_t(const Chain2Args &a, hipStream_t stream, const char *name) {
    const size_t lds = chain2_lds_bytes(TC);
    static DeviceOnce once;
    if (int rc = ensure_dynamic_lds(reinterpret_cast<const void *>(chain2_kernel<MODE, CMASK, TC>), 80 * 1024, once)) return rc;
    int grid = num_cus() * 2;
    if (grid > a.n_cells) grid = a.n_cells;
    if (grid < 1) return ICNV_OK;
    KernelTimer kt(name, stream);
    hipLaunchKernelGGL((chain2_kernel<MODE, CMASK, TC>), dim3(grid), dim3(C2_NT), lds, stream, a);
    ICNV_HIP(hipGetLastError());
    return ICNV_OK;
}

}  // namespace

// ---------------------------------------------------------------------------------------------------------------------
// Host: plan of the sub-blocks for this chromosome layout (position-only: built once per chain plan)
bool chain2_build_plan(const int32_t *chr_start, int32_t n_chr, int32_t G, int32_t T, std::vector<uint32_t> &plan,
                       std::vector<double> &dict) {
    plan.clear();
    dict.clear();
    if (T != 50 || (G & 1) || G < 4 || n_chr < 1) return false;
    // Opt-in (ICNV_CHAIN2=1): measured on MI355X this kernel is SLOWER than chain_kernel.inc (3.0 vs 2.36 ms for 45 000
    // cells; DESIGN.md section 4 has the phase profile and the ablations), so the product path does not use it.  It stays
    // in the tree, parity-tested, as the measured answer to "a second resident cell per CU".
    {
        const char *e = std::getenv("ICNV_CHAIN2");
        if (!(e && e[0] == '1')) return false;
    }
    const int PAD = (T + 3) & ~1, HALO = PAD, CORE = C2_WIN - 2 * HALO;
    std::vector<int32_t> chr_of((size_t)G);
    for (int k = 0; k < n_chr; ++k)
        for (int g = chr_start[k]; g < chr_start[k + 1]; ++g) chr_of[(size_t)g] = k;
    auto pp = [&](int g) -> int64_t { return (int64_t)g + (int64_t)PAD * (chr_of[(size_t)g] + 1); };
    const int64_t npos = (int64_t)G + (int64_t)PAD * (n_chr + 1);
    // gene at a padded position, or -1 (padding)
    std::vector<int32_t> gene_at((size_t)npos, -1);
    for (int g = 0; g < G; ++g) gene_at[(size_t)pp(g)] = g;
    // 1 / denominator per gene and its dictionary
    const int64_t full = (int64_t)(T + 1) * (T + 1);
    std::vector<double> inv((size_t)G);
    std::map<double, int64_t> freq;
    for (int k = 0; k < n_chr; ++k) {
        const int n = chr_start[k + 1] - chr_start[k];
        for (int i = 0; i < n; ++i) {
            double v;
            if (n <= 1) v = 1.0 / (double)(T + 1);   // single-gene chromosome: untouched (R/inferCNV_ops.R:2417); A = (T+1) x
            else {
                const int64_t rl = std::max(T - i, 0), rr = std::max(T - (n - 1 - i), 0);
                v = 1.0 / (double)(full - rl * (rl + 1) / 2 - rr * (rr + 1) / 2);
            }
            inv[(size_t)(chr_start[k] + i)] = v;
            ++freq[v];
        }
    }
    if (freq.size() > 254) return false;
    dict.assign(256, 0.0);
    std::map<double, uint32_t> code_of;
    {
        std::vector<std::pair<int64_t, double>> order;
        for (auto &kv : freq) order.emplace_back(-kv.second, kv.first);
        std::sort(order.begin(), order.end());
        for (size_t i = 0; i < order.size(); ++i) { code_of[order[i].second] = (uint32_t)i + 1; dict[i + 1] = order[i].second; }
    }
    // sub-blocks: core gene ranges with even bounds whose padded span fits the window's core
    std::vector<int32_t> cg(1, 0);
    while (cg.back() < G) {
        const int32_t b = cg.back();
        int32_t e = b;
        while (e + 2 <= G && pp(e + 1) - pp(b) < CORE && (e + 2 - b) <= C2_NCS * 128) e += 2;
        if (e == b) return false;
        cg.push_back(e);
        if ((int)cg.size() - 1 > C2_NSBT) return false;
    }
    while ((int)cg.size() - 1 < C2_NSBT) cg.push_back(G);   // empty sub-blocks at the end
    plan.assign((size_t)C2_P_END, 0u);
    for (int k = 0; k < C2_NSBT; ++k) {
        const int32_t b = cg[(size_t)k], e = cg[(size_t)k + 1];
        int32_t *hdr = reinterpret_cast<int32_t *>(plan.data()) + C2_P_HDR + k * C2_HDR;
        hdr[0] = b;
        hdr[1] = e;
        const bool empty = e == b;
        // window base: the first core gene sits at window index HALO
        const int64_t Wb = empty ? 0 : pp(b) - HALO;
        const int64_t need_end = empty ? 0 : (pp(e - 1) - Wb) + 1 + HALO;   // needed positions: [0, need_end)
        auto wi_of = [&](int g) -> int64_t { return pp(g) - Wb; };
        // core slot codes
        for (int lane = 0; lane < 64; ++lane) {
            uint32_t codes[C2_NCS];
            for (int s = 0; s < C2_NCS; ++s) {
                const int g = b + 2 * (64 * s + lane);
                uint32_t c = (uint32_t)HALO;   // a pair that does not exist re-reads the first core position
                if (!empty && g + 1 < e + 0 + 1 && g < e) {
                    const int64_t w0 = wi_of(g), w1 = wi_of(g + 1);
                    c = (uint32_t)w0 | C2_V0 | C2_V1;
                    if (w1 != w0 + 1) {
                        if (w1 != w0 + 1 + PAD) return false;   // (cannot happen: a pair spans at most one boundary)
                        c |= C2_STRAD;
                    }
                    if (w1 >= C2_WIN) return false;
                }
                codes[s] = c;
            }
            uint32_t *dst = plan.data() + C2_P_LANE + (k * 64 + lane) * 8;
            for (int j = 0; j < 3; ++j) dst[j] = codes[2 * j] | (codes[2 * j + 1] << 16);
            // halo slot: lanes 0..31 the pairs below the core, 32..63 the pairs above it
            uint32_t hcode = 0;
            if (!empty) {
                const int hp = lane < 32 ? b / 2 - 32 + lane : e / 2 + lane - 32;
                if (hp >= 0 && hp < G / 2) {
                    const int g = 2 * hp;
                    const int64_t w0 = wi_of(g), w1 = wi_of(g + 1);
                    const bool ok0 = w0 >= 0 && w0 < need_end && (g < b || g >= e);
                    const bool ok1 = w1 >= 0 && w1 < need_end && (g + 1 < b || g + 1 >= e);
                    if (ok0 || ok1) {
                        // element 0's index is the base of both writes
                        int64_t base = w0;
                        bool strad = false;
                        if (w1 != w0 + 1) { if (w1 != w0 + 1 + PAD) return false; strad = true; }
                        if (!ok0) {   // only element 1: its own index as the base, written through the V1 bit
                            base = w1 - 1 - (strad ? PAD : 0);
                            if (base < 0) {   // the base would leave the window: shift it into range with the straddle offset off
                                base = w1 - 1;
                                strad = false;
                                if (base < 0) return false;
                            }
                        }
                        if (base < 0 || base > 0x3FF) return false;
                        hcode = (uint32_t)base | (ok0 ? C2_V0 : 0u) | (ok1 ? C2_V1 : 0u) | (strad ? C2_STRAD : 0u);
                    }
                }
            }
            dst[3] = hcode;
            // 1 / denominator codes of this lane's chunk: core genes only
            uint32_t ic[4] = {0, 0, 0, 0};
            for (int q = 0; q < C2_L; ++q) {
                const int64_t p = Wb + (int64_t)C2_L * lane + q;
                if (empty || p < 0 || p >= npos) continue;
                const int g = gene_at[(size_t)p];
                if (g < b || g >= e) continue;
                ic[q >> 2] |= code_of[inv[(size_t)g]] << (8 * (q & 3));
            }
            for (int j = 0; j < 4; ++j) dst[4 + j] = ic[j];
        }
        // zero runs: padding positions inside [0, need_end), in pieces of at most 64
        int nz = 0;
        for (int64_t p = 0; p < need_end;) {
            const int64_t ap = Wb + p;
            const bool pad = ap < 0 || ap >= npos || gene_at[(size_t)ap] < 0;
            if (!pad) { ++p; continue; }
            int64_t q = p;
            while (q < need_end && q - p < 64) {
                const int64_t aq = Wb + q;
                if (!(aq < 0 || aq >= npos || gene_at[(size_t)aq] < 0)) break;
                ++q;
            }
            if (nz >= C2_MAXZR) return false;
            hdr[4 + 2 * nz] = (int32_t)p;
            hdr[5 + 2 * nz] = (int32_t)(q - p);
            ++nz;
            p = q;
        }
        hdr[2] = nz;
        // every needed gene position must be covered by a core or halo slot
        for (int64_t p = 0; p < need_end; ++p) {
            const int64_t ap = Wb + p;
            if (ap < 0 || ap >= npos) continue;
            const int g = gene_at[(size_t)ap];
            if (g < 0 || (g >= b && g < e)) continue;
            const int hp = g / 2;
            const bool left = g < b;
            const int lane = left ? hp - (b / 2 - 32) : hp - e / 2 + 32;
            if (left ? (lane < 0 || lane >= 32) : (lane < 32 || lane >= 64)) return false;
        }
    }
    for (int wv = 0; wv < C2_NW; ++wv)
        for (int lane = 0; lane < 64; ++lane) {
            uint32_t m = 0;
            for (int sb = 0; sb < C2_NSB; ++sb) {
                const int k = C2_NSB * wv + sb;
                for (int s = 0; s < C2_NCS; ++s)
                    if (cg[(size_t)k] + 2 * (64 * s + lane) < cg[(size_t)k + 1]) m |= 1u << (sb * C2_NCS + s);
            }
            plan[(size_t)(C2_P_VM + wv * 64 + lane)] = m;
        }
    return true;
}

// Runs the request on the chain2 kernels when they cover it; CHAIN_NOT_INSTANTIATED otherwise.
int launch_chain2(const ChainArgs &a, int mode, hipStream_t stream) {
    if (!a.plan2 || !a.dict2 || a.T != 50 || (a.G & 1)) return CHAIN_NOT_INSTANTIATED;
    Chain2Args c;
    c.in = a.in; c.out = a.out; c.pre_out = a.pre_out; c.G = a.G; c.cells = a.cells; c.n_cells = a.n_cells;
    c.in_by_pos = a.in_by_pos; c.out_by_pos = a.out_by_pos; c.max_thresh = a.max_thresh; c.b1 = a.b1; c.b2 = a.b2; c.denoise = a.denoise;
    c.plan = a.plan2; c.inv_dict = a.dict2;
    const uint32_t m = a.mask;
    if (mode == MODE_APPLY && m == 0x7Fu) return launch_chain2_t<MODE_APPLY, 0x7F, 50>(c, stream, "chain2_apply");
    if (mode == MODE_APPLY && m == 0x3Fu) return launch_chain2_t<MODE_APPLY, 0x3F, 50>(c, stream, "chain2_apply");
    if (mode == MODE_APPLY && m == 0x0Fu) return launch_chain2_t<MODE_APPLY, 0x0F, 50>(c, stream, "chain2_stage_ref");   // steps 8-11 of the reference cells into their cache
    return CHAIN_NOT_INSTANTIATED;
}

}  // namespace icnv

// Developer / test hook (not part of include/icnv.h): the sub-block plan of a chromosome layout, so that the host logic
// can be checked without a GPU (tests/test_host.py emulates the kernel's data movement from this image).
// Returns the number of plan words (0: the layout is not covered); plan_out may be null to ask for the size.
#ifdef ICNV_CHAIN_PROFILE
extern "C" int icnv_debug_chain2_profile(unsigned long long *out32, int reset) {
    if (out32 && hipMemcpyFromSymbol(out32, HIP_SYMBOL(icnv::g_chain_prof), 32 * sizeof(unsigned long long)) != hipSuccess) return 2;
    if (reset) {
        unsigned long long z[32] = {0};
        if (hipMemcpyToSymbol(HIP_SYMBOL(icnv::g_chain_prof), z, sizeof(z)) != hipSuccess) return 2;
    }
    return 0;
}
#endif

extern "C" int icnv_debug_chain2_plan(const int32_t *chr_start, int32_t n_chr, int32_t G, int32_t T, uint32_t *plan_out,
                                      int32_t plan_cap, double *dict_out256) {
    std::vector<uint32_t> plan;
    std::vector<double> dict;
    if (!icnv::chain2_build_plan(chr_start, n_chr, G, T, plan, dict)) return 0;
    if (plan_out) {
        if ((size_t)plan_cap < plan.size()) return -1;
        std::copy(plan.begin(), plan.end(), plan_out);
    }
    if (dict_out256) std::copy(dict.begin(), dict.end(), dict_out256);
    return (int)plan.size();
}
