// Certified fast path of the per-cell HMM Viterbi (Viterbi.dthmm.adj, R/inferCNV_HMM.R:1101-1176).
//
// The exact kernel (viterbi_kernels.hip) spends ~1.1 k fp64 instructions per gene and wavefront on
// the reference's emission arithmetic (K x [Cody's pnorm + log + two divisions + log]).  Its OUTPUT
// is discrete: the arg-max decisions of the max-plus recurrence.  This kernel computes the same
// decisions from scores that are within a CERTIFIED distance eps of the exact kernel's scores
//   - emission scores relative to state 1 (a term common to all states changes no decision): K - 1
//     polynomials per observation from the table of emission_table.{h,cpp} (built in 80-bit arithmetic
//     from the exact functions, verified through these very double operations; eps_tab), plus twice the
//     distance of the exact kernel's own arithmetic from the exact functions (2 eps_spec: a difference
//     of two of its scores);
//   - the recurrence on nu_i - i b (the diagonal log transition b is a shift common to all states, so it is
//     dropped: one add, max, add per state; two roundings and the rounding of a - b per step, against the
//     exact kernel's two),
// and tests every decision it takes against the accumulated error bound: with
//   E_i <= (i + 1) * (eps + 6 u B)   (u = 2^-53, B >= any |value| of the recurrence)
// bounding |nu_fast - nu_exact| after gene i, a decision whose winner leads by more than 2 E_i is
// the decision the exact arithmetic takes (max is 1-Lipschitz).  A sequence with any decision inside
// the band (or an observation outside the table's domain, or a non-finite one) is FLAGGED and
// recomputed by the exact kernel (viterbi_redo_kernel) -- so the states are those of the exact
// kernel, bit for bit, always; the fast path only decides how many sequences take the slow road
// (~1e-4 of them on real-valued data).  DESIGN.md "Certified fast Viterbi" has the derivation.
//
// Requires the transition matrix of .get_HMM / .i3HMM_get_HMM (R/inferCNV_HMM.R:230-265,
// R/inferCNV_i3HMM.R:99-156): one off-diagonal value a = log t and one diagonal value b, b > a.
// Then  max_j (nu_j + logPi[j,k]) = max(nu_k + b, max_{j != k} nu_j + a), and because the best
// predecessor overall (i1) beats every other off-diagonal candidate, row k needs only the
// comparison  nu_k + b  vs  nu_i1 + a  (row i1 keeps itself: b - a is far outside the band):
// 2 K + 5 (K - 1) operations instead of 4 K^2.  Back-pointers shrink to K bits + i1 (uint16).
//
// Mapping: persistent workgroups of 12 wavefronts -- three per SIMD -- (the table fills most of the CU's LDS),
// every wavefront pulls (chromosome, 64-column block) tasks, longest chromosomes first, one lane per
// sequence as in the exact kernel.
#include <algorithm>
#include <cmath>
#include <cstring>

#include "icnv_internal.h"
#include "emission_table.h"
#include <type_traits>
#include "viterbi_trace.h"

#pragma clang fp contract(off)

namespace icnv {

namespace {

// Launch geometry and chunk sizes (measured choices; experiments with other values are built as replacement translation
// units or with -DVF_NT / -DVF_CH by scripts/build_variant.sh, never shipped).
// Round 4: 768 threads = THREE wavefronts per SIMD (<= 168 registers), 128-byte observation chunks.  What round 4's
// ablations showed (docs/KERNEL_LOG.md): with the observations served from 256 L2-resident columns the 1024-thread kernel
// of round 3 takes 1.73 instead of 2.1 ms, without the back-pointer stores it takes the same 2.1 -- the launch was paced
// by the observation stream (1024 lanes per CU each fetching HALF a 128-byte line per request from a column of its own:
// 4 MiB of lines in flight per XCD = the whole L2, every half line fetched twice, DRAM rows opened for 64 bytes), not by
// the vector or LDS pipes: taking 10 % of the vector instructions out (block summaries for the traceback) or hiding both
// LDS round trips of a gene behind the decision bookkeeping changed nothing at 1024 threads.  With the gene step software-
// pipelined (below) three wavefronts per SIMD are enough to keep the pipes as busy, the 40 registers they free hold a
// whole 128-byte line per lane and request, and the lines in flight fit the L2: 2.17-2.29 -> 2.00-2.07 ms on one box
// (1024 threads / 64-byte chunks -> 768 / 128; 512 threads 2.08, 768 / 64-byte 2.13-2.16, 256-byte chunks spill).
#ifndef VF_NT
#define VF_NT 768
#endif
constexpr int FAST_NT = VF_NT;
constexpr int FAST_TB = 16;   // genes per block of the uniform-alignment traceback (16: 2.54 ms, 32: 2.58, 64: 2.61)
#ifndef VF_CH
#define VF_CH 16
#endif
constexpr int FAST_CH = VF_CH;    // genes per observation chunk: 128 bytes = one whole cache line per lane and request
constexpr int NCF = EMIS_DEG + 1;
// coefficient record of one interval: (K - 1) x NCF doubles (scores relative to state 1, whose row is not stored),
// state after state, padded to an ODD number of 16-byte bank groups (the lanes' random intervals then spread over all LDS
// banks; K = 6: 25 doubles -> 208 bytes) -- read as 16-byte pairs at immediate offsets of one address
constexpr int rec_doubles(int K) { return (((K - 1) * NCF / 2) | 1) * 2; }
typedef double dbl2_t __attribute__((ext_vector_type(2)));

// v_min_f64 / v_max_f64 without the compiler's canonicalisation of both inputs (every operand here is the
// result of an arithmetic instruction; a NaN takes the flagged path anyway)
__device__ inline double min_raw(double x, double y) {
    double r;
    asm("v_min_f64 %0, %1, %2" : "=v"(r) : "v"(x), "v"(y));
    return r;
}
__device__ inline double max_raw(double x, double y) {
    double r;
    asm("v_max_f64 %0, %1, %2" : "=v"(r) : "v"(x), "v"(y));
    return r;
}

// the same with a wave-uniform second operand (kept in scalar registers)
__device__ inline double min_raw_s(double x, double y) {
    double r;
    asm("v_min_f64 %0, %1, %2" : "=v"(r) : "v"(x), "s"(y));
    return r;
}
__device__ inline double max_raw_s(double x, double y) {
    double r;
    asm("v_max_f64 %0, %1, %2" : "=v"(r) : "v"(x), "s"(y));
    return r;
}
// Which states are "near the top" (e_k >= t2, i.e. nu_k >= m1 - thr): bit k of the result, per lane.  The K comparisons
// leave wave masks in scalar registers, K v_addc_co_u32 (2 b + carry) shift them into the lane's word -- the best
// predecessor is then the lowest set bit, "more than one state near the top" a population count: two instructions per
// state and three per gene, where selects into i1 plus the scalar and / or tree over the masks took two per state and
// fifteen per gene.  One block: the compiler does not look inside asm for the wait states gfx950 wants between a vector
// instruction that writes a scalar register and a vector instruction that reads it -- every add is K instructions
// behind its comparison.
template <int K>
__device__ inline uint32_t near_top_bits(const double (&e)[K], double t2);
template <>
__device__ inline uint32_t near_top_bits<6>(const double (&e)[6], double t2) {
    uint32_t b = 0;
    uint64_t m0, m1, m2, m3, m4, m5;
    asm("v_cmp_ge_f64_e64 %1, %7, %13\n\t"
        "v_cmp_ge_f64_e64 %2, %8, %13\n\t"
        "v_cmp_ge_f64_e64 %3, %9, %13\n\t"
        "v_cmp_ge_f64_e64 %4, %10, %13\n\t"
        "v_cmp_ge_f64_e64 %5, %11, %13\n\t"
        "v_cmp_ge_f64_e64 %6, %12, %13\n\t"
        "v_addc_co_u32_e64 %0, vcc, %0, %0, %1\n\t"
        "v_addc_co_u32_e64 %0, vcc, %0, %0, %2\n\t"
        "v_addc_co_u32_e64 %0, vcc, %0, %0, %3\n\t"
        "v_addc_co_u32_e64 %0, vcc, %0, %0, %4\n\t"
        "v_addc_co_u32_e64 %0, vcc, %0, %0, %5\n\t"
        "v_addc_co_u32_e64 %0, vcc, %0, %0, %6"
        : "+v"(b), "=&s"(m5), "=&s"(m4), "=&s"(m3), "=&s"(m2), "=&s"(m1), "=&s"(m0)
        : "v"(e[5]), "v"(e[4]), "v"(e[3]), "v"(e[2]), "v"(e[1]), "v"(e[0]), "s"(t2)
        : "vcc");
    return b;
}
template <>
__device__ inline uint32_t near_top_bits<3>(const double (&e)[3], double t2) {
    uint32_t b = 0;
    uint64_t m0, m1, m2;
    asm("v_cmp_ge_f64_e64 %1, %4, %7\n\t"
        "v_cmp_ge_f64_e64 %2, %5, %7\n\t"
        "v_cmp_ge_f64_e64 %3, %6, %7\n\t"
        "v_addc_co_u32_e64 %0, vcc, %0, %0, %1\n\t"
        "v_addc_co_u32_e64 %0, vcc, %0, %0, %2\n\t"
        "v_addc_co_u32_e64 %0, vcc, %0, %0, %3"
        : "+v"(b), "=&s"(m2), "=&s"(m1), "=&s"(m0)
        : "v"(e[2]), "v"(e[1]), "v"(e[0]), "s"(t2)
        : "vcc");
    return b;
}
// STAGE (round 5): the observations of a chunk reach the lanes THROUGH LDS.  A lane that walks a column of its own asks for
// 16 bytes of each of 64 cache lines per request, eight requests per line; the memory system serves that pattern at 2.8-3.7
// TB/s (scripts/ubench/column_walk).  Requested BY ROWS -- request q fetches the eight lines of the columns 8 q .. 8 q + 7
// whole, eight lanes x 16 bytes per line -- the same 64 lines arrive at 4.6 TB/s, and at 5.5 TB/s when the requests are LDS-DMA
// (global_load_lds_dwordx4: lane l's 16 bytes land at M0 + 16 l, no registers, no write instructions): the eight lines of a
// request lie behind each other in the wavefront's 8 KiB buffer, column after column, and every lane reads its own column
// back with eight ds_read_b128.  Within a line the eight fetching lanes are permuted (pair p of column r of request q sits
// at slot 8 r + (p ^ f), f = (r >> 1) | ((q & 1) << 2)) so that the sixteen lanes of a ds_read_b128 pass hit sixteen
// different bank groups.  Price: 12 x 8 KiB of the LDS, i.e. a table of 256 instead of 694 records (shorter tails; a batch
// whose data leave them is redone with the full table by the register variant -- gate_count below).
// The DMA is issued from inline asm: the compiler's wait-count pass knows no alias information for LDS and would make
// every table gather behind a DMA wait for it (a chunk's HBM latency exposed sixteen times per chunk instead of hidden).
template <int K, bool STAGE>
__global__ void __launch_bounds__(FAST_NT) viterbi_fast_kernel(const FastViterbiArgs A_in) {
    // The descriptor (~60 dwords) is read from the kernel-argument segment where it is used (scalar loads) instead of living
    // in scalar registers for the whole launch: held, it does not fit next to the loop state, the allocator parks it in
    // VGPR lanes / scratch and the gene loop reloads a wave-uniform double per gene.
    (void)A_in;
    typedef const FastViterbiArgs __attribute__((address_space(4))) *ArgsK;
    ArgsK ap = (ArgsK)__builtin_amdgcn_kernarg_segment_ptr();
#define A (*ap)
#define ARGS_HERE()                                                                                      \
    do {                                                                                                 \
        unsigned long long ap_bits_ = (unsigned long long)ap;                                            \
        const unsigned ap_lo_ = __builtin_amdgcn_readfirstlane((unsigned)ap_bits_);                       \
        const unsigned ap_hi_ = __builtin_amdgcn_readfirstlane((unsigned)(ap_bits_ >> 32));               \
        ap_bits_ = ((unsigned long long)ap_hi_ << 32) | ap_lo_;                                           \
        asm volatile("" : "+s"(ap_bits_));                                                                \
        ap = (ArgsK)ap_bits_;                                                                             \
    } while (0)
    extern __shared__ __attribute__((aligned(16))) double tab[];
    if (A.gate_count) {
        // second attempt of a batch (the full table after the staged kernel's short one): only if the first one flagged more
        // sequences than the redo kernel takes; otherwise its count is handed on and nothing runs
        const int32_t first = *A.gate_count;
        if (first <= A.gate_limit) {
            if (blockIdx.x == 0 && threadIdx.x == 0) *A.flag_count = first;
            return;
        }
    }
    {
        const int n_dbl = A.n_int * rec_doubles(K) + 2 * A.n_grid;   // the records, then the grid entries
        const double2 *src = reinterpret_cast<const double2 *>(A.table);
        double2 *dst = reinterpret_cast<double2 *>(tab);
        for (int i = threadIdx.x; i < n_dbl / 2; i += FAST_NT) dst[i] = src[i];
    }
    __syncthreads();
    const double *coef = tab;
    const int lane = threadIdx.x & 63;
    const int64_t ncg = (A.ncols + 63) >> 6;
    const int64_t n_tasks = ncg * A.n_chr;

    for (;;) {
        ARGS_HERE();
        int task = 0;
        if (lane == 0) task = atomicAdd(A.task_counter, 1);
        task = __builtin_amdgcn_readfirstlane(task);
        if (task >= n_tasks) break;
        const int chr = A.chr_order[task / ncg];
        int64_t col0 = (task % ncg) * 64;
        bool dup = false;   // STAGE: this lane repeats a column of the task in front (its results are identical, it flags nothing)
        if constexpr (STAGE) {
            // every lane fetches for other lanes: the last, partial group of columns is served as the 64 LAST columns of
            // the matrix (launch_viterbi_fast: ncols >= 64), overlapping its predecessor
            const int64_t last0 = A.ncols - 64;
            if (col0 > last0) {
                dup = last0 + lane < col0;
                col0 = last0;
            }
        }
        const int64_t col = col0 + lane;
        if (!STAGE && col >= A.ncols) continue;
        const int s0 = A.chr_start[chr];
        const int n = A.chr_start[chr + 1] - s0;
        const double *xc = A.x + col * A.ld_x + s0;
        uint8_t *st = A.states + col * A.ld_st + s0;
        if (n < 2) {  // R/inferCNV_HMM.R:1104-1107
            if (n == 1) st[0] = 3;
            continue;
        }
        // back-pointers of this task: n rows of 64 lanes, contiguous (one 128-byte line per gene and wavefront, the
        // rows of a task next to each other in memory -- the forward pass writes and the traceback reads a stream)
        uint16_t *bpc = A.bp + ((int64_t)s0 * (ncg * 64) + (task % ncg) * 64 * (int64_t)n + lane);
        // block summaries (viterbi_trace.h: viterbi_traceback_blocks): behind the G rows of words, chromosome c owns the rows
        // [(s0 >> 4) + 3 c, (s0_next >> 4) + 3 (c + 1)) -- at least (n >> 4) + 3 of them, a task touches at most (n >> 4) + 2
        // aligned 16-gene blocks -- laid out like the words: the rows of a task next to each other
        const int sum_rows = ((s0 + n) >> 4) - (s0 >> 4) + 3;
        uint16_t *bsum = A.bp + ((int64_t)A.G + (s0 >> 4) + 3 * chr) * (ncg * 64) + ((task % ncg) * 64 * (int64_t)sum_rows + lane);
        // The 16-gene blocks of the traceback (and their summaries) are laid out by LANE 0's byte position of gene 0 inside a 16-byte
        // word of its state column.  When the number of genes per column is a multiple of 16 every lane shares it and a block is
        // one aligned 16-byte store; otherwise (round 5 -- real gene counts are not multiples of 16) the other lanes store their
        // blocks unaligned, which the hardware splits where a store crosses a line, instead of every lane taking the byte-wise
        // walk over the per-gene words (1.39 -> ~0.95 ms per 20 000 cells at 9 939 genes)
        const int a0u = __builtin_amdgcn_readfirstlane((int)((uintptr_t)st & 15u));
            // decision band of this task: 4 (n + 1) (eps + 6 u B), B = |logDelta|max + |a| + (n + 1)(s_max + |b|)
        const double np1 = (double)(n + 1);
        const double B = A.b0 + np1 * A.s_step;
        const double thr = 4.0 * np1 * (A.eps + 0x1.8p-51 * B);
        const double ab = A.a - A.b;      // off-diagonal minus diagonal log transition
        const double t2 = -ab - thr;
        const double x_hi_s = A.x_hi, x_lo_s = A.x_lo, inv_w_s = A.inv_w;
        const double *gridp = tab + A.n_int * rec_doubles(K);   // (wave-uniform) the grid entries behind the records

        double nu[K];
        uint32_t sacc = 0;      // OR of the back-pointer words of the current 16-gene block
        uint64_t seqflag = 0;   // lanes with an observation the table cannot score: the whole sequence goes to the exact kernel
        // Scores of one observation from the table, in two stages so that the LDS round trips of several genes
        // overlap: locate() finds the interval (lookup cell -> segment record -> interval index) and the position
        // inside it; poly() evaluates the K polynomials.
        auto locate = [&](double xv, int &idx, double &tn) {
            // clamp into the table's domain; an observation the clamp changes (NaN included) flags its sequence
            const double xs = max_raw_s(min_raw_s(xv, x_hi_s), x_lo_s);
            seqflag |= __builtin_amdgcn_ballot_w64(!(xs == xv));
            // uniform grid: interval j, one 16-byte entry {the state mean inside it or +inf, the interval's record}; an
            // interval with a mean has two records, below the mean and from the mean on (x_hi is chosen so that j needs no clamp)
            const double u = (xs - x_lo_s) * inv_w_s;
            const int j = (int)u;
            const double2 ge = *reinterpret_cast<const double2 *>(gridp + 2 * j);   // boundary, {index of the record below it, of the record from it on}
            const unsigned long long rr = (unsigned long long)__double_as_longlong(ge.y);
            idx = (xs >= ge.x) ? (int)(uint32_t)(rr >> 32) : (int)(uint32_t)rr;
            tn = __builtin_amdgcn_fract(u) - 0.5;   // u >= 0: u - floor(u) = u - (double)(int)u, exactly
        };
        // Coefficients [f0, f1) of the record at cq (flat index f = (state - 1) * NCF + j, c0 first) as 16-byte pairs: for
        // NCF = 6 a state is three pairs of its own; an odd NCF makes pairs straddle two states (all indices are static)
        auto horner = [&](const dbl2_t *q, int p0, int base, double tn) -> double {
            auto cf = [&](int f) -> double { return (f & 1) ? q[f / 2 - p0].y : q[f / 2 - p0].x; };
            double p = cf(base + NCF - 1);
#pragma unroll
            for (int j = NCF - 2; j >= 0; --j) p = __builtin_fma(p, tn, cf(base + j));
            return p;
        };
        auto poly = [&](int idx, double tn, double (&sc)[K]) {
            const double *c = coef + idx;
            sc[0] = 0.0;   // the table holds s_k - s_1: a term common to all states changes no decision
            constexpr int NP = ((K - 1) * NCF + 1) / 2;
            dbl2_t q[NP];
#pragma unroll
            for (int p = 0; p < NP; ++p) q[p] = *reinterpret_cast<const dbl2_t *>(c + 2 * p);
#pragma unroll
            for (int k = 1; k < K; ++k) sc[k] = horner(q, 0, (k - 1) * NCF, tn);
        };
        // One step of the recurrence, on nu~_i = nu_i - i b (a shift common to all states: every decision and the
        // final arg-max are those of nu): nu~'_k = max(nu~_k, m1 + (a - b)) + s_k, m1 = max_j nu~_j.  Row k keeps itself
        // iff e_k = nu~_k - (m1 + (a - b)) >= 0.  Back-pointer word: bits 0..K-1 "row k does NOT keep itself" (the
        // sign bits of e_k, shifted in one v_alignbit each), bits 6..8 the best predecessor overall (i1), bits
        // 9..9+K-1 "row k's decision lies inside the error band".  Row k's decision is certain when |e_k| > thr and --
        // if the off-diagonal candidate wins -- the best predecessor leads the second best by more than thr as well
        // (exactly one state has nu~_k >= m1 - thr, i.e. e_k >= -(a - b) - thr).  The comparisons leave wave masks;
        // their combination is scalar work.
        //
        // One gene, register-lean (four wavefronts per SIMD: <= 128 registers): the bookkeeping that needs no score -- the
        // candidates e_k, the keep bits, the near-top word -- runs first, then the K - 1 score polynomials are gathered and
        // consumed in two batches (states 1-2, states 3-5), every score folded into its row of the recurrence as soon as it
        // exists.  Fake dependences (empty asm) keep batch B's gathers behind batch A's arithmetic and a gene behind its
        // predecessor: left to itself the scheduler requests all gathers of several genes at once (50-60 registers
        // per gene) -- the right thing with 256 registers, spills with 168 or 128.
        // Software pipeline across genes: the record of gene i + 1 is located while gene i runs (idx_c / tn_c hold the current
        // gene's), so a gene starts with its batch-A gathers already addressable; the instruction order below is pinned with
        // scheduling barriers:  gathers A, locate(i + 1) | candidates and keep bits | rows of batch A, gathers B | near-top word,
        // band test, back-pointer store | rows of batch B.  Both LDS round trips of the gene hide behind the ~50 vector
        // instructions of decision bookkeeping that need no score (before: gathers A waited for, rows A, gathers B waited for).
        int idx_c = 0;
        double tn_c = 0.0;
        auto gene = [&](double xv_next, int i) {
            constexpr int KA = (K - 1) < 2 ? (K - 1) : 2;    // states of batch A
            constexpr int KB0 = (K - 1 > KA) ? KA + 1 : 1;   // first state of batch B (if any)
            constexpr int pA1 = (KA * NCF + 1) / 2;                                   // pairs [0, pA1) hold the states 1..KA
            constexpr int pB0 = ((KB0 - 1) * NCF) / 2, pB1 = ((K - 1) * NCF + 1) / 2;  // pairs of the states KB0..K-1
            const int idx_here = idx_c;
            const double *cq = coef + idx_here;
            const double tn = tn_c;
            dbl2_t qa[pA1];
#pragma unroll
            for (int p = 0; p < pA1; ++p) qa[p] = *reinterpret_cast<const dbl2_t *>(cq + 2 * p);
            // the successor's grid entry rides behind the gathers; it is consumed together with them after the first block
            const double xs_n = max_raw_s(min_raw_s(xv_next, x_hi_s), x_lo_s);
            seqflag |= __builtin_amdgcn_ballot_w64(!(xs_n == xv_next));
            const double u_n = (xs_n - x_lo_s) * inv_w_s;
            const int j_n = (int)u_n;
            const double2 ge_n = *reinterpret_cast<const double2 *>(gridp + 2 * j_n);
            __builtin_amdgcn_sched_barrier(0);
            double m1 = nu[0];
#pragma unroll
            for (int k = 1; k < K; ++k) m1 = max_raw(m1, nu[k]);
            const double c = m1 + ab;
            uint32_t sb = 0;
            double e[K];
            double amin = 0.0;
#pragma unroll
            for (int k = K - 1; k >= 0; --k) {
                e[k] = nu[k] - c;
                sb = __builtin_amdgcn_alignbit(sb, (uint32_t)__double2hiint(e[k]), 31);
                amin = (k == K - 1) ? __builtin_fabs(e[k]) : __builtin_fmin(amin, __builtin_fabs(e[k]));
            }
            const uint64_t band = __builtin_amdgcn_ballot_w64(!(amin > thr));
            nu[0] = max_raw(nu[0], c);
            __builtin_amdgcn_sched_barrier(0);
            {
                const unsigned long long rr = (unsigned long long)__double_as_longlong(ge_n.y);
                idx_c = (xs_n >= ge_n.x) ? (int)(uint32_t)(rr >> 32) : (int)(uint32_t)rr;
                tn_c = __builtin_amdgcn_fract(u_n) - 0.5;   // u >= 0: u - floor(u) = u - (double)(int)u, exactly
            }
#pragma unroll
            for (int k = 1; k <= KA; ++k) nu[k] = max_raw(nu[k], c) + horner(qa, 0, (k - 1) * NCF, tn);
            dbl2_t qb[(K - 1 > KA) ? (pB1 - pB0) : 1];
            if (K - 1 > KA) {
                // batch B's gathers behind batch A's arithmetic (an empty asm makes their address depend on A's rows): with all
                // thirteen quads requested at once the step does not fit 128 registers, and a single scratch reload costs a
                // vmcnt(0) -- a drain of every outstanding store and observation load of the wavefront
                int idxb = idx_here;
                if (KA >= 2) asm volatile("" : "+v"(idxb) : "v"(nu[1]), "v"(nu[KA]));
                else asm volatile("" : "+v"(idxb) : "v"(nu[KA]));
                const double *cqb = coef + idxb;
#pragma unroll
                for (int p = pB0; p < pB1; ++p) {
                    // (an odd number of coefficients: the last pair's upper half is padding -- read 8 bytes, or the dead half of
                    // the destination is handed out again and its next writer waits for the LDS: an lgkmcnt(0) in mid-flight)
                    if (p == pB1 - 1 && (((K - 1) * NCF) & 1)) qb[p - pB0].x = *(cqb + 2 * p);
                    else qb[p - pB0] = *reinterpret_cast<const dbl2_t *>(cqb + 2 * p);
                }
            }
            __builtin_amdgcn_sched_barrier(0);
            const uint32_t near = near_top_bits<K>(e, t2);
            const uint32_t i1 = (uint32_t)__builtin_ctz(near | 0x80u);
            const uint64_t two_near = __builtin_amdgcn_ballot_w64(__builtin_popcount(near) != 1);
            uint32_t word = sb | (i1 << 6);
            if (__builtin_expect((band | two_near) != 0, 0)) {
                asm volatile("; uncertain decision in this wavefront" ::: "memory");
                const bool top_unsure = (two_near >> lane) & 1u;
#pragma unroll
                for (int k = 0; k < K; ++k) {
                    const bool unsure = !(__builtin_fabs(e[k]) > thr) || (!(e[k] >= 0.0) && top_unsure);
                    word |= unsure ? (512u << k) : 0u;
                }
            }
            // (one address per chunk + immediate offsets for its sixteen rows measured SLOWER: 2.03 against 1.96 ms)
            bpc[i * 64] = (uint16_t)word;
            sacc |= word;
            __builtin_amdgcn_sched_barrier(0);
            if (K - 1 > KA) {
#pragma unroll
                for (int k = KB0; k <= K - 1; ++k) nu[k] = max_raw(nu[k], c) + horner(qb, pB0, (k - 1) * NCF, tn);
            }
        };
        auto gene_s = [&](int i) {   // a gene outside the chunk loop: its successor's observation comes by itself; it may close a block
            gene(xc[(i + 1 < n) ? i + 1 : i], i);
            if (((a0u + i) & 15) == 15) {
                bsum[((a0u + i) >> 4) * 64] = (uint16_t)sacc;
                sacc = 0;
            }
        };
        {
            double sc[K], tn0;
            int idx0;
            locate(xc[0], idx0, tn0);
            poly(idx0, tn0, sc);
#pragma unroll
            for (int k = 0; k < K; ++k) nu[k] = A.logDelta[k] + sc[k];
            locate(xc[1], idx_c, tn_c);   // n >= 2
        }
        // Observations are streamed CH genes (128 bytes, one aligned cache line when G is a multiple of 16) per
        // lane and request: every lane walks its own column, so the memory system sees 64 streams per wavefront;
        // 8-byte requests would fetch each line sixteen times through an L1 that cannot hold 1024 of them, and
        // small unaligned pieces of a DRAM burst arrive as separate requests long after the burst was evicted.
        constexpr int CH = FAST_CH;
        auto load_chunk = [&](const double *p, double (&v)[CH]) {
#pragma unroll
            for (int j = 0; j < CH; j += 2) {
                // default cache policy (non-temporal loads measured 17 % slower with 64-byte chunks in round 2 and 50 % slower
                // with whole lines in round 4: 2.95 against 1.94 ms)
                const dbl2_t a0 = *reinterpret_cast<const dbl2_t *>(p + j);
                v[j] = a0.x;
                v[j + 1] = a0.y;
            }
        };
        int i = 1;
        {   // wave-uniform peel up to lane 0's next chunk (cache line) boundary
            const uint64_t a0 = (uint64_t)(uintptr_t)(xc + 1);
            constexpr uint32_t AL = CH * 8;
            int peel = (int)(((AL - (uint32_t)(a0 & (AL - 1))) & (AL - 1)) >> 3);
            peel = __builtin_amdgcn_readfirstlane(peel);
            for (; peel > 0 && i < n; --peel, ++i) gene_s(i);
        }
        // the chunks are the 16-gene blocks of the summaries when the observations' line alignment and the states' 16-byte
        // alignment go together (always, for G a multiple of 16 and aligned matrices): a block then ends with a chunk
        const bool use_sum = ((a0u + i) & (CH - 1)) == 0;
        if constexpr (STAGE) {
            static_assert(!STAGE || CH == 16, "a chunk is one 128-byte line per column");
            if (i + CH <= n) {
                double xcur[CH];
                // this wavefront's buffer: 64 columns x 128 bytes behind the table
                const int n_dbl = A.n_int * rec_doubles(K) + 2 * A.n_grid;
                const double *wbuf = tab + ((n_dbl + 1) & ~1) + (threadIdx.x >> 6) * 1024;
                const uint32_t lds_wave = __builtin_amdgcn_readfirstlane(
                    (uint32_t)(uintptr_t)(const __attribute__((address_space(3))) void *)wbuf);
                // what this lane reads back: its own column = line (lane & 7) of request (lane >> 3)
                const uint32_t rq = (uint32_t)lane >> 3, rr = (uint32_t)lane & 7u;
                const uint32_t rf = (rr >> 1) | ((rq & 1u) << 2);
                const char *rb = reinterpret_cast<const char *>(wbuf) + rq * 1024u + rr * 128u;
                // what this lane fetches in request q: 16 bytes (pair fp ^ f) of the column 8 q + (lane >> 3)
                const uint32_t fr = (uint32_t)lane >> 3, fp = (uint32_t)lane & 7u;
                const uint32_t voff_e = (fr * (uint32_t)A.ld_x + 2u * (fp ^ (fr >> 1))) * 8u;
                const uint32_t voff_o = (fr * (uint32_t)A.ld_x + 2u * (fp ^ ((fr >> 1) | 4u))) * 8u;
                const double *xt = A.x + col0 * A.ld_x + s0;   // (wave-uniform) gene 0 of the task's first column
                const int64_t qstride = 8 * A.ld_x;
                auto request = [&](int gi) {
                    const double *tb = xt + gi;
#pragma unroll
                    for (int q = 0; q < 8; ++q) {
                        // (M0 is not in the clobber list -- the compiler reserves it and warns; nothing else in this kernel uses it)
                        asm volatile("s_mov_b32 m0, %0\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %1, %2"
                                     :
                                     : "s"(lds_wave + (uint32_t)q * 1024u), "v"((q & 1) ? voff_o : voff_e), "s"(tb + q * qstride)
                                     : "memory");
                    }
                };
                auto fetch = [&]() {   // this lane's column of the chunk in the buffer
#pragma unroll
                    for (int p = 0; p < CH / 2; ++p) {
                        const dbl2_t v = *reinterpret_cast<const dbl2_t *>(rb + (((uint32_t)p ^ rf) << 4));
                        xcur[2 * p] = v.x;
                        xcur[2 * p + 1] = v.y;
                    }
                };
                request(i);
                asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
                for (; i + CH <= n; i += CH) {
                    // the chunk is in the buffer (waited for in front of the previous chunk's last gene, or above).  (Reading it
                    // in front of the previous chunk's last gene instead, so that the reads' latency hides behind a gene step:
                    // 1.876-1.897 against 1.885-1.887 ms in one call -- nothing.)
                    fetch();
                    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");   // every lane has its column: the buffer is free
                    const bool more = i + 2 * CH <= n;
                    // the observation behind the chunk: the next chunk's first, or -- last chunk -- the tail's first gene, if any
                    // (requested here; one load address that is either LDS or global would become a flat load, whose wait
                    // drains every outstanding store)
                    double xb_g = 0.0;
                    if (more) request(i + CH);
                    else xb_g = xc[(i + CH < n) ? i + CH : i + CH - 1];
#pragma unroll
                    for (int j = 0; j + 1 < CH; ++j) gene(xcur[j + 1], i + j);
                    // (fifteen back-pointer stores were issued behind the DMA; the counter returns in order)
                    if (more) asm volatile("s_waitcnt vmcnt(15)" ::: "memory");
                    const double xb_l = *reinterpret_cast<const double *>(rb + (rf << 4));   // (the old chunk's when there is no next)
                    const double x_behind = more ? xb_l : xb_g;
                    gene(x_behind, i + CH - 1);
                    if (use_sum && ((a0u + i + CH) & 15) == 0) {
                        bsum[((a0u + i) >> 4) * 64] = (uint16_t)sacc;
                        sacc = 0;
                    }
                }
            }
        } else if (i + CH <= n) {
            // the chunk behind the current one is requested before the current one's genes run: one chunk (16 gene steps) of lead.
            // (Two buffers taking turns without the copy -- the loop body twice -- measured 5 % slower: 2.10 against 2.00 ms.)
            double xcur[CH], xnext[CH];
            load_chunk(xc + i, xcur);
            for (; i + CH <= n; i += CH) {
                const bool more = i + 2 * CH <= n;
                if (more) load_chunk(xc + i + CH, xnext);
                // (the observation behind the chunk: the next chunk's first, or -- last chunk -- the tail's first gene, if any)
                const double x_behind = more ? xnext[0] : xc[(i + CH < n) ? i + CH : i + CH - 1];
#pragma unroll
                for (int j = 0; j < CH; ++j) gene((j + 1 < CH) ? xcur[j + 1] : x_behind, i + j);
                if (use_sum && ((a0u + i + CH) & 15) == 0) {
                    bsum[((a0u + i) >> 4) * 64] = (uint16_t)sacc;
                    sacc = 0;
                }
                if (more) {
#pragma unroll
                    for (int j = 0; j < CH; ++j) xcur[j] = xnext[j];
                }
            }
        }
        for (; i < n; ++i) gene_s(i);
        ARGS_HERE();
        // last row: R's which.max
        double m1 = nu[0], m2 = -__builtin_inf();
        int cur = 0;
#pragma unroll
        for (int k = 1; k < K; ++k) {
            const double v = nu[k];
            m2 = max_raw(m2, min_raw(m1, v));
            cur = (v > m1) ? k : cur;
            m1 = max_raw(m1, v);
        }
        // the traceback follows the decisions of ONE path: only an uncertain decision ON that path (or an
        // uncertain final arg-max) can make the exact arithmetic trace a different one
        uint32_t unsure = (((seqflag >> lane) & 1u) || !(m1 - m2 > thr)) ? 1u : 0u;
        {
            // OR of the words shifted by the traced state: bit 9 collects the "inside the band" bits of the path's rows
            uint32_t uacc = 0;
            auto load_bp = [&](int i) { return (uint32_t)bpc[i * 64]; };
            auto step_bp = [&](uint32_t w, int c) {
                const uint32_t tsh = w >> c;
                uacc |= tsh;
                return (tsh & 1u) ? (int)((w >> 6) & 7u) : c;
            };
            auto load_sum = [&](int b) { return (uint32_t)bsum[b * 64]; };
            auto note_sum = [&](uint32_t S, int c) { uacc |= S >> c; };
            if (use_sum) viterbi_traceback_blocks(st, n, cur, a0u, load_bp, load_sum, step_bp, note_sum);
            else viterbi_traceback_uniform<FAST_TB>(st, n, cur, a0u, load_bp, step_bp);
            unsure |= (uacc >> 9) & 1u;
        }
        if (unsure && !dup) {
            const int e = atomicAdd(A.flag_count, 1);
            A.flag_list[2 * (int64_t)e] = chr;
            A.flag_list[2 * (int64_t)e + 1] = (int32_t)col;
        }
    }
}

#undef A
#undef ARGS_HERE

}  // namespace

size_t viterbi_fast_scratch_bytes(int32_t G, int32_t n_chr, int64_t n_cols) {   // columns in blocks of 64
    // G rows of back-pointer words, then the block summaries: (G >> 4) + 3 n_chr rows (see the kernel)
    return ((size_t)G + (size_t)(G >> 4) + 3 * (size_t)n_chr + 1) * (size_t)((n_cols + 63) / 64 * 64) * sizeof(uint16_t);
}
constexpr size_t FAST_STAGE_BYTES = (size_t)(FAST_NT / 64) * 8192;   // staged variant: 64 columns x 128 bytes per wavefront
size_t viterbi_fast_lds_bytes(int K, int n_int, int n_grid, bool staged) {
    const size_t n_dbl = (size_t)n_int * rec_doubles(K) + 2 * (size_t)n_grid;
    return ((n_dbl + 1) & ~(size_t)1) * sizeof(double) + (staged ? FAST_STAGE_BYTES : 0);
}
int viterbi_fast_max_intervals(int K, bool staged) {
    // records the LDS can hold, each with its 16-byte grid entry; 8 KiB of the 160 KiB are left to the runtime
    return (int)((152 * 1024 - (staged ? FAST_STAGE_BYTES : 0)) / ((size_t)rec_doubles(K) * sizeof(double) + 16));
}

// Device image of the table: the coefficient records, then the grid entries {boundary, {rec, 0}}.
void viterbi_fast_table_image(const EmisTable &t, std::vector<double> &img) {
    const int rec = rec_doubles(t.K);
    img.assign((size_t)t.n_int * rec + 2 * (size_t)t.n_grid, 0.0);
    for (int i = 0; i < t.n_int; ++i)
        for (int k = 1; k < t.K; ++k)
            for (int j = 0; j < NCF; ++j)
                img[(size_t)i * rec + (k - 1) * NCF + j] = t.coef[((size_t)i * t.K + k) * NCF + j];
    const size_t g0 = (size_t)t.n_int * rec;
    for (int j = 0; j < t.n_grid; ++j) {
        // {boundary, index (in doubles) of the record below the boundary, index of the record from the boundary on}: the kernel
        // selects one of the two words -- no add and multiply per gene.  Without a mean in the interval (boundary = +inf)
        // both are the interval's one record.
        img[g0 + 2 * (size_t)j] = t.grid[(size_t)j].boundary;
        const int32_t lo = t.grid[(size_t)j].rec;
        const int32_t hi = std::isinf(t.grid[(size_t)j].boundary) ? lo : lo + 1;
        int32_t rp[2] = {lo * rec, hi * rec};
        std::memcpy(&img[g0 + 2 * (size_t)j + 1], rp, sizeof(rp));
    }
    if (img.size() & 1) img.push_back(0.0);   // the kernel copies 16 bytes at a time
}

int launch_viterbi_fast(const FastViterbiArgs &a, int K, bool staged, hipStream_t stream) {
    if (a.ncols <= 0 || a.n_chr <= 0) return ICNV_OK;
    const size_t lds = (viterbi_fast_lds_bytes(K, a.n_int, a.n_grid, staged) + 15) & ~(size_t)15;
    if (lds > 160 * 1024) ICNV_FAIL(ICNV_ERR_UNSUPPORTED, "emission table does not fit the LDS");
    if (staged && a.ncols < 64) ICNV_FAIL(ICNV_ERR_ARG, "the staged fast Viterbi needs at least 64 columns");
    if (a.ld_x < a.G || a.ld_st < a.G) ICNV_FAIL(ICNV_ERR_ARG, "fast Viterbi: leading dimension below the gene count");
    if (staged && a.ld_x * 8 * 8 >= ((int64_t)1 << 32)) ICNV_FAIL(ICNV_ERR_UNSUPPORTED, "too many genes for the staged fast Viterbi");
    const int64_t ncg = (a.ncols + 63) / 64;
    const int64_t tasks = ncg * a.n_chr;
    if (tasks > 0x7fffff00) ICNV_FAIL(ICNV_ERR_UNSUPPORTED, "too many Viterbi tasks for one launch");
    int grid = num_cus();
    const int64_t need = (tasks + FAST_NT / 64 - 1) / (FAST_NT / 64);
    if (grid > need) grid = (int)need;
    KernelTimer kt(a.gate_count ? "viterbi_full_table" : "viterbi", stream);
    if (K == 6 && staged) {
        static DeviceOnce once6s;
        if (int rc = ensure_dynamic_lds(reinterpret_cast<const void *>(viterbi_fast_kernel<6, true>), 160 * 1024, once6s)) return rc;
        hipLaunchKernelGGL((viterbi_fast_kernel<6, true>), dim3(grid), dim3(FAST_NT), lds, stream, a);
    } else if (K == 3 && staged) {
        static DeviceOnce once3s;
        if (int rc = ensure_dynamic_lds(reinterpret_cast<const void *>(viterbi_fast_kernel<3, true>), 160 * 1024, once3s)) return rc;
        hipLaunchKernelGGL((viterbi_fast_kernel<3, true>), dim3(grid), dim3(FAST_NT), lds, stream, a);
    } else if (K == 6) {
        static DeviceOnce once6;
        if (int rc = ensure_dynamic_lds(reinterpret_cast<const void *>(viterbi_fast_kernel<6, false>), 160 * 1024, once6)) return rc;
        hipLaunchKernelGGL((viterbi_fast_kernel<6, false>), dim3(grid), dim3(FAST_NT), lds, stream, a);
    } else if (K == 3) {
        static DeviceOnce once3;
        if (int rc = ensure_dynamic_lds(reinterpret_cast<const void *>(viterbi_fast_kernel<3, false>), 160 * 1024, once3)) return rc;
        hipLaunchKernelGGL((viterbi_fast_kernel<3, false>), dim3(grid), dim3(FAST_NT), lds, stream, a);
    } else {
        ICNV_FAIL(ICNV_ERR_UNSUPPORTED, "fast Viterbi is built for K = 6 and K = 3");
    }
    ICNV_HIP(hipGetLastError());
    return ICNV_OK;
}

}  // namespace icnv
