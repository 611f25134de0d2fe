// Traceback shared by the exact and the certified-fast Viterbi kernels: walks one sequence from its
// last gene to its first, emitting the 1-based state of every gene (uint8), eight consecutive genes
// per 8-byte store where the alignment allows.  The back-pointer words of a group of genes are requested
// a whole group before they are consumed (their addresses do not depend on the state being traced),
// so the walk is not one exposed memory latency per gene.
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

namespace icnv {

// load(i)  -> raw back-pointer word of gene i (1 <= i < n)
// step(w, cur) -> predecessor state (0-based) of state `cur` given gene i's word
template <int TG, class Load, class Step>
__device__ inline void viterbi_traceback(uint8_t *st, int n, int cur, Load load, Step step) {
    const uint64_t base = (uint64_t)(uintptr_t)st;
    const uint64_t last = base + (uint64_t)(n - 1);
    uint64_t word = 0;
    int i = n - 1;
    // TG genes per group: two groups of back-pointer lines in flight per wavefront hide the HBM latency of the walk
    uint32_t Wn[TG];
#pragma unroll
    for (int j = 0; j < TG; ++j) Wn[j] = (i - j > 0) ? load(i - j) : 0u;
    while (i >= 0) {
        uint32_t W[TG];
#pragma unroll
        for (int j = 0; j < TG; ++j) W[j] = Wn[j];
#pragma unroll
        for (int j = 0; j < TG; ++j) Wn[j] = (i - TG - j > 0) ? load(i - TG - j) : 0u;   // the next group, a group ahead
#pragma unroll
        for (int j = 0; j < TG; ++j) {
            const int g = i - j;
            if (g < 0) continue;
            const uint64_t addr = base + (uint64_t)g;
            const int b = (int)(addr & 7);
            word |= (uint64_t)(cur + 1) << (8 * b);
            if (b == 0 || g == 0) {
                const uint64_t w0 = addr - (uint64_t)b;
                const int hi = (int)((last - w0) < 7 ? (last - w0) : 7);
                if (b == 0 && hi == 7) {
                    *reinterpret_cast<uint64_t *>(st + g) = word;
                } else {
                    for (int bb = b; bb <= hi; ++bb) *reinterpret_cast<uint8_t *>(w0 + bb) = (uint8_t)(word >> (8 * bb));
                }
                word = 0;
            }
            if (g > 0) cur = step(W[j], cur);
        }
        i -= TG;
    }
}

// The same walk when every lane's state column has the SAME byte alignment a0 = address & 7 (always the case when
// the number of genes per column is a multiple of 8): the position of a gene inside its 8-byte output word is then
// wave-uniform, so a whole word is assembled with immediate shifts (one v_lshl_or per gene, +1 per byte added once
// per dword) and stored without per-gene alignment tests -- about 8 vector instructions per gene instead of 20.
// Genes in the partial words at either end of the sequence, and gene 0's group, are written byte by byte.
template <class Load, class Step>
__device__ inline void viterbi_traceback_uniform(uint8_t *st, int n, int cur, int a0, Load load, Step step) {
    int g = n - 1;
    while (g >= 0 && ((a0 + g) & 7) != 7) {   // top partial word
        st[g] = (uint8_t)(cur + 1);
        if (g > 0) cur = step(load(g), cur);
        --g;
    }
    uint32_t Wn[8];
    if (g >= 8) {
#pragma unroll
        for (int j = 0; j < 8; ++j) Wn[j] = load(g - j);
    }
    while (g >= 8) {   // genes g .. g-7 fill one aligned word, all of them have a predecessor
        uint32_t W[8];
#pragma unroll
        for (int j = 0; j < 8; ++j) W[j] = Wn[j];
        if (g - 8 >= 8) {
#pragma unroll
            for (int j = 0; j < 8; ++j) Wn[j] = load(g - 8 - j);   // the next word's back-pointers, a word ahead
        }
        uint32_t lo = 0, hi = 0;
#pragma unroll
        for (int j = 0; j < 8; ++j) {
            const int byte = 7 - j;
            if (byte >= 4) hi |= (uint32_t)cur << (8 * (byte - 4));
            else lo |= (uint32_t)cur << (8 * byte);
            cur = step(W[j], cur);
        }
        uint2 v;
        v.x = lo + 0x01010101u;
        v.y = hi + 0x01010101u;
        *reinterpret_cast<uint2 *>(st + g - 7) = v;
        g -= 8;
    }
    while (g >= 0) {   // bottom partial word and gene 0's group
        st[g] = (uint8_t)(cur + 1);
        if (g > 0) cur = step(load(g), cur);
        --g;
    }
}

}  // namespace icnv
