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

// The same walk when every lane's state column has the SAME byte alignment a0 = address & 15 (always the case when
// the number of genes per column is a multiple of 16): the position of a gene inside its 16-byte output word is then
// wave-uniform, so a block of 32 genes is assembled with immediate shifts (one v_lshl_or per gene, +1 per byte added
// once per dword) and stored as two aligned 16-byte stores without per-gene alignment tests -- about 8 vector
// instructions per gene instead of 20.  The back-pointer words of the NEXT
// block of 32 genes are requested before the current block is walked (their addresses do not depend on the traced
// state): 32 lines in flight per wavefront instead of one exposed latency per word.
// Genes in the partial blocks at either end of the sequence, and gene 0's group, are written byte by byte.
template <int TB, class Load, class Step>
__device__ inline void viterbi_traceback_uniform(uint8_t *st, int n, int cur, int a0, Load load, Step step) {
    static_assert(TB % 16 == 0, "whole 16-byte words");
    int g = n - 1;
    while (g >= 0 && ((a0 + g) & 15) != 15) {   // top partial word
        st[g] = (uint8_t)(cur + 1);
        if (g > 0) cur = step(load(g), cur);
        --g;
    }
    uint32_t Wn[TB];
    if (g >= TB) {
#pragma unroll
        for (int j = 0; j < TB; ++j) Wn[j] = load(g - j);
    }
    while (g >= TB) {   // genes g .. g-31 fill one aligned block, all of them have a predecessor
        uint32_t W[TB];
#pragma unroll
        for (int j = 0; j < TB; ++j) W[j] = Wn[j];
        if (g - TB >= TB) {
#pragma unroll
            for (int j = 0; j < TB; ++j) Wn[j] = load(g - TB - j);   // the next block's back-pointers, a block ahead
        }
        uint32_t d[TB / 4];
#pragma unroll
        for (int q = 0; q < TB / 4; ++q) d[q] = 0;
#pragma unroll
        for (int j = 0; j < TB; ++j) {
            const int byte = TB - 1 - j;
            d[byte >> 2] |= (uint32_t)cur << (8 * (byte & 3));
            cur = step(W[j], cur);
        }
        uint8_t *dst = st + g - (TB - 1);
#pragma unroll
        for (int q = 0; q < TB / 16; ++q) {
            uint4 v;
            v.x = d[4 * q] + 0x01010101u; v.y = d[4 * q + 1] + 0x01010101u;
            v.z = d[4 * q + 2] + 0x01010101u; v.w = d[4 * q + 3] + 0x01010101u;
            __builtin_memcpy(dst + 16 * q, &v, 16);   // one 16-byte store; aligned for the lane whose a0 this is, and for all when a0 is shared
        }
        g -= TB;
    }
    while (g >= 0) {   // bottom partial block and gene 0's group
        st[g] = (uint8_t)(cur + 1);
        if (g > 0) cur = step(load(g), cur);
        --g;
    }
}

// Traceback over BLOCK SUMMARIES (certified fast kernel, wave-uniform alignment a0 as above).  The forward pass also writes,
// for every aligned block of 16 genes, the OR of the block's back-pointer words.  Bit `cur` of a summary clear means
// "row cur kept itself at every gene of the block": the traced path stays in `cur` for all 16 genes, whatever the other
// rows did -- 16 output bytes from one 2-byte load and a handful of instructions instead of 16 loads and 16 dependent
// steps.  A path changes its state a few times per sequence, so nearly every block takes that road; the per-gene words
// are read only for a block in which some lane's row did not keep itself (wave-uniform test), and for the partial blocks
// at either end.
// load_sum(b) -> summary of block b = (a0 + gene) >> 4;  note_sum(S, cur): account the summary's "inside the band" bits
template <class Load, class LoadSum, class Step, class NoteSum>
__device__ inline void viterbi_traceback_blocks(uint8_t *st, int n, int cur, int a0, Load load, LoadSum load_sum, Step step,
                                                NoteSum note_sum) {
    constexpr int TB = 16;
    int g = n - 1;
    while (g >= 0 && ((a0 + g) & 15) != 15) {   // top partial word
        st[g] = (uint8_t)(cur + 1);
        if (g > 0) cur = step(load(g), cur);
        --g;
    }
    uint32_t Sn = 0;
    if (g >= TB) Sn = load_sum((a0 + g) >> 4);
    while (g >= TB) {   // genes g .. g-15 fill one aligned 16-byte word, all of them have a predecessor
        const uint32_t S = Sn;
        if (g - TB >= TB) Sn = load_sum((a0 + g - TB) >> 4);   // the next block's summary, a block ahead
        uint8_t *dst = st + g - (TB - 1);   // (16-byte aligned for the lane whose a0 this is, and for all lanes when a0 is shared)
        if (__builtin_amdgcn_ballot_w64(((S >> cur) & 1u) != 0) == 0) {
            note_sum(S, cur);
            const uint32_t v = (uint32_t)(cur + 1) * 0x01010101u;
            uint4 o;
            o.x = v; o.y = v; o.z = v; o.w = v;
            __builtin_memcpy(dst, &o, 16);
        } else {
            uint32_t W[TB];
#pragma unroll
            for (int j = 0; j < TB; ++j) W[j] = load(g - j);
            uint32_t d[TB / 4];
#pragma unroll
            for (int q = 0; q < TB / 4; ++q) d[q] = 0;
#pragma unroll
            for (int j = 0; j < TB; ++j) {
                const int byte = TB - 1 - j;
                d[byte >> 2] |= (uint32_t)cur << (8 * (byte & 3));
                cur = step(W[j], cur);
            }
            uint4 o;
            o.x = d[0] + 0x01010101u; o.y = d[1] + 0x01010101u; o.z = d[2] + 0x01010101u; o.w = d[3] + 0x01010101u;
            __builtin_memcpy(dst, &o, 16);
        }
        g -= TB;
    }
    while (g >= 0) {   // bottom partial block and gene 0's group
        st[g] = (uint8_t)(cur + 1);
        if (g > 0) cur = step(load(g), cur);
        --g;
    }
}

}  // namespace icnv
