// developer microbenchmark (round 4, a design question): would a LINE-TILED layout of the HMM input pay?
//   layout T: [block of 64 cells][line of 16 genes][cell in block][16 genes] -- a cell's 128-byte lines stay whole, but the 64
//   lines a Viterbi wavefront reads per visit (one per lane) are CONTIGUOUS (8 KB) instead of 80 000 bytes apart.
// (1) read side: the Viterbi's per-lane walk over layout T against the column walk over the plain matrix;
// (2) write side: a workgroup per cell writing its 625 lines -- plain: contiguous 80 KB; T: one line per 8 KB tile row -- next to
//     a contiguous read and a contiguous second output, like the fused smooth pass (1 read : 2 writes).
#include <hip/hip_runtime.h>
#include <cstdio>
typedef double dbl2 __attribute__((ext_vector_type(2)));
constexpr long long G = 10000, C = 49984;   // 781 blocks of 64 cells; G = 625 lines of 16 genes
template <int TILED>
__global__ void walk(const double *x, int *counter, double *sink) {
    const int lane = threadIdx.x & 63;
    double acc = 0.0;
    for (;;) {
        int task = 0;
        if (lane == 0) task = atomicAdd(counter, 1);
        task = __builtin_amdgcn_readfirstlane(task);
        if (task >= C / 64) break;
        for (long long line = 0; line < G / 16; ++line) {
            const double *p = TILED ? x + ((long long)task * (G / 16) + line) * 1024 + lane * 16
                                    : x + ((long long)task * 64 + lane) * G + line * 16;
#pragma unroll
            for (int j = 0; j < 8; ++j) { const dbl2 v = *reinterpret_cast<const dbl2 *>(p + 2 * j); acc += v.x + v.y; }
        }
    }
    if (acc == 12345.678) sink[0] = acc;
}
template <int TILED>
__global__ void write_cells(const dbl2 *in, dbl2 *o1, double *o2) {   // one workgroup per cell at a time, 1024 threads
    for (long long c = blockIdx.x; c < C; c += gridDim.x) {
        for (int s = 0; s < 5; ++s) {
            const long long pair = threadIdx.x + 1024ll * s;      // gene pair of this thread (S layout)
            if (pair >= G / 2) break;
            dbl2 v = __builtin_nontemporal_load(in + c * (G / 2) + pair);
            v.x += 1.0;
            __builtin_nontemporal_store(v, o1 + c * (G / 2) + pair);
            const long long g = 2 * pair;
            double *q = TILED ? o2 + ((c >> 6) * (G / 16) + (g >> 4)) * 1024 + (c & 63) * 16 + (g & 15) : o2 + c * G + g;
            __builtin_nontemporal_store(v, reinterpret_cast<dbl2 *>(q));
        }
    }
}
int main() {
    double *x, *o1, *o2, *sink; int *counter;
    const size_t bytes = (size_t)G * C * 8;
    hipMalloc(&x, bytes); hipMalloc(&o1, bytes); hipMalloc(&o2, bytes); hipMalloc(&sink, 8); hipMalloc(&counter, 4);
    hipMemset(x, 0, bytes);
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    auto timeit = [&](const char *name, auto launch, double gb) {
        float best = 1e30f;
        for (int r = 0; r < 4; ++r) {
            hipMemset(counter, 0, 4);
            hipEventRecord(e0); launch(); hipEventRecord(e1); hipEventSynchronize(e1);
            float ms; hipEventElapsedTime(&ms, e0, e1);
            if (r > 0 && ms < best) best = ms;
        }
        printf("%-64s %.3f ms  %.2f TB/s\n", name, best, gb / best / 1e9 * 1e3 / 1e3);
    };
    timeit("read, plain matrix (column walk, 768 threads x 256, 128 B visits)", [&] { hipLaunchKernelGGL(walk<0>, dim3(256), dim3(768), 0, 0, x, counter, sink); }, bytes / 1e0 / 1e3);
    timeit("read, line-tiled layout (same kernel, contiguous 8 KB per visit)", [&] { hipLaunchKernelGGL(walk<1>, dim3(256), dim3(768), 0, 0, x, counter, sink); }, bytes / 1e0 / 1e3);
    timeit("1 read : 2 writes per cell, both outputs plain (256 x 1024)", [&] { hipLaunchKernelGGL(write_cells<0>, dim3(256), dim3(1024), 0, 0, (const dbl2 *)x, (dbl2 *)o1, o2); }, 3.0 * bytes / 1e3);
    timeit("1 read : 2 writes per cell, second output line-tiled (256 x 1024)", [&] { hipLaunchKernelGGL(write_cells<1>, dim3(256), dim3(1024), 0, 0, (const dbl2 *)x, (dbl2 *)o1, o2); }, 3.0 * bytes / 1e3);
    timeit("... the same with 2048 workgroups", [&] { hipLaunchKernelGGL(write_cells<0>, dim3(2048), dim3(1024), 0, 0, (const dbl2 *)x, (dbl2 *)o1, o2); }, 3.0 * bytes / 1e3);
    timeit("... tiled, 2048 workgroups", [&] { hipLaunchKernelGGL(write_cells<1>, dim3(2048), dim3(1024), 0, 0, (const dbl2 *)x, (dbl2 *)o1, o2); }, 3.0 * bytes / 1e3);
    return 0;
}
