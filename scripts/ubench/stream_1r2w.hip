// developer microbenchmark: what HBM gives a 1-read : 2-write stream (the fused pass writes the denoised matrix and the HMM
// input for every matrix it reads), grid-stride with 16-byte accesses, plain and non-temporal
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#pragma clang diagnostic ignored "-Wunused-value"
typedef double dbl2 __attribute__((ext_vector_type(2)));
template <int NT_HINT, int NW>
__global__ void k(const dbl2 *in, dbl2 *o1, dbl2 *o2, size_t n) {
    for (size_t i = blockIdx.x * (size_t)blockDim.x + threadIdx.x; i < n; i += (size_t)gridDim.x * blockDim.x) {
        dbl2 v = NT_HINT ? __builtin_nontemporal_load(in + i) : in[i];
        v.x += 1.0;
        if (NT_HINT) { __builtin_nontemporal_store(v, o1 + i); if (NW > 1) __builtin_nontemporal_store(v, o2 + i); }
        else { o1[i] = v; if (NW > 1) o2[i] = v; }
    }
}
int main(int argc, char **argv) {
    const bool quick = argc > 1;   // `stream_1r2w quick`: the two best geometries only (bench.py measures its ceiling with it in every run)
    const size_t bytes = 3600000000ull, n = bytes / 16;
    dbl2 *in, *o1, *o2;
    if (hipMalloc(&in, bytes) != hipSuccess || hipMalloc(&o1, bytes) != hipSuccess || hipMalloc(&o2, bytes) != hipSuccess) { fprintf(stderr, "hipMalloc failed\n"); return 1; }
    hipMemset(in, 0, bytes);
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    auto run = [&](const char *name, auto kern, int nw, int grid, int block) {
        for (int r = 0; r < 2; ++r) hipLaunchKernelGGL(kern, dim3(grid), dim3(block), 0, 0, in, o1, o2, n);
        hipEventRecord(e0);
        for (int r = 0; r < 5; ++r) hipLaunchKernelGGL(kern, dim3(grid), dim3(block), 0, 0, in, o1, o2, n);
        hipEventRecord(e1); hipEventSynchronize(e1);
        float ms; hipEventElapsedTime(&ms, e0, e1); ms /= 5;
        printf("%-28s grid %5d x %4d  %.3f ms  %.2f TB/s\n", name, grid, block, ms, bytes * (1.0 + nw) / ms / 1e9);
    };
    for (int grid : {256, 1024, 4096, 16384}) {
        if (quick && grid < 4096) continue;
        run("1r1w plain", k<0, 1>, 1, grid, 1024);
        run("1r1w nt", k<1, 1>, 1, grid, 1024);
        run("1r2w plain", k<0, 2>, 2, grid, 1024);
        run("1r2w nt", k<1, 2>, 2, grid, 1024);
    }
    return 0;
}
