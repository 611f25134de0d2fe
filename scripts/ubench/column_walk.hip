// developer microbenchmark: what the memory system gives the Viterbi's access pattern -- every LANE walks a column of its
// own (80 000 bytes apart, like the cells of a 10 000-gene matrix), V bytes per visit, one persistent workgroup of NT
// threads per CU -- as a function of the visit size V and of the number of concurrent lane streams.  No arithmetic between
// the visits: this is the pattern's ceiling.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
typedef double dbl2 __attribute__((ext_vector_type(2)));
template <int V>   // bytes per visit (multiple of 16)
__global__ void walk(const double *x, long long G, long long ncols, int *counter, double *sink) {
    const int lane = threadIdx.x & 63;
    __shared__ int s_task[32];
    double acc = 0.0;
    const long long ncg = ncols / 64;
    for (;;) {
        int task = 0;
        if (lane == 0) task = atomicAdd(counter, 1);
        task = __builtin_amdgcn_readfirstlane(task);
        if (task >= ncg) break;
        const double *col = x + (task * 64ll + lane) * G;
        for (long long g = 0; g + V / 8 <= G; g += V / 8) {
#pragma unroll
            for (int j = 0; j < V / 16; ++j) {
                const dbl2 v = *reinterpret_cast<const dbl2 *>(col + g + 2 * j);
                acc += v.x + v.y;
            }
        }
    }
    if (acc == 12345.678) sink[0] = acc;
    (void)s_task;
}
int main(int argc, char **argv) {
    const bool quick = argc > 1;   // `column_walk quick`: the product's geometry only (768 lanes per CU, 128-byte visits)
    const long long G = 10000, C = 50000;
    double *x, *sink;
    int *counter;
    if (hipMalloc(&x, G * C * 8) != hipSuccess || hipMalloc(&sink, 8) != hipSuccess || hipMalloc(&counter, 4) != hipSuccess) { fprintf(stderr, "hipMalloc failed\n"); return 1; }
    hipMemset(x, 0, G * C * 8);
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    auto run = [&](const char *name, auto kern, int nt, int cus) {
        float best = 1e30f;
        for (int r = 0; r < 4; ++r) {
            hipMemset(counter, 0, 4);
            hipEventRecord(e0);
            hipLaunchKernelGGL(kern, dim3(cus), dim3(nt), 0, 0, x, G, (C / 64) * 64, counter, sink);
            hipEventRecord(e1); hipEventSynchronize(e1);
            float ms; hipEventElapsedTime(&ms, e0, e1);
            if (r > 0 && ms < best) best = ms;
        }
        printf("%-22s %4d threads x %3d workgroups  %.3f ms  %.2f TB/s\n", name, nt, cus, best, (C / 64) * 64 * G * 8.0 / best / 1e9);
    };
    if (quick) { for (int r = 0; r < 4; ++r) run("visit 128 B", walk<128>, 768, 256); return 0; }   // (the clocks ramp up over the first runs: bench.py takes the best line)
    for (int nt : {512, 768, 1024})
        for (int cus : {128, 256}) {
            run("visit  64 B", walk<64>, nt, cus);
            run("visit 128 B", walk<128>, nt, cus);
            run("visit 256 B", walk<256>, nt, cus);
            run("visit 512 B", walk<512>, nt, cus);
        }
    return 0;
}
