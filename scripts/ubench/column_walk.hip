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
// The same 64 lines per wavefront and visit, requested by rows: instruction q fetches the 128-byte lines of the columns
// 8 q .. 8 q + 7 whole (eight lanes x 16 bytes per line) instead of 16 bytes of each of the 64 lines -- every line is asked
// for once, by one instruction (round 5: would a transposing load shape lift the pattern's ceiling?).
__global__ void walk_rows(const double *x, long long G, long long ncols, int *counter, double *sink) {
    const int lane = threadIdx.x & 63;
    double acc = 0.0;
    const long long ncg = ncols / 64;
    for (;;) {
        int task = 0;
        if (lane == 0) task = atomicAdd(counter, 1);
        task = __builtin_amdgcn_readfirstlane(task);
        if (task >= ncg) break;
        const double *col = x + (task * 64ll + (lane >> 3)) * G + 2 * (lane & 7);
        for (long long g = 0; g + 16 <= G; g += 16) {
#pragma unroll
            for (int q = 0; q < 8; ++q) {
                const dbl2 v = *reinterpret_cast<const dbl2 *>(col + (long long)(8 * q) * G + g);
                acc += v.x + v.y;
            }
        }
    }
    if (acc == 12345.678) sink[0] = acc;
}
// ... and the form a kernel could use: the eight row requests go straight into LDS (global_load_lds_dwordx4: lane l's 16 bytes land
// at base + 16 l, so a request's eight lines lie behind each other, column after column), every lane then reads its own column back
// (8 x ds_read_b128).  Swizzle: the lane that fetches pair p of column r of request q sits at slot 8 r + (p ^ f), f = (r >> 1) | ((q & 1) << 2),
// so that the sixteen lanes of a ds_read_b128 pass hit sixteen different bank groups.  `lds_reserved` bytes stand for the emission table.
__global__ void walk_lds(const double *x, long long G, long long ncols, int *counter, double *sink, int lds_reserved) {
    extern __shared__ __attribute__((aligned(16))) double dyn[];
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    double *wbuf = dyn + lds_reserved / 8 + wave * 1024;   // 8 KiB per wavefront
    typedef __attribute__((address_space(3))) void *lds_t;
    typedef const __attribute__((address_space(1))) void *glb_t;
    double acc = 0.0;
    const long long ncg = ncols / 64;
    const int r = lane >> 3, pp = lane & 7;
    // what this lane fetches in request q: column 8 q + r, pair pp ^ f(r, q)
    const int pe = pp ^ (r >> 1), po = pp ^ ((r >> 1) | 4);
    // what this lane (= column `lane` of the task) reads back: request q = lane >> 3, column r' = lane & 7
    const int rq = lane >> 3, rr = lane & 7;
    const int rf = (rr >> 1) | ((rq & 1) << 2);
    const double *rbase = wbuf + rq * 128 + rr * 16;
    for (;;) {
        int task = 0;
        if (lane == 0) task = atomicAdd(counter, 1);
        task = __builtin_amdgcn_readfirstlane(task);
        if (task >= ncg) break;
        const double *ce = x + (task * 64ll + r) * G + 2 * pe;
        const double *co = x + (task * 64ll + r) * G + 2 * po;
        auto request = [&](long long g) {
#pragma unroll
            for (int q = 0; q < 8; ++q)
                __builtin_amdgcn_global_load_lds((glb_t)(((q & 1) ? co : ce) + (long long)(8 * q) * G + g), (lds_t)(wbuf + q * 128), 16, 0, 0);
        };
        request(0);
        for (long long g = 0; g + 16 <= G; g += 16) {
            __builtin_amdgcn_s_waitcnt(0x0070);   // vmcnt(0)
            dbl2 v[8];
#pragma unroll
            for (int p = 0; p < 8; ++p) v[p] = *reinterpret_cast<const dbl2 *>(rbase + 2 * (p ^ rf));
            asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
            if (g + 32 <= G) request(g + 16);
#pragma unroll
            for (int p = 0; p < 8; ++p) acc += v[p].x + v[p].y;
        }
    }
    if (acc == 12345.678) sink[0] = acc;
}
int main(int argc, char **argv) {
    const bool quick = argc > 1 && argv[1][0] == 'q';   // `column_walk quick`: the product's geometry only (768 lanes per CU, 128-byte visits)
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
    auto run_lds = [&](int nt) {
        (void)hipFuncSetAttribute(reinterpret_cast<const void *>(walk_lds), hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
        const int reserved = 56 * 1024;   // stands for the staged Viterbi's 256-record table
        float best = 1e30f;
        for (int r = 0; r < 4; ++r) {
            (void)hipMemset(counter, 0, 4);
            (void)hipEventRecord(e0);
            hipLaunchKernelGGL(walk_lds, dim3(256), dim3(nt), reserved + (nt / 64) * 8192, 0, x, G, (C / 64) * 64, counter, sink, reserved);
            (void)hipEventRecord(e1); (void)hipEventSynchronize(e1);
            float ms; (void)hipEventElapsedTime(&ms, e0, e1);
            if (r > 0 && ms < best) best = ms;
        }
        printf("rows via LDS-DMA       %4d threads x 256 workgroups  %.3f ms  %.2f TB/s\n", nt, best, (C / 64) * 64 * G * 8.0 / best / 1e9);
    };
    // `column_walk quick`: the product's two geometries only -- per-lane 128-byte visits (register kernel) and whole lines by
    // rows through LDS-DMA (staged kernel), 768 lanes per CU (the clocks ramp up over the first runs: bench.py takes the best line)
    if (quick) { for (int r = 0; r < 3; ++r) { run("visit 128 B", walk<128>, 768, 256); run_lds(768); } return 0; }
    if (argc > 1 && argv[1][0] == 'r') {
        for (int nt : {512, 768, 1024}) { run("visit 128 B", walk<128>, nt, 256); run("128 B by rows", walk_rows, nt, 256); }
        for (int nt : {512, 768}) run_lds(nt);
        return 0;
    }
    for (int nt : {512, 768, 1024})
        for (int cus : {128, 256}) {
            run("visit  64 B", walk<64>, nt, cus);
            run("visit 128 B", walk<128>, nt, cus);
            run("visit 256 B", walk<256>, nt, cus);
            run("visit 512 B", walk<512>, nt, cus);
        }
    return 0;
}
