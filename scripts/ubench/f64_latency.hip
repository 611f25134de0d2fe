// Microbenchmark: latency of DEPENDENT vector instructions on gfx950 (developer tool): a chain of NCH independent
// dependent-chains per wavefront (NCH = 1: every instruction waits for the previous one), W wavefronts per SIMD.
//   hipcc --offload-arch=gfx950 -O3 -o f64_latency f64_latency.hip && ./f64_latency
#include <hip/hip_runtime.h>
#include <cstdio>
#pragma clang diagnostic ignored "-Wunused-value"
template <int OP, int NCH>
__global__ void k(double *out, int iters, double seed) {
    double a[NCH];
#pragma unroll
    for (int i = 0; i < NCH; ++i) a[i] = seed + threadIdx.x * 1e-3 + i;
    double b = seed * 0.5 + 1.0;
    unsigned u = threadIdx.x;
    for (int it = 0; it < iters; ++it) {
#pragma unroll
        for (int r = 0; r < 16 / NCH; ++r)
#pragma unroll
            for (int i = 0; i < NCH; ++i) {
                if (OP == 0) asm volatile("v_fma_f64 %0, %0, %1, %1" : "+v"(a[i]) : "v"(b));
                if (OP == 1) asm volatile("v_add_f64 %0, %0, %1" : "+v"(a[i]) : "v"(b));
                if (OP == 2) asm volatile("v_max_f64 %0, %0, %1" : "+v"(a[i]) : "v"(b));
                if (OP == 3) asm volatile("v_cmp_lt_f64 vcc, %0, %1\n\tv_cndmask_b32 %2, %2, %2, vcc\n\tv_add_f64 %0, %0, %1" : "+v"(a[i]) : "v"(b), "v"(u) : "vcc");
                if (OP == 4) asm volatile("v_add_u32 %0, %0, %1" : "+v"(u) : "v"(u));
            }
    }
    double s = u;
#pragma unroll
    for (int i = 0; i < NCH; ++i) s += a[i];
    out[blockIdx.x * blockDim.x + threadIdx.x] = s;
}
template <int OP, int NCH>
void run(const char *name, double *d, int waves_per_simd, int instr_per_op) {
    const int iters = 20000, blocks = 256, threads = 256 * waves_per_simd;
    hipEvent_t e0, e1;
    hipEventCreate(&e0); hipEventCreate(&e1);
    hipLaunchKernelGGL((k<OP, NCH>), dim3(blocks), dim3(threads), 0, 0, d, 100, 1.0);
    hipEventRecord(e0);
    hipLaunchKernelGGL((k<OP, NCH>), dim3(blocks), dim3(threads), 0, 0, d, iters, 1.0);
    hipEventRecord(e1);
    hipEventSynchronize(e1);
    float ms; hipEventElapsedTime(&ms, e0, e1);
    // per wavefront: iters * 16 ops in sequence
    const double ns_per_op = ms * 1e6 / ((double)iters * 16);
    printf("%-26s chains=%2d waves/SIMD=%d  %.2f ns per op and wavefront (= %.1f cycles at 2.4 GHz; %d instr per op)\n", name, NCH,
           waves_per_simd, ns_per_op, ns_per_op * 2.4, instr_per_op);
}
int main() {
    double *d; hipMalloc(&d, 256 * 1024 * sizeof(double));
    run<0, 1>("v_fma_f64 dependent", d, 1, 1);
    run<0, 2>("v_fma_f64", d, 1, 1);
    run<0, 4>("v_fma_f64", d, 1, 1);
    run<0, 1>("v_fma_f64 dependent", d, 2, 1);
    run<0, 1>("v_fma_f64 dependent", d, 4, 1);
    run<1, 1>("v_add_f64 dependent", d, 1, 1);
    run<2, 1>("v_max_f64 dependent", d, 1, 1);
    run<4, 1>("v_add_u32 dependent", d, 1, 1);
    run<3, 1>("cmp+cndmask+add dependent", d, 1, 3);
    run<3, 1>("cmp+cndmask+add dependent", d, 2, 3);
    return 0;
}
