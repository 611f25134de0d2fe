// Microbenchmark: issue rate of fp64 vector instructions on gfx950 (developer tool).
//   hipcc --offload-arch=gfx950 -O3 -o f64_rate f64_rate.hip && ./f64_rate
#include <hip/hip_runtime.h>
#include <cstdio>
template <int OP>
__global__ void k(double *out, int iters, double seed) {
    double a[16];
#pragma unroll
    for (int i = 0; i < 16; ++i) a[i] = seed + threadIdx.x * 1e-3 + i;
    double b = seed * 0.5 + 1.0;
    for (int it = 0; it < iters; ++it) {
#pragma unroll
        for (int i = 0; i < 16; ++i) {
            if (OP == 0) asm volatile("v_fma_f64 %0, %0, %1, %1" : "+v"(a[i]) : "v"(b));
            if (OP == 1) asm volatile("v_min_f64 %0, %0, %1" : "+v"(a[i]) : "v"(b));
            if (OP == 2) asm volatile("v_max_f64 %0, %0, %1" : "+v"(a[i]) : "v"(b));
            if (OP == 3) asm volatile("v_add_f64 %0, %0, %1" : "+v"(a[i]) : "v"(b));
            if (OP == 4) asm volatile("v_cmp_lt_f64 vcc, %0, %1" : "+v"(a[i]) : "v"(b) : "vcc");
            if (OP == 5) asm volatile("v_mul_f64 %0, %0, %1" : "+v"(a[i]) : "v"(b));
            if (OP == 6) asm volatile("v_pk_min_u16 %0, %0, %1" : "+v"(*reinterpret_cast<unsigned *>(&a[i])) : "v"(*reinterpret_cast<unsigned *>(&b)));
            if (OP == 7) asm volatile("v_pk_max_u16 %0, %0, %1" : "+v"(*reinterpret_cast<unsigned *>(&a[i])) : "v"(*reinterpret_cast<unsigned *>(&b)));
            if (OP == 8) asm volatile("v_min_u32 %0, %0, %1" : "+v"(*reinterpret_cast<unsigned *>(&a[i])) : "v"(*reinterpret_cast<unsigned *>(&b)));
            if (OP == 9) asm volatile("v_max_u32 %0, %0, %1" : "+v"(*reinterpret_cast<unsigned *>(&a[i])) : "v"(*reinterpret_cast<unsigned *>(&b)));
            if (OP == 10) asm volatile("v_cmp_lt_u64 vcc, %0, %1" : "+v"(a[i]) : "v"(b) : "vcc");
            if (OP == 11) asm volatile("v_pk_min_i16 %0, %0, %1" : "+v"(*reinterpret_cast<unsigned *>(&a[i])) : "v"(*reinterpret_cast<unsigned *>(&b)));
            if (OP == 12) asm volatile("v_min3_u32 %0, %0, %1, %1" : "+v"(*reinterpret_cast<unsigned *>(&a[i])) : "v"(*reinterpret_cast<unsigned *>(&b)));
            if (OP == 13) asm volatile("v_med3_u32 %0, %0, %1, %1" : "+v"(*reinterpret_cast<unsigned *>(&a[i])) : "v"(*reinterpret_cast<unsigned *>(&b)));
        }
    }
    double s = 0;
#pragma unroll
    for (int i = 0; i < 16; ++i) s += a[i];
    out[blockIdx.x * blockDim.x + threadIdx.x] = s;
}
template <int OP>
void run(const char *name, double *d, int per_iter) {
    const int iters = 20000, blocks = 256 * 8, threads = 256;
    hipEvent_t e0, e1;
    hipEventCreate(&e0); hipEventCreate(&e1);
    hipLaunchKernelGGL(k<OP>, dim3(blocks), dim3(threads), 0, 0, d, 100, 1.0);
    hipEventRecord(e0);
    hipLaunchKernelGGL(k<OP>, dim3(blocks), dim3(threads), 0, 0, d, iters, 1.0);
    hipEventRecord(e1);
    hipEventSynchronize(e1);
    float ms; hipEventElapsedTime(&ms, e0, e1);
    const double wave_instr = (double)blocks * threads / 64 * iters * 16 * per_iter;
    // 256 CUs x 4 SIMDs
    const double per_simd = wave_instr / 1024.0;
    printf("%-22s %8.3f ms  %.2f ns per wave-instruction per SIMD  (= %.2f cycles at 2.4 GHz)\n", name, ms,
           ms * 1e6 / per_simd, ms * 1e6 / per_simd * 2.4);
}
int main() {
    double *d; hipMalloc(&d, 256 * 8 * 256 * sizeof(double));
    run<0>("v_fma_f64", d, 1);
    run<1>("v_min_f64", d, 1);
    run<2>("v_max_f64", d, 1);
    run<3>("v_add_f64", d, 1);
    run<5>("v_mul_f64", d, 1);
    run<4>("v_cmp_lt_f64", d, 1);
    run<6>("v_pk_min_u16", d, 1);
    run<7>("v_pk_max_u16", d, 1);
    run<11>("v_pk_min_i16", d, 1);
    run<8>("v_min_u32", d, 1);
    run<9>("v_max_u32", d, 1);
    run<12>("v_min3_u32", d, 1);
    run<13>("v_med3_u32", d, 1);
    run<10>("v_cmp_lt_u64", d, 1);
    return 0;
}
