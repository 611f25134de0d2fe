// LDS atomic throughput on gfx950 (developer microbenchmark; build: hipcc --offload-arch=gfx950 -O3 -o lds_atomic lds_atomic.hip).
// One workgroup of 768 threads per CU (like the chain kernel); every thread issues ROUNDS x 15 ds_add_u32 into a 2048-bin
// LDS histogram.  mode 0: random bins, all lanes; 1: random bins, every 4th lane active; 2: all lanes one address;
// 3: 8 lanes of each wavefront share one address, the rest random; 4: ds_read_b64 instead of the atomic (reference rate);
// 5: random bins, bins clustered in a 600-bin window (the median histogram's shape).
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
constexpr int NT = 768, ROUNDS = 200, PER = 15, NB = 2048;
__global__ void __launch_bounds__(NT) k(int mode, unsigned long long *cycles, unsigned *sink) {
    __shared__ unsigned hist[NB];
    __shared__ double vals[NT * PER];
    const int t = threadIdx.x;
    for (int i = t; i < NB; i += NT) hist[i] = 0;
    for (int i = t; i < NT * PER; i += NT) vals[i] = i;
    __syncthreads();
    unsigned s = t * 2654435761u + blockIdx.x * 40503u + 12345u;
    unsigned idx[PER];
    for (int q = 0; q < PER; ++q) {
        s = s * 1664525u + 1013904223u;
        unsigned b = (s >> 10) % NB;
        if (mode == 5) b = 700 + (s >> 10) % 600;
        if (mode == 2) b = 77;
        if (mode == 3 && (t & 63) < 8) b = 77;
        idx[q] = b;
    }
    double acc = 0;
    __syncthreads();
    const unsigned long long t0 = __builtin_readcyclecounter();
    for (int r = 0; r < ROUNDS; ++r) {
#pragma unroll
        for (int q = 0; q < PER; ++q) {
            if (mode == 4) acc += vals[(idx[q] * 5 + t) % (NT * PER)];
            else if (mode != 1 || (t & 3) == 0) atomicAdd(&hist[idx[q]], 1u);
        }
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
    }
    __syncthreads();
    const unsigned long long t1 = __builtin_readcyclecounter();
    if (t == 0) cycles[blockIdx.x] = t1 - t0;
    if (acc == 12345.678) sink[0] = 1;
    if (t < 4) sink[1 + t] = hist[t];
}
int main() {
    unsigned long long *c; unsigned *s;
    hipMalloc(&c, 256 * 8); hipMalloc(&s, 64);
    for (int mode = 0; mode < 6; ++mode) {
        hipLaunchKernelGGL(k, dim3(256), dim3(NT), 0, 0, mode, c, s);
        hipDeviceSynchronize();
        unsigned long long h[256];
        hipMemcpy(h, c, sizeof(h), hipMemcpyDeviceToHost);
        double avg = 0; for (int i = 0; i < 256; ++i) avg += h[i];
        avg /= 256;
        const double per_instr = avg / (ROUNDS * PER * (NT / 64));   // cycles of the CU's LDS per wavefront-instruction
        printf("mode %d: %.0f cycles per block, %.1f cycles per wavefront instruction (12 wavefronts sharing the LDS)\n", mode, avg, per_instr);
    }
    return 0;
}
