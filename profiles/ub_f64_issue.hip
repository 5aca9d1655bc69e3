// Microbenchmark (diagnostic, not part of the product): what a dependent chain of f64 vector ops costs a wave on gfx950, what
// several independent chains cost it, and what the whole chip reaches with 1 / 4 waves per SIMD.
//   hipcc --offload-arch=gfx950 -O3 -ffp-contract=off profiles/ub_f64_issue.hip -o /tmp/ub && /tmp/ub
// Read on one MI355X (profiles/r03_ab_notes.txt 13): a dependent v_fma_f64 every 8 cycles, four independent chains one per
// 4.9; 1,024 blocks x 16 waves of ONE dependent chain each 72.9 TFLOP/s, one wave per SIMD with four chains 63 (peak 78.6):
// the render kernel's idle issue cycles are not the latency of its f64 chains.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>
template <int CHAINS, int KIND>
__global__ void k(double *out, unsigned long long *cyc, int iters, double seed) {
    double x[CHAINS];
    for (int c = 0; c < CHAINS; c++) x[c] = seed + c + threadIdx.x;
    const double a = 1.0000001, b = 1e-9;
    unsigned long long t0 = __builtin_readcyclecounter(); unsigned long long r0 = __builtin_amdgcn_s_memrealtime();
    for (int i = 0; i < iters; i++) {
#pragma unroll
        for (int u = 0; u < 8; u++) {
#pragma unroll
            for (int c = 0; c < CHAINS; c++) {
                if (KIND == 0) x[c] = __builtin_fma(x[c], a, b);
                else if (KIND == 1) x[c] = x[c] * a;
                else if (KIND == 2) x[c] = x[c] + b;
                else if (KIND == 3) x[c] = __builtin_amdgcn_rsq(x[c]) + 2.0;
                else if (KIND == 4) x[c] = __builtin_amdgcn_rcp(x[c]) + 2.0;
                else if (KIND == 5) x[c] = __builtin_sqrt(x[c]) + 2.0;
                else if (KIND == 6) x[c] = 3.0 / x[c] + 2.0;
            }
        }
    }
    unsigned long long t1 = __builtin_readcyclecounter(); unsigned long long r1 = __builtin_amdgcn_s_memrealtime();
    double s = 0; for (int c = 0; c < CHAINS; c++) s += x[c];
    out[blockIdx.x * blockDim.x + threadIdx.x] = s;
    if (threadIdx.x == 0) { cyc[2*blockIdx.x] = t1 - t0; cyc[2*blockIdx.x+1] = r1 - r0; }
}
template <int CHAINS, int KIND>
void run(const char *name, int waves_per_block) {
    double *out; unsigned long long *cyc;
    hipMalloc(&out, 1 << 20); hipMalloc(&cyc, 8 * 1024);
    const int iters = 20000, blocks = 1;
    k<CHAINS, KIND><<<blocks, 64 * waves_per_block>>>(out, cyc, iters, 1.5);
    k<CHAINS, KIND><<<blocks, 64 * waves_per_block>>>(out, cyc, iters, 1.5);
    hipDeviceSynchronize();
    unsigned long long hh[2]; hipMemcpy(hh, cyc, 16, hipMemcpyDeviceToHost); unsigned long long h = hh[0];
    const double ops = (double)iters * 8 * CHAINS;
    printf("%-10s chains %d waves/block %d: %.2f ticks per op per wave; %.2f ns per op per wave (clock %.2f GHz)\n", name, CHAINS, waves_per_block, (double)h / ops, (double)hh[1] * 10.0 / ops, (double)h / ((double)hh[1] * 10.0));
    hipFree(out); hipFree(cyc);
}
template <int CHAINS>
void chip(int blocks, int wpb) {
    double *out; unsigned long long *cyc;
    hipMalloc(&out, (size_t)blocks * wpb * 64 * 8); hipMalloc(&cyc, (size_t)blocks * 16);
    const int iters = 20000;
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    k<CHAINS, 0><<<blocks, 64 * wpb>>>(out, cyc, iters, 1.5);
    hipEventRecord(e0);
    k<CHAINS, 0><<<blocks, 64 * wpb>>>(out, cyc, iters, 1.5);
    hipEventRecord(e1); hipDeviceSynchronize();
    float ms; hipEventElapsedTime(&ms, e0, e1);
    const double fma = (double)blocks * wpb * 64 * iters * 8 * CHAINS;
    printf("chip: %d blocks x %d waves, %d chains: %.3f ms, %.1f TFLOP/s f64 (2 flops per fma)\n", blocks, wpb, CHAINS, ms, fma * 2 / ms / 1e9);
}
int main() {
    chip<4>(256, 16); chip<4>(1024, 16); chip<1>(1024, 16); chip<4>(256, 4); chip<8>(256, 8);
    run<1, 0>("fma", 1); run<2, 0>("fma", 1); run<4, 0>("fma", 1); run<8, 0>("fma", 1);
    run<1, 0>("fma", 4); run<1, 0>("fma", 8); run<1, 0>("fma", 16); run<4, 0>("fma", 16);
    run<1, 1>("mul", 1); run<4, 1>("mul", 1); run<1, 2>("add", 1); run<4, 2>("add", 1);
    run<1, 3>("rsq+add", 1); run<4, 3>("rsq+add", 1); run<1, 4>("rcp+add", 1); run<4, 4>("rcp+add", 1);
    run<1, 5>("sqrt+add", 1); run<4, 5>("sqrt+add", 1); run<1, 6>("div+add", 1); run<4, 6>("div+add", 1);
    run<4, 5>("sqrt+add", 16); run<4, 6>("div+add", 16);
    return 0;
}
