// Microbenchmark: cost of DEPENDENT vector instructions as a function of the independent chains a wave offers and of the waves
// resident per SIMD.  N waves in all, each running `work` rounds of CH independent fma chains (CH fmas per round, each depending
// on the previous round's result of its chain).  If the SIMD interleaved waves freely, the time would be work * CH * 4 cycles * N /
// 1024 SIMDs whatever CH is.
// hipcc --offload-arch=gfx950 -O3 tools/micro/dep_issue.hip -o tools/micro/dep_issue.bin
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>

constexpr int UNROLL = 32;   // rounds per loop trip: the loop's own scalar instructions and its taken branch (~30 cycles of a wave) stay below 3 %
template <int CH, int WG, int OP>
__global__ void __launch_bounds__(WG) k(const float* __restrict__ in, float* __restrict__ out, int work)
{
    __shared__ float lds[64];
    float v[CH];
    const float s = in[threadIdx.x & 63], t = in[64 + (threadIdx.x & 63)];
#pragma unroll
    for (int c = 0; c < CH; c++) v[c] = s + (float)c;
    for (int i = 0; i < work / UNROLL; i++) {
#pragma unroll
        for (int u = 0; u < UNROLL; u++)
#pragma unroll
        for (int c = 0; c < CH; c++) {
            if constexpr (OP == 0) asm volatile("v_fma_f32 %0, %0, %1, %2" : "+v"(v[c]) : "v"(s), "v"(t));
            if constexpr (OP == 1) asm volatile("v_mul_f32 %0, %0, %1" : "+v"(v[c]) : "v"(s));
            if constexpr (OP == 2) asm volatile("v_exp_f32 %0, %0" : "+v"(v[c]));
        }
    }
    float r = 0.f;
#pragma unroll
    for (int c = 0; c < CH; c++) r += v[c];
    if (r == 12345.678f) out[blockIdx.x] = r + lds[threadIdx.x & 63];
}

template <int CH, int WG, int OP> void run(int waves, int instrs, const float* in, float* out, int lds_pad)
{
    const int n = waves / (WG / 64), work = instrs / CH;
    hipEvent_t a, b; (void)hipEventCreate(&a); (void)hipEventCreate(&b);
    k<CH, WG, OP><<<n, WG, lds_pad>>>(in, out, work);
    (void)hipEventRecord(a);
    for (int r = 0; r < 10; r++) k<CH, WG, OP><<<n, WG, lds_pad>>>(in, out, work);
    (void)hipEventRecord(b); (void)hipEventSynchronize(b);
    float ms; (void)hipEventElapsedTime(&ms, a, b);
    const double us = ms * 100.0, cyc = us * 2400.0 * 1024.0 / ((double)waves * instrs);
    printf("  op %d  chains %d  workgroup %4d  dyn LDS %6d : %8.2f us = %5.2f cycles (at 2.4 GHz) per wave-instruction per SIMD\n", OP, CH, WG, lds_pad, us, cyc);
}

int main(int argc, char** argv)
{
    const int waves = 57344, instrs = 1024;   // 56 waves per SIMD
    float *in, *out; (void)hipMalloc(&in, 4096 * 4); (void)hipMalloc(&out, (size_t)waves * 4); (void)hipMemset(in, 0, 4096 * 4);
    printf("%d waves x %d instructions\n", waves, instrs);
    run<1, 64, 0>(waves, instrs, in, out, 0);
    run<2, 64, 0>(waves, instrs, in, out, 0);
    run<4, 64, 0>(waves, instrs, in, out, 0);
    run<8, 64, 0>(waves, instrs, in, out, 0);
    run<1, 256, 0>(waves, instrs, in, out, 0);
    run<2, 256, 0>(waves, instrs, in, out, 0);
    run<4, 256, 0>(waves, instrs, in, out, 0);
    run<1, 1024, 0>(waves, instrs, in, out, 0);
    run<4, 1024, 0>(waves, instrs, in, out, 0);
    printf("one-wave workgroups held to 4 / 2 / 1 per SIMD by dynamic LDS:\n");
    run<1, 64, 0>(waves, instrs, in, out, 9 * 1024);
    run<1, 64, 0>(waves, instrs, in, out, 19 * 1024);
    run<1, 64, 0>(waves, instrs, in, out, 39 * 1024);
    run<4, 64, 0>(waves, instrs, in, out, 39 * 1024);
    printf("multiply / exp:\n");
    run<1, 64, 1>(waves, instrs, in, out, 0);
    run<4, 64, 1>(waves, instrs, in, out, 0);
    run<1, 64, 2>(waves, instrs, in, out, 0);
    run<4, 64, 2>(waves, instrs, in, out, 0);
    return 0;
}
