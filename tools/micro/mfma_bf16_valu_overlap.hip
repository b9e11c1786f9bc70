// Microbenchmark: do v_mfma_f32_16x16x32_bf16 and ordinary vector instructions of DIFFERENT waves on one SIMD overlap?
// hipcc --offload-arch=gfx950 -O3 tools/micro/mfma_bf16_valu_overlap.hip -o tools/micro/mfma_valu_overlap.bin
// One workgroup of 8 waves per CU = two waves per SIMD (wave w -> SIMD w % 4).  Roles per wave:
//   mode 0: all eight waves run MFMA chains        mode 1: all eight run FMA chains
//   mode 2: waves 0-3 MFMA, waves 4-7 FMA (one of each per SIMD)
//   mode 3: every wave alternates 16 MFMA and 64 FMA (independent of each other) in its own stream
//   mode 4 / 5: only four waves (one per SIMD) run MFMA resp. FMA
#include <hip/hip_runtime.h>
#include <cstdio>

typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));

__device__ __forceinline__ void mfma_block(f32x4& a0, f32x4& a1, bf16x8 x, bf16x8 y)
{
#pragma unroll
    for (int t = 0; t < 8; t++) {
        a0 = __builtin_amdgcn_mfma_f32_16x16x32_bf16(x, y, a0, 0, 0, 0);
        a1 = __builtin_amdgcn_mfma_f32_16x16x32_bf16(y, x, a1, 0, 0, 0);
    }
}

__device__ __forceinline__ void fma_block(float (&v)[8], float s)
{
#pragma unroll
    for (int t = 0; t < 8; t++) {
#pragma unroll
        for (int i = 0; i < 8; i++) v[i] = __builtin_fmaf(v[i], s, 0.5f);
    }
}

template <int MODE>
__global__ void __launch_bounds__(512) k(int iters, float* out)
{
    const int wave = threadIdx.x >> 6;
    f32x4 a0 = {0.f, 0.f, 0.f, 0.f}, a1 = a0;
    float v[8];
    for (int i = 0; i < 8; i++) v[i] = (float)(threadIdx.x + i);
    const float y = 0.999f;
    bf16x8 bx, by;
    for (int i = 0; i < 8; i++) { bx[i] = (__bf16)(1.0f + threadIdx.x * 1e-3f + i); by[i] = (__bf16)(0.5f + i); }
    const bool do_m = MODE == 0 || (MODE == 2 && wave < 4) || MODE == 3 || (MODE == 4 && wave < 4);
    const bool do_f = MODE == 1 || (MODE == 2 && wave >= 4) || MODE == 3 || (MODE == 5 && wave < 4);
    for (int it = 0; it < iters; it++) {
        if (do_m) mfma_block(a0, a1, bx, by);   // 16 bf16 MFMA (16x16x32)
        if (do_f) fma_block(v, y);              // 64 FMA  = 64 x 4 cycles of vector pipe
    }
    float r = a0[0] + a1[1] + a0[2] + a1[3];
    for (int i = 0; i < 8; i++) r += v[i];
    if (r == 12345.678f) out[0] = r;
}

template <int MODE> float run(int iters, float* d_out)
{
    hipEvent_t a, b; hipEventCreate(&a); hipEventCreate(&b);
    k<MODE><<<256, 512>>>(10, d_out);
    hipEventRecord(a);
    k<MODE><<<256, 512>>>(iters, d_out);
    hipEventRecord(b); hipEventSynchronize(b);
    float ms; hipEventElapsedTime(&ms, a, b); return ms;
}

int main()
{
    float* d_out; hipMalloc(&d_out, 4096);
    const int iters = 20000;
    const float t[6] = {run<0>(iters, d_out), run<1>(iters, d_out), run<2>(iters, d_out), run<3>(iters, d_out), run<4>(iters, d_out),
                        run<5>(iters, d_out)};
    const char* names[6] = {"8 waves MFMA", "8 waves FMA", "4 MFMA + 4 FMA waves", "each wave MFMA then FMA", "4 waves MFMA", "4 waves FMA"};
    for (int m = 0; m < 6; m++) printf("%-26s %8.3f ms  = %7.1f ns per iteration (16 MFMA and/or 64 FMA per wave)\n", names[m], t[m], t[m] * 1e6 / iters);
    return 0;
}
