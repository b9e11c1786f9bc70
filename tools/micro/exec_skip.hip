// Microbenchmark: does a wave64 vector instruction cost less when only some 16-lane quarters of EXEC are active?
// hipcc --offload-arch=gfx950 -O3 tools/micro/exec_skip.hip -o tools/micro/exec_skip.bin
// 256 workgroups x 1024 threads = 4 waves per SIMD, each running 8 independent chains of v_fma_f32 under a fixed EXEC mask.
#include <hip/hip_runtime.h>
#include <cstdio>

template <int OP>
__global__ void __launch_bounds__(1024) k(int iters, float* out, float s, unsigned long long mask)
{
    float v[8];
    for (int i = 0; i < 8; i++) v[i] = (float)(threadIdx.x + i) * 1e-3f + 0.5f;
    unsigned long long saved;
    asm volatile("s_mov_b64 %0, exec\n s_mov_b64 exec, %1" : "=s"(saved) : "s"(mask));
    for (int it = 0; it < iters; it++) {
#pragma unroll
        for (int rep = 0; rep < 8; rep++) {
#pragma unroll
            for (int i = 0; i < 8; i++) {
                if constexpr (OP == 0) asm volatile("v_fma_f32 %0, %0, %1, %1" : "+v"(v[i]) : "v"(s));
                if constexpr (OP == 1) asm volatile("v_mul_f32 %0, %0, %1" : "+v"(v[i]) : "v"(s));
                if constexpr (OP == 2) asm volatile("v_exp_f32 %0, %0" : "+v"(v[i]));
            }
        }
    }
    asm volatile("s_mov_b64 exec, %0" : : "s"(saved));
    float r = 0;
    for (int i = 0; i < 8; i++) r += v[i];
    if (r == 12345.678f) out[0] = r;
}

template <int OP> float run(int iters, float* d_out, unsigned long long mask)
{
    hipEvent_t a, b; (void)hipEventCreate(&a); (void)hipEventCreate(&b);
    k<OP><<<256, 1024>>>(10, d_out, 0.999f, mask);
    (void)hipEventRecord(a);
    k<OP><<<256, 1024>>>(iters, d_out, 0.999f, mask);
    (void)hipEventRecord(b); (void)hipEventSynchronize(b);
    float ms; (void)hipEventElapsedTime(&ms, a, b); return ms;
}

int main()
{
    float* d_out; (void)hipMalloc(&d_out, 4096);
    const int iters = 4000;
    const unsigned long long masks[] = {~0ull, 0xffffffffull, 0xffffull, 0xffull, 0x1ull, 0x0000ffff0000ffffull, 0x00ff00ff00ff00ffull};
    const char* names[] = {"all 64 lanes", "lanes 0-31", "lanes 0-15", "lanes 0-7", "lane 0", "lanes 0-15 + 32-47", "8 of every 16"};
    for (int m = 0; m < 7; m++) {
        const float t0 = run<0>(iters, d_out, masks[m]), t1 = run<1>(iters, d_out, masks[m]), t2 = run<2>(iters, d_out, masks[m]);
        const double c = 1e6 / (4.0 * iters * 64) * 2.4;
        printf("%-22s v_fma_f32 %5.2f  v_mul_f32 %5.2f  v_exp_f32 %5.2f  cycles per wave-instruction per SIMD\n", names[m], t0 * c, t1 * c, t2 * c);
    }
    return 0;
}
