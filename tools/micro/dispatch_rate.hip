// Microbenchmark: what does LAUNCHING a one-wave workgroup cost?  blend_bwd runs 4 U ~ 55 000 workgroups of 64 threads with 5 KB of
// LDS per view; this times N such workgroups that (0) return at once, (1) do one scalar + one vector load and return, (2) spin for
// `work` iterations of dependent fmas (a stand-in for a unit's life) -- as a function of the registers the kernel declares.
// hipcc --offload-arch=gfx950 -O3 tools/micro/dispatch_rate.hip -o tools/micro/dispatch_rate.bin
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>

template <int MODE, int LDSB>
__global__ void __launch_bounds__(64) k(const float* __restrict__ in, float* __restrict__ out, int work)
{
    __shared__ float lds[LDSB / 4];
    if (MODE == 0) { if (work == -7) out[blockIdx.x] = lds[threadIdx.x]; return; }
    float v = in[blockIdx.x & 1023] + in[1024 + threadIdx.x];
    if (MODE == 2) for (int i = 0; i < work; i++) asm volatile("v_fma_f32 %0, %0, %0, %0" : "+v"(v));
    lds[threadIdx.x] = v;
    if (v == 12345.678f) out[blockIdx.x] = lds[threadIdx.x ^ 1];
}

template <int MODE, int LDSB> float run(int n, int work, const float* in, float* out, int reps)
{
    hipEvent_t a, b; (void)hipEventCreate(&a); (void)hipEventCreate(&b);
    k<MODE, LDSB><<<n, 64>>>(in, out, work);
    (void)hipEventRecord(a);
    for (int r = 0; r < reps; r++) k<MODE, LDSB><<<n, 64>>>(in, out, work);
    (void)hipEventRecord(b); (void)hipEventSynchronize(b);
    float ms; (void)hipEventElapsedTime(&ms, a, b); return ms * 1000.f / reps;
}

int main(int argc, char** argv)
{
    const int n = argc > 1 ? atoi(argv[1]) : 54640;
    float *in, *out; (void)hipMalloc(&in, 4096 * 4); (void)hipMalloc(&out, (size_t)n * 4); (void)hipMemset(in, 0, 4096 * 4);
    printf("%d workgroups of 64 threads (us per launch, mean of 20)\n", n);
    printf("  return at once, 5 KB LDS : %7.2f\n", run<0, 5120>(n, 0, in, out, 20));
    printf("  return at once, 64 B LDS : %7.2f\n", run<0, 64>(n, 0, in, out, 20));
    printf("  two loads,      5 KB LDS : %7.2f\n", run<1, 5120>(n, 0, in, out, 20));
    for (int w : {250, 1000, 4000})
        printf("  %4d dependent fmas, 5 KB : %7.2f   (%d x 4 cycles x %d waves / 1024 SIMDs / 2.4 GHz = %.1f us of issue)\n", w,
               run<2, 5120>(n, w, in, out, 20), w, n, (double)w * 4 * n / 1024 / 2400.0);
    return 0;
}
