// Microbenchmark: throughput of LDS atomics (float / u32 / u64 add) and per-lane ds_read_b128 gathers on gfx950.
// hipcc --offload-arch=gfx950 -O3 -munsafe-fp-atomics tools/micro/lds_atomics.hip -o /tmp/lds_atomics && /tmp/lds_atomics
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdint>
#include <vector>

template <int MODE>
__global__ void __launch_bounds__(256) k(const int* __restrict__ idx, int iters, float* out)
{
    __shared__ float tab[4096];
    __shared__ unsigned long long tab64[2048];
    for (int i = threadIdx.x; i < 4096; i += 256) tab[i] = 0.f;
    for (int i = threadIdx.x; i < 2048; i += 256) tab64[i] = 0ull;
    __syncthreads();
    const int lane = threadIdx.x & 63;
    int a = idx[lane];             // slot per lane (pattern from the host)
    float v = 1.0f + lane;
    float acc = 0.f;
    for (int it = 0; it < iters; it++) {
        const int base = (a * 9) & 4095;
        if (MODE == 0) {
#pragma unroll
            for (int m = 0; m < 9; m++) __hip_atomic_fetch_add(&tab[(base + m) & 4095], v, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
        } else if (MODE == 1) {
#pragma unroll
            for (int m = 0; m < 9; m++) __hip_atomic_fetch_add((unsigned*)&tab[(base + m) & 4095], (unsigned)lane, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
        } else if (MODE == 2) {
#pragma unroll
            for (int m = 0; m < 9; m++) __hip_atomic_fetch_add(&tab64[(base + m) & 2047], (unsigned long long)lane, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
        } else if (MODE == 3) {   // plain stores
#pragma unroll
            for (int m = 0; m < 9; m++) tab[(base + m) & 4095] = v;
            __builtin_amdgcn_wave_barrier();
        } else if (MODE == 4) {   // b128 gathers
            const float4* p = reinterpret_cast<const float4*>(tab) + ((a * 3) & 1023);
            const float4 x = p[0], y = p[1], z = p[2];
            acc += x.x + y.y + z.z;
        }
        a = (a + 7) & 255;
    }
    __syncthreads();
    if (threadIdx.x == 0) out[blockIdx.x] = tab[5] + acc + (float)tab64[3];
    else if (MODE == 4 && acc == 123.f) out[0] = acc;
}

template <int MODE> float run(const int* d_idx, int iters, float* d_out, int blocks)
{
    hipEvent_t a, b; hipEventCreate(&a); hipEventCreate(&b);
    k<MODE><<<blocks, 256>>>(d_idx, 10, d_out);
    hipEventRecord(a);
    k<MODE><<<blocks, 256>>>(d_idx, iters, d_out);
    hipEventRecord(b); hipEventSynchronize(b);
    float ms; hipEventElapsedTime(&ms, a, b); return ms;
}

int main()
{
    const int blocks = 256 * 4, iters = 2000;   // 4 workgroups of 4 waves per CU
    int* d_idx; float* d_out;
    hipMalloc(&d_idx, 64 * 4); hipMalloc(&d_out, blocks * 4);
    const char* names[5] = {"ds_add_f32", "ds_add_u32", "ds_add_u64", "ds_write_b32", "ds_read_b128 x3 gather"};
    for (int pat = 0; pat < 3; pat++) {
        std::vector<int> h(64);
        for (int l = 0; l < 64; l++) h[l] = pat == 0 ? l : pat == 1 ? (l * 7) % 28 : l / 4;   // distinct | 28 slots (~2.3 lanes each) | 16 slots x 4 lanes
        hipMemcpy(d_idx, h.data(), 256, hipMemcpyHostToDevice);
        printf("pattern %d (%s)\n", pat, pat == 0 ? "64 distinct slots" : pat == 1 ? "28 slots" : "16 slots x 4 adjacent lanes");
        float ms[5] = {run<0>(d_idx, iters, d_out, blocks), run<1>(d_idx, iters, d_out, blocks), run<2>(d_idx, iters, d_out, blocks),
                       run<3>(d_idx, iters, d_out, blocks), run<4>(d_idx, iters, d_out, blocks)};
        for (int m = 0; m < 5; m++) {
            const double n_inst = (double)blocks * 4 * iters * (m == 4 ? 3 : 9);     // wave-instructions
            const double cyc = ms[m] * 1e-3 * 2.4e9;                                   // cycles (2.4 GHz nominal)
            printf("  %-24s %8.3f ms  -> %.1f CU-cycles per wave-instruction\n", names[m], ms[m], cyc / (n_inst / 256.0));
        }
    }
    return 0;
}
