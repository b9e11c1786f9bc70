// Microbenchmark + exactness check: the bf16 hi/mid/lo split of an f32 (gsr_blend_bwd.hip) with v_dot2_f32_bf16.
// The residual x - hi(x) (hi = upper 16 bits of x) costs v_and + v_sub; with the packed pair {hi(x0), hi(x1)} already formed
// for the matrix operand (v_perm), v_dot2_f32_bf16(pair, {-1, 0}, x0) = x0 - hi(x0) is ONE instruction -- if the dot unit keeps
// all 24 bits of the addend.  This program checks that bit for bit over random inputs and measures the issue cost.
// hipcc --offload-arch=gfx950 -O3 tools/micro/dot2_split.hip -o tools/micro/dot2_split.bin
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <vector>

__device__ __forceinline__ unsigned pair_hi(float lo_elem, float hi_elem)
{
    return __builtin_amdgcn_perm(__float_as_uint(hi_elem), __float_as_uint(lo_elem), 0x07060302u);
}

__global__ void check(const float* x0, const float* x1, float* ref, float* got, int n)
{
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    const float a = x0[i], b = x1[i];
    // reference: and + sub, twice
    const float ya = a - __uint_as_float(__float_as_uint(a) & 0xffff0000u), yb = b - __uint_as_float(__float_as_uint(b) & 0xffff0000u);
    const float za = ya - __uint_as_float(__float_as_uint(ya) & 0xffff0000u), zb = yb - __uint_as_float(__float_as_uint(yb) & 0xffff0000u);
    ref[4 * i] = ya; ref[4 * i + 1] = yb; ref[4 * i + 2] = za; ref[4 * i + 3] = zb;
    const unsigned p = pair_hi(a, b);
    float da, db;
    asm volatile("v_dot2_f32_bf16 %0, %1, %2, %3" : "=v"(da) : "v"(p), "v"(0x0000BF80u), "v"(a));
    asm volatile("v_dot2_f32_bf16 %0, %1, %2, %3" : "=v"(db) : "v"(p), "v"(0xBF800000u), "v"(b));
    const unsigned q = pair_hi(da, db);
    float ea, eb;
    asm volatile("v_dot2_f32_bf16 %0, %1, %2, %3" : "=v"(ea) : "v"(q), "v"(0x0000BF80u), "v"(da));
    asm volatile("v_dot2_f32_bf16 %0, %1, %2, %3" : "=v"(eb) : "v"(q), "v"(0xBF800000u), "v"(db));
    got[4 * i] = da; got[4 * i + 1] = db; got[4 * i + 2] = ea; got[4 * i + 3] = eb;
}

// the forms the compiler emits for gsr_blend_bwd.hip's bf16_rest_lo / bf16_rest_hi (v_dot2c_f32_bf16 with the inline
// constant -1.0 for {-1, 0} and a 32-bit literal for {0, -1})
typedef __bf16 bf16x2 __attribute__((ext_vector_type(2)));
__global__ void check_builtin(const float* x0, const float* x1, float* got, int n)
{
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    const float a = x0[i], b = x1[i];
    const unsigned p = pair_hi(a, b);
    const float da = __builtin_amdgcn_fdot2_f32_bf16(__builtin_bit_cast(bf16x2, p), __builtin_bit_cast(bf16x2, 0x0000BF80u), a, false);
    const float db = __builtin_amdgcn_fdot2_f32_bf16(__builtin_bit_cast(bf16x2, p), __builtin_bit_cast(bf16x2, 0xBF800000u), b, false);
    const unsigned q = pair_hi(da, db);
    const float ea = __builtin_amdgcn_fdot2_f32_bf16(__builtin_bit_cast(bf16x2, q), __builtin_bit_cast(bf16x2, 0x0000BF80u), da, false);
    const float eb = __builtin_amdgcn_fdot2_f32_bf16(__builtin_bit_cast(bf16x2, q), __builtin_bit_cast(bf16x2, 0xBF800000u), db, false);
    got[4 * i] = da; got[4 * i + 1] = db; got[4 * i + 2] = ea; got[4 * i + 3] = eb;
}
// the constant pair in a SCALAR register (NOT used): 41 % of the residuals come out wrong
__global__ void check_sgpr(const float* x0, const float* x1, float* got, int n)
{
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    const float a = x0[i], b = x1[i];
    const unsigned p = pair_hi(a, b);
    float da, db, ea, eb;
    asm("v_dot2_f32_bf16 %0, %1, %2, %3" : "=v"(da) : "v"(p), "s"(0x0000BF80u), "v"(a));
    asm("v_dot2_f32_bf16 %0, %1, %2, %3" : "=v"(db) : "v"(p), "s"(0xBF800000u), "v"(b));
    const unsigned q = pair_hi(da, db);
    asm("v_dot2_f32_bf16 %0, %1, %2, %3" : "=v"(ea) : "v"(q), "s"(0x0000BF80u), "v"(da));
    asm("v_dot2_f32_bf16 %0, %1, %2, %3" : "=v"(eb) : "v"(q), "s"(0xBF800000u), "v"(db));
    got[4 * i] = da; got[4 * i + 1] = db; got[4 * i + 2] = ea; got[4 * i + 3] = eb;
}
// gsr_blend_bwd.hip's spelling: the compiler builtin, the constant pairs in vector registers the compiler cannot fold
__global__ void check_kernel_spelling(const float* x0, const float* x1, float* got, int n)
{
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    const float a = x0[i], b = x1[i];
    unsigned k0 = 0x0000BF80u, k1 = 0xBF800000u;
    asm volatile("" : "+v"(k0), "+v"(k1));
    const unsigned p = pair_hi(a, b);
    const float da = __builtin_amdgcn_fdot2_f32_bf16(__builtin_bit_cast(bf16x2, p), __builtin_bit_cast(bf16x2, k0), a, false);
    const float db = __builtin_amdgcn_fdot2_f32_bf16(__builtin_bit_cast(bf16x2, p), __builtin_bit_cast(bf16x2, k1), b, false);
    const unsigned q = pair_hi(da, db);
    const float ea = __builtin_amdgcn_fdot2_f32_bf16(__builtin_bit_cast(bf16x2, q), __builtin_bit_cast(bf16x2, k0), da, false);
    const float eb = __builtin_amdgcn_fdot2_f32_bf16(__builtin_bit_cast(bf16x2, q), __builtin_bit_cast(bf16x2, k1), db, false);
    got[4 * i] = da; got[4 * i + 1] = db; got[4 * i + 2] = ea; got[4 * i + 3] = eb;
}

template <int MODE>
__global__ void __launch_bounds__(1024) rate(int iters, float* out, float s)
{
    float v[8];
    for (int i = 0; i < 8; i++) v[i] = (float)(threadIdx.x + i) * 1e-3f + 0.5f;
    unsigned p = __float_as_uint(s);
    for (int it = 0; it < iters; it++) {
#pragma unroll
        for (int rep = 0; rep < 8; rep++) {
#pragma unroll
            for (int i = 0; i < 8; i++) {
                if constexpr (MODE == 0) asm volatile("v_dot2_f32_bf16 %0, %1, %2, %0" : "+v"(v[i]) : "v"(p), "v"(0x0000BF80u));
                if constexpr (MODE == 1) asm volatile("v_dot2c_f32_bf16 %0, %1, %2" : "+v"(v[i]) : "v"(p), "v"(0x0000BF80u));
                if constexpr (MODE == 2) asm volatile("v_and_b32 %1, 0xffff0000, %0\n v_sub_f32 %0, %0, %1" : "+v"(v[i]), "=v"(p));
                if constexpr (MODE == 3) asm volatile("v_sub_f32 %0, %0, %1" : "+v"(v[i]) : "v"(s));
            }
        }
    }
    float r = 0;
    for (int i = 0; i < 8; i++) r += v[i];
    if (r == 12345.678f) out[0] = r + (float)p;
}

template <int MODE> float run(int iters, float* d_out)
{
    hipEvent_t a, b; (void)hipEventCreate(&a); (void)hipEventCreate(&b);
    rate<MODE><<<256, 1024>>>(10, d_out, 0.999f);
    (void)hipEventRecord(a);
    rate<MODE><<<256, 1024>>>(iters, d_out, 0.999f);
    (void)hipEventRecord(b); (void)hipEventSynchronize(b);
    float ms; (void)hipEventElapsedTime(&ms, a, b); return ms;
}

int main()
{
    const int n = 1 << 22;
    std::vector<float> h0(n), h1(n);
    srand(1);
    for (int i = 0; i < n; i++) {
        // random bit patterns over a wide exponent range (no inf / nan), both signs; every 16th value has a short mantissa
        unsigned u0 = ((unsigned)rand() << 16) ^ (unsigned)rand(), u1 = ((unsigned)rand() << 16) ^ (unsigned)rand();
        unsigned e0 = 40 + rand() % 170, e1 = 40 + rand() % 170;   // biased exponents 40 .. 209
        u0 = (u0 & 0x807fffffu) | (e0 << 23); u1 = (u1 & 0x807fffffu) | (e1 << 23);
        if ((i & 15) == 0) { u0 &= 0xffffff00u; u1 &= 0xffff0000u; }
        if ((i & 1023) == 0) { u0 = 0; }
        memcpy(&h0[i], &u0, 4); memcpy(&h1[i], &u1, 4);
    }
    float *d0, *d1, *dr, *dg, *d_out;
    (void)hipMalloc(&d0, 4 * n); (void)hipMalloc(&d1, 4 * n); (void)hipMalloc(&dr, 16 * (size_t)n); (void)hipMalloc(&dg, 16 * (size_t)n);
    (void)hipMalloc(&d_out, 4096);
    (void)hipMemcpy(d0, h0.data(), 4 * n, hipMemcpyHostToDevice); (void)hipMemcpy(d1, h1.data(), 4 * n, hipMemcpyHostToDevice);
    check<<<n / 256, 256>>>(d0, d1, dr, dg, n);
    std::vector<float> r(4 * (size_t)n), g(4 * (size_t)n);
    (void)hipMemcpy(r.data(), dr, 16 * (size_t)n, hipMemcpyDeviceToHost); (void)hipMemcpy(g.data(), dg, 16 * (size_t)n, hipMemcpyDeviceToHost);
    long long bad = 0;
    for (size_t i = 0; i < 4 * (size_t)n; i++)
        if (memcmp(&r[i], &g[i], 4) != 0 && !(r[i] == 0.f && g[i] == 0.f)) {
            if (bad < 8) printf("mismatch at %zu (x = %.9g): and/sub %.9g  dot2 %.9g\n", i, (i & 1) ? h1[i / 4] : h0[i / 4], r[i], g[i]);
            bad++;
        }
    printf("exactness (inline assembly, constant pairs in vector registers): %lld of %zu residuals differ\n", bad, 4 * (size_t)n);
    check_kernel_spelling<<<n / 256, 256>>>(d0, d1, dg, n);
    (void)hipMemcpy(g.data(), dg, 16 * (size_t)n, hipMemcpyDeviceToHost);
    long long bad4 = 0;
    for (size_t i = 0; i < 4 * (size_t)n; i++)
        if (memcmp(&r[i], &g[i], 4) != 0 && !(r[i] == 0.f && g[i] == 0.f)) bad4++;
    printf("exactness (builtin, constant pairs in opaque vector registers -- the kernel's spelling): %lld of %zu residuals differ\n", bad4, 4 * (size_t)n);
    bad += bad4;
    check_builtin<<<n / 256, 256>>>(d0, d1, dg, n);
    (void)hipMemcpy(g.data(), dg, 16 * (size_t)n, hipMemcpyDeviceToHost);
    long long bad2 = 0;
    for (size_t i = 0; i < 4 * (size_t)n; i++)
        if (memcmp(&r[i], &g[i], 4) != 0 && !(r[i] == 0.f && g[i] == 0.f)) bad2++;
    printf("exactness (builtin with literal constants -> inline constant -1.0; NOT used): %lld of %zu residuals differ\n", bad2, 4 * (size_t)n);
    check_sgpr<<<n / 256, 256>>>(d0, d1, dg, n);
    (void)hipMemcpy(g.data(), dg, 16 * (size_t)n, hipMemcpyDeviceToHost);
    long long bad3 = 0;
    for (size_t i = 0; i < 4 * (size_t)n; i++)
        if (memcmp(&r[i], &g[i], 4) != 0 && !(r[i] == 0.f && g[i] == 0.f)) bad3++;
    printf("exactness (VOP3P, constant pair in a scalar register; NOT used): %lld of %zu residuals differ\n", bad3, 4 * (size_t)n);
    const int iters = 4000;
    const char* names[4] = {"v_dot2_f32_bf16", "v_dot2c_f32_bf16", "v_and_b32 + v_sub_f32 (two instructions)", "v_sub_f32"};
    float t[4] = {run<0>(iters, d_out), run<1>(iters, d_out), run<2>(iters, d_out), run<3>(iters, d_out)};
    for (int m = 0; m < 4; m++)
        printf("%-44s %6.2f ns per wave-issue per SIMD = %5.1f cycles at 2.4 GHz\n", names[m], t[m] * 1e6 / (4.0 * iters * 64),
               t[m] * 1e6 / (4.0 * iters * 64) * 2.4);
    return bad != 0;
}
