// Microbenchmark: issue cost (cycles per wave64 instruction per SIMD) of the vector instructions the blend kernels are made of.
// hipcc --offload-arch=gfx950 -O3 tools/micro/valu_rates.hip -o tools/micro/valu_rates.bin
// 256 workgroups x 1024 threads = 4 waves per SIMD, each running 8 independent chains of one instruction.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <utility>


template <int MODE>
__global__ void __launch_bounds__(1024) k(int iters, float* out, float s, int si, unsigned long long mk)
{
    float v[8], w2[2] = {s, s};
    int u[8];
    for (int i = 0; i < 8; i++) { v[i] = (float)(threadIdx.x + i) * 1e-3f + 0.5f; u[i] = threadIdx.x * 7 + i; }
    unsigned long long m = 0;
    for (int it = 0; it < iters; it++) {
#pragma unroll
        for (int rep = 0; rep < 8; rep++) {
#pragma unroll
            for (int i = 0; i < 8; i++) {
                if constexpr (MODE == 0) { asm volatile("v_fma_f32 %0, %0, %1, %1" : "+v"(v[i]) : "v"(s)); }
                if constexpr (MODE == 1) { asm volatile("v_fma_f32 %0, %0, %1, 1.0" : "+v"(v[i]) : "v"(s)); }
                if constexpr (MODE == 2) { asm volatile("v_fmac_f32 %0, %1, %1" : "+v"(v[i]) : "v"(s)); }
                if constexpr (MODE == 3) { asm volatile("v_sub_f32 %0, %0, %1" : "+v"(v[i]) : "v"(s)); }
                if constexpr (MODE == 4) { asm volatile("v_mul_f32 %0, %0, %1" : "+v"(v[i]) : "v"(s)); }
                if constexpr (MODE == 5) { asm volatile("v_min_f32 %0, %0, %1" : "+v"(v[i]) : "v"(s)); }
                if constexpr (MODE == 6) { asm volatile("v_max_f32 %0, %0, %1" : "+v"(v[i]) : "v"(s)); }
                if constexpr (MODE == 7) { asm volatile("v_exp_f32 %0, %0" : "+v"(v[i])); }
                if constexpr (MODE == 8) { asm volatile("v_rcp_f32 %0, %0" : "+v"(v[i])); }
                if constexpr (MODE == 9) { asm volatile("v_cndmask_b32 %0, %0, %1, vcc" : "+v"(v[i]) : "v"(s)); }
                if constexpr (MODE == 10) { asm volatile("v_cndmask_b32 %0, %0, %1, %2" : "+v"(v[i]) : "v"(s), "s"(mk)); }
                if constexpr (MODE == 11) { asm volatile("v_cmp_lt_f32 vcc, %0, %1" : : "v"(v[i]), "v"(s) : "vcc"); }
                if constexpr (MODE == 12) { asm volatile("v_cmp_lt_f32 %0, %1, %2" : "=s"(m) : "v"(v[i]), "v"(s)); }
                if constexpr (MODE == 13) { asm volatile("v_cmp_lt_f32 vcc, %0, %1\n v_cndmask_b32 %0, %0, %1, vcc" : "+v"(v[i]) : "v"(s) : "vcc"); }
                if constexpr (MODE == 14) { asm volatile("v_mov_b32 %0, %1" : "=v"(v[i]) : "v"(s)); }
                if constexpr (MODE == 15) { asm volatile("v_and_b32 %0, %0, %1" : "+v"(u[i]) : "v"(si)); }
                if constexpr (MODE == 16) { asm volatile("v_or_b32 %0, %0, %1" : "+v"(u[i]) : "v"(si)); }
                if constexpr (MODE == 17) { asm volatile("v_add_u32 %0, %0, %1" : "+v"(u[i]) : "v"(si)); }
                if constexpr (MODE == 18) { asm volatile("v_sub_u32 %0, %0, %1" : "+v"(u[i]) : "v"(si)); }
                if constexpr (MODE == 19) { asm volatile("v_lshlrev_b32 %0, 1, %0" : "+v"(u[i])); }
                if constexpr (MODE == 20) { asm volatile("v_lshl_add_u32 %0, %0, 2, %1" : "+v"(u[i]) : "v"(si)); }
                if constexpr (MODE == 21) { asm volatile("v_and_or_b32 %0, %0, %1, %1" : "+v"(u[i]) : "v"(si)); }
                if constexpr (MODE == 22) { asm volatile("v_bfe_u32 %0, %0, 3, 5" : "+v"(u[i])); }
                if constexpr (MODE == 23) { asm volatile("v_ffbl_b32 %0, %0" : "+v"(u[i])); }
                if constexpr (MODE == 24) { asm volatile("v_bcnt_u32_b32 %0, %0, %1" : "+v"(u[i]) : "v"(si)); }
                if constexpr (MODE == 25) { asm volatile("v_perm_b32 %0, %0, %1, %1" : "+v"(u[i]) : "v"(si)); }
                if constexpr (MODE == 26) { asm volatile("v_mul_u32_u24 %0, %0, %1" : "+v"(u[i]) : "v"(si)); }
                if constexpr (MODE == 27) { asm volatile("v_mad_u32_u24 %0, %0, %1, %1" : "+v"(u[i]) : "v"(si)); }
                if constexpr (MODE == 28) { asm volatile("v_mul_lo_u32 %0, %0, %1" : "+v"(u[i]) : "v"(si)); }
                if constexpr (MODE == 29) { asm volatile("v_cvt_f32_u32 %0, %0" : "+v"(v[i])); }
                if constexpr (MODE == 30) { asm volatile("v_pk_fma_f32 %0, %0, %1, %1" : "+v"(*(double*)&v[i & 6]) : "v"(*(double*)&w2[0])); }
                if constexpr (MODE == 31) { asm volatile("v_pk_mul_f32 %0, %0, %1" : "+v"(*(double*)&v[i & 6]) : "v"(*(double*)&w2[0])); }
                if constexpr (MODE == 32) { asm volatile("v_pk_add_f32 %0, %0, %1" : "+v"(*(double*)&v[i & 6]) : "v"(*(double*)&w2[0])); }
                if constexpr (MODE == 33) { asm volatile("v_mov_b32_dpp %0, %0 quad_perm:[1,0,3,2] row_mask:0xf bank_mask:0xf" : "+v"(v[i])); }
                if constexpr (MODE == 34) { asm volatile("v_add_f32_dpp %0, %0, %0 row_shr:1 row_mask:0xf bank_mask:0xf" : "+v"(v[i])); }
                if constexpr (MODE == 35) { asm volatile("v_readlane_b32 %0, %1, 3" : "=s"(si) : "v"(u[i])); }
                if constexpr (MODE == 36) { asm volatile("v_readfirstlane_b32 %0, %1" : "=s"(si) : "v"(u[i])); }
                if constexpr (MODE == 37) { asm volatile("s_and_b64 %0, %0, %1" : "+s"(m) : "s"(mk) : "scc"); }
                if constexpr (MODE == 38) { asm volatile("s_ff1_i32_b64 %0, %1" : "=s"(si) : "s"(mk)); }
            }
        }
    }
    float r = 0;
    for (int i = 0; i < 8; i++) r += v[i] + (float)u[i];
    if (r == 12345.678f || m == 77ull) out[0] = r + si;
}

template <int MODE> float run(int iters, float* d_out)
{
    hipEvent_t a, b; (void)hipEventCreate(&a); (void)hipEventCreate(&b);
    k<MODE><<<256, 1024>>>(10, d_out, 0.999f, 0x7fffffff, 0x5555555555555555ull);
    (void)hipEventRecord(a);
    k<MODE><<<256, 1024>>>(iters, d_out, 0.999f, 0x7fffffff, 0x5555555555555555ull);
    (void)hipEventRecord(b); (void)hipEventSynchronize(b);
    float ms; (void)hipEventElapsedTime(&ms, a, b); return ms;
}

template <int... I> void run_all(std::integer_sequence<int, I...>, int iters, float* d_out, float* t) { ((t[I] = run<I>(iters, d_out)), ...); }

int main()
{
    float* d_out; (void)hipMalloc(&d_out, 4096);
    const int iters = 4000;
    constexpr int N = 39;
    const char* names[N] = {"v_fma_f32 (3 vgpr)", "v_fma_f32 (2 vgpr + const)", "v_fmac_f32", "v_sub_f32", "v_mul_f32", "v_min_f32", "v_max_f32", "v_exp_f32", "v_rcp_f32", "v_cndmask_b32 vcc", "v_cndmask_b32 sgpr", "v_cmp_lt_f32 vcc", "v_cmp_lt_f32 sgpr", "v_cmp + v_cndmask pair", "v_mov_b32", "v_and_b32", "v_or_b32", "v_add_u32", "v_sub_u32", "v_lshlrev_b32", "v_lshl_add_u32", "v_and_or_b32", "v_bfe_u32", "v_ffbl_b32", "v_bcnt_u32_b32", "v_perm_b32", "v_mul_u32_u24", "v_mad_u32_u24", "v_mul_lo_u32", "v_cvt_f32_u32", "v_pk_fma_f32", "v_pk_mul_f32", "v_pk_add_f32", "v_mov_b32 dpp quad_perm", "v_add_f32 dpp row_shr", "v_readlane_b32", "v_readfirstlane_b32", "s_and_b64 (scalar)", "s_ff1_i32_b64 (scalar)"};
    float t[N];
    run_all(std::make_integer_sequence<int, N>{}, iters, d_out, t);
    // per SIMD: 4 waves x iters x 64 instructions
    for (int m = 0; m < N; m++)
        printf("%-28s %6.2f ns per wave-instruction per SIMD = %5.1f cycles at 2.4 GHz\n", names[m], t[m] * 1e6 / (4.0 * iters * 64),
               t[m] * 1e6 / (4.0 * iters * 64) * 2.4);
    return 0;
}
