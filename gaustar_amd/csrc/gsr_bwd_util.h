// gsr_bwd_util.h -- helpers shared by the backward blend kernel (gsr_blend_bwd.hip: uniform pair loop) and the experiment
// variants under tools/variants/: bf16 splits, LDS queue-slot layout, explicit LDS requests, wave-wide OR.
#pragma once
#include "gsr_internal.h"
#include <cstdlib>
#include <type_traits>
#include <utility>

namespace gsr {

typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
typedef uint32_t u32x4 __attribute__((ext_vector_type(4)));

#ifndef GSR_BWD_BF16
#define GSR_BWD_BF16 1
#endif
#ifndef GSR_BWD_DOT2
#ifndef GSR_BWD_BF16_TILES2
#define GSR_BWD_BF16_TILES2 1   // four channels: the same contraction with a SECOND B tile for channel 3 (0: f32 MFMA)
#endif
#define GSR_BWD_DOT2 1   // residuals of the bf16 split by v_dot2_f32_bf16 (0: v_and + v_sub)
#endif
// Three channels: the contraction runs on the bf16 matrix pipe WITHOUT giving up f32 accuracy.  An f32 is exactly
// hi + mid + lo with eight significant bits each (truncate to the upper 16 bits, subtract, twice: the second remainder has
// at most eight bits left), the monomials are exact in bf16, dL_dpix takes three columns per channel (6 + 3 C = 15 <= 16).
// Six v_mfma_f32_16x16x32_bf16 (K = 32 pixels x {hi, mid, lo} of the A operand, products exact, f32 accumulation) replace
// sixteen v_mfma_f32_16x16x4_f32: ~100 instead of 512 cycles of the SIMD per eight instances, paid for with 5.5 vector
// instructions per table value for the splits (tools/micro/mfma_bf16_valu_overlap.hip).  Four channels
// take a second column tile for channel 3 (six more matrix issues on the same split A operand, 0.140 -> 0.134 ms); six
// channels gain nothing from it and keep the f32 instruction.
__device__ __forceinline__ uint32_t bf16_pair(float lo_elem, float hi_elem)   // upper halves of two floats, element order (lo, hi)
{
    return __builtin_amdgcn_perm(__float_as_uint(hi_elem), __float_as_uint(lo_elem), 0x07060302u);
}
__device__ __forceinline__ float bf16_rest(float x) { return x - __uint_as_float(__float_as_uint(x) & 0xffff0000u); }
// The same residuals for BOTH elements of an already packed pair {hi(x0), hi(x1)}: x - hi(x) = dot2(pair, {-1, 0}, x0) resp.
// dot2(pair, {0, -1}, x1) -- one v_dot2_f32_bf16 each (4.7 cycles) instead of v_and + v_sub (6.4), and bit-identical: the
// dot unit keeps all 24 bits of the f32 addend (tools/micro/dot2_split.hip checks 16.8 M residuals of both levels).
// The constant pairs {-1, 0} / {0, -1} must reach the instruction in VECTOR registers the compiler cannot see through
// (`asm volatile("" : "+v"(k))` at the use site): given the literal, __builtin_amdgcn_fdot2_f32_bf16 folds {-1, 0} into the
// inline constant -1.0, which the hardware does not read as that bf16 pair, and a scalar-register operand is not read as
// one either (tools/micro/dot2_split.hip checks each spelling bit for bit).  The builtin rather than inline assembly: the
// compiler has to know the opcode to keep the wait states dot instructions need next to matrix instructions.
typedef __bf16 bf16x2 __attribute__((ext_vector_type(2)));
__device__ __forceinline__ float bf16_rest_of(uint32_t pair, float x, uint32_t minus_one_at)
{
    return __builtin_amdgcn_fdot2_f32_bf16(__builtin_bit_cast(bf16x2, pair), __builtin_bit_cast(bf16x2, minus_one_at), x, false);
}

// LDS queue slot of one fetched instance, as floats:
//   [0] x  [1] y  [2] gaussian id (uint bits)  [3..5] conic a, b, c in the exp2 domain (conic_to_exp2)  [6] opacity
//   [7] list position (uint bits)  [8 ..] colour channels            -> 12 floats (C = 3), 16 floats (C = 6)
// Once the instance's w and r are out only x, y and the id are still needed: the 6 + C moments overwrite
// floats [3 .. 8 + C] (no separate moment table -> more resident waves).
template <int C> struct SlotLayout {
    static constexpr int NM = 6 + C;        // moments per instance
    static constexpr int MOM0 = 3;          // first overwritten float
    static constexpr int IN_FLOATS = (8 + C + 3) / 4 * 4, OUT_FLOATS = (MOM0 + NM + 3) / 4 * 4;
    static constexpr int FLOATS = IN_FLOATS > OUT_FLOATS ? IN_FLOATS : OUT_FLOATS;   // C = 4: 12 in, 13 out -> 16
    static constexpr int VECS = FLOATS / 4;
    static constexpr int IN_VECS = IN_FLOATS / 4;   // what the pair loop reads of a slot (C = 4: three of its four float4)
    static_assert(MOM0 + NM <= FLOATS, "moments must fit the slot");
};

// A queue slot is read with explicit ds_read_b128 (uniform address): left to the compiler the colour words are read
// inside the conditional "live" block, where the full LDS latency is exposed once per live pair, and nothing can be
// requested one pair ahead across the branches.  lds_wait() is the matching s_waitcnt; the operands tie the uses to it,
// and the "memory" clobber keeps the compiler's own LDS stores (queue fill, moments) on their side of a request.
__device__ __forceinline__ uint32_t lds_byte_address(const void* p) { return (uint32_t)(size_t)p; }   // low half of the flat address
template <int VECS> struct SlotRegs { f32x4 v[VECS]; };
template <int VECS, int OFF>   // OFF: compile-time byte offset from `addr` (one address register serves a whole group)
__device__ __forceinline__ void lds_request(SlotRegs<VECS>& r, uint32_t addr)
{
    // (early-clobber outputs: the address register must survive the request, it serves the whole group -- otherwise the
    // last read lands on it and every pair re-materialises it)
    asm volatile("ds_read_b128 %0, %1 offset:%2" : "=&v"(r.v[0]) : "v"(addr), "n"(OFF) : "memory");
    asm volatile("ds_read_b128 %0, %1 offset:%2" : "=&v"(r.v[1]) : "v"(addr), "n"(OFF + 16) : "memory");
    if constexpr (VECS > 2) asm volatile("ds_read_b128 %0, %1 offset:%2" : "=&v"(r.v[2]) : "v"(addr), "n"(OFF + 32) : "memory");
    if constexpr (VECS > 3) asm volatile("ds_read_b128 %0, %1 offset:%2" : "=&v"(r.v[3]) : "v"(addr), "n"(OFF + 48) : "memory");
}
template <int OFF>
__device__ __forceinline__ void lds_store_b32(uint32_t addr, float v)
{
    asm volatile("ds_write_b32 %0, %1 offset:%2" : : "v"(addr), "v"(v), "n"(OFF) : "memory");
}
// PENDING: LDS operations issued AFTER the request that may still be in flight when the data are used (LDS operations
// complete in order, so "at most PENDING outstanding" implies the older request has landed).  The pair loop passes 2 --
// the previous pair's two table stores; waiting for those as well (lgkmcnt(0)) stalled every pair for a store's latency.
// The "memory" clobber keeps the compiler from moving those stores behind the wait.
template <int PENDING, int VECS>
__device__ __forceinline__ void lds_wait(SlotRegs<VECS>& r)
{
    static_assert(PENDING == 0 || PENDING == 2, "");
    if constexpr (PENDING == 0) {
        if constexpr (VECS == 3) asm volatile("s_waitcnt lgkmcnt(0)" : "+v"(r.v[0]), "+v"(r.v[1]), "+v"(r.v[2]) : : "memory");
        else asm volatile("s_waitcnt lgkmcnt(0)" : "+v"(r.v[0]), "+v"(r.v[1]), "+v"(r.v[2]), "+v"(r.v[3]) : : "memory");
    } else {
        if constexpr (VECS == 3) asm volatile("s_waitcnt lgkmcnt(2)" : "+v"(r.v[0]), "+v"(r.v[1]), "+v"(r.v[2]) : : "memory");
        else asm volatile("s_waitcnt lgkmcnt(2)" : "+v"(r.v[0]), "+v"(r.v[1]), "+v"(r.v[2]), "+v"(r.v[3]) : : "memory");
    }
}

// OR over the 64 lanes of a wave of a 64-bit value, through LDS: every lane clears the scratch word (same value, one
// instruction), ORs its own in (ds_or_b64, lanes serialise inside the one instruction) and reads the result back; LDS
// operations of one wave execute in order, so no barrier is needed.  -> wave-uniform.
__device__ __forceinline__ unsigned long long wave_or_u64_lds(uint32_t scratch_addr, uint32_t lo, uint32_t hi)
{
    typedef uint32_t u32x2_ __attribute__((ext_vector_type(2)));
    const u32x2_ zero = {0u, 0u}, mine = {lo, hi};
    u32x2_ all;
    asm volatile("ds_write_b64 %1, %2\n\t"
                 "ds_or_b64 %1, %3\n\t"
                 "ds_read_b64 %0, %1\n\t"
                 "s_waitcnt lgkmcnt(0)"
                 : "=&v"(all) : "v"(scratch_addr), "v"(zero), "v"(mine) : "memory");
    return ((unsigned long long)(uint32_t)__builtin_amdgcn_readfirstlane((int)all[1]) << 32) |
           (uint32_t)__builtin_amdgcn_readfirstlane((int)all[0]);
}

__device__ __forceinline__ void atomic_add_f32(float* p, float v)
{
    __hip_atomic_fetch_add(p, v, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
}

#ifndef GSR_BWD_GRP
#define GSR_BWD_GRP 8
#endif
template <int... I, class F> __device__ __forceinline__ void static_for_impl(std::integer_sequence<int, I...>, F&& f)
{
    (f(std::integral_constant<int, I>{}), ...);
}
template <int N, class F> __device__ __forceinline__ void static_for(F&& f) { static_for_impl(std::make_integer_sequence<int, N>{}, f); }

constexpr int GRP = GSR_BWD_GRP;   // instances per MFMA group: A-operand rows 0..GRP-1 carry their r, the next GRP rows their w
constexpr int RSTRIDE = 68;     // floats per row of the r|w table: 64 pixels + 4 (16-byte aligned, spreads banks)

constexpr int BSEG = SNAP_SEG;
static_assert(BSEG == 64, "a unit is one 64-instance fetch batch (the forward's mask words have that granularity)");

}  // namespace gsr
