// fast_f32.hpp -- correctly rounded float32 reciprocal and square root in 3 / 5 instructions, range-guarded (round 4).
//
// hipcc expands `1.0f / x` into v_div_scale x2, v_rcp, 6 fma-class instructions, v_div_fmas, v_div_fixup (11) and
// `sqrtf(x)` into scale-in, v_sqrt, a +-1 ulp fix-up with two fma / compare / select pairs, scale-out and a class test
// (15 + hazard nops).  The ORCA solve of the latency-bound env kernels is ONE dependent instruction stream in which a
// lone wavefront pays ~7 cycles per instruction whatever it is; it walks through ~4 roots and ~3 reciprocals per step
// (half-plane construction: |w|, 1/|w|, the tangent leg, 1/dist^2; the speed-disc clip 1/|v_pref|; the 1-D LP's
// discriminant).
//
//   rcp3(x)  = v_rcp_f32 + one fma Newton step (3 instructions)
//   sqrt5(x) = v_rsq_f32, g = x y, h = y / 2, one fma residual + one fma correction (5 instructions)
//
// tools/microbench/fast_math_exhaustive.hip compares both with the IEEE expansions for ALL 2^32 bit patterns on gfx950
// (profiles/r04_fast_math.txt): rcp3 returns the same bits for every operand with biased exponent 1 .. 252 (all normal
// x with |x| < 2^126, both signs), sqrt5 for every positive operand with biased exponent 25 .. 254 (x >= 2^-102).  A
// proof by exhaustion for this hardware -- and the only kind available: v_rcp_f32 / v_rsq_f32 are specified to 1 ulp,
// not bit by bit.  Outside those ranges (zero, denormals, huge, inf, NaN) the guarded wrappers below take the IEEE
// expansion: ONE wave-uniform branch per guard, so the float state stays bit-identical to the oracle for every input.
// `used` = lanes whose result is consumed; the operands of the others (empty neighbour slots, discarded select sides)
// never force the slow path.  MCN_FAST_F32 = 0 compiles the IEEE expansions everywhere (A/B runs).
#pragma once
#include <hip/hip_runtime.h>

#ifndef MCN_FAST_F32
#define MCN_FAST_F32 1
#endif

namespace mcn {

__device__ __forceinline__ float rcp3(float x)
{
    const float r0 = __builtin_amdgcn_rcpf(x);
    const float e = __builtin_fmaf(-x, r0, 1.0f);
    return __builtin_fmaf(e, r0, r0);
}

__device__ __forceinline__ float sqrt5(float x)
{
    const float y = __builtin_amdgcn_rsqf(x);
    const float g = x * y, h = 0.5f * y;
    const float d = __builtin_fmaf(-g, g, x);
    return __builtin_fmaf(d, h, g);
}

// v_cmp_class_f32 masks
constexpr int kClsPosNormal = 1 << 8, kClsNormal = (1 << 3) | (1 << 8);

// x in [2^-102, 2^128): x * 2^-24 is a positive normal number exactly then (a smaller x lands in the denormals, inf / NaN
// stay what they are) -- one multiply + one class test for the whole two-sided condition
__device__ __forceinline__ bool sqrt5_ok(float x) { return __builtin_amdgcn_classf(x * 0x1p-24f, kClsPosNormal); }
// x normal and |x| < 2^126
__device__ __forceinline__ bool rcp3_ok(float x) { return __builtin_amdgcn_classf(x, kClsNormal) & (__builtin_fabsf(x) < 0x1p126f); }

// no active lane that uses its result holds an operand outside the fast range
// (two ballots and a scalar and-not: `used & !ok` as booleans goes through vector registers)
__device__ __forceinline__ bool wave_fast(bool used, bool ok)
{
    return (__builtin_amdgcn_ballot_w64(used) & ~__builtin_amdgcn_ballot_w64(ok)) == 0;
}

// == sqrtf(x), bit for bit
__device__ __forceinline__ float sqrt_f32(float x, bool used = true)
{
#if MCN_FAST_F32
    if (wave_fast(used, sqrt5_ok(x))) return sqrt5(x);
#endif
    return sqrtf(x);
}

// == 1.0f / x, bit for bit
__device__ __forceinline__ float rcp_f32(float x, bool used = true)
{
#if MCN_FAST_F32
    if (wave_fast(used, rcp3_ok(x))) return rcp3(x);
#endif
    return 1.0f / x;
}

// == 1.0f / sqrtf(x), bit for bit (two roundings, as written); the root of an x in sqrt5's range lies in [2^-51, 2^64):
// inside rcp3's range, so one guard covers both
__device__ __forceinline__ float rcp_sqrt_f32(float x, bool used = true)
{
#if MCN_FAST_F32
    if (wave_fast(used, sqrt5_ok(x))) return rcp3(sqrt5(x));
#endif
    return 1.0f / sqrtf(x);
}

}  // namespace mcn
