// Device-side helpers shared by the gfx950 kernels of libmixq_mi355x.so.
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

namespace mixq {

typedef int v4i __attribute__((ext_vector_type(4)));
typedef int v16i __attribute__((ext_vector_type(16)));
typedef float v16f __attribute__((ext_vector_type(16)));
typedef float v4f __attribute__((ext_vector_type(4)));
typedef _Float16 v8h __attribute__((ext_vector_type(8)));
typedef _Float16 v4h __attribute__((ext_vector_type(4)));
typedef _Float16 v2h __attribute__((ext_vector_type(2)));

#define MIXQ_GLOBAL_PTR(p) ((const __attribute__((address_space(1))) void*)(p))
#define MIXQ_LDS_PTR(p) ((__attribute__((address_space(3))) void*)(p))

__device__ __forceinline__ float h2f(uint16_t bits)
{
    _Float16 h;
    __builtin_memcpy(&h, &bits, 2);
    return (float)h;
}
__device__ __forceinline__ uint16_t f2h_bits(float f)
{
    _Float16 h = (_Float16)f; // v_cvt_f16_f32, round-to-nearest-even
    uint16_t b;
    __builtin_memcpy(&b, &h, 2);
    return b;
}

// fp32 -> fp16 with the fp32 value pinned in a VGPR first.  Without the (empty) asm hipcc folds
// "fptrunc(fma(...))" into v_fma_mixlo/mixhi_f16, whose result is not the twice-rounded value
// fp16(fp32(fma)) the reference computes (__float2half of an fp32 FMA) -- measured: rare 1-ulp differences.
__device__ __forceinline__ uint16_t f2h_bits_of_f32_result(float f)
{
    asm("" : "+v"(f));
    return f2h_bits(f);
}

// 16-byte async copy global -> LDS.  LDS destination = wave-uniform base + lane*16 (hardware rule); the
// per-lane part lives entirely in the global source address.
__device__ __forceinline__ void glds16(const void* gsrc, void* lds_wave_base)
{
    __builtin_amdgcn_global_load_lds(MIXQ_GLOBAL_PTR(gsrc), MIXQ_LDS_PTR(lds_wave_base), 16, 0, 0);
}

// One fp16 quotient with the reference's semantics (kernel/i8gemm.cu:103-104):
//   (int8) __half2int_rn( __hdiv(x, s) )
// __hdiv = correctly rounded fp16 division = RNE_fp16(fp32 IEEE quotient) (innocuous double rounding);
// __half2int_rn: RNE, NaN -> 0, +-inf saturate to INT_MAX/INT_MIN; the int8 cast keeps the low 8 bits.
__device__ __forceinline__ int quant_one(float x, float s)
{
    float q = x / s;                    // IEEE-correct fp32 division (hipcc default, no fast-math)
    float qh = (float)((_Float16)q);    // RNE to fp16 (overflow -> inf), back to fp32 exactly
    float r = __builtin_rintf(qh);      // v_rndne_f32
    // clamp first: (int) of inf/NaN is undefined in C++; every finite fp16 is within +-65504
    int i = (int)__builtin_fminf(__builtin_fmaxf(r, -65536.f), 65536.f);
    i = (qh != qh) ? 0 : i;
    i = (__builtin_isinf(qh)) ? (qh > 0.f ? 0x7fffffff : (int)0x80000000) : i;
    return i & 0xff;
}

} // namespace mixq
