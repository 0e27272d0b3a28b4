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

// Measurement only (NULL in production): thread 0 of every workgroup writes the 100 MHz wall clock (s_memrealtime: one
// time base across kernels and CUs) into slot `slot` of its 8-slot record (tools/small_m_timeline.py).
__device__ __forceinline__ void dbg_stamp(void* buf, int slot)
{
    if (buf != nullptr && threadIdx.x == 0)
        static_cast<unsigned long long*>(buf)[(size_t)blockIdx.x * 8 + slot] = wall_clock64();
}

// 16-byte weight-stream load, optionally NON-TEMPORAL (no L2 / Infinity-Cache allocation): for weight tensors so large that a model's
// decode step, which reads every layer's weights exactly once, can never find them cache-resident (w8a16_gemm_kernels.hip, NTW)
template <bool NT>
__device__ __forceinline__ uint4 wload16(const void* ptr)
{
    if constexpr (NT) return __builtin_bit_cast(uint4, __builtin_nontemporal_load(reinterpret_cast<const v4i*>(ptr)));
    else return *reinterpret_cast<const uint4*>(ptr);
}

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

// SiLU as the reference writes it (linear_combination_dequant.h:167-170, cult.cu:2301-2304): x / (1 + expf(-x)) with the
// accurate library expf (<= 1 ulp, like CUDA's expf without -use_fast_math) and an IEEE division -- not the
// hardware v_exp_f32 approximation (__expf).
__device__ __forceinline__ float silu_f32(float x) { return x / (1.f + expf(-x)); }

// fp32 -> fp16 with the fp32 value pinned in a VGPR first.  Without the (empty) asm hipcc folds
// "fptrunc(fma(...))" into v_fma_mixlo/mixhi_f16, whose result is not the twice-rounded value
// fp16(fp32(fma)) the reference computes (__float2half of an fp32 FMA) -- measured: rare 1-ulp differences.
__device__ __forceinline__ uint16_t f2h_bits_of_f32_result(float f)
{
    asm("" : "+v"(f));
    return f2h_bits(f);
}

// Two fp32 results -> packed fp16 (RNE), both pinned first (same reason); one v_cvt_pk_f16_f32.
__device__ __forceinline__ v2h f2h2_of_f32_results(float a, float b)
{
    asm("" : "+v"(a), "+v"(b));
    typedef float v2f_ __attribute__((ext_vector_type(2)));
    return __builtin_convertvector(v2f_{a, b}, v2h);
}

// Two fp32 values that are NOT results of an fp32 FMA/add in this kernel (e.g. MFMA accumulators) -> packed fp16 (RNE).
__device__ __forceinline__ v2h f2h2(float a, float b)
{
    typedef float v2f_ __attribute__((ext_vector_type(2)));
    return __builtin_convertvector(v2f_{a, b}, v2h);
}

// 16-byte async copy global -> LDS.  LDS destination = wave-uniform base + lane*16 (hardware rule); the
// per-lane part lives entirely in the global source address.
__device__ __forceinline__ void glds16(const void* gsrc, void* lds_wave_base)
{
    __builtin_amdgcn_global_load_lds(MIXQ_GLOBAL_PTR(gsrc), MIXQ_LDS_PTR(lds_wave_base), 16, 0, 0);
}

// (non-temporal form: aux = 2 is the NT cache-policy bit of gfx94x / gfx950 -- for weight tiles that exactly one workgroup row reads)
__device__ __forceinline__ void glds16_nt(const void* gsrc, void* lds_wave_base)
{
    __builtin_amdgcn_global_load_lds(MIXQ_GLOBAL_PTR(gsrc), MIXQ_LDS_PTR(lds_wave_base), 16, 0, 2);
}

// The same copy in asm form: wave-uniform 64-bit base in SGPRs + per-lane 32-bit byte offset; LDS destination = M0
// (wave-uniform LDS byte address) + lane * 16.  M0 is saved / restored inside the statement (it is compiler-reserved).
// WHY asm: the compiler's s_waitcnt insertion treats the builtin as a FLAT access that may touch LDS and memory, and
// while one is pending every dependency wait becomes vmcnt(0) / lgkmcnt(0) -- a loop that keeps DMA copies in flight
// loses its whole register prefetch (every use of a loaded register drains all copies) and every LDS read waits for the
// youngest one.  The asm form is not counted: every consumer of the copied bytes must sit behind an explicit
// `s_waitcnt vmcnt(n)` (n counted over ALL VMEM operations of the thread, these included) and a barrier.
__device__ __forceinline__ void glds16_sbase(const char* sbase, unsigned voff, unsigned lds_addr)
{
    unsigned keep;
    asm volatile("s_mov_b32 %0, m0\n\t"
                 "s_mov_b32 m0, %3\n\t"
                 "s_nop 0\n\t"
                 "global_load_lds_dwordx4 %1, %2\n\t"
                 "s_mov_b32 m0, %0"
                 : "=&s"(keep)
                 : "v"(voff), "s"(sbase), "s"(lds_addr)
                 : "memory");
}

// (non-temporal form of glds16_sbase: weight tiles that exactly one workgroup row reads)
__device__ __forceinline__ void glds16_sbase_nt(const char* sbase, unsigned voff, unsigned lds_addr)
{
    unsigned keep;
    asm volatile("s_mov_b32 %0, m0\n\t"
                 "s_mov_b32 m0, %3\n\t"
                 "s_nop 0\n\t"
                 "global_load_lds_dwordx4 %1, %2 nt\n\t"
                 "s_mov_b32 m0, %0"
                 : "=&s"(keep)
                 : "v"(voff), "s"(sbase), "s"(lds_addr)
                 : "memory");
}

// The asm form with a per-lane 64-bit source address (for copies whose source is not base + 32-bit offset, e.g. a row
// clamp or a redirection to the zero page).  Same rules: not counted by the compiler.
__device__ __forceinline__ void glds16_vaddr(const void* gsrc, unsigned lds_addr)
{
    unsigned keep;
    asm volatile("s_mov_b32 %0, m0\n\t"
                 "s_mov_b32 m0, %2\n\t"
                 "s_nop 0\n\t"
                 "global_load_lds_dwordx4 %1, off\n\t"
                 "s_mov_b32 m0, %0"
                 : "=&s"(keep)
                 : "v"(gsrc), "s"(lds_addr)
                 : "memory");
}

__device__ __forceinline__ void glds16_vaddr_nt(const void* gsrc, unsigned lds_addr)
{
    unsigned keep;
    asm volatile("s_mov_b32 %0, m0\n\t"
                 "s_mov_b32 m0, %2\n\t"
                 "s_nop 0\n\t"
                 "global_load_lds_dwordx4 %1, off nt\n\t"
                 "s_mov_b32 m0, %0"
                 : "=&s"(keep)
                 : "v"(gsrc), "s"(lds_addr)
                 : "memory");
}

// 16-byte load into registers, asm form (wave-uniform base + per-lane byte offset + immediate): NOT counted by the
// compiler either, for loops that keep register prefetches in flight next to asm-form LDS-DMA copies (a counted load's
// wait would be computed without the copies and drain part of them).  The destination must not be read before an explicit
// `s_waitcnt vmcnt(n)` followed by vmem_landed() on it (which makes every later use depend on a statement behind the wait).
typedef unsigned v4u __attribute__((ext_vector_type(4)));
template <int IMM>
__device__ __forceinline__ void gload16_sbase(v4u& dst, const char* sbase, unsigned voff)
{
    asm volatile("global_load_dwordx4 %0, %1, %2 offset:%3" : "=v"(dst) : "v"(voff), "s"(sbase), "n"(IMM) : "memory");
}
__device__ __forceinline__ void vmem_landed(v4u& a, v4u& b, v4u& c, v4u& d)
{
    asm volatile("" : "+v"(a), "+v"(b), "+v"(c), "+v"(d)::"memory");
}

// One fp16 quotient with the reference's semantics (kernel/i8gemm.cu:103-104):
//   (int8) __half2int_rn( __hdiv(x, s) )
// __hdiv = correctly rounded fp16 division = RNE_fp16(fp32 IEEE quotient) (innocuous double rounding);
// __half2int_rn: RNE, NaN -> 0, +-inf saturate to INT_MAX/INT_MIN; the int8 cast keeps the low 8 bits.
__device__ __forceinline__ int half_to_int8_bits(float q)
{
    float qh = (float)((_Float16)q);    // RNE to fp16 (overflow -> inf), back to fp32 exactly
    float r = __builtin_rintf(qh);      // v_rndne_f32
    int i;
    asm("v_cvt_i32_f32 %0, %1" : "=v"(i) : "v"(r)); // hardware semantics: NaN -> 0, out of range saturates
    return i & 0xff;
}

__device__ __forceinline__ int quant_one(float x, float s)
{
    return half_to_int8_bits(x / s);    // IEEE-correct fp32 division (hipcc default, no fast-math)
}

// Same result as quant_one(x, s) given rs = fl32(1/s), without a division in the common case.
// q0 = fl32(x * rs) is within 2^-23 relative (< 2 ulp32) of the exact quotient T; RNE_fp16(q0) can differ from
// RNE_fp16(T) only if an fp16 rounding breakpoint lies between them.  In the normal fp16 range every breakpoint has
// fp32 mantissa bits [12:0] == 0x1000, so q0 is safe unless its low 13 bits are within +-3 of 0x1000; then the exact
// division is used.  Below 2^-14 both round to an fp16 < 0.5 -> integer 0 either way; zero / inf / NaN scales give
// the same inf / NaN patterns as the division (x*inf, 0*inf, x*0).  (Argument in DESIGN.md 2.2.)
__device__ __forceinline__ int quant_one_fast(float x, float s, float rs)
{
    float q0 = x * rs;
    const unsigned low = (__builtin_bit_cast(unsigned, q0) + 3u - 0x1000u) & 0x1fffu;
    if (__builtin_expect(low <= 6u, 0)) q0 = x / s;
    return half_to_int8_bits(q0);
}

// max of the two 16-bit halves of a and b, per half (v_pk_max_u16)
__device__ __forceinline__ unsigned pk_max_u16(unsigned a, unsigned b)
{
    typedef unsigned short v2u16 __attribute__((ext_vector_type(2)));
    return __builtin_bit_cast(unsigned, __builtin_elementwise_max(__builtin_bit_cast(v2u16, a), __builtin_bit_cast(v2u16, b)));
}

// 8 fp16 -> 8 int8 with the semantics of quant_one(), for rows in which every element is finite and the scale s is
// finite and non-zero (then |x / s| < 512).  Per pair of elements: fp32 products with rs = fl32(1/s) (v_pk_mul_f32),
// one RNE conversion to packed fp16 (= fp16(x / s) unless the product sits next to a rounding breakpoint, see
// quant_one_fast), and round-to-integer + low-8-bits in ONE packed fp16 add: h + 1536 has ulp 1, so its mantissa
// field is 512 + rint(h) and the low byte of the bit pattern is the two's-complement int8.  The fp16 value is pinned
// in a VGPR before the add (a fused fp32 -> fp16 add would round once instead of twice).  Elements whose product lies
// within 3 ulp32 of a breakpoint are recomputed with the exact division.
__device__ __forceinline__ uint2 quant_vec8_finite(const uint4& xv, float s, float rs)
{
    typedef float v2f_ __attribute__((ext_vector_type(2)));
    const unsigned w[4] = {xv.x, xv.y, xv.z, xv.w};
    unsigned rb[4];
    unsigned rmin = 0xffffffffu;
#pragma unroll
    for (int e = 0; e < 4; ++e) {
        const v2h hx = __builtin_bit_cast(v2h, w[e]);
        const v2f_ q = v2f_{(float)hx[0], (float)hx[1]} * v2f_{rs, rs};
        const float q0 = q[0], q1 = q[1]; // (copies: __builtin_bit_cast of a vector-element lvalue reads element 0)
        const unsigned a0 = (__builtin_bit_cast(unsigned, q0) + 3u - 0x1000u) & 0x1fffu;
        const unsigned a1 = (__builtin_bit_cast(unsigned, q1) + 3u - 0x1000u) & 0x1fffu;
        rmin = min(rmin, min(a0, a1));
        unsigned hb = __builtin_bit_cast(unsigned, __builtin_convertvector(q, v2h));
        asm("" : "+v"(hb));
        const v2h r = __builtin_bit_cast(v2h, hb) + v2h{(_Float16)1536.f, (_Float16)1536.f};
        rb[e] = __builtin_bit_cast(unsigned, r);
    }
    uint2 o;
    o.x = __builtin_amdgcn_perm(rb[1], rb[0], 0x06040200u);
    o.y = __builtin_amdgcn_perm(rb[3], rb[2], 0x06040200u);
    if (__builtin_expect(rmin <= 6u, 0)) {
        unsigned ow[2] = {o.x, o.y};
#pragma unroll
        for (int e = 0; e < 8; ++e) {
            const float x = h2f((uint16_t)(w[e >> 1] >> ((e & 1) * 16)));
            const unsigned low = (__builtin_bit_cast(unsigned, x * rs) + 3u - 0x1000u) & 0x1fffu;
            if (low <= 6u) {
                const unsigned b = (unsigned)quant_one(x, s);
                ow[e >> 2] = (ow[e >> 2] & ~(0xffu << ((e & 3) * 8))) | (b << ((e & 3) * 8));
            }
        }
        o.x = ow[0], o.y = ow[1];
    }
    return o;
}

} // namespace mixq
