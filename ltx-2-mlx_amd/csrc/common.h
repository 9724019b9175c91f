// Shared device/host helpers for the gfx950 (CDNA4) kernels of the LTX-2 hot path.
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <type_traits>

// `bf16` names the 16-bit activation / weight element type of THIS build of the library: bfloat16 in libltx2hip.so (the default), IEEE
// half in libltx2hip_f16.so (the same sources compiled with -DLTX2_F16: the reference's default compute dtype is float16,
// scripts/generate.py:1006).  Accumulation, the residual stream, statistics and tables are fp32 in both.  The C ABI is the same; dtype
// code LTX2_DTYPE_BF16 means "this build's 16-bit type".
#ifdef LTX2_F16
typedef _Float16 bf16;
#define LTX2_MFMA_32x32x16 __builtin_amdgcn_mfma_f32_32x32x16_f16
#define LTX2_MFMA_16x16x32 __builtin_amdgcn_mfma_f32_16x16x32_f16
#define LTX2_DT "f16"
#else
typedef __bf16 bf16;
#define LTX2_MFMA_32x32x16 __builtin_amdgcn_mfma_f32_32x32x16_bf16
#define LTX2_MFMA_16x16x32 __builtin_amdgcn_mfma_f32_16x16x32_bf16
#define LTX2_DT "bf16"
#endif
typedef bf16 bf16x8 __attribute__((ext_vector_type(8)));
typedef bf16 bf16x4 __attribute__((ext_vector_type(4)));
typedef bf16 bf16x2 __attribute__((ext_vector_type(2)));
typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef float f32x2 __attribute__((ext_vector_type(2)));
typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef unsigned int u32x4 __attribute__((ext_vector_type(4)));
typedef unsigned int u32x2 __attribute__((ext_vector_type(2)));

#define LTX2_OK 0
#define LTX2_E_INVALID (-1)    // bad argument / unsupported shape
#define LTX2_E_HIP (-2)        // HIP runtime error
#define LTX2_E_STATE (-3)      // call order / missing weight / workspace too small

// thread-local last-error string, readable through ltx2_last_error()
void ltx2_set_error(const char* fmt, ...);

#define LTX2_CHECK_ARG(cond, ...)            \
    do {                                     \
        if (!(cond)) {                       \
            ltx2_set_error(__VA_ARGS__);     \
            return LTX2_E_INVALID;           \
        }                                    \
    } while (0)

#define LTX2_CHECK_LAUNCH(name)                                                   \
    do {                                                                          \
        hipError_t e_ = hipGetLastError();                                        \
        if (e_ != hipSuccess) {                                                   \
            ltx2_set_error("%s: launch failed: %s", name, hipGetErrorString(e_)); \
            return LTX2_E_HIP;                                                    \
        }                                                                         \
    } while (0)

typedef __attribute__((address_space(3))) void* lds_ptr_t;

// Launchers raise a kernel's dynamic-LDS limit once per DEVICE (hipFuncSetAttribute acts on the current device): returns
// true the first time a given flag set sees the current device.
struct PerDeviceOnce {
    bool seen[64] = {};
    bool first() {
        int dev = 0;
        (void)hipGetDevice(&dev);
        if (dev < 0 || dev >= 64) return true;
        const bool f = !seen[dev];
        seen[dev] = true;
        return f;
    }
};

// Hand-issued LDS reads and waits. hipcc only ever emits `s_waitcnt lgkmcnt(0)` around ds_read_b128 in our
// loops, i.e. every wait drains ALL fragment reads in flight. With the read and the counted wait both in asm the
// compiler tracks neither; lds_wait<N> names the fragment it guards so the consuming MFMA cannot move above it.
// LGKM returns in order for LDS operations, so "at most N younger operations outstanding" proves the guarded
// read has landed; extra compiler-issued LGKM traffic only over-waits.
template <int OFF>
__device__ __forceinline__ u32x4 lds_read16(unsigned addr) {
    u32x4 v;
    asm volatile("ds_read_b128 %0, %1 offset:%2" : "=v"(v) : "v"(addr), "n"(OFF) : "memory");
    return v;
}
template <int N>
__device__ __forceinline__ void lds_wait(u32x4& v) {
    asm volatile("s_waitcnt lgkmcnt(%1)" : "+v"(v) : "n"(N) : "memory");
}
template <int I0, int I1, class F>
__device__ __forceinline__ void static_for(F&& f) {
    if constexpr (I0 < I1) {
        f(std::integral_constant<int, I0>{});
        static_for<I0 + 1, I1>(f);
    }
}
__device__ __forceinline__ bf16x8 as_bf16x8(u32x4 v) { return __builtin_bit_cast(bf16x8, v); }

__device__ __forceinline__ float bf2f(bf16 v) { return (float)v; }
__device__ __forceinline__ bf16 f2bf(float v) { return (bf16)v; }

// float8_e4m3fn code -> fp32 (OCP e4m3fn: bias 7, no infinities, 0x7f / 0xff = NaN); shared by the load-time dequantiser and
// the kernels that keep fp8 weights resident, so both produce bf16(f32(code) * scale) from the same arithmetic
__device__ __forceinline__ float e4m3fn_to_f32(unsigned int b) {
    const unsigned int sign = (b & 0x80u) << 24, exp = (b >> 3) & 0xfu, man = b & 7u;
    if ((b & 0x7fu) == 0x7fu) return __uint_as_float(0x7fc00000u | sign);
    if (exp == 0) return __uint_as_float(__float_as_uint((float)man * 0.001953125f) | sign);     // man * 2^-9
    return __uint_as_float(sign | ((exp + 120u) << 23) | (man << 20));
}


__device__ __forceinline__ float gelu_tanh(float x) {
    // 0.5 x (1 + tanh(sqrt(2/pi) (x + 0.044715 x^3)))  == x * sigmoid(2u) == x / (1 + 2^(-2u log2 e)).
    // v_exp_f32 + v_rcp_f32 (1 ulp each) instead of __expf and an IEEE division (a ~10-instruction sequence): the FFN-up
    // epilogue runs this 56.6 M times per launch with the matrix pipes idle.
    const float u = 0.7978845608028654f * (x + 0.044715f * x * x * x);
    return x * __builtin_amdgcn_rcpf(1.0f + __builtin_amdgcn_exp2f(-2.8853900817779268f * u));
}
// two elements per VALU op where the ISA has a packed fp32 form (v_pk_mul_f32 / v_pk_fma_f32 / v_pk_add_f32); v_exp_f32 and v_rcp_f32
// stay scalar.  Same formula as gelu_tanh().
__device__ __forceinline__ f32x2 gelu_tanh2(f32x2 x) {
    const f32x2 k0 = {0.044715f, 0.044715f}, one = {1.0f, 1.0f}, c = {-2.8853900817779268f * 0.7978845608028654f, -2.8853900817779268f * 0.7978845608028654f};
    const f32x2 t = __builtin_elementwise_fma(x * x, k0, one);          // 1 + 0.044715 x^2
    const f32x2 a = (x * c) * t;                                        // -2 u log2(e)
    const f32x2 e = {__builtin_amdgcn_exp2f(a[0]), __builtin_amdgcn_exp2f(a[1])};
    const f32x2 d = e + one;
    const f32x2 r = {__builtin_amdgcn_rcpf(d[0]), __builtin_amdgcn_rcpf(d[1])};
    return x * r;
}
__device__ __forceinline__ float silu_f(float x) { return x * __builtin_amdgcn_rcpf(1.0f + __builtin_amdgcn_exp2f(-1.4426950408889634f * x)); }

// v as seen through a DPP control word (row = 16 lanes): 0xB1 / 0x4E quad swaps, 0x141 row_half_mirror, 0x140 row_mirror -- four adds sum a row of 16 lanes
template <int CTRL>
__device__ __forceinline__ float dpp_f32(float v) {
    return __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, v), CTRL, 0xf, 0xf, true));
}

__device__ __forceinline__ float wave_sum(float v) {
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) v += __shfl_xor(v, o);
    return v;
}

// Bijective XCD-aware remap of a 1-D block id: consecutive logical ids land on the same
// XCD (observed placement: hardware block b -> XCD b % 8), so neighbouring tiles share an L2.
__device__ __forceinline__ int xcd_remap(int bid, int nwg) {
    const int xcd = bid & 7, q = nwg >> 3, r = nwg & 7;
    const int base = (xcd < r) ? xcd * (q + 1) : r * (q + 1) + (xcd - r) * q;
    return base + (bid >> 3);
}
