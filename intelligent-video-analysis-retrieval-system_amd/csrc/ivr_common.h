// Internal helpers shared by the libivr_hip.so translation units (gfx950 only).
#pragma once
#include <hip/hip_runtime.h>

#include <cstdarg>
#include <cstdint>
#include <cstdio>
#include <map>
#include <mutex>
#include <string>
#include <vector>

#include "../../include/ivr_api.h"

struct ivr_ctx {
    int device = 0;
    int cu_count = 256;
    int64_t hbm_bytes = 0;
    char arch[64] = {0};
    std::mutex mu;
    // Device scratch (preprocess intermediates): ONE grow-only block per stream.  Work of one stream is ordered, so calls
    // on the same stream may share a block; calls on different streams (the reference drives encode_images from a 4-thread
    // pool, unified_index.py:773) never do.  An outgrown block may still be referenced by kernels in flight or by a captured
    // HIP graph (StreamingSession), so it is only retired here and freed in ivr_destroy.
    struct Scratch {
        void *ptr = nullptr;
        size_t bytes = 0;
    };
    std::map<hipStream_t, Scratch> scratch;
    std::vector<void *> retired;
    std::mutex enqueue_mu;                 // held across a whole multi-launch enqueue that uses a scratch block
    std::map<std::string, float *> luts;   // preprocess value tables, keyed by the mean/std bytes
};

std::string &ivr_err_slot();
int ivr_fail(int code, const char *fmt, ...);

#define IVR_HIP(expr)                                                                              \
    do {                                                                                           \
        hipError_t e__ = (expr);                                                                   \
        if (e__ != hipSuccess)                                                                     \
            return ivr_fail(e__ == hipErrorOutOfMemory ? IVR_ERR_OOM : IVR_ERR_HIP, "%s: %s (%s:%d)", #expr, \
                            hipGetErrorString(e__), __FILE__, __LINE__);                           \
    } while (0)

#define IVR_REQUIRE(cond, ...)                                     \
    do {                                                           \
        if (!(cond)) return ivr_fail(IVR_ERR_INVALID, __VA_ARGS__); \
    } while (0)

#define IVR_LAUNCH_CHECK() IVR_HIP(hipGetLastError())

int ivr_ctx_scratch(ivr_ctx *ctx, hipStream_t stream, size_t bytes, void **out);
// hipFuncSetAttribute(MaxDynamicSharedMemorySize) once per (device, kernel) and size: the opt-in is per device, and towers /
// indexes of several devices may live in one process
int ivr_func_max_lds(const void *fn, int bytes);

// Opt-in per-kernel timing with HIP events on the launch stream (bench.py's roofline figures).
// `work` is the launch's algorithmic work: bytes for HBM-bound kernels, FLOP for MFMA-bound ones.
// Level 1 (ivr_profile_enable(ctx, 1)) brackets the kernels that carry the step - GEMMs, LayerNorm, attention, the index scans,
// the preprocess emit; level 2 also the short launches of the search tail and the index append (`minor`): an event pair costs a few
// microseconds ON the stream, which is nothing beside a GEMM and a quarter of a 0.28 ms search.
int ivr_prof_level();
void ivr_prof_begin(const char *name, hipStream_t s, double work);
void ivr_prof_end(hipStream_t s);
struct IvrProf {
    hipStream_t s;
    bool on;
    IvrProf(const char *name, hipStream_t st, double work, bool minor = false) : s(st), on(ivr_prof_level() >= (minor ? 2 : 1)) {
        if (on) ivr_prof_begin(name, s, work);
    }
    ~IvrProf() {
        if (on) ivr_prof_end(s);
    }
};

static inline int64_t ivr_ceil_div(int64_t a, int64_t b) { return (a + b - 1) / b; }
static inline int64_t ivr_round_up(int64_t a, int64_t b) { return ivr_ceil_div(a, b) * b; }

typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef short bf16x8 __attribute__((ext_vector_type(8)));
typedef short bf16x4 __attribute__((ext_vector_type(4)));

// float <-> order-preserving uint32 (larger float -> larger key); -0.0 is folded onto +0.0 first
__device__ __forceinline__ uint32_t ivr_f2ord(float f) {
    uint32_t u = __float_as_uint(f + 0.0f);
    return (u & 0x80000000u) ? ~u : (u | 0x80000000u);
}
__device__ __forceinline__ float ivr_ord2f(uint32_t o) {
    return __uint_as_float((o & 0x80000000u) ? (o & 0x7fffffffu) : ~o);
}

__device__ __forceinline__ float ivr_bf16_to_f32(unsigned short b) { return __uint_as_float(((uint32_t)b) << 16); }
// round-to-nearest-even; NaN stays NaN
__device__ __forceinline__ unsigned short ivr_f32_to_bf16(float f) {
    uint32_t u = __float_as_uint(f);
    if ((u & 0x7fffffffu) > 0x7f800000u) return (unsigned short)((u >> 16) | 0x40);
    return (unsigned short)((u + 0x7fffu + ((u >> 16) & 1u)) >> 16);
}

// two floats -> packed bf16 pair with the hardware converter (v_cvt_pk_bf16_f32: round-to-nearest-even, NaN stays NaN)
__device__ __forceinline__ uint32_t ivr_pack_bf16x2(float lo, float hi) {
    typedef __bf16 bf16x2_t __attribute__((ext_vector_type(2)));
    bf16x2_t v = {(__bf16)lo, (__bf16)hi};
    return __builtin_bit_cast(uint32_t, v);
}

// Wave-wide reductions on the DPP / permlane paths (a few cycles per step) instead of six ds_bpermute round trips through the
// LDS pipe.  Butterfly over lane bits 0..3 inside a row of 16 (quad_perm, quad_perm, row_half_mirror, row_mirror), then
// v_permlane16_swap / v_permlane32_swap across rows.  Every lane ends with the result.
template <int CTRL>
__device__ __forceinline__ float ivr_dpp(float x) {
    return __uint_as_float((unsigned)__builtin_amdgcn_update_dpp(0, (int)__float_as_uint(x), CTRL, 0xf, 0xf, false));
}
__device__ __forceinline__ float ivr_wave_sum(float v) {
    v += ivr_dpp<0xB1>(v);       // quad_perm [1,0,3,2]
    v += ivr_dpp<0x4E>(v);       // quad_perm [2,3,0,1]
    v += ivr_dpp<0x141>(v);      // row_half_mirror
    v += ivr_dpp<0x140>(v);      // row_mirror
    auto a = __builtin_amdgcn_permlane16_swap(__float_as_uint(v), __float_as_uint(v), false, false);
    v = __uint_as_float(a[0]) + __uint_as_float(a[1]);
    auto b = __builtin_amdgcn_permlane32_swap(__float_as_uint(v), __float_as_uint(v), false, false);
    return __uint_as_float(b[0]) + __uint_as_float(b[1]);
}
template <int CTRL>
__device__ __forceinline__ uint32_t ivr_dpp_u(uint32_t x) {
    return (uint32_t)__builtin_amdgcn_update_dpp(0, (int)x, CTRL, 0xf, 0xf, false);
}
__device__ __forceinline__ uint32_t ivr_wave_max_u32(uint32_t v) {
    v = max(v, ivr_dpp_u<0xB1>(v));
    v = max(v, ivr_dpp_u<0x4E>(v));
    v = max(v, ivr_dpp_u<0x141>(v));
    v = max(v, ivr_dpp_u<0x140>(v));
    auto a = __builtin_amdgcn_permlane16_swap(v, v, false, false);
    v = max((uint32_t)a[0], (uint32_t)a[1]);
    auto b = __builtin_amdgcn_permlane32_swap(v, v, false, false);
    return max((uint32_t)b[0], (uint32_t)b[1]);
}
__device__ __forceinline__ float ivr_wave_max(float v) {
    v = fmaxf(v, ivr_dpp<0xB1>(v));
    v = fmaxf(v, ivr_dpp<0x4E>(v));
    v = fmaxf(v, ivr_dpp<0x141>(v));
    v = fmaxf(v, ivr_dpp<0x140>(v));
    auto a = __builtin_amdgcn_permlane16_swap(__float_as_uint(v), __float_as_uint(v), false, false);
    v = fmaxf(__uint_as_float(a[0]), __uint_as_float(a[1]));
    auto b = __builtin_amdgcn_permlane32_swap(__float_as_uint(v), __float_as_uint(v), false, false);
    return fmaxf(__uint_as_float(b[0]), __uint_as_float(b[1]));
}
