// Kernels of the encoder towers (E1/E2/E3 + N1 of SURVEY.md section 8a), gfx950 only.
//
// The arithmetic follows what the reference obtains from HuggingFace at core.py:1619-1620,
// core.py:1541-1542 and video_frame_filter.py:31-32 (modeling_clip.py:138-218, 259-383, 594-656,
// 719-753, 541-581; modeling_vit.py:261-281, 348, 385):
//   residual stream, LayerNorm statistics, softmax and every accumulation in float32;
//   GEMM operands bf16 on v_mfma_f32_16x16x32_bf16 (IVR_COMPUTE_BF16) or float32 on
//   v_mfma_f32_16x16x4_f32 (IVR_COMPUTE_F32, verification mode).
//
// GEMM: C[m][n] = sum_k X[m][k] * W[n][k]  (nn.Linear: both operands K-contiguous).
//   128 x 128 x 128-byte tile, 4 waves as 2(m) x 2(n), each wave 64 x 64 = 4 x 4 MFMA tiles.
//   The MFMA is issued with W as the A operand and X as the B operand, so a lane ends up holding
//   4 consecutive n of one row m: bias/activation/residual epilogues are float4-wide and the
//   residual stream is updated in place.
//   Staging is LDS-DMA (global_load_lds_dwordx4, 1 KiB = 8 full 128-byte rows per wave instruction) into two
//   LDS buffers: the next tile's DMA is in flight under the current tile's MFMAs, one barrier per K step.
//   LDS rows are 128 bytes; 16-byte chunk c of row r sits at chunk position (c ^ (r & 7)) - applied on the DMA
//   source address and on the fragment read - which makes the ds_read_b128 of MFMA fragments bank-conflict free.
//   Workgroup ids are remapped so that the tiles sharing an X row panel run on one XCD (one L2).
#include "tower.h"
#include "tower_kernels.h"

#include <map>
#include <mutex>
#include <type_traits>

#include <algorithm>
#include <cstdlib>

namespace {

// ---------------------------------------------------------------------------------------------
// element helpers: T = unsigned short (bf16 bits) or float
// ---------------------------------------------------------------------------------------------
template <typename T>
struct El;
template <>
struct El<unsigned short> {
    static __device__ __forceinline__ void unpack(const uint4 &u, float (&f)[8]) {
        f[0] = __uint_as_float(u.x << 16);
        f[1] = __uint_as_float(u.x & 0xffff0000u);
        f[2] = __uint_as_float(u.y << 16);
        f[3] = __uint_as_float(u.y & 0xffff0000u);
        f[4] = __uint_as_float(u.z << 16);
        f[5] = __uint_as_float(u.z & 0xffff0000u);
        f[6] = __uint_as_float(u.w << 16);
        f[7] = __uint_as_float(u.w & 0xffff0000u);
    }
    static __device__ __forceinline__ void store4(unsigned short *p, const float (&v)[4]) {
        uint2 o;
        o.x = ivr_pack_bf16x2(v[0], v[1]);
        o.y = ivr_pack_bf16x2(v[2], v[3]);
        *reinterpret_cast<uint2 *>(p) = o;
    }
};
// four floats -> four OCP e4m3 bytes, saturated to +-448 (v_med3 first: the converter's overflow result is mode-dependent)
__device__ __forceinline__ uint32_t ivr_pack_fp8x4(float a, float b, float c, float d) {
    a = __builtin_amdgcn_fmed3f(a, -448.f, 448.f);
    b = __builtin_amdgcn_fmed3f(b, -448.f, 448.f);
    c = __builtin_amdgcn_fmed3f(c, -448.f, 448.f);
    d = __builtin_amdgcn_fmed3f(d, -448.f, 448.f);
    int w = 0;
    w = __builtin_amdgcn_cvt_pk_fp8_f32(a, b, w, false);
    w = __builtin_amdgcn_cvt_pk_fp8_f32(c, d, w, true);
    return (uint32_t)w;
}
template <>
struct El<unsigned char> {       // e4m3 activations of the fp8 mode
    static __device__ __forceinline__ void store4(unsigned char *p, const float (&v)[4]) {
        *reinterpret_cast<uint32_t *>(p) = ivr_pack_fp8x4(v[0], v[1], v[2], v[3]);
    }
};
template <>
struct El<float> {
    static __device__ __forceinline__ void store4(float *p, const float (&v)[4]) {
        *reinterpret_cast<float4 *>(p) = make_float4(v[0], v[1], v[2], v[3]);
    }
};

// FAST = bf16 output: v_exp + v_rcp (1 ulp each) instead of an IEEE division; the f32 verification mode keeps it exact
template <bool FAST>
__device__ __forceinline__ float act_fn(float x, int act) {
    if (act == IVR_ACT_QUICK_GELU) {                                          // x * sigmoid(1.702 x)
        if (FAST) {
            // exp(-1.702 x) = exp2(x * (-1.702 * log2 e)): ONE multiply in front of the raw v_exp_f32 instead of two (the fc1
            // epilogue is VALU-bound: 128 elements per lane at ~34 issue cycles each were 10 k of a 41 k-cycle tile)
            const float e = __builtin_amdgcn_exp2f(x * -2.4554669595930157f);
            return x * __builtin_amdgcn_rcpf(1.0f + e);
        }
        const float e = __expf(-1.702f * x);
        return x / (1.0f + e);
    }
    if (FAST) {
        // exact-GELU with erf from Abramowitz & Stegun 7.1.26 (|error| <= 1.5e-7, far inside the bf16 / e4m3 output rounding):
        // 2 transcendentals + 7 FMAs per element instead of the ~40-instruction erff expansion, which made the fc1 epilogue of
        // the DINO tower (K = 384: six K steps per tile) three times as long as its K loop.
        const float z = fabsf(x) * 0.70710678118654752f;
        const float tt = __builtin_amdgcn_rcpf(fmaf(0.3275911f, z, 1.0f));
        float p = fmaf(1.061405429f, tt, -1.453152027f);
        p = fmaf(p, tt, 1.421413741f);
        p = fmaf(p, tt, -0.284496736f);
        p = fmaf(p, tt, 0.254829592f);
        const float e = __builtin_amdgcn_exp2f(-z * z * 1.4426950408889634f);
        const float erf_abs = fmaf(-p * tt, e, 1.0f);
        return 0.5f * x * (1.0f + copysignf(erf_abs, x));
    }
    return 0.5f * x * (1.0f + erff(x * 0.70710678118654752f));
}

// Four adjacent outputs at once, bf16-output epilogues: quick_gelu written on float2 vectors so that the multiplies and the add run
// as v_pk_mul_f32 / v_pk_add_f32 (two elements per issue slot; hipcc does not pack them on its own around the transcendentals).
// Same operations per element as act_fn<true>, so the results are identical.
__device__ __forceinline__ void act4_fast(float (&v)[4], int act) {
    if (act == IVR_ACT_QUICK_GELU) {
        typedef float f32x2_t __attribute__((ext_vector_type(2)));
        const f32x2_t c = {-2.4554669595930157f, -2.4554669595930157f}, one = {1.0f, 1.0f};
        f32x2_t a = {v[0], v[1]}, b = {v[2], v[3]};
        const f32x2_t ta = a * c, tb = b * c;
        const f32x2_t da = f32x2_t{__builtin_amdgcn_exp2f(ta.x), __builtin_amdgcn_exp2f(ta.y)} + one;
        const f32x2_t db = f32x2_t{__builtin_amdgcn_exp2f(tb.x), __builtin_amdgcn_exp2f(tb.y)} + one;
        a = a * f32x2_t{__builtin_amdgcn_rcpf(da.x), __builtin_amdgcn_rcpf(da.y)};
        b = b * f32x2_t{__builtin_amdgcn_rcpf(db.x), __builtin_amdgcn_rcpf(db.y)};
        v[0] = a.x;
        v[1] = a.y;
        v[2] = b.x;
        v[3] = b.y;
        return;
    }
#pragma unroll
    for (int i = 0; i < 4; ++i) v[i] = act_fn<true>(v[i], act);
}

// ---------------------------------------------------------------------------------------------
// embeddings
// ---------------------------------------------------------------------------------------------
__global__ __launch_bounds__(256) void vision_cls_kernel(float *__restrict__ resid, const float *__restrict__ cls,
                                                         const float *__restrict__ pos, int T, int D) {
    float *row = resid + (int64_t)blockIdx.x * T * D;
    for (int k = threadIdx.x; k < D; k += blockDim.x) row[k] = cls[k] + pos[k];
}

__global__ __launch_bounds__(256) void text_embed_kernel(float *__restrict__ resid, const int64_t *__restrict__ ids,
                                                         const float *__restrict__ tok, const float *__restrict__ pos,
                                                         int T, int D, int vocab) {
    const int64_t r = blockIdx.x;          // q*T + t
    const int t = (int)(r % T);
    int64_t id = ids[r];
    id = id < 0 ? 0 : (id >= vocab ? vocab - 1 : id);
    const float *e = tok + id * D;
    for (int k = threadIdx.x; k < D; k += blockDim.x) resid[r * D + k] = e[k] + pos[(int64_t)t * D + k];
}

// first position of eos_id in each row, 0 when absent (modeling_clip.py:566-576: (ids == eos).int().argmax(-1))
__global__ void eos_pos_kernel(const int64_t *__restrict__ ids, int q, int T, int eos, int *__restrict__ out) {
    const int r = blockIdx.x * blockDim.x + threadIdx.x;
    if (r >= q) return;
    int p = 0;
    for (int t = 0; t < T; ++t)
        if (ids[(int64_t)r * T + t] == eos) {
            p = t;
            break;
        }
    out[r] = p;
}

// ---------------------------------------------------------------------------------------------
// LayerNorm: one wave per row, row held in registers, two-pass statistics in float32
// source row of output row r: r * row_mul + (offs ? offs[r] : 0)
// ---------------------------------------------------------------------------------------------
template <typename TOut>
__global__ __launch_bounds__(256) void layernorm_kernel(const float *__restrict__ x, int row_mul, const int *__restrict__ offs,
                                                        const float *__restrict__ g, const float *__restrict__ b, float eps,
                                                        TOut *__restrict__ out, int rows, int D, int reverse) {
    const int lane = threadIdx.x & 63;
    int row = blockIdx.x * 4 + (threadIdx.x >> 6);
    if (row >= rows) return;
    if (reverse) row = rows - 1 - row;
    const int64_t srow = (int64_t)row * row_mul + (offs ? offs[row] : 0);
    const float4 *src = reinterpret_cast<const float4 *>(x + srow * D);
    const int nv = D >> 2;
    float4 v[8];
    float s = 0.f;
#pragma unroll
    for (int i = 0; i < 8; ++i) {
        const int idx = lane + 64 * i;
        v[i] = make_float4(0.f, 0.f, 0.f, 0.f);
        if (idx < nv) {
            v[i] = src[idx];
            s += (v[i].x + v[i].y) + (v[i].z + v[i].w);
        }
    }
    const float mean = ivr_wave_sum(s) / (float)D;
    float ss = 0.f;
#pragma unroll
    for (int i = 0; i < 8; ++i) {
        const int idx = lane + 64 * i;
        if (idx < nv) {
            const float a = v[i].x - mean, bb = v[i].y - mean, c = v[i].z - mean, d = v[i].w - mean;
            ss += (a * a + bb * bb) + (c * c + d * d);
        }
    }
    const float rstd = 1.0f / sqrtf(ivr_wave_sum(ss) / (float)D + eps);
    TOut *orow = out + (int64_t)row * D;
#pragma unroll
    for (int i = 0; i < 8; ++i) {
        const int idx = lane + 64 * i;
        if (idx < nv) {
            const float4 gg = reinterpret_cast<const float4 *>(g)[idx], bv = reinterpret_cast<const float4 *>(b)[idx];
            float y[4];
            y[0] = (v[i].x - mean) * rstd * gg.x + bv.x;
            y[1] = (v[i].y - mean) * rstd * gg.y + bv.y;
            y[2] = (v[i].z - mean) * rstd * gg.z + bv.z;
            y[3] = (v[i].w - mean) * rstd * gg.w + bv.w;
            El<TOut>::store4(orow + idx * 4, y);
        }
    }
}

// F.normalize(x, p=2, dim=1): x / max(||x||, 1e-12)  (core.py:1620)
__global__ __launch_bounds__(256) void f_normalize_kernel(const float *__restrict__ x, float *__restrict__ out, int n, int d,
                                                          int normalize) {
    const int lane = threadIdx.x & 63;
    const int row = blockIdx.x * 4 + (threadIdx.x >> 6);
    if (row >= n) return;
    const float *p = x + (int64_t)row * d;
    float ss = 0.f;
    for (int k = lane; k < d; k += 64) ss = fmaf(p[k], p[k], ss);
    ss = ivr_wave_sum(ss);
    const float den = normalize ? fmaxf(sqrtf(ss), 1e-12f) : 1.0f;
    for (int k = lane; k < d; k += 64) out[(int64_t)row * d + k] = p[k] / den;
}

// bf16 -> saturated e4m3, 8 elements per thread (a bf16 fc1 feeding an e4m3 fc2: error-budget configurations of the fp8 mode)
__global__ __launch_bounds__(256) void bf16_to_e4m3_kernel(const uint4 *__restrict__ src, uint2 *__restrict__ dst, int64_t n8) {
    const int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x;
    if (i >= n8) return;
    float f[8];
    El<unsigned short>::unpack(src[i], f);
    uint2 o;
    o.x = ivr_pack_fp8x4(f[0], f[1], f[2], f[3]);
    o.y = ivr_pack_fp8x4(f[4], f[5], f[6], f[7]);
    dst[i] = o;
}

// ---------------------------------------------------------------------------------------------
// GEMM
// ---------------------------------------------------------------------------------------------
typedef uint32_t u32x4 __attribute__((ext_vector_type(4)));
typedef uint32_t u32x2 __attribute__((ext_vector_type(2)));
constexpr int BM = 128, BN = 128, ROWB = 128;            // ROWB: bytes of K per LDS row and K step
constexpr int TILE_BYTES = BM * ROWB;                     // 16 KiB per operand per stage
constexpr int GEMM_LDS = 2 * 2 * TILE_BYTES;              // 64 KiB

template <typename T>
__device__ __forceinline__ void mma_chunk(const u32x4 &w, const u32x4 &x, f32x4 &acc);
template <>
__device__ __forceinline__ void mma_chunk<unsigned short>(const u32x4 &w, const u32x4 &x, f32x4 &acc) {
    typedef __bf16 bf16x8_t __attribute__((ext_vector_type(8)));
    acc = __builtin_amdgcn_mfma_f32_16x16x32_bf16(__builtin_bit_cast(bf16x8_t, w), __builtin_bit_cast(bf16x8_t, x), acc, 0, 0, 0);
}
template <>
__device__ __forceinline__ void mma_chunk<float>(const u32x4 &w, const u32x4 &x, f32x4 &acc) {
    acc = __builtin_amdgcn_mfma_f32_16x16x4f32(__uint_as_float(w.x), __uint_as_float(x.x), acc, 0, 0, 0);
    acc = __builtin_amdgcn_mfma_f32_16x16x4f32(__uint_as_float(w.y), __uint_as_float(x.y), acc, 0, 0, 0);
    acc = __builtin_amdgcn_mfma_f32_16x16x4f32(__uint_as_float(w.z), __uint_as_float(x.z), acc, 0, 0, 0);
    acc = __builtin_amdgcn_mfma_f32_16x16x4f32(__uint_as_float(w.w), __uint_as_float(x.w), acc, 0, 0, 0);
}

template <typename T, int EPI, int ACT>
__global__ __launch_bounds__(256, 2) void gemm_kernel(GemmArgs g) {
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);   // provably wave-uniform: keeps the DMA addressing scalar
    const int wm = wave >> 1, wn = wave & 1;
    const int MT = (g.M + BM - 1) / BM, NT = (g.N + BN - 1) / BN;
    // Tile order.  Workgroups b, b+8, b+16, ... share an XCD (one L2).  XCD x owns the row panels p = x (mod 8); inside
    // an XCD the panels are walked in groups of GROUP_M: a group's X panels (GROUP_M x 128 rows x K) stay in that L2 while
    // its column tiles are swept, and neighbouring workgroups use the same W tile.
    int tm, tn;
    {
        const int xcd = blockIdx.x & 7, i = blockIdx.x >> 3;
        const int lx = MT > xcd ? (MT - xcd + 7) >> 3 : 0;      // panels owned by this XCD
        if (i >= lx * NT) return;
        const int gm = g.group_m;
        const int per = gm * NT, grp = i / per, within = i - grp * per;
        const int gme = min(gm, lx - grp * gm);
        tm = xcd + 8 * (grp * gm + within % gme);
        tn = within / gme;
        if (g.reverse_m) tm = MT - 1 - tm;
    }
    const int m0 = tm * BM, n0 = tn * BN;
    constexpr int EPR = ROWB / (int)sizeof(T);   // elements of K per step
    const int KT = g.K / EPR;
    const unsigned lds0 = (unsigned)(size_t)smem;

    // staging: LDS-DMA (buffer_load_dwordx4 ... lds).  One wave instruction fills 8 rows x 128 B = 1 KiB: lane l lands at
    // (wave-uniform M0 base) + 16*l, i.e. row 8i + (l >> 3), chunk position l & 7.  The XOR swizzle therefore goes on the
    // SOURCE: the lane fetches logical chunk (l & 7) ^ (row & 7) of its row, and fragment reads apply the same involution.
    // 16 such pieces per operand per K step; wave w issues pieces 4w .. 4w+3 of X and of W.  The per-lane address part is
    // a constant voffset; tile origin and K step are a scalar soffset; rows past the matrix end read as zeros.
    unsigned voffX[4], voffW[4];
#pragma unroll
    for (int j = 0; j < 4; ++j) {
        const int r = 8 * (wave * 4 + j) + (lane >> 3), c = (lane & 7) ^ (lane >> 3);
        voffX[j] = (unsigned)(r * g.lda) * (unsigned)sizeof(T) + c * 16;
        voffW[j] = (unsigned)(r * g.ldw) * (unsigned)sizeof(T) + c * 16;
    }
    const auto rsX = __builtin_amdgcn_make_buffer_rsrc(const_cast<void *>(g.A), 0, (int)((int64_t)g.M * g.lda * sizeof(T)), 0x00020000);
    const auto rsW = __builtin_amdgcn_make_buffer_rsrc(const_cast<void *>(g.W), 0, (int)((int64_t)g.N * g.ldw * sizeof(T)), 0x00020000);
    const unsigned sx0 = (unsigned)m0 * (unsigned)g.lda * (unsigned)sizeof(T), sw0 = (unsigned)n0 * (unsigned)g.ldw * (unsigned)sizeof(T);
    auto stage = [&](int kt, int buf) {
        unsigned char *base = smem + buf * 2 * TILE_BYTES + (wave * 4) * 1024;
        const unsigned adv = (unsigned)kt * ROWB;
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            __builtin_amdgcn_raw_ptr_buffer_load_lds(rsX, (__attribute__((address_space(3))) void *)(base + j * 1024), 16, voffX[j],
                                                     sx0 + adv, 0, 0);
            __builtin_amdgcn_raw_ptr_buffer_load_lds(rsW, (__attribute__((address_space(3))) void *)(base + TILE_BYTES + j * 1024), 16,
                                                     voffW[j], sw0 + adv, 0, 0);
        }
    };
    // per-lane fragment offsets (operand, kk) inside a buffer; the four 16-row tiles are immediate offsets
    unsigned foX[2], foW[2];
#pragma unroll
    for (int kk = 0; kk < 2; ++kk) {
        const unsigned f = (lane & 15) * ROWB + ((((kk << 2) + (lane >> 4)) ^ (lane & 7)) << 4);
        foX[kk] = lds0 + (wm * 64) * ROWB + f;
        foW[kk] = lds0 + TILE_BYTES + (wn * 64) * ROWB + f;
    }

    f32x4 acc[4][4];   // [nt][mt]
#pragma unroll
    for (int a = 0; a < 4; ++a)
#pragma unroll
        for (int b = 0; b < 4; ++b) acc[a][b] = f32x4{0.f, 0.f, 0.f, 0.f};

    stage(0, 0);
    for (int kt = 0; kt < KT; ++kt) {
        // tile kt has landed for every wave, and every wave is done reading the other buffer (tile kt-1)
        asm volatile("s_waitcnt vmcnt(0)\n\ts_barrier" ::: "memory");
        if (kt + 1 < KT) stage(kt + 1, (kt + 1) & 1);          // in flight under this tile's MFMAs
        // Fragment reads are inline asm on purpose: hipcc cannot prove that a ds_read of the current buffer does not
        // alias the LDS-DMA just issued into the other one and would put s_waitcnt vmcnt(0) in front of the first
        // compiler-visible LDS read, serialising DMA and MFMA.  lgkmcnt is counted by hand; the sched_barrier keeps the
        // MFMAs below the wait (they only have register operands, a "memory" clobber does not order them).
        const unsigned boff = (kt & 1) * 2 * TILE_BYTES;
        u32x4 xf[2][4], wf[2][4];
#pragma unroll
        for (int kk = 0; kk < 2; ++kk) {
            const unsigned xa = foX[kk] + boff, wa = foW[kk] + boff;
            asm volatile("ds_read_b128 %0, %1" : "=v"(xf[kk][0]) : "v"(xa));
            asm volatile("ds_read_b128 %0, %1" : "=v"(wf[kk][0]) : "v"(wa));
            asm volatile("ds_read_b128 %0, %1 offset:2048" : "=v"(xf[kk][1]) : "v"(xa));
            asm volatile("ds_read_b128 %0, %1 offset:2048" : "=v"(wf[kk][1]) : "v"(wa));
            asm volatile("ds_read_b128 %0, %1 offset:4096" : "=v"(xf[kk][2]) : "v"(xa));
            asm volatile("ds_read_b128 %0, %1 offset:4096" : "=v"(wf[kk][2]) : "v"(wa));
            asm volatile("ds_read_b128 %0, %1 offset:6144" : "=v"(xf[kk][3]) : "v"(xa));
            asm volatile("ds_read_b128 %0, %1 offset:6144" : "=v"(wf[kk][3]) : "v"(wa));
        }
        asm volatile("s_waitcnt lgkmcnt(8)" ::: "memory");
        __builtin_amdgcn_sched_barrier(0);
#pragma unroll
        for (int nt = 0; nt < 4; ++nt)
#pragma unroll
            for (int mt = 0; mt < 4; ++mt) mma_chunk<T>(wf[0][nt], xf[0][mt], acc[nt][mt]);
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
        __builtin_amdgcn_sched_barrier(0);
#pragma unroll
        for (int nt = 0; nt < 4; ++nt)
#pragma unroll
            for (int mt = 0; mt < 4; ++mt) mma_chunk<T>(wf[1][nt], xf[1][mt], acc[nt][mt]);
    }

    // epilogue: lane holds C[m][n..n+3], m = .. + (lane & 15), n = .. + 4 * (lane >> 4).
    // All global reads of the epilogue (bias, residual, position rows) are issued back to back before the first use so
    // that their latency is paid once per tile, not once per fragment.
    int mrow[4], ncol[4];
    bool mok[4], nok[4];
    float4 bv[4];
#pragma unroll
    for (int t = 0; t < 4; ++t) {
        mrow[t] = m0 + wm * 64 + t * 16 + (lane & 15);
        ncol[t] = n0 + wn * 64 + t * 16 + 4 * (lane >> 4);
        mok[t] = mrow[t] < g.M;
        nok[t] = ncol[t] < g.N;
        mrow[t] = min(mrow[t], g.M - 1);
        ncol[t] = min(ncol[t], g.N - 4);
        bv[t] = g.bias ? *reinterpret_cast<const float4 *>(g.bias + ncol[t]) : make_float4(0.f, 0.f, 0.f, 0.f);
    }
    if (EPI == EPI_RESID || EPI == EPI_PATCH) {
        float4 rv[4][4];
        float *rowp[4];
#pragma unroll
        for (int mt = 0; mt < 4; ++mt) {
            if (EPI == EPI_RESID) {
                rowp[mt] = g.resid + (int64_t)mrow[mt] * g.ldr;
#pragma unroll
                for (int nt = 0; nt < 4; ++nt) rv[mt][nt] = *reinterpret_cast<const float4 *>(rowp[mt] + ncol[nt]);
            } else {
                const int img = mrow[mt] / g.G2, pch = mrow[mt] % g.G2;
                rowp[mt] = g.resid + ((int64_t)img * g.T + 1 + pch) * g.ldr;
#pragma unroll
                for (int nt = 0; nt < 4; ++nt)
                    rv[mt][nt] = *reinterpret_cast<const float4 *>(g.pos + (int64_t)(1 + pch) * g.N + ncol[nt]);
            }
        }
#pragma unroll
        for (int mt = 0; mt < 4; ++mt)
#pragma unroll
            for (int nt = 0; nt < 4; ++nt) {
                float4 r = rv[mt][nt];
                r.x += acc[nt][mt][0] + bv[nt].x;
                r.y += acc[nt][mt][1] + bv[nt].y;
                r.z += acc[nt][mt][2] + bv[nt].z;
                r.w += acc[nt][mt][3] + bv[nt].w;
                if (mok[mt] && nok[nt] && !(EPI == EPI_RESID && g.skip_mod && mrow[mt] % g.skip_mod == 0))
                    *reinterpret_cast<float4 *>(rowp[mt] + ncol[nt]) = r;
            }
    } else {
#pragma unroll
        for (int mt = 0; mt < 4; ++mt)
#pragma unroll
            for (int nt = 0; nt < 4; ++nt) {
                float v[4] = {acc[nt][mt][0] + bv[nt].x, acc[nt][mt][1] + bv[nt].y, acc[nt][mt][2] + bv[nt].z,
                              acc[nt][mt][3] + bv[nt].w};
                if (ACT >= 0) {
#pragma unroll
                    for (int i = 0; i < 4; ++i) v[i] = act_fn<sizeof(T) == 2>(v[i], ACT);
                }
                if (mok[mt] && nok[nt]) {
                    if (EPI == EPI_STORE)
                        El<T>::store4(reinterpret_cast<T *>(g.out) + (int64_t)mrow[mt] * g.ldo + ncol[nt], v);
                    else   // EPI_F32
                        *reinterpret_cast<float4 *>(reinterpret_cast<float *>(g.out) + (int64_t)mrow[mt] * g.ldo + ncol[nt]) =
                            make_float4(v[0], v[1], v[2], v[3]);
                }
            }
    }
}

// Wide epilogue of the 128 x 64 wave block (8 x 4 MFMA tiles): the block goes through 16 KiB of the (by then idle) LDS
// owned by the wave, so that every global store / residual load covers whole rows of the block (bf16: 8 rows x 128 B per
// instruction, f32 residual: 4 rows x 256 B) instead of 16 rows x 32 or 64 B straight from the MFMA layout.  The bf16
// store tail is bound by the number of vector-memory instructions and the lines each one touches, not by bytes (s_memtime
// stamps: 18.9k -> 6.3k cycles of a 55k-cycle qkv tile).  LDS images are XOR-swizzled by row: writes and reads are
// conflict-free.
template <typename T, int EPI, int ACT, bool SCALED = false, bool OUT8 = false, bool SKIP = false>
__device__ __forceinline__ void wide_epilogue(const GemmArgs &g, f32x4 (&acc)[4][8], unsigned char *wb, int row0, int col0, int lane, bool inside) {
    // Called by every wave of the workgroup: it contains the barrier that separates the last fragment reads of the K loop from the
    // epilogue's use of LDS.  `inside`: the wave's 64-column block lies inside N (a block is all in or all out).  The residual
    // epilogue issues its first loads BEFORE that barrier: they go to registers, and the wait for the slowest wave then overlaps
    // with their latency.
    const int r = lane & 15, gq = lane >> 4;
    if (EPI != EPI_RESID) {
        __builtin_amdgcn_s_barrier();
        if (!inside) return;
    }
    float4 bv[4];
#pragma unroll
    for (int nt = 0; nt < 4; ++nt)
        bv[nt] = g.bias && inside ? *reinterpret_cast<const float4 *>(g.bias + col0 + nt * 16 + 4 * gq) : make_float4(0.f, 0.f, 0.f, 0.f);
    if (SCALED) {                 // fp8 GEMM: per-column dequantisation scale of the weight rows
#pragma unroll
        for (int nt = 0; nt < 4; ++nt) {
            const float4 sv = g.colscale && inside ? *reinterpret_cast<const float4 *>(g.colscale + col0 + nt * 16 + 4 * gq) : make_float4(1.f, 1.f, 1.f, 1.f);
#pragma unroll
            for (int mt = 0; mt < 8; ++mt) {
                acc[nt][mt][0] *= sv.x;
                acc[nt][mt][1] *= sv.y;
                acc[nt][mt][2] *= sv.z;
                acc[nt][mt][3] *= sv.w;
            }
        }
    }
    if (EPI == EPI_STORE && OUT8) {
        // e4m3 output: 64-byte rows in LDS, 16-B chunk nt at position nt ^ ((row >> 1) & 3); read back 16 rows x 64 B per store
#pragma unroll
        for (int mt = 0; mt < 8; ++mt)
#pragma unroll
            for (int nt = 0; nt < 4; ++nt) {
                const f32x4 a = acc[nt][mt];
                float v[4] = {a[0] + bv[nt].x, a[1] + bv[nt].y, a[2] + bv[nt].z, a[3] + bv[nt].w};
                if (ACT >= 0) act4_fast(v, ACT);
                *reinterpret_cast<uint32_t *>(wb + (mt * 16 + r) * 64 + ((nt ^ ((r >> 1) & 3)) << 4) + gq * 4) =
                    ivr_pack_fp8x4(v[0], v[1], v[2], v[3]);
            }
        unsigned char *outp = reinterpret_cast<unsigned char *>(g.out) + col0 + (lane & 3) * 16;
#pragma unroll
        for (int it = 0; it < 8; ++it) {
            const int R = it * 16 + (lane >> 2);
            const uint4 v = *reinterpret_cast<const uint4 *>(wb + R * 64 + (((lane & 3) ^ ((R >> 1) & 3)) << 4));
            if (row0 + R < g.M) *reinterpret_cast<uint4 *>(outp + (int64_t)(row0 + R) * g.ldo) = v;
        }
        return;
    }
    if (EPI == EPI_STORE) {
#pragma unroll
        for (int mt = 0; mt < 8; ++mt)
#pragma unroll
            for (int nt = 0; nt < 4; ++nt) {
                const f32x4 a = acc[nt][mt];
                float v[4] = {a[0] + bv[nt].x, a[1] + bv[nt].y, a[2] + bv[nt].z, a[3] + bv[nt].w};
                if (ACT >= 0) act4_fast(v, ACT);
                uint2 o;
                o.x = ivr_pack_bf16x2(v[0], v[1]);
                o.y = ivr_pack_bf16x2(v[2], v[3]);
                const int chunk = (nt * 2 + (gq >> 1)) ^ (r & 7);
                *reinterpret_cast<uint2 *>(wb + (mt * 16 + r) * 128 + chunk * 16 + (gq & 1) * 8) = o;
            }
        unsigned short *outp = reinterpret_cast<unsigned short *>(g.out) + col0 + (lane & 7) * 8;
#pragma unroll
        for (int it = 0; it < 16; ++it) {
            const int R = it * 8 + (lane >> 3);
            const uint4 v = *reinterpret_cast<const uint4 *>(wb + R * 128 + (((lane & 7) ^ (R & 7)) << 4));
            if (row0 + R < g.M) *reinterpret_cast<uint4 *>(outp + (int64_t)(row0 + R) * g.ldo) = v;
        }
    } else {
        // Residual read-modify-write of the 128 x 64 block in four batches of 32 rows (8 float4 per lane each).  The batches are software
        // pipelined over two register sets: both batches of the first half are in flight before the workgroup's barrier and the staging
        // of the accumulators, those of the second half are issued as soon as a set has been stored - one partly exposed memory round
        // trip per tile instead of four (the fragment registers of the K loop are dead here: 64 registers of residual fit beside the 128
        // accumulators; a third set was tried and spills 13 registers, whose reloads wait for every load in flight).
        // The block is addressed through its own buffer descriptor (wave-uniform base = its first row, 32-bit offsets, no 64-bit address
        // arithmetic); the descriptor ends with the last valid row, so rows past M load zeros and their stores are dropped by the range
        // check.  The row offset is added into the VGPR offset: the hardware's range check does not look at the scalar offset.
        typedef unsigned u32x4_t __attribute__((ext_vector_type(4)));
        float *blk = g.resid + (int64_t)row0 * g.ldr + col0;
        const int vrows = min(128, g.M - row0);
        const auto rsR = __builtin_amdgcn_make_buffer_rsrc(blk, 0, vrows > 0 ? ((vrows - 1) * g.ldr + 64) * 4 : 0, 0x00020000);
        const unsigned voffR = (unsigned)((lane >> 4) * g.ldr + (lane & 15) * 4) * 4u;
        const unsigned rowB = (unsigned)g.ldr * 4u;
        u32x4_t rvA[8], rvB[8];
        auto loadb = [&](u32x4_t (&rv)[8], int half, int b8) {
#pragma unroll
            for (int it = 0; it < 8; ++it)
                rv[it] = __builtin_amdgcn_raw_buffer_load_b128(rsR, voffR + (unsigned)(half * 64 + (b8 * 8 + it) * 4) * rowB, 0, 0);
        };
        auto stage = [&](int half) {
#pragma unroll
            for (int mt = 0; mt < 4; ++mt)
#pragma unroll
                for (int nt = 0; nt < 4; ++nt) {
                    const f32x4 a = acc[nt][half * 4 + mt];
                    const int chunk = (nt * 4 + gq) ^ r;
                    *reinterpret_cast<float4 *>(wb + (mt * 16 + r) * 256 + chunk * 16) =
                        make_float4(a[0] + bv[nt].x, a[1] + bv[nt].y, a[2] + bv[nt].z, a[3] + bv[nt].w);
                }
        };
        auto storeb = [&](const u32x4_t (&rv)[8], int half, int b8) {
#pragma unroll
            for (int it = 0; it < 8; ++it) {
                const int R = (b8 * 8 + it) * 4 + (lane >> 4), grow = row0 + half * 64 + R;
                const float4 v = *reinterpret_cast<const float4 *>(wb + R * 256 + (((lane & 15) ^ (R & 15)) << 4));
                u32x4_t o;
                o.x = __float_as_uint(__uint_as_float(rv[it].x) + v.x);
                o.y = __float_as_uint(__uint_as_float(rv[it].y) + v.y);
                o.z = __float_as_uint(__uint_as_float(rv[it].z) + v.z);
                o.w = __float_as_uint(__uint_as_float(rv[it].w) + v.w);
                if (!(SKIP && grow % g.skip_mod == 0))
                    __builtin_amdgcn_raw_buffer_store_b128(o, rsR, voffR + (unsigned)(half * 64 + (b8 * 8 + it) * 4) * rowB, 0, 0);
            }
        };
        if (inside) {
            loadb(rvA, 0, 0);
            loadb(rvB, 0, 1);
        }
        __builtin_amdgcn_s_barrier();
        if (!inside) return;
        stage(0);
        storeb(rvA, 0, 0);
        loadb(rvA, 1, 0);
        storeb(rvB, 0, 1);
        loadb(rvB, 1, 1);
        stage(1);                 // the wave's LDS image is reused: its reads of the first half were issued before, LDS works in order
        storeb(rvA, 1, 0);
        storeb(rvB, 1, 1);
    }
}

#ifdef IVR_GEMM_STAMPS
// diagnostic build only: per-workgroup s_memtime stamps (entry, first stage landed, K loop done, stores drained)
__device__ unsigned long long ivr_gemm_stamps[16384][8];   // [4], [5]: s_memrealtime (100 MHz) at stamps 0 and 3; [6]: extra stamp
#define IVR_STAMP(I)                                                                                       \
    if (threadIdx.x == 0 && blockIdx.x < 16384) {                                                          \
        ivr_gemm_stamps[blockIdx.x][I] = __builtin_amdgcn_s_memtime();                                     \
        if ((I) == 0) ivr_gemm_stamps[blockIdx.x][4] = __builtin_amdgcn_s_memrealtime();                   \
        if ((I) == 3) ivr_gemm_stamps[blockIdx.x][5] = __builtin_amdgcn_s_memrealtime();                   \
    }
#else
#define IVR_STAMP(I)
#endif

template <int N>
__device__ __forceinline__ void wait_vm_barrier() {
    asm volatile("s_waitcnt vmcnt(%0)\n\ts_barrier" ::"n"(N) : "memory");
}

// ---------------------------------------------------------------------------------------------
// GEMM, large-tile version: 256 x 256 tile, 8 waves as 2(m) x 4(n), each wave 128 x 64 = 8 x 4 MFMA tiles.
// Same staging scheme as gemm_kernel (two LDS buffers of 64 KiB filled by buffer_load ... lds, swizzled source, asm
// fragment reads), but half the operand bytes per FLOP, 0.375 instead of 0.5 fragment reads per MFMA, and a quarter of
// the tile prologues / epilogues per FLOP.  One workgroup per CU (128 KiB of LDS), two waves per SIMD.
//   * software pipeline across the barrier: per stage a wave holds two fragment sets (the two 32-wide K halves).  Set 1
//     of stage kt is read under the MFMAs of set 0, and set 0 of stage kt+1 under the MFMAs of set 1, so no MFMA waits
//     for a read issued after a barrier.  The single wait + barrier of a stage sits in its MIDDLE: by then every wave
//     has read both sets of stage kt, so that buffer can take the DMA of stage kt+2, and stage kt+1 (issued one stage
//     earlier) has landed.
//   * the 8 DMA pieces and 12 fragment reads a wave issues per stage are spread between groups of four MFMAs instead of
//     being issued back to back behind the barrier (64 pieces at once from the 8 waves hold every wave's issue with no
//     MFMA queued behind them).  The last two pieces of a stage go out early in the following stage.
//   * s_memtime stamps (-DIVR_GEMM_STAMPS, tools/gemm_stamps.py) on a K=768 qkv tile: prologue 2.5k, K loop 30.8k
//     (2566 cycles per stage, 2048 = MFMA-bound), epilogue 6.7k cycles.
// ---------------------------------------------------------------------------------------------
constexpr int LBM = 256, LBN = 256, LX_BYTES = LBM * ROWB, LW_BYTES = LBN * ROWB;
constexpr int DEEP_LDS = 3 * LX_BYTES + 2 * LW_BYTES;   // 160 KiB (bf16 / f32 kernel: three X slots, two W slots)

// SKIP: EPI_RESID leaves rows r % skip_mod == 0 alone (a template parameter: the runtime test cost the plain instantiation 68 bytes
// of scratch per lane)
template <typename T, int EPI, int ACT, bool SKIP = false>
__device__ __forceinline__ void gemm_big_tile(const GemmArgs &g, const int vb, unsigned char *smem) {
    IVR_STAMP(0)
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int wm = wave >> 2, wn = wave & 3;
    const int MT = (g.M + LBM - 1) / LBM, NT = (g.N + LBN - 1) / LBN;
    int tm, tn;
    {
        const int xcd = vb & 7, i = vb >> 3;          // vb: the tile's position in the launch order (blockIdx.x of the plain kernel)
        const int lx = MT > xcd ? (MT - xcd + 7) >> 3 : 0;
        if (i >= lx * NT) return;
        const int gm = g.group_m;
        const int per = gm * NT, grp = i / per, within = i - grp * per;
        const int gme = min(gm, lx - grp * gm);
        tm = xcd + 8 * (grp * gm + within % gme);
        tn = within / gme;
        if (g.reverse_m) tm = MT - 1 - tm;
    }
    const int m0 = tm * LBM, n0 = tn * LBN;
    constexpr int EPR = ROWB / (int)sizeof(T);
    const int KT = g.K / EPR;
    const unsigned lds0 = (unsigned)(size_t)smem;

    // 32 + 32 pieces of 1 KiB per stage; wave w issues pieces 4w .. 4w+3 of X and of W
    unsigned voffX[4], voffW[4];
#pragma unroll
    for (int j = 0; j < 4; ++j) {
        const int r = 8 * (wave * 4 + j) + (lane >> 3), c = (lane & 7) ^ (lane >> 3);
        voffX[j] = (unsigned)(r * g.lda) * (unsigned)sizeof(T) + c * 16;
        voffW[j] = (unsigned)(r * g.ldw) * (unsigned)sizeof(T) + c * 16;
    }
    const auto rsX = __builtin_amdgcn_make_buffer_rsrc(const_cast<void *>(g.A), 0, (int)((int64_t)g.M * g.lda * sizeof(T)), 0x00020000);
    const auto rsW = __builtin_amdgcn_make_buffer_rsrc(const_cast<void *>(g.W), 0, (int)((int64_t)g.N * g.ldw * sizeof(T)), 0x00020000);
    const unsigned sx0 = (unsigned)m0 * (unsigned)g.lda * (unsigned)sizeof(T), sw0 = (unsigned)n0 * (unsigned)g.ldw * (unsigned)sizeof(T);
    // LDS: three X slots (the activation operand is usually HBM-fed: its DMA runs TWO stages ahead) and two W slots (the
    // weight panel is L2 / Infinity-Cache resident: one stage ahead), 3 x 32 + 2 x 32 = 160 KiB.
    // s_memtime stamps of the two-slot version: per stage the waves waited ~1000 cycles for the X DMA on the fc2 / attn-out
    // shapes (A read once from HBM by 3 column tiles) against ~150 on qkv / fc1 (A panel shared by 9 - 12 tiles through L2).
    constexpr int WBASE = 3 * LX_BYTES;
    auto piece = [&](int kt, int xs, int j) {          // one 1 KiB DMA piece of stage kt: j < 4 X (slot xs), else W (slot kt & 1)
        const unsigned adv = (unsigned)kt * ROWB;
        if (j < 4)
            __builtin_amdgcn_raw_ptr_buffer_load_lds(rsX, (__attribute__((address_space(3))) void *)(smem + xs * LX_BYTES + (wave * 4 + j) * 1024),
                                                     16, voffX[j], sx0 + adv, 0, 0);
        else
            __builtin_amdgcn_raw_ptr_buffer_load_lds(
                rsW, (__attribute__((address_space(3))) void *)(smem + WBASE + (kt & 1) * LW_BYTES + (wave * 4 + j - 4) * 1024), 16,
                voffW[j - 4], sw0 + adv, 0, 0);
    };
    unsigned foX[2], foW[2];
#pragma unroll
    for (int kk = 0; kk < 2; ++kk) {
        const unsigned f = (lane & 15) * ROWB + ((((kk << 2) + (lane >> 4)) ^ (lane & 7)) << 4);
        foX[kk] = lds0 + (wm * 128) * ROWB + f;
        foW[kk] = lds0 + WBASE + (wn * 64) * ROWB + f;
    }

    f32x4 acc[4][8];   // [nt][mt]
#pragma unroll
    for (int a = 0; a < 4; ++a)
#pragma unroll
        for (int b = 0; b < 8; ++b) acc[a][b] = f32x4{0.f, 0.f, 0.f, 0.f};

#define IVR_ROW(XF, WF, MT)                                                                                \
    _Pragma("unroll") for (int nt = 0; nt < 4; ++nt) mma_chunk<T>(WF[nt], XF[MT], acc[nt][MT]);            \
    __builtin_amdgcn_sched_barrier(0);
#define IVR_RD4(DST, ADDR, O0)                                                                             \
    asm volatile("ds_read_b128 %0, %1 offset:%2" : "=v"(DST[0]) : "v"(ADDR), "n"(O0));                      \
    asm volatile("ds_read_b128 %0, %1 offset:%2" : "=v"(DST[1]) : "v"(ADDR), "n"(O0 + 2048));               \
    asm volatile("ds_read_b128 %0, %1 offset:%2" : "=v"(DST[2]) : "v"(ADDR), "n"(O0 + 4096));               \
    asm volatile("ds_read_b128 %0, %1 offset:%2" : "=v"(DST[3]) : "v"(ADDR), "n"(O0 + 6144));
#define IVR_LGKM(N)                                                                                        \
    asm volatile("s_waitcnt lgkmcnt(" #N ")" ::: "memory");                                                \
    __builtin_amdgcn_sched_barrier(0);

    u32x4 xa0[8], wa0[4], xa1[8], wa1[4];
    u32x4 *x0lo = xa0, *x0hi = xa0 + 4, *x1lo = xa1, *x1hi = xa1 + 4;
    // prologue: stage 0, then X(1), W(1), X(2) in that order (the order the counted waits below rely on), all issued before
    // the first wait: every slot is free at the start of a tile, and stage 1 then has the whole latency of stage 0 as a head start
#pragma unroll
    for (int j = 0; j < 8; ++j) piece(0, 0, j);
    if (KT > 1) {
#pragma unroll
        for (int j = 0; j < 8; ++j) piece(1, 1, j);
    }
    if (KT > 2) {
#pragma unroll
        for (int j = 0; j < 4; ++j) piece(2, 2, j);
    }
    if (KT > 2) asm volatile("s_waitcnt vmcnt(12)\n\ts_barrier" ::: "memory");
    else if (KT > 1) asm volatile("s_waitcnt vmcnt(8)\n\ts_barrier" ::: "memory");
    else asm volatile("s_waitcnt vmcnt(0)\n\ts_barrier" ::: "memory");
    IVR_STAMP(1)
    IVR_RD4(wa0, foW[0], 0)
    IVR_RD4(x0lo, foX[0], 0)
    IVR_RD4(x0hi, foX[0], 8192)
    int xs = 0;                                         // X slot of stage kt = kt % 3
    for (int kt = 0; kt < KT; ++kt) {
        const int xs1 = xs == 2 ? 0 : xs + 1;           // slot of stage kt + 1; stage kt + 3 reuses slot xs
        const unsigned xoff = xs * LX_BYTES, woff = (kt & 1) * LW_BYTES, nxoff = xs1 * LX_BYTES, nwoff = ((kt + 1) & 1) * LW_BYTES;
        // DMA windows: after the mid-stage barrier of stage s a wave issues W(s+2) x 4 and then X(s+3) x 4; the last two X
        // pieces go out early in stage s+1 (`tail`, none for the X(2) the prologue issued whole)
        const bool tail = kt >= 1 && kt + 2 < KT;
        const bool morew = kt + 2 < KT, morex = kt + 3 < KT, next = kt + 1 < KT;
        const unsigned wa = foW[1] + woff, xa = foX[1] + xoff, nwa = foW[0] + nwoff, nxa = foX[0] + nxoff;
        IVR_LGKM(4)                     // W + first four X fragments of set 0
        IVR_ROW(xa0, wa0, 0)
        if (tail) piece(kt + 2, xs1 == 2 ? 0 : xs1 + 1, 2);
        IVR_ROW(xa0, wa0, 1)
        IVR_RD4(wa1, wa, 0)
        IVR_ROW(xa0, wa0, 2)
        if (tail) piece(kt + 2, xs1 == 2 ? 0 : xs1 + 1, 3);
        IVR_ROW(xa0, wa0, 3)
        IVR_RD4(x1lo, xa, 0)
        IVR_LGKM(8)                     // all of set 0
        IVR_ROW(xa0, wa0, 4)
        IVR_ROW(xa0, wa0, 5)
        IVR_RD4(x1hi, xa, 8192)
        IVR_ROW(xa0, wa0, 6)
        IVR_ROW(xa0, wa0, 7)
        IVR_LGKM(0)                     // this wave has read everything it needs from the stage's buffers
        if (next) {
            // X(kt+1) and W(kt+1) must have landed; the four youngest pieces in flight are X(kt+2), which may stay
            if (kt + 2 < KT) asm volatile("s_waitcnt vmcnt(4)\n\ts_barrier" ::: "memory");
            else asm volatile("s_waitcnt vmcnt(0)\n\ts_barrier" ::: "memory");
        }
        IVR_ROW(xa1, wa1, 0)
        if (morew) piece(kt + 2, 0, 4);
        IVR_ROW(xa1, wa1, 1)
        if (next) { IVR_RD4(wa0, nwa, 0) }
        if (morew) piece(kt + 2, 0, 5);
        IVR_ROW(xa1, wa1, 2)
        if (morew) piece(kt + 2, 0, 6);
        IVR_ROW(xa1, wa1, 3)
        if (next) { IVR_RD4(x0lo, nxa, 0) }
        if (morew) piece(kt + 2, 0, 7);
        IVR_ROW(xa1, wa1, 4)
        if (morex) piece(kt + 3, xs, 0);
        IVR_ROW(xa1, wa1, 5)
        if (next) { IVR_RD4(x0hi, nxa, 8192) }
        IVR_ROW(xa1, wa1, 6)
        if (morex) piece(kt + 3, xs, 1);
        IVR_ROW(xa1, wa1, 7)
        xs = xs1;
    }
    IVR_LGKM(0)
    IVR_STAMP(2)
#undef IVR_ROW
#undef IVR_RD4
#undef IVR_LGKM

    if (sizeof(T) == 2 && (EPI == EPI_STORE || EPI == EPI_RESID) && g.wide_epi) {
        // (contains the barrier after which every wave has read its last fragments; N is a multiple of 64: a wave block is all in or all out)
        wide_epilogue<T, EPI, ACT, false, false, SKIP>(g, acc, smem + wave * 16384, m0 + wm * 128, n0 + wn * 64, lane, n0 + wn * 64 < g.N);
#ifdef IVR_GEMM_STAMPS
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        IVR_STAMP(3)
#endif
        return;
    }
    // epilogue in two halves of four row tiles (keeps the batched residual loads at 64 registers)
    int ncol[4];
    bool nok[4];
    float4 bv[4];
#pragma unroll
    for (int t = 0; t < 4; ++t) {
        ncol[t] = n0 + wn * 64 + t * 16 + 4 * (lane >> 4);
        nok[t] = ncol[t] < g.N;
        ncol[t] = min(ncol[t], g.N - 4);
        bv[t] = g.bias ? *reinterpret_cast<const float4 *>(g.bias + ncol[t]) : make_float4(0.f, 0.f, 0.f, 0.f);
    }
#pragma unroll
    for (int half = 0; half < 2; ++half) {
        int mrow[4];
        bool mok[4];
#pragma unroll
        for (int t = 0; t < 4; ++t) {
            mrow[t] = m0 + wm * 128 + (half * 4 + t) * 16 + (lane & 15);
            mok[t] = mrow[t] < g.M;
            mrow[t] = min(mrow[t], g.M - 1);
        }
        if (EPI == EPI_RESID || EPI == EPI_PATCH) {
            float4 rv[4][4];
            float *rowp[4];
#pragma unroll
            for (int mt = 0; mt < 4; ++mt) {
                if (EPI == EPI_RESID) {
                    rowp[mt] = g.resid + (int64_t)mrow[mt] * g.ldr;
#pragma unroll
                    for (int nt = 0; nt < 4; ++nt) rv[mt][nt] = *reinterpret_cast<const float4 *>(rowp[mt] + ncol[nt]);
                } else {
                    const int img = mrow[mt] / g.G2, pch = mrow[mt] % g.G2;
                    rowp[mt] = g.resid + ((int64_t)img * g.T + 1 + pch) * g.ldr;
#pragma unroll
                    for (int nt = 0; nt < 4; ++nt)
                        rv[mt][nt] = *reinterpret_cast<const float4 *>(g.pos + (int64_t)(1 + pch) * g.N + ncol[nt]);
                }
            }
#pragma unroll
            for (int mt = 0; mt < 4; ++mt)
#pragma unroll
                for (int nt = 0; nt < 4; ++nt) {
                    float4 r = rv[mt][nt];
                    const f32x4 a = acc[nt][half * 4 + mt];
                    r.x += a[0] + bv[nt].x;
                    r.y += a[1] + bv[nt].y;
                    r.z += a[2] + bv[nt].z;
                    r.w += a[3] + bv[nt].w;
                    if (mok[mt] && nok[nt] && !(SKIP && mrow[mt] % g.skip_mod == 0))
                        *reinterpret_cast<float4 *>(rowp[mt] + ncol[nt]) = r;
                }
        } else {
#pragma unroll
            for (int mt = 0; mt < 4; ++mt)
#pragma unroll
                for (int nt = 0; nt < 4; ++nt) {
                    const f32x4 a = acc[nt][half * 4 + mt];
                    float v[4] = {a[0] + bv[nt].x, a[1] + bv[nt].y, a[2] + bv[nt].z, a[3] + bv[nt].w};
                    if (ACT >= 0) {
#pragma unroll
                        for (int i = 0; i < 4; ++i) v[i] = act_fn<sizeof(T) == 2>(v[i], ACT);
                    }
                    if (mok[mt] && nok[nt]) {
                        if (EPI == EPI_STORE)
                            El<T>::store4(reinterpret_cast<T *>(g.out) + (int64_t)mrow[mt] * g.ldo + ncol[nt], v);
                        else
                            *reinterpret_cast<float4 *>(reinterpret_cast<float *>(g.out) + (int64_t)mrow[mt] * g.ldo + ncol[nt]) =
                                make_float4(v[0], v[1], v[2], v[3]);
                    }
                }
        }
    }
#ifdef IVR_GEMM_STAMPS
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    IVR_STAMP(3)
#endif
}

template <typename T, int EPI, int ACT, bool SKIP = false>
__global__ __launch_bounds__(512, 2) void gemm_big_kernel(GemmArgs g) {
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    gemm_big_tile<T, EPI, ACT, SKIP>(g, (int)blockIdx.x, smem);
}

// ---------------------------------------------------------------------------------------------
// fp8 GEMM (IVR_COMPUTE_FP8, BASELINE config 5): the 256 x 256 kernel above with e4m3 operands and
// v_mfma_scale_f32_16x16x128_f8f6f4 (unit block scales), i.e. twice the K per MFMA at the same cycles per stage.
//   * a stage is still 128 bytes of K per row = 128 e4m3 elements, the LDS images, swizzle and DMA are unchanged;
//   * an MFMA wants 32 bytes of K per lane.  Lane group g takes chunks g and g+4 of the row (the two reads of the bf16
//     kernel) rather than the contiguous pair 2g, 2g+1, which would be a 2-way bank conflict; A and B use the same
//     permutation of K, so the contraction is unchanged;
//   * one MFMA per (row tile, column tile) per stage needs BOTH halves of both fragments, so the pipeline is split by row
//     tiles: G0 = row tiles 0-3, G1 = row tiles 4-7, both walking the column tiles in the outer loop.  X tiles 4-7 are read
//     under G0, X tiles 0-3 of the next stage under G1, and the next stage's W[nt] right after G1 has issued its last MFMA
//     on W[nt]; the wait + barrier sits between G0 and G1 as in the bf16 kernel.
//   * epilogue: acc * colscale[n] + bias[n] through the row-wide LDS-staged path (bf16, e4m3 or f32 residual output).
// ---------------------------------------------------------------------------------------------
template <int EPI, int ACT, bool OUT8, bool SKIP = false>
__global__ __launch_bounds__(512, 2) void gemm_big8_kernel(GemmArgs g) {
    typedef int v4i __attribute__((ext_vector_type(4)));
    typedef int v8i __attribute__((ext_vector_type(8)));
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int wm = wave >> 2, wn = wave & 3;
    const int MT = (g.M + LBM - 1) / LBM, NT = (g.N + LBN - 1) / LBN;
    int tm, tn;
    {
        const int xcd = blockIdx.x & 7, i = blockIdx.x >> 3;
        const int lx = MT > xcd ? (MT - xcd + 7) >> 3 : 0;
        if (i >= lx * NT) return;
        const int gm = g.group_m;
        const int per = gm * NT, grp = i / per, within = i - grp * per;
        const int gme = min(gm, lx - grp * gm);
        tm = xcd + 8 * (grp * gm + within % gme);
        tn = within / gme;
        if (g.reverse_m) tm = MT - 1 - tm;
    }
    const int m0 = tm * LBM, n0 = tn * LBN;
    const int KT = g.K / ROWB;                           // 128 e4m3 elements per stage
    const unsigned lds0 = (unsigned)(size_t)smem;

    unsigned voffX[4], voffW[4];
#pragma unroll
    for (int j = 0; j < 4; ++j) {
        const int r = 8 * (wave * 4 + j) + (lane >> 3), c = (lane & 7) ^ (lane >> 3);
        voffX[j] = (unsigned)(r * g.lda) + c * 16;
        voffW[j] = (unsigned)(r * g.ldw) + c * 16;
    }
    // The LDS-DMA is issued from inline asm: the compiler then sees no pending LDS write, so the fragment reads can be
    // ordinary LDS loads (two ds_read_b128 land directly in the halves of an 8-register MFMA operand, and lgkmcnt is
    // counted by the compiler) without the s_waitcnt vmcnt(0) it would put in front of them.  Ordering is explicit: every
    // asm below clobbers memory, and the data is only read after the asm vmcnt(0) + barrier that follows its DMA.
    const unsigned long long xb = (unsigned long long)g.A, wb8 = (unsigned long long)g.W;
    const v4i rsX = {(int)(unsigned)xb, (int)((xb >> 32) & 0xffff), (int)((int64_t)g.M * g.lda), 0x00020000};
    const v4i rsW = {(int)(unsigned)wb8, (int)((wb8 >> 32) & 0xffff), (int)((int64_t)g.N * g.ldw), 0x00020000};
    const unsigned sx0 = (unsigned)m0 * (unsigned)g.lda, sw0 = (unsigned)n0 * (unsigned)g.ldw;
    // LDS as in the bf16 kernel: three X slots (DMA two stages ahead) + two W slots.  `on` = false turns a piece into a
    // no-op without a branch (zero-length descriptor: nothing is fetched, zeros are delivered), which keeps a whole K step
    // in one basic block for the scheduler barriers below and the vmcnt counts uniform.  Disabled pieces only occur for
    // stages past the end of K, i.e. they write into slots nobody reads any more; the epilogue waits for them (vmcnt(0))
    // before it reuses the LDS.
    constexpr int WBASE = 3 * LX_BYTES;
    auto piece = [&](int kt, int xs, int j, bool on) {
        const unsigned adv = (unsigned)kt * ROWB;
        v4i rs = j < 4 ? rsX : rsW;
        rs[2] = on ? rs[2] : 0;
        const unsigned dst = j < 4 ? lds0 + xs * LX_BYTES + (wave * 4 + j) * 1024 : lds0 + WBASE + (kt & 1) * LW_BYTES + (wave * 4 + j - 4) * 1024;
        asm volatile("s_mov_b32 m0, %0\n\tbuffer_load_dwordx4 %1, %2, %3 offen lds" ::"s"(dst), "v"(j < 4 ? voffX[j] : voffW[j - 4]), "s"(rs),
                     "s"((j < 4 ? sx0 : sw0) + adv)
                     : "memory", "m0");
    };
    // fragment offsets: row (lane & 15) of a 16-row tile, 16-B chunks g and g+4 of the 128-byte row (swizzled)
    unsigned fo[2];
#pragma unroll
    for (int kk = 0; kk < 2; ++kk) fo[kk] = (lane & 15) * ROWB + ((((kk << 2) + (lane >> 4)) ^ (lane & 7)) << 4);
    const unsigned offX = (wm * 128) * ROWB, offW = WBASE + (wn * 64) * ROWB;
    auto frag = [&](unsigned base, int tile) -> v8i {
        const v4i lo = *reinterpret_cast<const v4i *>(smem + base + fo[0] + tile * 2048);
        const v4i hi = *reinterpret_cast<const v4i *>(smem + base + fo[1] + tile * 2048);
        return v8i{lo[0], lo[1], lo[2], lo[3], hi[0], hi[1], hi[2], hi[3]};
    };
    f32x4 acc[4][8];   // [nt][mt]
#pragma unroll
    for (int a = 0; a < 4; ++a)
#pragma unroll
        for (int b = 0; b < 8; ++b) acc[a][b] = f32x4{0.f, 0.f, 0.f, 0.f};
    v8i xf[8], wf[4];
// four MFMAs of one column tile; the empty asm ties the group to the program order of the (volatile) DMA / barrier asm
#define IVR_MMA8(NT, LO)                                                                                   \
    _Pragma("unroll") for (int mt = LO; mt < LO + 4; ++mt)                                                 \
        acc[NT][mt] = __builtin_amdgcn_mfma_scale_f32_16x16x128_f8f6f4(wf[NT], xf[mt], acc[NT][mt], 0, 0, 0, 0x7f7f7f7f, 0, 0x7f7f7f7f); \
    asm volatile("" : "+v"(acc[NT][LO]), "+v"(acc[NT][LO + 1]), "+v"(acc[NT][LO + 2]), "+v"(acc[NT][LO + 3]));  \
    __builtin_amdgcn_sched_barrier(0);

    // stage 0, X(1), W(1), then the first half of X(2), all before the first wait (ten younger pieces may stay in flight);
    // every later window is W(s+2) x 4, X(s+3) x 4 with the last two X pieces issued early in the following stage, so at every
    // mid-stage barrier the four youngest pieces are X two stages ahead
#pragma unroll
    for (int j = 0; j < 8; ++j) piece(0, 0, j, true);
#pragma unroll
    for (int j = 0; j < 8; ++j) piece(1, 1, j, KT > 1);
    piece(2, 2, 0, KT > 2);
    piece(2, 2, 1, KT > 2);
    asm volatile("s_waitcnt vmcnt(10)\n\ts_barrier" ::: "memory");
#pragma unroll
    for (int i = 0; i < 4; ++i) xf[i] = frag(offX, i);
#pragma unroll
    for (int i = 0; i < 4; ++i) wf[i] = frag(offW, i);
    int xs = 0;                                          // X slot of stage kt = kt % 3
    for (int kt = 0; kt < KT; ++kt) {
        const int xs1 = xs == 2 ? 0 : xs + 1, xs2 = xs1 == 2 ? 0 : xs1 + 1;
        const unsigned xoff = xs * LX_BYTES + offX, nxoff = xs1 * LX_BYTES + offX, nwoff = ((kt + 1) & 1) * LW_BYTES + offW;
        const bool on2 = kt + 2 < KT, on3 = kt + 3 < KT;
        IVR_MMA8(0, 0)
        piece(kt + 2, xs2, 2, on2);
        xf[4] = frag(xoff, 4);
        xf[5] = frag(xoff, 5);
        IVR_MMA8(1, 0)
        piece(kt + 2, xs2, 3, on2);
        xf[6] = frag(xoff, 6);
        xf[7] = frag(xoff, 7);
        IVR_MMA8(2, 0)
        IVR_MMA8(3, 0)
        // every read of this stage has been issued and must have landed before its slots are handed to the DMA; X(kt+1) and
        // W(kt+1) must have landed, the four youngest pieces (X(kt+2)) may stay in flight.
        // (The loads of the next stage's fragments are unconditional: in the last iteration they fetch stale bytes that
        // nothing uses - a conditional 8-register update costs a register copy per lane per tile.)
        asm volatile("s_waitcnt lgkmcnt(0)\n\ts_waitcnt vmcnt(4)\n\ts_barrier" ::: "memory");
#pragma unroll
        for (int i = 0; i < 4; ++i) xf[i] = frag(nxoff, i);
        IVR_MMA8(0, 4)
        piece(kt + 2, 0, 4, on2);
        wf[0] = frag(nwoff, 0);
        piece(kt + 2, 0, 5, on2);
        IVR_MMA8(1, 4)
        piece(kt + 2, 0, 6, on2);
        wf[1] = frag(nwoff, 1);
        piece(kt + 2, 0, 7, on2);
        IVR_MMA8(2, 4)
        piece(kt + 3, xs, 0, on3);
        wf[2] = frag(nwoff, 2);
        IVR_MMA8(3, 4)
        piece(kt + 3, xs, 1, on3);
        wf[3] = frag(nwoff, 3);
        xs = xs1;
    }
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");       // disabled pieces included: nothing may land in LDS after this
#undef IVR_MMA8
    // (contains the barrier after which every wave has read its last fragments)
    wide_epilogue<unsigned short, EPI, ACT, true, OUT8, SKIP>(g, acc, smem + wave * 16384, m0 + wm * 128, n0 + wn * 64, lane, n0 + wn * 64 < g.N);
}

// ---------------------------------------------------------------------------------------------
// attention for short sequences (T = 50 / 77 / 197 / 257, head_dim 64): one workgroup per
// (head, image); K and V of the head staged in LDS, one query row per lane, float32 online softmax.
// The 1/sqrt(64) scale is folded into the Q weights at upload (exact: a power of two).
// ---------------------------------------------------------------------------------------------
template <typename T>
__device__ __forceinline__ void load_row64(const T *p, float (&f)[64]);
template <>
__device__ __forceinline__ void load_row64<unsigned short>(const unsigned short *p, float (&f)[64]) {
#pragma unroll
    for (int c = 0; c < 8; ++c) {
        float t[8];
        El<unsigned short>::unpack(reinterpret_cast<const uint4 *>(p)[c], t);
#pragma unroll
        for (int i = 0; i < 8; ++i) f[c * 8 + i] = t[i];
    }
}
template <>
__device__ __forceinline__ void load_row64<float>(const float *p, float (&f)[64]) {
#pragma unroll
    for (int c = 0; c < 16; ++c) {
        const float4 v = reinterpret_cast<const float4 *>(p)[c];
        f[c * 4 + 0] = v.x;
        f[c * 4 + 1] = v.y;
        f[c * 4 + 2] = v.z;
        f[c * 4 + 3] = v.w;
    }
}

// float32 verification / query mode.  KS lanes share a query row: lane part p takes the keys j = p (mod KS) with its own online-softmax
// state (m, l, o[64]); the KS states of a row sit in adjacent lanes and are merged by log2(KS) xor-shuffle rounds.  One text query is 77
// rows x 8 heads: with one lane per row (KS = 1) a head was a 77-key serial chain on two waves (40 us, a third of the float32 text
// tower's latency); KS = 4 cuts the chain to 20 keys.  K / V rows are padded to 68 floats in LDS: the KS rows a wave reads at once fall
// into distinct banks.
constexpr int AF_STRIDE = 68;
template <int KS>
__global__ __launch_bounds__(512) void attention_f32_kernel(const float *__restrict__ qkv, float *__restrict__ att, int Tn, int D, int causal) {
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    float *Ks = reinterpret_cast<float *>(smem);
    float *Vs = Ks + (size_t)Tn * AF_STRIDE;
    const int h = blockIdx.x, img = blockIdx.y;
    const int64_t base = (int64_t)img * Tn * 3 * D + h * 64;
    for (int i = threadIdx.x; i < Tn * 16; i += blockDim.x) {
        const int t = i >> 4, c = i & 15;
        const float4 *kp = reinterpret_cast<const float4 *>(qkv + base + (int64_t)t * 3 * D + D) + c;
        const float4 *vp = reinterpret_cast<const float4 *>(qkv + base + (int64_t)t * 3 * D + 2 * D) + c;
        reinterpret_cast<float4 *>(Ks + t * AF_STRIDE)[c] = *kp;
        reinterpret_cast<float4 *>(Vs + t * AF_STRIDE)[c] = *vp;
    }
    __syncthreads();
    const int t = threadIdx.x / KS, part = threadIdx.x % KS;
    const bool active = t < Tn;
    float q[64], o[64];
    if (active) load_row64<float>(qkv + base + (int64_t)t * 3 * D, q);
#pragma unroll
    for (int i = 0; i < 64; ++i) {
        o[i] = 0.f;
        if (!active) q[i] = 0.f;
    }
    float m = -INFINITY, l = 0.f;
    // wave-uniform key bound: all keys, or (last query row of this wave) + 1 under the causal mask
    int jmax = Tn;
    if (causal) jmax = min(Tn, (int)((threadIdx.x | 63) / KS + 1));
    for (int j0 = 0; j0 < jmax; j0 += KS) {
        const int j = j0 + part;
        const bool jin = j < jmax;
        float kv[64];
        load_row64<float>(Ks + (size_t)min(j, Tn - 1) * AF_STRIDE, kv);
        float s0 = 0.f, s1 = 0.f, s2 = 0.f, s3 = 0.f;
#pragma unroll
        for (int i = 0; i < 64; i += 4) {
            s0 = fmaf(q[i], kv[i], s0);
            s1 = fmaf(q[i + 1], kv[i + 1], s1);
            s2 = fmaf(q[i + 2], kv[i + 2], s2);
            s3 = fmaf(q[i + 3], kv[i + 3], s3);
        }
        float s = (s0 + s1) + (s2 + s3);
        if (!jin || (causal && j > t)) s = -INFINITY;
        if (__any(s > m)) {
            const float mn = fmaxf(m, s);
            const float alpha = (m == -INFINITY) ? 0.f : __expf(m - mn);
            m = mn;
            l *= alpha;
#pragma unroll
            for (int i = 0; i < 64; ++i) o[i] *= alpha;
        }
        const float p = (s == -INFINITY) ? 0.f : __expf(s - m);
        l += p;
        load_row64<float>(Vs + (size_t)min(j, Tn - 1) * AF_STRIDE, kv);
#pragma unroll
        for (int i = 0; i < 64; ++i) o[i] = fmaf(p, kv[i], o[i]);
    }
    // merge the KS partial states of a row (adjacent lanes)
#pragma unroll
    for (int off = 1; off < KS; off <<= 1) {
        const float m2 = __shfl_xor(m, off, 64), l2 = __shfl_xor(l, off, 64);
        const float mn = fmaxf(m, m2);
        const float a = (m == -INFINITY) ? 0.f : __expf(m - mn), b2 = (m2 == -INFINITY) ? 0.f : __expf(m2 - mn);
        l = l * a + l2 * b2;
#pragma unroll
        for (int i = 0; i < 64; ++i) o[i] = o[i] * a + __shfl_xor(o[i], off, 64) * b2;
        m = mn;
    }
    if (active && part == 0) {
        const float inv = 1.0f / l;
        float *op = att + ((int64_t)img * Tn + t) * D + h * 64;
#pragma unroll
        for (int i = 0; i < 64; i += 4) *reinterpret_cast<float4 *>(op + i) = make_float4(o[i] * inv, o[i + 1] * inv, o[i + 2] * inv, o[i + 3] * inv);
    }
}

// ---------------------------------------------------------------------------------------------
// bf16 attention on MFMA (production path): one wave per (image, head, 64-query block), flash-style loop
// over 64-key blocks.  Both products are issued transposed so that the query index stays on the lane:
//   S^T[key][query] = K Q^T      A = K rows, B = Q rows          -> lane holds 16 keys of ONE query per query tile
//   O^T[dh][query]  = V^T P^T    A = V^T rows (LDS), B = P^T     -> the softmax'd accumulator IS the B operand
// The k order of the second product is permuted (element j of lane group g = key 16*(2ks + (j>>2)) + 4g + (j&3)) so
// that P goes from accumulator registers to operand registers with a bf16 pack and no cross-lane traffic; V^T is
// staged per wave in LDS (rows padded to 136 B: conflict-free ds_read_b64) with the same permutation on the read.
// Row max / sum across the 4 lanes of a query are two wave shuffles; the online-softmax rescale is lane-local.
// ---------------------------------------------------------------------------------------------
constexpr int VT_STRIDE = 136;                    // bytes per V^T row (64 keys * 2 B + 8 B pad)
constexpr int VT_BYTES = 64 * VT_STRIDE;          // per wave

__global__ __launch_bounds__(256) void attention_mfma_kernel(const unsigned short *__restrict__ qkv, unsigned short *__restrict__ att,
                                                             int n, int Tn, int D, int heads, int causal) {
    typedef __bf16 bf16x8_t __attribute__((ext_vector_type(8)));
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int nqb = (Tn + 63) >> 6;
    const int64_t item = (int64_t)blockIdx.x * 4 + wave;
    if (item >= (int64_t)n * heads * nqb) return;          // no workgroup barrier below: waves are independent
    const int qb = (int)(item % nqb);
    const int h = (int)((item / nqb) % heads);
    const int img = (int)(item / ((int64_t)nqb * heads));
    unsigned char *vt = smem + wave * VT_BYTES;
    const int g = lane >> 4, c = lane & 15;
    const int64_t rs = 3 * (int64_t)D;                     // qkv row stride (elements)
    const unsigned short *base = qkv + (int64_t)img * Tn * rs + h * 64;

    // Q fragments (B operand of S^T): query = qb*64 + 16*qt + c, dh = 32*ks + 8*g .. +7
    uint4 qf[2][4];
#pragma unroll
    for (int qt = 0; qt < 4; ++qt) {
        const int q = min(qb * 64 + qt * 16 + c, Tn - 1);
#pragma unroll
        for (int ks = 0; ks < 2; ++ks) qf[ks][qt] = *reinterpret_cast<const uint4 *>(base + q * rs + ks * 32 + g * 8);
    }
    f32x4 o[4][4];        // [dh tile][query tile]: O^T[16*nt + 4g + r][16*qt + c]
    float m[4], l[4];
#pragma unroll
    for (int a = 0; a < 4; ++a) {
        m[a] = -INFINITY;
        l[a] = 0.f;
#pragma unroll
        for (int b = 0; b < 4; ++b) o[a][b] = f32x4{0.f, 0.f, 0.f, 0.f};
    }
    const int kend = causal ? min(Tn, qb * 64 + 64) : Tn;
    for (int k0 = 0; k0 < kend; k0 += 64) {
        // ---- S^T block = K Q^T
        f32x4 s[4][4];    // [key tile][query tile]: key = k0 + 16*kt + 4g + r, query = 16*qt + c
#pragma unroll
        for (int kt = 0; kt < 4; ++kt) {
            const int key = min(k0 + kt * 16 + c, Tn - 1);
            uint4 kf[2];
#pragma unroll
            for (int ks = 0; ks < 2; ++ks) kf[ks] = *reinterpret_cast<const uint4 *>(base + D + key * rs + ks * 32 + g * 8);
#pragma unroll
            for (int qt = 0; qt < 4; ++qt) {
                f32x4 acc = {0.f, 0.f, 0.f, 0.f};
                acc = __builtin_amdgcn_mfma_f32_16x16x32_bf16(__builtin_bit_cast(bf16x8_t, kf[0]), __builtin_bit_cast(bf16x8_t, qf[0][qt]), acc, 0, 0, 0);
                acc = __builtin_amdgcn_mfma_f32_16x16x32_bf16(__builtin_bit_cast(bf16x8_t, kf[1]), __builtin_bit_cast(bf16x8_t, qf[1][qt]), acc, 0, 0, 0);
                s[kt][qt] = acc;
            }
        }
        // ---- V^T block -> LDS: lane owns key k0 + lane, scatters its 64 dh values down a column
        {
            const int key = min(k0 + lane, Tn - 1);
            const uint4 *vp = reinterpret_cast<const uint4 *>(base + 2 * D + key * rs);
#pragma unroll
            for (int ch = 0; ch < 8; ++ch) {
                const uint4 v = vp[ch];
                const uint32_t w[4] = {v.x, v.y, v.z, v.w};
#pragma unroll
                for (int i = 0; i < 4; ++i) {
                    *reinterpret_cast<unsigned short *>(vt + (ch * 8 + 2 * i) * VT_STRIDE + lane * 2) = (unsigned short)(w[i] & 0xffffu);
                    *reinterpret_cast<unsigned short *>(vt + (ch * 8 + 2 * i + 1) * VT_STRIDE + lane * 2) = (unsigned short)(w[i] >> 16);
                }
            }
        }
        // ---- masks + online softmax (per query column: 16 values in this lane, 4 lanes per query)
        uint4 pf[2][4];   // P^T fragments (B operand of O^T), k-step ks covers key tiles 2ks, 2ks+1
#pragma unroll
        for (int qt = 0; qt < 4; ++qt) {
            const int q = qb * 64 + qt * 16 + c;
            float mx = -INFINITY;
#pragma unroll
            for (int kt = 0; kt < 4; ++kt)
#pragma unroll
                for (int r = 0; r < 4; ++r) {
                    const int key = k0 + kt * 16 + g * 4 + r;
                    if (key >= Tn || (causal && key > q)) s[kt][qt][r] = -INFINITY;
                    mx = fmaxf(mx, s[kt][qt][r]);
                }
            mx = fmaxf(mx, __shfl_xor(mx, 16, 64));
            mx = fmaxf(mx, __shfl_xor(mx, 32, 64));
            const float mn = fmaxf(m[qt], mx);
            // mn stays -inf only for a query that has seen no valid key yet (cannot happen: key 0 is always visible)
            const float alpha = (m[qt] == -INFINITY) ? 0.f : __expf(m[qt] - mn);
            m[qt] = mn;
            float sum = 0.f;
#pragma unroll
            for (int kt = 0; kt < 4; ++kt)
#pragma unroll
                for (int r = 0; r < 4; ++r) {
                    const float p = (s[kt][qt][r] == -INFINITY) ? 0.f : __expf(s[kt][qt][r] - mn);
                    s[kt][qt][r] = p;
                    sum += p;
                }
            sum += __shfl_xor(sum, 16, 64);
            sum += __shfl_xor(sum, 32, 64);
            l[qt] = l[qt] * alpha + sum;
#pragma unroll
            for (int nt = 0; nt < 4; ++nt) {
                o[nt][qt][0] *= alpha;
                o[nt][qt][1] *= alpha;
                o[nt][qt][2] *= alpha;
                o[nt][qt][3] *= alpha;
            }
#pragma unroll
            for (int ks = 0; ks < 2; ++ks) {
                uint4 f;
                f.x = ivr_pack_bf16x2(s[2 * ks][qt][0], s[2 * ks][qt][1]);
                f.y = ivr_pack_bf16x2(s[2 * ks][qt][2], s[2 * ks][qt][3]);
                f.z = ivr_pack_bf16x2(s[2 * ks + 1][qt][0], s[2 * ks + 1][qt][1]);
                f.w = ivr_pack_bf16x2(s[2 * ks + 1][qt][2], s[2 * ks + 1][qt][3]);
                pf[ks][qt] = f;
            }
        }
        // ---- O^T += V^T P^T   (the wave's own LDS writes above are complete before these reads: same wave, lgkmcnt)
#pragma unroll
        for (int nt = 0; nt < 4; ++nt) {
#pragma unroll
            for (int ks = 0; ks < 2; ++ks) {
                const unsigned char *row = vt + (nt * 16 + c) * VT_STRIDE + ks * 64 + g * 8;
                const uint2 lo = *reinterpret_cast<const uint2 *>(row), hi = *reinterpret_cast<const uint2 *>(row + 32);
                const uint4 vf = make_uint4(lo.x, lo.y, hi.x, hi.y);
#pragma unroll
                for (int qt = 0; qt < 4; ++qt)
                    o[nt][qt] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(__builtin_bit_cast(bf16x8_t, vf), __builtin_bit_cast(bf16x8_t, pf[ks][qt]),
                                                                        o[nt][qt], 0, 0, 0);
            }
        }
    }
    // ---- store: lane holds O[query 16*qt + c][dh 16*nt + 4g .. +3]
#pragma unroll
    for (int qt = 0; qt < 4; ++qt) {
        const int q = qb * 64 + qt * 16 + c;
        if (q >= Tn) continue;
        const float inv = 1.0f / l[qt];
        unsigned short *op = att + ((int64_t)img * Tn + q) * D + h * 64 + g * 4;
#pragma unroll
        for (int nt = 0; nt < 4; ++nt) {
            const float v[4] = {o[nt][qt][0] * inv, o[nt][qt][1] * inv, o[nt][qt][2] * inv, o[nt][qt][3] * inv};
            El<unsigned short>::store4(op + nt * 16, v);
        }
    }
}

// ---------------------------------------------------------------------------------------------
// bf16 attention, head-resident version (64 < T <= 640: ViT-L/14 T = 257, DINO T = 197, CLIP text T = 77): one workgroup per
// (image, head[, query split]); the head's whole K and V (T x 64 bf16 each) are staged ONCE into LDS by LDS-DMA, then every
// wave runs a flash loop over 32-key blocks for its own QC query tiles of 16.  Products are transposed as in the kernels
// above (S^T = K Q^T, O^T = V^T P^T) so a query stays on one lane column.
//   * K image: 128-B rows, 16-B chunk ch at position ch ^ (row & 7)  (conflict-free ds_read_b128 of the A fragments).
//   * V image: row-major [key][64 dh]; the V^T fragments come out of it with ds_read_b64_tr_b16 (hardware transpose: the
//     16 lanes of a group read a 4-key x 16-dh block, lane i receives column i).  32-B segment s of a row sits at position
//     s ^ ((row >> 1) & 3): the 8 rows a 32-lane half touches land on distinct banks.
//     Both swizzles are applied on the DMA *source* address (the LDS side of an LDS-DMA is lane-linear).
//   * rows past T read as zeros (buffer descriptor bounds), so no garbage ever enters an MFMA.
//   * the key order inside an O^T MFMA is permuted (element j of lane group g = key 16*(j>>2) + 4g + (j&3)) so that P goes
//     from the S^T accumulators to the B operand with a bf16 pack and no cross-lane traffic; the tr reads follow the same order.
//   * softmax in base 2 (one fma + v_exp per score); the running max is shared by the 4 lanes of a query through
//     v_permlane32_swap / v_permlane16_swap; row sums stay lane-local until the end; the O rescale is skipped (exactly: the
//     factor would be 1) in blocks where no query of the wave raised its maximum; masks only in the tail / diagonal blocks.
// ---------------------------------------------------------------------------------------------
__device__ __forceinline__ float vmax2(float a, float b) {                // plain v_max_f32 (fmaxf adds two canonicalising moves)
    float r;
    asm("v_max_f32 %0, %1, %2" : "=v"(r) : "v"(a), "v"(b));
    return r;
}
__device__ __forceinline__ float quad_max(float x) {      // max over lanes c, c+16, c+32, c+48
    auto a = __builtin_amdgcn_permlane32_swap(__float_as_uint(x), __float_as_uint(x), false, false);
    const float y = vmax2(__uint_as_float(a[0]), __uint_as_float(a[1]));
    auto b = __builtin_amdgcn_permlane16_swap(__float_as_uint(y), __float_as_uint(y), false, false);
    return vmax2(__uint_as_float(b[0]), __uint_as_float(b[1]));
}
__device__ __forceinline__ float vmax3(float a, float b, float c) {      // no IEEE canonicalisation moves around it
    float r;
    asm("v_max3_f32 %0, %1, %2, %3" : "=v"(r) : "v"(a), "v"(b), "v"(c));
    return r;
}
__device__ __forceinline__ float quad_sum(float x) {
    auto a = __builtin_amdgcn_permlane32_swap(__float_as_uint(x), __float_as_uint(x), false, false);
    const float y = __uint_as_float(a[0]) + __uint_as_float(a[1]);
    auto b = __builtin_amdgcn_permlane16_swap(__float_as_uint(y), __float_as_uint(y), false, false);
    return __uint_as_float(b[0]) + __uint_as_float(b[1]);
}

template <int QC, typename TOut, int MINW = 3>
__global__ __launch_bounds__(384, MINW) void attention_head_kernel(const unsigned short *__restrict__ qkv, TOut *__restrict__ att,
                                                                int Tn, int D, int heads, int causal, int nsplit, int Tp) {
    typedef __bf16 bf16x8_t __attribute__((ext_vector_type(8)));
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    const int lane = threadIdx.x & 63;
    const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6), nw = blockDim.x >> 6;
    int b = blockIdx.x;
    const int split = b % nsplit;
    b /= nsplit;
    const int h = b % heads, img = b / heads;
    const int c = lane & 15, g = lane >> 4;
    unsigned char *Ks = smem, *Vs = smem + Tp * 128;
    const int64_t rs = 3 * (int64_t)D;                     // qkv row stride (elements)
    const unsigned short *ibase = qkv + (int64_t)img * Tn * rs;
    const unsigned rowB = (unsigned)rs * 2;

    // ---- stage K and V of the head: pieces of 8 rows x 128 B, lane l -> row l >> 3, LDS slot l & 7
    {
        const auto rsrc = __builtin_amdgcn_make_buffer_rsrc(const_cast<unsigned short *>(ibase), 0, (int)((int64_t)Tn * rowB), 0x00020000);
        const int r = lane >> 3, pos = lane & 7;
        const int chK = pos ^ r;
        const int chV = ((((pos >> 1) ^ ((r >> 1) & 3)) << 1) | (pos & 1));
        const unsigned voffK = r * rowB + (unsigned)(D + h * 64) * 2 + chK * 16;
        const unsigned voffV = r * rowB + (unsigned)(2 * D + h * 64) * 2 + chV * 16;
        const int npieces = Tp >> 3;
        for (int p = wave; p < npieces; p += nw) {
            const unsigned soff = (unsigned)p * 8 * rowB;
            __builtin_amdgcn_raw_ptr_buffer_load_lds(rsrc, (__attribute__((address_space(3))) void *)(Ks + p * 1024), 16, voffK, soff, 0, 0);
            __builtin_amdgcn_raw_ptr_buffer_load_lds(rsrc, (__attribute__((address_space(3))) void *)(Vs + p * 1024), 16, voffV, soff, 0, 0);
        }
    }
    // ---- this wave's query tiles
    const int ntiles = (Tn + 15) >> 4, tps = (ntiles + nsplit - 1) / nsplit;
    const int tend = min(ntiles, (split + 1) * tps), t0 = split * tps + wave * QC;
    uint4 qf[2][QC];      // B operand of S^T: query = 16*(t0+qt) + c, dh = 32*ks + 8*g .. +7
#pragma unroll
    for (int qt = 0; qt < QC; ++qt) {
        const int q = min(16 * (t0 + qt) + c, Tn - 1);
#pragma unroll
        for (int ks = 0; ks < 2; ++ks) qf[ks][qt] = *reinterpret_cast<const uint4 *>(ibase + q * rs + h * 64 + ks * 32 + g * 8);
    }
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __syncthreads();
    if (t0 >= tend) return;                                // (no barrier below)
    const int nq = min(QC, tend - t0);

    f32x4 o[4][QC];       // O^T[16*nt + 4g + r][query c of tile qt]
    float m[QC], l[QC];
#pragma unroll
    for (int qt = 0; qt < QC; ++qt) {
        m[qt] = -INFINITY;
        l[qt] = 0.f;
#pragma unroll
        for (int nt = 0; nt < 4; ++nt) o[nt][qt] = f32x4{0.f, 0.f, 0.f, 0.f};
    }
    const unsigned lds0 = (unsigned)(size_t)smem;
    // per-lane LDS offsets: K row c, chunk 4ks+g;  V row 4g + (c>>2) (+16 for the second half), segment nt, 8-B piece c&3
    unsigned offK[2], offV[4];
#pragma unroll
    for (int ks = 0; ks < 2; ++ks) offK[ks] = lds0 + c * 128 + (((4 * ks + g) ^ (c & 7)) << 4);
    {
        const int vr = 4 * g + (c >> 2), xr = (2 * g + (c >> 3)) & 3;
#pragma unroll
        for (int nt = 0; nt < 4; ++nt) offV[nt] = lds0 + Tp * 128 + vr * 128 + ((nt ^ xr) << 5) + (c & 3) * 8;
    }
    constexpr float L2E = 1.4426950408889634f;
    const int kend = causal ? min(Tn, 16 * (t0 + nq)) : Tn;
    for (int k0 = 0; k0 < kend; k0 += 32) {
        const unsigned kb = (unsigned)k0 * 128;
        // ---- fragments of the block: 4 K reads, 8 transposed V reads
        uint4 kf[2][2];
        uint2 vlo[4], vhi[4];
        asm volatile("ds_read_b128 %0, %1" : "=v"(kf[0][0]) : "v"(offK[0] + kb));
        asm volatile("ds_read_b128 %0, %1" : "=v"(kf[0][1]) : "v"(offK[1] + kb));
        asm volatile("ds_read_b128 %0, %1 offset:2048" : "=v"(kf[1][0]) : "v"(offK[0] + kb));
        asm volatile("ds_read_b128 %0, %1 offset:2048" : "=v"(kf[1][1]) : "v"(offK[1] + kb));
#pragma unroll
        for (int nt = 0; nt < 4; ++nt) {
            asm volatile("ds_read_b64_tr_b16 %0, %1" : "=v"(vlo[nt]) : "v"(offV[nt] + kb));
            asm volatile("ds_read_b64_tr_b16 %0, %1 offset:2048" : "=v"(vhi[nt]) : "v"(offV[nt] + kb));
        }
        asm volatile("s_waitcnt lgkmcnt(8)" ::: "memory");
        __builtin_amdgcn_sched_barrier(0);
        // ---- S^T block = K Q^T : s[kt][qt], key = k0 + 16*kt + 4g + r, query = 16*qt + c
        f32x4 s[2][QC];
#pragma unroll
        for (int kt = 0; kt < 2; ++kt)
#pragma unroll
            for (int qt = 0; qt < QC; ++qt) {
                f32x4 acc = {0.f, 0.f, 0.f, 0.f};
                acc = __builtin_amdgcn_mfma_f32_16x16x32_bf16(__builtin_bit_cast(bf16x8_t, kf[kt][0]), __builtin_bit_cast(bf16x8_t, qf[0][qt]), acc, 0, 0, 0);
                acc = __builtin_amdgcn_mfma_f32_16x16x32_bf16(__builtin_bit_cast(bf16x8_t, kf[kt][1]), __builtin_bit_cast(bf16x8_t, qf[1][qt]), acc, 0, 0, 0);
                s[kt][qt] = acc;
            }
        if (k0 + 32 > Tn || (causal && k0 + 31 > 16 * t0)) {
#pragma unroll
            for (int qt = 0; qt < QC; ++qt) {
                const int q = 16 * (t0 + qt) + c;
#pragma unroll
                for (int kt = 0; kt < 2; ++kt)
#pragma unroll
                    for (int r = 0; r < 4; ++r) {
                        const int key = k0 + kt * 16 + g * 4 + r;
                        if (key >= Tn || (causal && key > q)) s[kt][qt][r] = -INFINITY;
                    }
            }
        }
        // ---- online softmax, base 2
        float mn[QC];
        bool grew = false;
#pragma unroll
        for (int qt = 0; qt < QC; ++qt) {
            float mx = vmax3(s[0][qt][0], s[0][qt][1], s[0][qt][2]);
            mx = vmax3(mx, s[0][qt][3], s[1][qt][0]);
            mx = vmax3(mx, s[1][qt][1], s[1][qt][2]);
            mx = quad_max(vmax2(mx, s[1][qt][3]));
            mn[qt] = vmax2(m[qt], mx);
            grew |= mn[qt] > m[qt];
        }
        if (__any(grew)) {
#pragma unroll
            for (int qt = 0; qt < QC; ++qt) {
                const float alpha = __builtin_amdgcn_exp2f((m[qt] - mn[qt]) * L2E);      // m = -inf (first block) -> 0
                m[qt] = mn[qt];
                l[qt] *= alpha;
#pragma unroll
                for (int nt = 0; nt < 4; ++nt) {
                    o[nt][qt][0] *= alpha;
                    o[nt][qt][1] *= alpha;
                    o[nt][qt][2] *= alpha;
                    o[nt][qt][3] *= alpha;
                }
            }
        }
        uint4 pf[QC];     // P^T fragments (B operand of O^T)
#pragma unroll
        for (int qt = 0; qt < QC; ++qt) {
            const float mb = -m[qt] * L2E;
            float pv[2][4], sum = 0.f;
#pragma unroll
            for (int kt = 0; kt < 2; ++kt)
#pragma unroll
                for (int r = 0; r < 4; ++r) {
                    pv[kt][r] = __builtin_amdgcn_exp2f(fmaf(s[kt][qt][r], L2E, mb));       // v_exp_f32; masked (-inf) -> 0
                    sum += pv[kt][r];
                }
            l[qt] += sum;
            pf[qt].x = ivr_pack_bf16x2(pv[0][0], pv[0][1]);
            pf[qt].y = ivr_pack_bf16x2(pv[0][2], pv[0][3]);
            pf[qt].z = ivr_pack_bf16x2(pv[1][0], pv[1][1]);
            pf[qt].w = ivr_pack_bf16x2(pv[1][2], pv[1][3]);
        }
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
        __builtin_amdgcn_sched_barrier(0);
        // ---- O^T += V^T P^T
#pragma unroll
        for (int nt = 0; nt < 4; ++nt) {
            const uint4 vf = make_uint4(vlo[nt].x, vlo[nt].y, vhi[nt].x, vhi[nt].y);
#pragma unroll
            for (int qt = 0; qt < QC; ++qt)
                o[nt][qt] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(__builtin_bit_cast(bf16x8_t, vf), __builtin_bit_cast(bf16x8_t, pf[qt]), o[nt][qt], 0, 0, 0);
        }
    }
    // ---- store: lane holds O[query 16*qt + c][dh 16*nt + 4g .. +3]
#pragma unroll
    for (int qt = 0; qt < QC; ++qt) {
        const float inv = __builtin_amdgcn_rcpf(quad_sum(l[qt]));
        const int q = 16 * (t0 + qt) + c;
        if (qt < nq && q < Tn) {
            TOut *op = att + ((int64_t)img * Tn + q) * D + h * 64 + g * 4;
#pragma unroll
            for (int nt = 0; nt < 4; ++nt) {
                const float v[4] = {o[nt][qt][0] * inv, o[nt][qt][1] * inv, o[nt][qt][2] * inv, o[nt][qt][3] * inv};
                El<TOut>::store4(op + nt * 16, v);
            }
        }
    }
}

// kernel choice, overridable for A/B runs and so that the parity tests can drive every kernel with every shape:
// IVR_GEMM=0 the 128 x 128 kernel, 4 the 256 x 256 kernel; default by problem size.  Read on every launch (a getenv is
// ~100 ns next to a multi-microsecond launch) so a test can flip it inside one process.
int env_int(const char *name, int dflt) {
    const char *e = getenv(name);
    return e && *e ? atoi(e) : dflt;
}
int gemm_mode() {
    const int v = env_int("IVR_GEMM", -1);
    return v == 0 || v == 4 ? v : -1;
}

// Short-sequence variant (T <= 64: CLIP ViT-B/32 has T = 50): one key block, so no online-softmax state has to survive a
// loop.  The K fragments of the head stay in registers (32) and the four 16-query tiles are processed one after the other
// (S^T tile -> softmax -> P^T -> O^T tile -> store), which keeps the kernel at <= 128 VGPRs = 4 waves per SIMD: the kernel
// is latency-bound (each wave touches 19 KB once), occupancy is what hides it.
template <typename TOut>
__global__ __launch_bounds__(256, 5) void attention_mfma_short_kernel(const unsigned short *__restrict__ qkv,
                                                                      TOut *__restrict__ att, int n, int Tn, int D, int heads,
                                                                      int causal) {
    typedef __bf16 bf16x8_t __attribute__((ext_vector_type(8)));
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int64_t item = (int64_t)blockIdx.x * 4 + wave;
    if (item >= (int64_t)n * heads) return;                 // waves are independent: no workgroup barrier below
    const int h = (int)(item % heads), img = (int)(item / heads);
    unsigned char *vt = smem + wave * 8192;                 // 64 keys x 128 B of V per wave
    const int g = lane >> 4, c = lane & 15;
    const int64_t rs = 3 * (int64_t)D;
    const unsigned short *base = qkv + (int64_t)img * Tn * rs + h * 64;

    // K and V rows of the head: eight coalesced loads each (one instruction = 8 key rows x 128 B; lane -> row 8 i + (lane >> 3),
    // 16-B chunk lane & 7), all sixteen in flight together.  K goes through the wave's LDS region first (chunk ch of a row at
    // position ch ^ (row & 7): conflict-free ds_read_b128 of the A fragments), then V takes the region over.
    uint4 kf[2][4];
    {
        const int ch = lane & 7;
        uint4 kv[8], vv[8];
#pragma unroll
        for (int i = 0; i < 8; ++i) {
            const int key = min(8 * i + (lane >> 3), Tn - 1);
            kv[i] = *reinterpret_cast<const uint4 *>(base + D + key * rs + ch * 8);
        }
#pragma unroll
        for (int i = 0; i < 8; ++i) {
            const int key = min(8 * i + (lane >> 3), Tn - 1);
            vv[i] = *reinterpret_cast<const uint4 *>(base + 2 * D + key * rs + ch * 8);
        }
#pragma unroll
        for (int i = 0; i < 8; ++i) {
            const int row = 8 * i + (lane >> 3);
            *reinterpret_cast<uint4 *>(vt + row * 128 + ((ch ^ (row & 7)) << 4)) = kv[i];
        }
        // K fragments (A operand of S^T): key = 16*kt + c, dh = 32*ks + 8*g .. +7
#pragma unroll
        for (int kt = 0; kt < 4; ++kt)
#pragma unroll
            for (int ks = 0; ks < 2; ++ks)
                kf[ks][kt] = *reinterpret_cast<const uint4 *>(vt + (kt * 16 + c) * 128 + (((4 * ks + g) ^ (c & 7)) << 4));
        // V -> LDS row-major; the V^T fragments are read back with the transposing ds_read_b64_tr_b16.  32-B segment s of a row
        // sits at position s ^ ((row >> 1) & 3) (conflict-free transposed reads).  Same wave, LDS in order: the K reads above
        // have returned their data before these stores are issued (the compiler waits for kf before its first use or keeps the
        // order; the explicit wait makes the dependence unconditional).
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
#pragma unroll
        for (int i = 0; i < 8; ++i) {
            const int row = 8 * i + (lane >> 3), xr = (row >> 1) & 3;
            *reinterpret_cast<uint4 *>(vt + row * 128 + ((((ch >> 1) ^ xr) << 5) | ((ch & 1) << 4))) = vv[i];
        }
    }
    unsigned offV[4];
    {
        const int vr = 4 * g + (c >> 2), xr = (2 * g + (c >> 3)) & 3;
#pragma unroll
        for (int nt = 0; nt < 4; ++nt) offV[nt] = (unsigned)(size_t)vt + vr * 128 + ((nt ^ xr) << 5) + (c & 3) * 8;
    }
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");       // the wave's own V rows are in LDS before any transposed read
    constexpr float L2E = 1.4426950408889634f;
    const int nqt = (Tn + 15) >> 4;
#pragma unroll 1
    for (int qt = 0; qt < nqt; ++qt) {
        const int q = qt * 16 + c;
        const int qc = min(q, Tn - 1);
        uint4 qf[2];
#pragma unroll
        for (int ks = 0; ks < 2; ++ks) qf[ks] = *reinterpret_cast<const uint4 *>(base + qc * rs + ks * 32 + g * 8);
        f32x4 s[4];
        float mx = -INFINITY;
#pragma unroll
        for (int kt = 0; kt < 4; ++kt) {
            f32x4 acc = {0.f, 0.f, 0.f, 0.f};
            acc = __builtin_amdgcn_mfma_f32_16x16x32_bf16(__builtin_bit_cast(bf16x8_t, kf[0][kt]), __builtin_bit_cast(bf16x8_t, qf[0]), acc, 0, 0, 0);
            acc = __builtin_amdgcn_mfma_f32_16x16x32_bf16(__builtin_bit_cast(bf16x8_t, kf[1][kt]), __builtin_bit_cast(bf16x8_t, qf[1]), acc, 0, 0, 0);
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                const int key = kt * 16 + g * 4 + r;
                if (key >= Tn || (causal && key > q)) acc[r] = -INFINITY;
                mx = fmaxf(mx, acc[r]);
            }
            s[kt] = acc;
        }
        mx = quad_max(mx);                    // finite: key 0 is visible to every query
        const float mb = -mx * L2E;
        float sum = 0.f;
#pragma unroll
        for (int kt = 0; kt < 4; ++kt)
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                const float p = __builtin_amdgcn_exp2f(fmaf(s[kt][r], L2E, mb));      // masked (-inf) -> 0
                s[kt][r] = p;
                sum += p;
            }
        sum = quad_sum(sum);
        uint4 pf[2];
#pragma unroll
        for (int ks = 0; ks < 2; ++ks) {
            pf[ks].x = ivr_pack_bf16x2(s[2 * ks][0], s[2 * ks][1]);
            pf[ks].y = ivr_pack_bf16x2(s[2 * ks][2], s[2 * ks][3]);
            pf[ks].z = ivr_pack_bf16x2(s[2 * ks + 1][0], s[2 * ks + 1][1]);
            pf[ks].w = ivr_pack_bf16x2(s[2 * ks + 1][2], s[2 * ks + 1][3]);
        }
        const float inv = __builtin_amdgcn_rcpf(sum);
        TOut *op = att + ((int64_t)img * Tn + qc) * D + h * 64 + g * 4;
#pragma unroll
        for (int nt = 0; nt < 4; ++nt) {
            f32x4 o = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
            for (int ks = 0; ks < 2; ++ks) {
                uint2 lo, hi;
                if (ks == 0)
                    asm volatile("ds_read_b64_tr_b16 %0, %2\n\tds_read_b64_tr_b16 %1, %2 offset:2048\n\ts_waitcnt lgkmcnt(0)"
                                 : "=&v"(lo), "=&v"(hi)
                                 : "v"(offV[nt]));
                else
                    asm volatile("ds_read_b64_tr_b16 %0, %2 offset:4096\n\tds_read_b64_tr_b16 %1, %2 offset:6144\n\ts_waitcnt lgkmcnt(0)"
                                 : "=&v"(lo), "=&v"(hi)
                                 : "v"(offV[nt]));
                const uint4 vf = make_uint4(lo.x, lo.y, hi.x, hi.y);
                o = __builtin_amdgcn_mfma_f32_16x16x32_bf16(__builtin_bit_cast(bf16x8_t, vf), __builtin_bit_cast(bf16x8_t, pf[ks]), o, 0, 0, 0);
            }
            if (q < Tn) {
                const float v[4] = {o[0] * inv, o[1] * inv, o[2] * inv, o[3] * inv};
                El<TOut>::store4(op + nt * 16, v);
            }
        }
    }
}

// ---------------------------------------------------------------------------------------------
// Fused QKV projection + attention for short sequences (T <= 64: CLIP ViT-B/32 has T = 50), bf16.
// Replaces gemm_big_kernel<bf16, EPI_STORE> (qkv) + attention_mfma_short_kernel: the [rows, 3D] QKV activations never go to
// HBM (4608 B written + 4608 B read per token row and layer at D = 768; the attention kernel was a pure HBM round trip).
//   * one workgroup = (G whole images, one head): an M tile of G*T rows (G = 256 / T: 250 rows at T = 50) times the 192 output
//     columns q_h | k_h | v_h of head h.  The K loop is gemm_big_kernel's (three X slots, two W slots, LDS-DMA, fragment sets
//     software-pipelined across the mid-stage barrier) with three column tiles per wave instead of four;
//   * after the K loop the accumulators (+ bias) go to LDS as bf16 - the same rounding the QKV buffer had - in the layouts the
//     attention products read without bank conflicts: Q and K rows with the GEMM's chunk ^ (row & 7) swizzle (ds_read_b128
//     fragments), V rows with the 32-byte segment swizzle of the transposing ds_read_b64_tr_b16;
//   * the (image, 16-query tile) units of the tile are dealt to the eight waves; each unit is the arithmetic of
//     attention_mfma_short_kernel, operation for operation (S^T = K Q^T, base-2 softmax, P packed to bf16, O^T = V^T P^T), so
//     the result is bit-identical to the unfused path; only att[rows, 64] is written.
// Tile order as in the GEMMs: the 12 head workgroups of a row panel run on one XCD and share the panel through its L2.
// ---------------------------------------------------------------------------------------------
struct QkvAttnArgs {
    const void *X = nullptr;      // LayerNorm output [rows, D] bf16
    const void *W = nullptr;      // fused QKV weight [3D, D] bf16 (rows: q | k | v, attention scale folded into q)
    const float *bias = nullptr;  // [3D]
    void *att = nullptr;          // [rows, D] bf16 (or e4m3 bytes when out8)
    int M = 0, D = 0, T = 0, heads = 0, G = 0, group_m = 4, reverse = 0;
};

constexpr int QA_WBYTES = 192 * ROWB;                         // one W stage: q_h, k_h, v_h rows = 24 KiB
constexpr int QA_LDS = 3 * LX_BYTES + 2 * QA_WBYTES;          // 144 KiB
constexpr int QA_REGION = 264 * ROWB;                         // Q / K / V image after the K loop: 256 rows + 8 zeroed pad rows

template <typename TOut>
__global__ __launch_bounds__(512, 2) void qkv_attn_kernel(QkvAttnArgs g) {
    typedef __bf16 bf16x8_t __attribute__((ext_vector_type(8)));
    typedef unsigned short T;
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    IVR_STAMP(0)
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int wm = wave >> 2, wn = wave & 3;
    const int RT = g.G * g.T;                                  // rows per tile
    const int MT = (g.M + RT - 1) / RT, NT = g.heads;
    int tm, h;
    {
        const int xcd = blockIdx.x & 7, i = blockIdx.x >> 3;
        const int lx = MT > xcd ? (MT - xcd + 7) >> 3 : 0;
        if (i >= lx * NT) return;
        const int gm = g.group_m;
        const int per = gm * NT, grp = i / per, within = i - grp * per;
        const int gme = min(gm, lx - grp * gm);
        tm = xcd + 8 * (grp * gm + within % gme);
        h = within / gme;
        if (g.reverse) tm = MT - 1 - tm;
    }
    const int m0 = tm * RT;
    const int KT = g.D / 64;
    const unsigned lds0 = (unsigned)(size_t)smem;

    // DMA pieces of 1 KiB (8 rows x 128 B): 32 of X, 24 of W per stage; wave w issues X pieces 4w..4w+3 and W pieces 3w..3w+2.
    // W piece p covers rows 8p..8p+7 of the 192-row head tile: third p >> 3 (q, k, v), rows h*64 + 8*(p & 7) of that third.
    unsigned voffX[4], voffW;
    {
        const int c = (lane & 7) ^ (lane >> 3);
#pragma unroll
        for (int j = 0; j < 4; ++j) voffX[j] = (unsigned)((8 * (wave * 4 + j) + (lane >> 3)) * g.D) * 2u + c * 16;
        voffW = (unsigned)((lane >> 3) * g.D) * 2u + c * 16;
    }
    unsigned swW[3];
#pragma unroll
    for (int j = 0; j < 3; ++j) {
        const int p = wave * 3 + j;
        swW[j] = (unsigned)(((p >> 3) * g.D + h * 64 + 8 * (p & 7)) * g.D) * 2u;
    }
    const auto rsX = __builtin_amdgcn_make_buffer_rsrc(const_cast<void *>(g.X), 0, (int)((int64_t)g.M * g.D * 2), 0x00020000);
    const auto rsW = __builtin_amdgcn_make_buffer_rsrc(const_cast<void *>(g.W), 0, (int)((int64_t)3 * g.D * g.D * 2), 0x00020000);
    const unsigned sx0 = (unsigned)m0 * (unsigned)g.D * 2u;
    constexpr int WBASE = 3 * LX_BYTES;
    auto piece = [&](int kt, int xs, int j) {          // j < 4: X piece (slot xs); 4..6: W piece (slot kt & 1)
        const unsigned adv = (unsigned)kt * ROWB;
        if (j < 4)
            __builtin_amdgcn_raw_ptr_buffer_load_lds(rsX, (__attribute__((address_space(3))) void *)(smem + xs * LX_BYTES + (wave * 4 + j) * 1024),
                                                     16, voffX[j], sx0 + adv, 0, 0);
        else
            __builtin_amdgcn_raw_ptr_buffer_load_lds(
                rsW, (__attribute__((address_space(3))) void *)(smem + WBASE + (kt & 1) * QA_WBYTES + (wave * 3 + j - 4) * 1024), 16, voffW,
                swW[j - 4] + adv, 0, 0);
    };
    unsigned foX[2], foW[2];
#pragma unroll
    for (int kk = 0; kk < 2; ++kk) {
        const unsigned f = (lane & 15) * ROWB + ((((kk << 2) + (lane >> 4)) ^ (lane & 7)) << 4);
        foX[kk] = lds0 + (wm * 128) * ROWB + f;
        foW[kk] = lds0 + WBASE + (wn * 48) * ROWB + f;
    }

    f32x4 acc[3][8];   // [nt][mt]
#pragma unroll
    for (int a = 0; a < 3; ++a)
#pragma unroll
        for (int b = 0; b < 8; ++b) acc[a][b] = f32x4{0.f, 0.f, 0.f, 0.f};
    // the bias of the wave's three column tiles, fetched now so that its latency is long gone when the epilogue needs it
    float4 bq[3];
#pragma unroll
    for (int nt = 0; nt < 3; ++nt) {
        const int col = wn * 48 + nt * 16 + 4 * (lane >> 4);
        bq[nt] = *reinterpret_cast<const float4 *>(g.bias + (col >> 6) * g.D + h * 64 + (col & 63));
    }

#define QA_ROW(XF, WF, MTI)                                                                                \
    _Pragma("unroll") for (int nt = 0; nt < 3; ++nt) mma_chunk<T>(WF[nt], XF[MTI], acc[nt][MTI]);          \
    __builtin_amdgcn_sched_barrier(0);
#define QA_RD4(DST, ADDR, O0)                                                                              \
    asm volatile("ds_read_b128 %0, %1 offset:%2" : "=v"(DST[0]) : "v"(ADDR), "n"(O0));                      \
    asm volatile("ds_read_b128 %0, %1 offset:%2" : "=v"(DST[1]) : "v"(ADDR), "n"(O0 + 2048));               \
    asm volatile("ds_read_b128 %0, %1 offset:%2" : "=v"(DST[2]) : "v"(ADDR), "n"(O0 + 4096));               \
    asm volatile("ds_read_b128 %0, %1 offset:%2" : "=v"(DST[3]) : "v"(ADDR), "n"(O0 + 6144));
#define QA_RD3(DST, ADDR)                                                                                  \
    asm volatile("ds_read_b128 %0, %1" : "=v"(DST[0]) : "v"(ADDR));                                         \
    asm volatile("ds_read_b128 %0, %1 offset:2048" : "=v"(DST[1]) : "v"(ADDR));                             \
    asm volatile("ds_read_b128 %0, %1 offset:4096" : "=v"(DST[2]) : "v"(ADDR));
#define QA_LGKM(N)                                                                                         \
    asm volatile("s_waitcnt lgkmcnt(" #N ")" ::: "memory");                                                \
    __builtin_amdgcn_sched_barrier(0);

    u32x4 xa0[8], wa0[3], xa1[8], wa1[3];
    u32x4 *x0lo = xa0, *x0hi = xa0 + 4, *x1lo = xa1, *x1hi = xa1 + 4;
    // prologue (KT >= 3 is checked by the launcher): stage 0, stage 1, X(2), in the order the counted waits rely on
#pragma unroll
    for (int j = 0; j < 7; ++j) piece(0, 0, j);
#pragma unroll
    for (int j = 0; j < 7; ++j) piece(1, 1, j);
#pragma unroll
    for (int j = 0; j < 4; ++j) piece(2, 2, j);
    asm volatile("s_waitcnt vmcnt(11)\n\ts_barrier" ::: "memory");
    IVR_STAMP(1)
    QA_RD3(wa0, foW[0])
    QA_RD4(x0lo, foX[0], 0)
    QA_RD4(x0hi, foX[0], 8192)
    int xs = 0;
    for (int kt = 0; kt < KT; ++kt) {
        const int xs1 = xs == 2 ? 0 : xs + 1;
        const unsigned xoff = xs * LX_BYTES, woff = (kt & 1) * QA_WBYTES, nxoff = xs1 * LX_BYTES, nwoff = ((kt + 1) & 1) * QA_WBYTES;
        const bool tail = kt >= 1 && kt + 2 < KT;
        const bool morew = kt + 2 < KT, morex = kt + 3 < KT, next = kt + 1 < KT;
        const unsigned wa = foW[1] + woff, xa = foX[1] + xoff, nwa = foW[0] + nwoff, nxa = foX[0] + nxoff;
        QA_LGKM(4)                      // W + first four X fragments of set 0
        QA_ROW(xa0, wa0, 0)
        if (tail) piece(kt + 2, xs1 == 2 ? 0 : xs1 + 1, 2);
        QA_ROW(xa0, wa0, 1)
        QA_RD3(wa1, wa)
        QA_ROW(xa0, wa0, 2)
        if (tail) piece(kt + 2, xs1 == 2 ? 0 : xs1 + 1, 3);
        QA_ROW(xa0, wa0, 3)
        QA_RD4(x1lo, xa, 0)
        QA_LGKM(7)                      // all of set 0
        QA_ROW(xa0, wa0, 4)
        QA_ROW(xa0, wa0, 5)
        QA_RD4(x1hi, xa, 8192)
        QA_ROW(xa0, wa0, 6)
        QA_ROW(xa0, wa0, 7)
        QA_LGKM(0)
        if (next) {
            if (kt + 2 < KT) asm volatile("s_waitcnt vmcnt(4)\n\ts_barrier" ::: "memory");
            else asm volatile("s_waitcnt vmcnt(0)\n\ts_barrier" ::: "memory");
        }
        QA_ROW(xa1, wa1, 0)
        if (morew) piece(kt + 2, 0, 4);
        QA_ROW(xa1, wa1, 1)
        if (next) { QA_RD3(wa0, nwa) }
        if (morew) piece(kt + 2, 0, 5);
        QA_ROW(xa1, wa1, 2)
        if (morew) piece(kt + 2, 0, 6);
        QA_ROW(xa1, wa1, 3)
        if (next) { QA_RD4(x0lo, nxa, 0) }
        QA_ROW(xa1, wa1, 4)
        if (morex) piece(kt + 3, xs, 0);
        QA_ROW(xa1, wa1, 5)
        if (next) { QA_RD4(x0hi, nxa, 8192) }
        QA_ROW(xa1, wa1, 6)
        if (morex) piece(kt + 3, xs, 1);
        QA_ROW(xa1, wa1, 7)
        xs = xs1;
    }
    QA_LGKM(0)
#undef QA_ROW
#undef QA_RD4
#undef QA_RD3
#undef QA_LGKM
    IVR_STAMP(2)
    __builtin_amdgcn_s_barrier();                             // every wave has read its last fragments: the LDS is free

    // ---- accumulators + bias -> bf16 Q | K | V images in LDS
    unsigned char *const qreg = smem, *const kreg = smem + QA_REGION, *const vreg = smem + 2 * QA_REGION;
    {
        const int r = lane & 15, gq = lane >> 4;
#pragma unroll
        for (int nt = 0; nt < 3; ++nt) {
            const int col = wn * 48 + nt * 16 + 4 * gq;       // 0..191: region col >> 6, head dim col & 63
            const int reg3 = col >> 6, dh = col & 63, ch = dh >> 3;
            const float4 bv = bq[nt];
            unsigned char *base = smem + reg3 * QA_REGION + (dh & 7) * 2;
#pragma unroll
            for (int mt = 0; mt < 8; ++mt) {
                const int m = wm * 128 + mt * 16 + r;
                const f32x4 a = acc[nt][mt];
                uint2 o;
                o.x = ivr_pack_bf16x2(a[0] + bv.x, a[1] + bv.y);
                o.y = ivr_pack_bf16x2(a[2] + bv.z, a[3] + bv.w);
                const unsigned pos = reg3 == 2 ? (unsigned)((((ch >> 1) ^ ((m >> 1) & 3)) << 5) | ((ch & 1) << 4))
                                               : (unsigned)((ch ^ (m & 7)) << 4);
                *reinterpret_cast<uint2 *>(base + m * ROWB + pos) = o;
            }
        }
        // rows 256..263 behind the K and V images: key blocks of the tile's last image may reach them (masked scores, P = 0)
        if (tid < 128) *reinterpret_cast<uint4 *>((tid < 64 ? kreg : vreg) + 256 * ROWB + (tid & 63) * 16) = make_uint4(0u, 0u, 0u, 0u);
    }
    __syncthreads();
#ifdef IVR_GEMM_STAMPS
    if (threadIdx.x == 0 && blockIdx.x < 16384) ivr_gemm_stamps[blockIdx.x][6] = __builtin_amdgcn_s_memtime();
#endif

    // ---- attention: units (image, 16-query tile) dealt round-robin to the waves
    const int Tn = g.T, nqt = (Tn + 15) >> 4, units = g.G * nqt;
    const int gl = lane >> 4, c = lane & 15;
    constexpr float L2E = 1.4426950408889634f;
    for (int u = wave; u < units; u += 8) {
        const int img = u / nqt, qt = u - img * nqt;
        const int R0 = img * Tn;                                // first tile row of the image
        if (m0 + R0 >= g.M) continue;                           // image past the end of the batch (last panel)
        const int q = qt * 16 + c;
        // All LDS reads of the unit go out together - Q fragments (B operand of S^T: row R0 + q, chunks 4 ks + gl), K fragments
        // (A operand: rows R0 + 16 kt + c) and the sixteen transposed V^T pieces, which do not depend on the softmax - and are
        // waited for once: issued one dependent group at a time the unit was a chain of ten LDS round trips.
        uint4 qf[2], kf[2][4];
        uint2 vt[4][4];                                         // [dh tile nt][16-key block]
        {
            const int rq = R0 + q;
            const unsigned qa = (unsigned)(size_t)qreg + rq * ROWB, qx = rq & 7;
#pragma unroll
            for (int ks = 0; ks < 2; ++ks) asm volatile("ds_read_b128 %0, %1" : "=v"(qf[ks]) : "v"(qa + (((4 * ks + gl) ^ qx) << 4)));
            // the four key tiles are 16 rows = 2048 bytes apart and share (row & 7)
            const int rk = R0 + c;
            const unsigned ka = (unsigned)(size_t)kreg + rk * ROWB, kx = rk & 7;
#pragma unroll
            for (int ks = 0; ks < 2; ++ks) {
                const unsigned a0 = ka + (((4 * ks + gl) ^ kx) << 4);
                asm volatile("ds_read_b128 %0, %1" : "=v"(kf[ks][0]) : "v"(a0));
                asm volatile("ds_read_b128 %0, %1 offset:2048" : "=v"(kf[ks][1]) : "v"(a0));
                asm volatile("ds_read_b128 %0, %1 offset:4096" : "=v"(kf[ks][2]) : "v"(a0));
                asm volatile("ds_read_b128 %0, %1 offset:6144" : "=v"(kf[ks][3]) : "v"(a0));
            }
            // V^T: this lane addresses key row 4 gl + (c >> 2) of a 16-key block, 8 bytes (c & 3) of the 32-byte segment nt
            const int vrow = R0 + 4 * gl + (c >> 2);
            const unsigned vbase = (unsigned)(size_t)vreg + (unsigned)vrow * ROWB + (c & 3) * 8;
            const int xr = (vrow >> 1) & 3;                      // key blocks start 16 rows apart: (row >> 1) & 3 is the same in all four
#pragma unroll
            for (int nt = 0; nt < 4; ++nt) {
                const unsigned va = vbase + ((nt ^ xr) << 5);
                asm volatile("ds_read_b64_tr_b16 %0, %1" : "=v"(vt[nt][0]) : "v"(va));
                asm volatile("ds_read_b64_tr_b16 %0, %1 offset:2048" : "=v"(vt[nt][1]) : "v"(va));
                asm volatile("ds_read_b64_tr_b16 %0, %1 offset:4096" : "=v"(vt[nt][2]) : "v"(va));
                asm volatile("ds_read_b64_tr_b16 %0, %1 offset:6144" : "=v"(vt[nt][3]) : "v"(va));
            }
            asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
            __builtin_amdgcn_sched_barrier(0);
            // Key blocks of the tile's LAST images can reach past the 8 zeroed pad rows (T = 10: row 303, T = 26: 271) into the
            // next region or stale staging bytes.  Their scores are masked below (P = 0), but 0 x Inf/NaN is NaN inside the MFMA:
            // select zeros into the V elements of keys >= T whenever the shape can reach past row 263 (wave-uniform; never for
            // T = 50, where (G-1) T + 63 = 263).  Element r of vt[nt][kb] is key 16 kb + 4 gl + r.
            if ((g.G - 1) * Tn + 63 > 263) {
#pragma unroll
                for (int kb = 0; kb < 4; ++kb) {
                    if (kb * 16 + 15 < Tn) continue;
                    const int k0 = kb * 16 + gl * 4;
                    const unsigned mlo = (k0 < Tn ? 0x0000ffffu : 0u) | (k0 + 1 < Tn ? 0xffff0000u : 0u);
                    const unsigned mhi = (k0 + 2 < Tn ? 0x0000ffffu : 0u) | (k0 + 3 < Tn ? 0xffff0000u : 0u);
#pragma unroll
                    for (int nt = 0; nt < 4; ++nt) {
                        vt[nt][kb].x &= mlo;
                        vt[nt][kb].y &= mhi;
                    }
                }
            }
        }
        f32x4 sc[4];
        float mx = -INFINITY;
#pragma unroll
        for (int kt = 0; kt < 4; ++kt) {
            f32x4 a = {0.f, 0.f, 0.f, 0.f};
            a = __builtin_amdgcn_mfma_f32_16x16x32_bf16(__builtin_bit_cast(bf16x8_t, kf[0][kt]), __builtin_bit_cast(bf16x8_t, qf[0]), a, 0, 0, 0);
            a = __builtin_amdgcn_mfma_f32_16x16x32_bf16(__builtin_bit_cast(bf16x8_t, kf[1][kt]), __builtin_bit_cast(bf16x8_t, qf[1]), a, 0, 0, 0);
            if (kt * 16 + 15 >= Tn) {             // wave-uniform: only a tile that reaches past the sequence needs the mask
#pragma unroll
                for (int r = 0; r < 4; ++r)
                    if (kt * 16 + gl * 4 + r >= Tn) a[r] = -INFINITY;
            }
            mx = vmax3(mx, a[0], a[1]);
            mx = vmax3(mx, a[2], a[3]);
            sc[kt] = a;
        }
        mx = quad_max(mx);                    // finite: key 0 is visible to every query
        const float mb = -mx * L2E;
        float sum = 0.f;
#pragma unroll
        for (int kt = 0; kt < 4; ++kt)
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                const float pv = __builtin_amdgcn_exp2f(fmaf(sc[kt][r], L2E, mb));      // masked (-inf) -> 0
                sc[kt][r] = pv;
                sum += pv;
            }
        sum = quad_sum(sum);
        uint4 pf[2];
#pragma unroll
        for (int ks = 0; ks < 2; ++ks) {
            pf[ks].x = ivr_pack_bf16x2(sc[2 * ks][0], sc[2 * ks][1]);
            pf[ks].y = ivr_pack_bf16x2(sc[2 * ks][2], sc[2 * ks][3]);
            pf[ks].z = ivr_pack_bf16x2(sc[2 * ks + 1][0], sc[2 * ks + 1][1]);
            pf[ks].w = ivr_pack_bf16x2(sc[2 * ks + 1][2], sc[2 * ks + 1][3]);
        }
        const float inv = __builtin_amdgcn_rcpf(sum);
        const int64_t grow = (int64_t)m0 + R0 + q;
        TOut *op = reinterpret_cast<TOut *>(g.att) + grow * g.D + h * 64 + gl * 4;
#pragma unroll
        for (int nt = 0; nt < 4; ++nt) {
            f32x4 o = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
            for (int ks = 0; ks < 2; ++ks) {
                const uint4 vf = make_uint4(vt[nt][2 * ks].x, vt[nt][2 * ks].y, vt[nt][2 * ks + 1].x, vt[nt][2 * ks + 1].y);
                o = __builtin_amdgcn_mfma_f32_16x16x32_bf16(__builtin_bit_cast(bf16x8_t, vf), __builtin_bit_cast(bf16x8_t, pf[ks]), o, 0, 0, 0);
            }
            if (q < Tn && grow < g.M) {
                const float v[4] = {o[0] * inv, o[1] * inv, o[2] * inv, o[3] * inv};
                El<TOut>::store4(op + nt * 16, v);
            }
        }
    }
#ifdef IVR_GEMM_STAMPS
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __syncthreads();                                            // the slowest wave of the workgroup
    IVR_STAMP(3)
#endif
}

// ---------------------------------------------------------------------------------------------
// Persistent form of qkv_attn_kernel (round 3): one workgroup per CU walks its (row tile, head) items; the LDS is laid out so that
// stage 0 of the K-loop ring (bottom 56 KiB) and the Q | K | V images (top 99 KiB) are disjoint, and the NEXT item's stage 0 is
// fetched by LDS-DMA during the attention phase of the current one: a tile's 4.2 k-cycle prologue (11 % of its 37.9 k cycles, stamps
// of round 2) becomes the landing time of stage 1 behind an already resident stage 0.  Everything else - K loop, accumulators ->
// images, attention - is qkv_attn_kernel's code operation for operation, so the output is bit-identical to it (and to the unfused
// path): tests/test_fused_qkv_attention_gpu.py runs both.  IVR_QKV_PERS=0 keeps the one-tile-per-workgroup kernel.
// ---------------------------------------------------------------------------------------------
constexpr int QP_LDS = 160 * 1024;

template <typename TOut>
__global__ __launch_bounds__(512, 2) void qkv_attn_pers_kernel(QkvAttnArgs g) {
    typedef __bf16 bf16x8_t __attribute__((ext_vector_type(8)));
    typedef unsigned short T;
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int wm = wave >> 2, wn = wave & 3;
    const int RT = g.G * g.T;                                  // rows per tile
    const int MT = (g.M + RT - 1) / RT, NT = g.heads;
    // persistent: one workgroup per CU walks items slot, slot + nslots, ... of its XCD's tile list (the order of qkv_attn_kernel)
    const int xcd = blockIdx.x & 7, slot = blockIdx.x >> 3, nslots = gridDim.x >> 3;
    const int lx = MT > xcd ? (MT - xcd + 7) >> 3 : 0;
    const int nitems = lx * NT;
    if (slot >= nitems) return;
    const int count = (nitems - slot + nslots - 1) / nslots;
    auto locate = [&](int n, int &tm_, int &h_) {
        const int i = slot + n * nslots;
        const int gm = g.group_m;
        const int per = gm * NT, grp = i / per, within = i - grp * per;
        const int gme = min(gm, lx - grp * gm);
        tm_ = xcd + 8 * (grp * gm + within % gme);
        h_ = within / gme;
        if (g.reverse) tm_ = MT - 1 - tm_;
    };
    const int KT = g.D / 64;
    const unsigned lds0 = (unsigned)(size_t)smem;

    // DMA pieces of 1 KiB (8 rows x 128 B): 32 of X, 24 of W per stage; wave w issues X pieces 4w..4w+3 and W pieces 3w..3w+2.
    // W piece p covers rows 8p..8p+7 of the 192-row head tile: third p >> 3 (q, k, v), rows h*64 + 8*(p & 7) of that third.
    unsigned voffX[4], voffW;
    {
        const int c = (lane & 7) ^ (lane >> 3);
#pragma unroll
        for (int j = 0; j < 4; ++j) voffX[j] = (unsigned)((8 * (wave * 4 + j) + (lane >> 3)) * g.D) * 2u + c * 16;
        voffW = (unsigned)((lane >> 3) * g.D) * 2u + c * 16;
    }
    // scalar offsets of an item's operands: X rows of its tile, W rows (q | k | v of its head) of this wave's three pieces
    auto item_offsets = [&](int tm_, int h_, unsigned &sx, unsigned (&sw)[3]) {
        sx = (unsigned)(tm_ * RT) * (unsigned)g.D * 2u;
#pragma unroll
        for (int j = 0; j < 3; ++j) {
            const int p = wave * 3 + j;
            sw[j] = (unsigned)(((p >> 3) * g.D + h_ * 64 + 8 * (p & 7)) * g.D) * 2u;
        }
    };
    const auto rsX = __builtin_amdgcn_make_buffer_rsrc(const_cast<void *>(g.X), 0, (int)((int64_t)g.M * g.D * 2), 0x00020000);
    const auto rsW = __builtin_amdgcn_make_buffer_rsrc(const_cast<void *>(g.W), 0, (int)((int64_t)3 * g.D * g.D * 2), 0x00020000);
    // LDS (160 KiB), laid out so that stage 0 of the ring (X slot 0 + W slot 0 = the bottom 56 KiB) is disjoint from the Q | K | V
    // images (the top 99 KiB): the NEXT item's stage 0 is fetched while the attention phase of the current one reads the images.
    auto xbase = [](int xs) { return xs == 0 ? 0 : xs == 1 ? 57344 : 90112; };         // X slots: 0, 56 K, 88 K (32 KiB each)
    auto wbase = [](int b) { return b ? 122880 : 32768; };                              // W slots: 32 K, 120 K (24 KiB each)
    constexpr int IMG = 62464;                                                           // images: 61 K .. 160 K
    auto piece = [&](unsigned sx, const unsigned (&sw)[3], int kt, int xs, int j) {      // j < 4: X piece (slot xs); 4..6: W piece (slot kt & 1)
        const unsigned adv = (unsigned)kt * ROWB;
        if (j < 4)
            __builtin_amdgcn_raw_ptr_buffer_load_lds(rsX, (__attribute__((address_space(3))) void *)(smem + xbase(xs) + (wave * 4 + j) * 1024), 16,
                                                     voffX[j], sx + adv, 0, 0);
        else
            __builtin_amdgcn_raw_ptr_buffer_load_lds(rsW, (__attribute__((address_space(3))) void *)(smem + wbase(kt & 1) + (wave * 3 + j - 4) * 1024),
                                                     16, voffW, sw[j - 4] + adv, 0, 0);
    };
    unsigned foX[2], foW[2];
#pragma unroll
    for (int kk = 0; kk < 2; ++kk) {
        const unsigned f = (lane & 15) * ROWB + ((((kk << 2) + (lane >> 4)) ^ (lane & 7)) << 4);
        foX[kk] = lds0 + (wm * 128) * ROWB + f;
        foW[kk] = lds0 + (wn * 48) * ROWB + f;
    }

    int tm, h;
    locate(0, tm, h);
    unsigned sx0, swW[3];
    item_offsets(tm, h, sx0, swW);
#define QA_ROW(XF, WF, MTI)                                                                                \
    _Pragma("unroll") for (int nt = 0; nt < 3; ++nt) mma_chunk<T>(WF[nt], XF[MTI], acc[nt][MTI]);          \
    __builtin_amdgcn_sched_barrier(0);
#define QA_RD4(DST, ADDR, O0)                                                                              \
    asm volatile("ds_read_b128 %0, %1 offset:%2" : "=v"(DST[0]) : "v"(ADDR), "n"(O0));                      \
    asm volatile("ds_read_b128 %0, %1 offset:%2" : "=v"(DST[1]) : "v"(ADDR), "n"(O0 + 2048));               \
    asm volatile("ds_read_b128 %0, %1 offset:%2" : "=v"(DST[2]) : "v"(ADDR), "n"(O0 + 4096));               \
    asm volatile("ds_read_b128 %0, %1 offset:%2" : "=v"(DST[3]) : "v"(ADDR), "n"(O0 + 6144));
#define QA_RD3(DST, ADDR)                                                                                  \
    asm volatile("ds_read_b128 %0, %1" : "=v"(DST[0]) : "v"(ADDR));                                         \
    asm volatile("ds_read_b128 %0, %1 offset:2048" : "=v"(DST[1]) : "v"(ADDR));                             \
    asm volatile("ds_read_b128 %0, %1 offset:4096" : "=v"(DST[2]) : "v"(ADDR));
#define QA_LGKM(N)                                                                                         \
    asm volatile("s_waitcnt lgkmcnt(" #N ")" ::: "memory");                                                \
    __builtin_amdgcn_sched_barrier(0);

    u32x4 xa0[8], wa0[3], xa1[8], wa1[3];
    u32x4 *x0lo = xa0, *x0hi = xa0 + 4, *x1lo = xa1, *x1hi = xa1 + 4;
    // stage 0 of the first item; every later item finds its stage 0 fetched during the previous item's attention phase
#pragma unroll
    for (int j = 0; j < 7; ++j) piece(sx0, swW, 0, 0, j);
  for (int n = 0; n < count; ++n) {
    const int m0 = tm * RT;
    f32x4 acc[3][8];   // [nt][mt]
#pragma unroll
    for (int a = 0; a < 3; ++a)
#pragma unroll
        for (int b = 0; b < 8; ++b) acc[a][b] = f32x4{0.f, 0.f, 0.f, 0.f};
    float4 bq[3];
#pragma unroll
    for (int nt = 0; nt < 3; ++nt) {
        const int col = wn * 48 + nt * 16 + 4 * (lane >> 4);
        bq[nt] = *reinterpret_cast<const float4 *>(g.bias + (col >> 6) * g.D + h * 64 + (col & 63));
    }
    // stage 1 and X(2), then wait for stage 0: the 11 pieces just issued are the youngest LOADS in flight and may stay (the stores of
    // the previous attention phase count in vmcnt too, but can only lengthen the wait: scanq_kernel)
#pragma unroll
    for (int j = 0; j < 7; ++j) piece(sx0, swW, 1, 1, j);
#pragma unroll
    for (int j = 0; j < 4; ++j) piece(sx0, swW, 2, 2, j);
    asm volatile("s_waitcnt vmcnt(11)\n\ts_barrier" ::: "memory");
    QA_RD3(wa0, foW[0] + wbase(0))
    QA_RD4(x0lo, foX[0] + xbase(0), 0)
    QA_RD4(x0hi, foX[0] + xbase(0), 8192)
    int xs = 0;
    for (int kt = 0; kt < KT; ++kt) {
        const int xs1 = xs == 2 ? 0 : xs + 1;
        const unsigned xoff = xbase(xs), woff = wbase(kt & 1), nxoff = xbase(xs1), nwoff = wbase((kt + 1) & 1);
        const bool tail = kt >= 1 && kt + 2 < KT;
        const bool morew = kt + 2 < KT, morex = kt + 3 < KT, next = kt + 1 < KT;
        const unsigned wa = foW[1] + woff, xa = foX[1] + xoff, nwa = foW[0] + nwoff, nxa = foX[0] + nxoff;
        QA_LGKM(4)                      // W + first four X fragments of set 0
        QA_ROW(xa0, wa0, 0)
        if (tail) piece(sx0, swW, kt + 2, xs1 == 2 ? 0 : xs1 + 1, 2);
        QA_ROW(xa0, wa0, 1)
        QA_RD3(wa1, wa)
        QA_ROW(xa0, wa0, 2)
        if (tail) piece(sx0, swW, kt + 2, xs1 == 2 ? 0 : xs1 + 1, 3);
        QA_ROW(xa0, wa0, 3)
        QA_RD4(x1lo, xa, 0)
        QA_LGKM(7)                      // all of set 0
        QA_ROW(xa0, wa0, 4)
        QA_ROW(xa0, wa0, 5)
        QA_RD4(x1hi, xa, 8192)
        QA_ROW(xa0, wa0, 6)
        QA_ROW(xa0, wa0, 7)
        QA_LGKM(0)
        if (next) {
            if (kt + 2 < KT) asm volatile("s_waitcnt vmcnt(4)\n\ts_barrier" ::: "memory");
            else asm volatile("s_waitcnt vmcnt(0)\n\ts_barrier" ::: "memory");
        }
        QA_ROW(xa1, wa1, 0)
        if (morew) piece(sx0, swW, kt + 2, 0, 4);
        QA_ROW(xa1, wa1, 1)
        if (next) { QA_RD3(wa0, nwa) }
        if (morew) piece(sx0, swW, kt + 2, 0, 5);
        QA_ROW(xa1, wa1, 2)
        if (morew) piece(sx0, swW, kt + 2, 0, 6);
        QA_ROW(xa1, wa1, 3)
        if (next) { QA_RD4(x0lo, nxa, 0) }
        QA_ROW(xa1, wa1, 4)
        if (morex) piece(sx0, swW, kt + 3, xs, 0);
        QA_ROW(xa1, wa1, 5)
        if (next) { QA_RD4(x0hi, nxa, 8192) }
        QA_ROW(xa1, wa1, 6)
        if (morex) piece(sx0, swW, kt + 3, xs, 1);
        QA_ROW(xa1, wa1, 7)
        xs = xs1;
    }
    QA_LGKM(0)
#undef QA_ROW
#undef QA_RD4
#undef QA_RD3
#undef QA_LGKM
    __builtin_amdgcn_s_barrier();                             // every wave has read its last fragments: the LDS is free

    // ---- accumulators + bias -> bf16 Q | K | V images in LDS
    unsigned char *const qreg = smem + IMG, *const kreg = smem + IMG + QA_REGION, *const vreg = smem + IMG + 2 * QA_REGION;
    {
        const int r = lane & 15, gq = lane >> 4;
#pragma unroll
        for (int nt = 0; nt < 3; ++nt) {
            const int col = wn * 48 + nt * 16 + 4 * gq;       // 0..191: region col >> 6, head dim col & 63
            const int reg3 = col >> 6, dh = col & 63, ch = dh >> 3;
            const float4 bv = bq[nt];
            unsigned char *base = smem + IMG + reg3 * QA_REGION + (dh & 7) * 2;
#pragma unroll
            for (int mt = 0; mt < 8; ++mt) {
                const int m = wm * 128 + mt * 16 + r;
                const f32x4 a = acc[nt][mt];
                uint2 o;
                o.x = ivr_pack_bf16x2(a[0] + bv.x, a[1] + bv.y);
                o.y = ivr_pack_bf16x2(a[2] + bv.z, a[3] + bv.w);
                const unsigned pos = reg3 == 2 ? (unsigned)((((ch >> 1) ^ ((m >> 1) & 3)) << 5) | ((ch & 1) << 4))
                                               : (unsigned)((ch ^ (m & 7)) << 4);
                *reinterpret_cast<uint2 *>(base + m * ROWB + pos) = o;
            }
        }
        // rows 256..263 behind the K and V images: key blocks of the tile's last image may reach them (masked scores, P = 0)
        if (tid < 128) *reinterpret_cast<uint4 *>((tid < 64 ? kreg : vreg) + 256 * ROWB + (tid & 63) * 16) = make_uint4(0u, 0u, 0u, 0u);
    }
    __syncthreads();
    // the next item's stage 0 goes into the bottom 56 KiB (nobody reads the ring any more) while the attention below reads the images
    const int h_cur = h;
    if (n + 1 < count) {
        locate(n + 1, tm, h);
        item_offsets(tm, h, sx0, swW);
#pragma unroll
        for (int j = 0; j < 7; ++j) piece(sx0, swW, 0, 0, j);
    }

    // ---- attention: units (image, 16-query tile) dealt round-robin to the waves
    const int Tn = g.T, nqt = (Tn + 15) >> 4, units = g.G * nqt;
    const int gl = lane >> 4, c = lane & 15;
    constexpr float L2E = 1.4426950408889634f;
    for (int u = wave; u < units; u += 8) {
        const int img = u / nqt, qt = u - img * nqt;
        const int R0 = img * Tn;                                // first tile row of the image
        if (m0 + R0 >= g.M) continue;                           // image past the end of the batch (last panel)
        const int q = qt * 16 + c;
        // All LDS reads of the unit go out together - Q fragments (B operand of S^T: row R0 + q, chunks 4 ks + gl), K fragments
        // (A operand: rows R0 + 16 kt + c) and the sixteen transposed V^T pieces, which do not depend on the softmax - and are
        // waited for once: issued one dependent group at a time the unit was a chain of ten LDS round trips.
        uint4 qf[2], kf[2][4];
        uint2 vt[4][4];                                         // [dh tile nt][16-key block]
        {
            const int rq = R0 + q;
            const unsigned qa = (unsigned)(size_t)qreg + rq * ROWB, qx = rq & 7;
#pragma unroll
            for (int ks = 0; ks < 2; ++ks) asm volatile("ds_read_b128 %0, %1" : "=v"(qf[ks]) : "v"(qa + (((4 * ks + gl) ^ qx) << 4)));
            // the four key tiles are 16 rows = 2048 bytes apart and share (row & 7)
            const int rk = R0 + c;
            const unsigned ka = (unsigned)(size_t)kreg + rk * ROWB, kx = rk & 7;
#pragma unroll
            for (int ks = 0; ks < 2; ++ks) {
                const unsigned a0 = ka + (((4 * ks + gl) ^ kx) << 4);
                asm volatile("ds_read_b128 %0, %1" : "=v"(kf[ks][0]) : "v"(a0));
                asm volatile("ds_read_b128 %0, %1 offset:2048" : "=v"(kf[ks][1]) : "v"(a0));
                asm volatile("ds_read_b128 %0, %1 offset:4096" : "=v"(kf[ks][2]) : "v"(a0));
                asm volatile("ds_read_b128 %0, %1 offset:6144" : "=v"(kf[ks][3]) : "v"(a0));
            }
            // V^T: this lane addresses key row 4 gl + (c >> 2) of a 16-key block, 8 bytes (c & 3) of the 32-byte segment nt
            const int vrow = R0 + 4 * gl + (c >> 2);
            const unsigned vbase = (unsigned)(size_t)vreg + (unsigned)vrow * ROWB + (c & 3) * 8;
            const int xr = (vrow >> 1) & 3;                      // key blocks start 16 rows apart: (row >> 1) & 3 is the same in all four
#pragma unroll
            for (int nt = 0; nt < 4; ++nt) {
                const unsigned va = vbase + ((nt ^ xr) << 5);
                asm volatile("ds_read_b64_tr_b16 %0, %1" : "=v"(vt[nt][0]) : "v"(va));
                asm volatile("ds_read_b64_tr_b16 %0, %1 offset:2048" : "=v"(vt[nt][1]) : "v"(va));
                asm volatile("ds_read_b64_tr_b16 %0, %1 offset:4096" : "=v"(vt[nt][2]) : "v"(va));
                asm volatile("ds_read_b64_tr_b16 %0, %1 offset:6144" : "=v"(vt[nt][3]) : "v"(va));
            }
            asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
            __builtin_amdgcn_sched_barrier(0);
            // Key blocks of the tile's LAST images can reach past the 8 zeroed pad rows (T = 10: row 303, T = 26: 271) into the
            // next region or stale staging bytes.  Their scores are masked below (P = 0), but 0 x Inf/NaN is NaN inside the MFMA:
            // select zeros into the V elements of keys >= T whenever the shape can reach past row 263 (wave-uniform; never for
            // T = 50, where (G-1) T + 63 = 263).  Element r of vt[nt][kb] is key 16 kb + 4 gl + r.
            if ((g.G - 1) * Tn + 63 > 263) {
#pragma unroll
                for (int kb = 0; kb < 4; ++kb) {
                    if (kb * 16 + 15 < Tn) continue;
                    const int k0 = kb * 16 + gl * 4;
                    const unsigned mlo = (k0 < Tn ? 0x0000ffffu : 0u) | (k0 + 1 < Tn ? 0xffff0000u : 0u);
                    const unsigned mhi = (k0 + 2 < Tn ? 0x0000ffffu : 0u) | (k0 + 3 < Tn ? 0xffff0000u : 0u);
#pragma unroll
                    for (int nt = 0; nt < 4; ++nt) {
                        vt[nt][kb].x &= mlo;
                        vt[nt][kb].y &= mhi;
                    }
                }
            }
        }
        f32x4 sc[4];
        float mx = -INFINITY;
#pragma unroll
        for (int kt = 0; kt < 4; ++kt) {
            f32x4 a = {0.f, 0.f, 0.f, 0.f};
            a = __builtin_amdgcn_mfma_f32_16x16x32_bf16(__builtin_bit_cast(bf16x8_t, kf[0][kt]), __builtin_bit_cast(bf16x8_t, qf[0]), a, 0, 0, 0);
            a = __builtin_amdgcn_mfma_f32_16x16x32_bf16(__builtin_bit_cast(bf16x8_t, kf[1][kt]), __builtin_bit_cast(bf16x8_t, qf[1]), a, 0, 0, 0);
            if (kt * 16 + 15 >= Tn) {             // wave-uniform: only a tile that reaches past the sequence needs the mask
#pragma unroll
                for (int r = 0; r < 4; ++r)
                    if (kt * 16 + gl * 4 + r >= Tn) a[r] = -INFINITY;
            }
            mx = vmax3(mx, a[0], a[1]);
            mx = vmax3(mx, a[2], a[3]);
            sc[kt] = a;
        }
        mx = quad_max(mx);                    // finite: key 0 is visible to every query
        const float mb = -mx * L2E;
        float sum = 0.f;
#pragma unroll
        for (int kt = 0; kt < 4; ++kt)
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                const float pv = __builtin_amdgcn_exp2f(fmaf(sc[kt][r], L2E, mb));      // masked (-inf) -> 0
                sc[kt][r] = pv;
                sum += pv;
            }
        sum = quad_sum(sum);
        uint4 pf[2];
#pragma unroll
        for (int ks = 0; ks < 2; ++ks) {
            pf[ks].x = ivr_pack_bf16x2(sc[2 * ks][0], sc[2 * ks][1]);
            pf[ks].y = ivr_pack_bf16x2(sc[2 * ks][2], sc[2 * ks][3]);
            pf[ks].z = ivr_pack_bf16x2(sc[2 * ks + 1][0], sc[2 * ks + 1][1]);
            pf[ks].w = ivr_pack_bf16x2(sc[2 * ks + 1][2], sc[2 * ks + 1][3]);
        }
        const float inv = __builtin_amdgcn_rcpf(sum);
        const int64_t grow = (int64_t)m0 + R0 + q;
        TOut *op = reinterpret_cast<TOut *>(g.att) + grow * g.D + h_cur * 64 + gl * 4;
#pragma unroll
        for (int nt = 0; nt < 4; ++nt) {
            f32x4 o = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
            for (int ks = 0; ks < 2; ++ks) {
                const uint4 vf = make_uint4(vt[nt][2 * ks].x, vt[nt][2 * ks].y, vt[nt][2 * ks + 1].x, vt[nt][2 * ks + 1].y);
                o = __builtin_amdgcn_mfma_f32_16x16x32_bf16(__builtin_bit_cast(bf16x8_t, vf), __builtin_bit_cast(bf16x8_t, pf[ks]), o, 0, 0, 0);
            }
            if (q < Tn && grow < g.M) {
                const float v[4] = {o[0] * inv, o[1] * inv, o[2] * inv, o[3] * inv};
                El<TOut>::store4(op + nt * 16, v);
            }
        }
    }
    if (n + 1 < count) __syncthreads();                         // every wave is done with the images: the ring may take stage 1 and X(2)
  }
}


// ---------------------------------------------------------------------------------------------
// Persistent form of the 256 x 256 bf16 GEMM (round 3).  s_memtime stamps of gemm_big_kernel (tools/exp_epilogue_contention.py,
// profiles/r03h_epilogue_contention.log): a tile's prologue is 4.6 - 7.3 k cycles (fc1: 10 % of the tile), and the f32 residual
// epilogue takes 16 k cycles with a quarter of the chip busy but 31 - 35 k with every CU in it at once - the lockstep of
// same-length tiles makes the chip's HBM idle during K loops and saturate during epilogues.  Here one workgroup per CU walks
// its tiles as ONE flattened sequence of K stages, exactly like scanq_kernel (search_scanq.hip):
//   * the LDS-DMA cursors run two / one stages ahead ACROSS tile boundaries: only the first tile of a workgroup has a prologue;
//   * the epilogues therefore may not use the LDS (the next tile's stages are live in it): the bf16 store epilogue pairs column
//     tiles with v_permlane16_swap so that a lane holds 8 consecutive columns (16 rows x 64 B per store), the f32 residual
//     epilogue reads / adds / writes float4 straight from the MFMA layout (16 rows x 64 B); the fragment prefetch of the next
//     stage is deferred behind the epilogue, which frees its 48 registers for the residual rows in flight;
//   * odd slots start half a tile late (GemmArgs.stagger), so that half the chip is in its K loop while the other half is in its
//     epilogue - the phase persists because every tile of a launch takes the same time.
// Tile order inside an XCD = gemm_big_kernel's (groups of group_m row panels, column tiles outermost inside a group), dealt to
// the XCD's slots round-robin: slots that run at the same time work on the same few panels and column tiles.
// Arithmetic per output element is gemm_big_kernel's (same K order, same epilogue expressions): results are bit-identical.
// Measured (tools/ab_gemm_pers.py, profiles/r03i_ab_gemm_persistent.log, interleaved rounds in one process, 4,096 frames):
//   fc1 (bf16 store + quick_gelu, K = 768) 0.990 -> 0.937 ms (-5 %), qkv 0.689 -> 0.685; patch (K = 3072: 48 stages, the prologue is
//   nothing) 0.767 -> 0.775; the RESIDUAL sites LOSE: attn-out 0.375 -> 0.466 ms, fc2 0.908 -> 1.32 ms - the 64-byte read-modify-write
//   straight from the MFMA layout is far slower than the 4-row x 256-byte LDS-staged one, and the staggered start changes nothing
//   (+-1 %; also with the tile-per-workgroup body in a persistent loop: 0.445 vs 0.439 ms, both slower than the plain launch, whose
//   dispatch order balances the uneven epilogue times dynamically).  So the launcher uses this kernel for the store epilogue with
//   K <= 1024 only; IVR_GEMM_PERS=0 switches it off, 2 forces it wherever the shape allows (both epilogues: the parity tests).
// ---------------------------------------------------------------------------------------------
template <int EPI, int ACT>
__global__ __launch_bounds__(512, 2) void gemm_pers_kernel(GemmArgs g) {
    typedef unsigned short T;
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int wm = wave >> 2, wn = wave & 3;
    const int MT = (g.M + LBM - 1) / LBM, NT = (g.N + LBN - 1) / LBN;
    const int xcd = blockIdx.x & 7, slot = blockIdx.x >> 3, nslots = gridDim.x >> 3;
    const int lx = MT > xcd ? (MT - xcd + 7) >> 3 : 0;                   // row panels of this XCD
    const int nitems = lx * NT;
    if (slot >= nitems) return;
    const int count = (nitems - slot + nslots - 1) / nslots;
    const int KT = g.K / 64;
    const int S = count * KT;
    const unsigned lds0 = (unsigned)(size_t)smem;
    const int gm = g.group_m;
    // item n of this workgroup = number slot + n * nslots of the XCD's tile list
    auto locate = [&](int n, int &tm, int &tn) {
        const unsigned i = (unsigned)(slot + n * nslots);
        const unsigned per = (unsigned)(gm * NT), grp = i / per, within = i - grp * per;
        const unsigned gme = (unsigned)min(gm, lx - (int)grp * gm);
        tm = xcd + 8 * (int)(grp * gm + within % gme);
        tn = (int)(within / gme);
        if (g.reverse_m) tm = MT - 1 - tm;
    };
    // per-lane part of a DMA piece's address (row lane >> 3 of the piece, swizzled 16-byte chunk): ONE register per operand; the
    // piece's own rows (8 * (4 wave + j)) go into the scalar offset (gemm_big_kernel keeps eight such registers)
    const unsigned voffX = (unsigned)((lane >> 3) * g.lda) * 2u + (((lane & 7) ^ (lane >> 3)) << 4);
    const unsigned voffW = (unsigned)((lane >> 3) * g.ldw) * 2u + (((lane & 7) ^ (lane >> 3)) << 4);
    const unsigned prowX = (unsigned)(8 * wave * 4 * g.lda) * 2u, pstepX = (unsigned)(8 * g.lda) * 2u;
    const unsigned prowW = (unsigned)(8 * wave * 4 * g.ldw) * 2u, pstepW = (unsigned)(8 * g.ldw) * 2u;
    const auto rsX = __builtin_amdgcn_make_buffer_rsrc(const_cast<void *>(g.A), 0, (int)((int64_t)g.M * g.lda * 2), 0x00020000);
    const auto rsW = __builtin_amdgcn_make_buffer_rsrc(const_cast<void *>(g.W), 0, (int)((int64_t)g.N * g.ldw * 2), 0x00020000);
    constexpr int WBASE = 3 * LX_BYTES;
    // DMA cursors: the next stage to issue for the row operand (X) and for the weights (W)
    int xn = 0, xkt = 0, wcn = 0, wkt = 0;
    unsigned sx0, sw0;
    {
        int tm, tn;
        locate(0, tm, tn);
        sx0 = (unsigned)(tm * LBM) * (unsigned)g.lda * 2u;
        sw0 = (unsigned)(tn * LBN) * (unsigned)g.ldw * 2u;
    }
    auto pieceX = [&](int xs, int j) {
        __builtin_amdgcn_raw_ptr_buffer_load_lds(rsX, (__attribute__((address_space(3))) void *)(smem + xs * LX_BYTES + (wave * 4 + j) * 1024), 16,
                                                 voffX, sx0 + prowX + (unsigned)j * pstepX + (unsigned)xkt * ROWB, 0, 0);
    };
    auto pieceW = [&](int ws, int j) {
        __builtin_amdgcn_raw_ptr_buffer_load_lds(rsW, (__attribute__((address_space(3))) void *)(smem + WBASE + ws * LW_BYTES + (wave * 4 + j) * 1024),
                                                 16, voffW, sw0 + prowW + (unsigned)j * pstepW + (unsigned)wkt * ROWB, 0, 0);
    };
    auto advanceX = [&]() {
        if (++xkt == KT) {
            xkt = 0;
            if (++xn < count) {
                int tm, tn;
                locate(xn, tm, tn);
                sx0 = (unsigned)(tm * LBM) * (unsigned)g.lda * 2u;
            }
        }
    };
    auto advanceW = [&]() {
        if (++wkt == KT) {
            wkt = 0;
            if (++wcn < count) {
                int tm, tn;
                locate(wcn, tm, tn);
                sw0 = (unsigned)(tn * LBN) * (unsigned)g.ldw * 2u;
            }
        }
    };
    // fragment addresses of the first K half; the second half is the same address with chunk bit 2 flipped (^ 64): two registers
    const unsigned f0 = (lane & 15) * ROWB + (((lane >> 4) ^ (lane & 7)) << 4);
    const unsigned foX0 = lds0 + (wm * 128) * ROWB + f0, foW0 = lds0 + WBASE + (wn * 64) * ROWB + f0;
    // odd slots start half a tile late: their epilogues fall into the other half's K loops
    if (g.stagger > 0 && (slot & 1)) {
        for (int i = 0; i < g.stagger; ++i) __builtin_amdgcn_s_sleep(127);
    }

    f32x4 acc[4][8];   // [nt][mt]
    const f32x4 zero = {0.f, 0.f, 0.f, 0.f};
    typedef __bf16 bf16x8_t __attribute__((ext_vector_type(8)));
#define GP_ROW(XF, WF, MT_, FIRST)                                                                                          \
    _Pragma("unroll") for (int nt = 0; nt < 4; ++nt) acc[nt][MT_] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(                 \
        __builtin_bit_cast(bf16x8_t, WF[nt]), __builtin_bit_cast(bf16x8_t, XF[MT_]), FIRST ? zero : acc[nt][MT_], 0, 0, 0);  \
    __builtin_amdgcn_sched_barrier(0);
#define GP_RD4(DST, ADDR, O0)                                                                              \
    asm volatile("ds_read_b128 %0, %1 offset:%2" : "=v"(DST[0]) : "v"(ADDR), "n"(O0));                      \
    asm volatile("ds_read_b128 %0, %1 offset:%2" : "=v"(DST[1]) : "v"(ADDR), "n"(O0 + 2048));               \
    asm volatile("ds_read_b128 %0, %1 offset:%2" : "=v"(DST[2]) : "v"(ADDR), "n"(O0 + 4096));               \
    asm volatile("ds_read_b128 %0, %1 offset:%2" : "=v"(DST[3]) : "v"(ADDR), "n"(O0 + 6144));
#define GP_LGKM(N)                                                                                         \
    asm volatile("s_waitcnt lgkmcnt(" #N ")" ::: "memory");                                                \
    __builtin_amdgcn_sched_barrier(0);

    u32x4 xa0[8], wa0[4], xa1[8], wa1[4];
    u32x4 *x0lo = xa0, *x0hi = xa0 + 4, *x1lo = xa1, *x1hi = xa1 + 4;
    // prologue (once per workgroup): stage 0, then X(1), W(1), X(2)
#pragma unroll
    for (int j = 0; j < 4; ++j) pieceX(0, j);
#pragma unroll
    for (int j = 0; j < 4; ++j) pieceW(0, j);
    advanceX();
    advanceW();
    if (S > 1) {
#pragma unroll
        for (int j = 0; j < 4; ++j) pieceX(1, j);
#pragma unroll
        for (int j = 0; j < 4; ++j) pieceW(1, j);
        advanceX();
        advanceW();
    }
    if (S > 2) {
#pragma unroll
        for (int j = 0; j < 4; ++j) pieceX(2, j);
        advanceX();
    }
    if (S > 2) asm volatile("s_waitcnt vmcnt(12)\n\ts_barrier" ::: "memory");
    else if (S > 1) asm volatile("s_waitcnt vmcnt(8)\n\ts_barrier" ::: "memory");
    else asm volatile("s_waitcnt vmcnt(0)\n\ts_barrier" ::: "memory");
    GP_RD4(wa0, foW0, 0)
    GP_RD4(x0lo, foX0, 0)
    GP_RD4(x0hi, foX0, 8192)

    int xs = 0, cn = 0, ckt = 0;
    for (int s = 0; s < S; ++s) {
        const int xs1 = xs == 2 ? 0 : xs + 1, xs2 = xs1 == 2 ? 0 : xs1 + 1;
        const unsigned xoff = xs * LX_BYTES, woff = (s & 1) * LW_BYTES, nxoff = xs1 * LX_BYTES, nwoff = ((s + 1) & 1) * LW_BYTES;
        const bool tail = s >= 1 && s + 2 < S;
        const bool morew = s + 2 < S, morex = s + 3 < S, next = s + 1 < S;
        const bool last = ckt == KT - 1;              // the item's last stage: its epilogue follows
        const unsigned wa = (foW0 ^ 64u) + woff, xa = (foX0 ^ 64u) + xoff, nwa = foW0 + nwoff, nxa = foX0 + nxoff;
#define GP_SET0(FIRST)                                                                                     \
        GP_LGKM(4)                                                                                         \
        GP_ROW(xa0, wa0, 0, FIRST)                                                                         \
        if (tail) pieceX(xs2, 2);                                                                          \
        GP_ROW(xa0, wa0, 1, FIRST)                                                                         \
        GP_RD4(wa1, wa, 0)                                                                                 \
        GP_ROW(xa0, wa0, 2, FIRST)                                                                         \
        if (tail) pieceX(xs2, 3);                                                                          \
        GP_ROW(xa0, wa0, 3, FIRST)                                                                         \
        GP_RD4(x1lo, xa, 0)                                                                                \
        GP_LGKM(8)                                                                                         \
        GP_ROW(xa0, wa0, 4, FIRST)                                                                         \
        GP_ROW(xa0, wa0, 5, FIRST)                                                                         \
        GP_RD4(x1hi, xa, 8192)                                                                             \
        GP_ROW(xa0, wa0, 6, FIRST)                                                                         \
        GP_ROW(xa0, wa0, 7, FIRST)
        if (ckt == 0) {
            GP_SET0(true)
        } else {
            GP_SET0(false)
        }
#undef GP_SET0
        GP_LGKM(0)
        if (s >= 1) advanceX();
        if (next) {
            // (an epilogue's global stores count in vmcnt too; they can only lengthen this wait, never release it early: scanq_kernel)
            if (s + 2 < S) asm volatile("s_waitcnt vmcnt(4)\n\ts_barrier" ::: "memory");
            else asm volatile("s_waitcnt vmcnt(0)\n\ts_barrier" ::: "memory");
        }
        GP_ROW(xa1, wa1, 0, false)
        if (morew) pieceW(s & 1, 0);
        GP_ROW(xa1, wa1, 1, false)
        if (next) { GP_RD4(wa0, nwa, 0) }
        if (morew) pieceW(s & 1, 1);
        GP_ROW(xa1, wa1, 2, false)
        if (morew) pieceW(s & 1, 2);
        GP_ROW(xa1, wa1, 3, false)
        if (next) { GP_RD4(x0lo, nxa, 0) }
        if (morew) pieceW(s & 1, 3);
        GP_ROW(xa1, wa1, 4, false)
        if (morex) pieceX(xs, 0);
        GP_ROW(xa1, wa1, 5, false)
        if (next) { GP_RD4(x0hi, nxa, 8192) }
        GP_ROW(xa1, wa1, 6, false)
        if (morex) pieceX(xs, 1);
        GP_ROW(xa1, wa1, 7, false)
        if (morew) advanceW();
        xs = xs1;
        if (!last) {
            ++ckt;
            continue;
        }
        // ---- tile finished: epilogue straight from the accumulators (no LDS: the next tile's stages are live in it) ----
        ckt = 0;
        int tm, tn;
        locate(cn, tm, tn);
        ++cn;
        // the lane-dependent addressing of the epilogue is derived from an opaque copy of the lane id: the compiler would otherwise
        // hoist it out of the stage loop and hold ~40 more registers across the MFMA code (259 spilled registers)
        int lane_e = lane;
        asm volatile("" : "+v"(lane_e));
        const int r = lane_e & 15, h = lane_e >> 4;
        const int row0 = tm * LBM + wm * 128, col0 = tn * LBN + wn * 64;
        if (col0 < g.N) {                                   // N is a multiple of 64: a wave's block is all in or all out
            float4 bv[4];
#pragma unroll
            for (int nt = 0; nt < 4; ++nt)
                bv[nt] = g.bias ? *reinterpret_cast<const float4 *>(g.bias + col0 + nt * 16 + 4 * h) : make_float4(0.f, 0.f, 0.f, 0.f);
            if (EPI == EPI_STORE) {
                // pair the column tiles (nt, nt+1): after the swaps lane (r, h) holds 8 consecutive columns of tile nt + (h & 1)
                T *outp = reinterpret_cast<T *>(g.out) + col0 + (h & 1) * 16 + (h >> 1) * 8;
#pragma unroll
                for (int mt = 0; mt < 8; ++mt) {
                    const int row = row0 + mt * 16 + r;
#pragma unroll
                    for (int np = 0; np < 2; ++np) {
                        uint32_t pk[2][2];
#pragma unroll
                        for (int e = 0; e < 2; ++e) {
                            const f32x4 a = acc[2 * np + e][mt];
                            const float4 b = bv[2 * np + e];
                            float v[4] = {a[0] + b.x, a[1] + b.y, a[2] + b.z, a[3] + b.w};
                            if (ACT >= 0) act4_fast(v, ACT);
                            pk[e][0] = ivr_pack_bf16x2(v[0], v[1]);
                            pk[e][1] = ivr_pack_bf16x2(v[2], v[3]);
                        }
                        const auto s0 = __builtin_amdgcn_permlane16_swap(pk[0][0], pk[1][0], false, false);
                        const auto s1 = __builtin_amdgcn_permlane16_swap(pk[0][1], pk[1][1], false, false);
                        if (row < g.M) *reinterpret_cast<uint4 *>(outp + (int64_t)row * g.ldo + np * 32) = make_uint4(s0[0], s1[0], s0[1], s1[1]);
                    }
                }
            } else {
                float *resp = g.resid + col0 + 4 * h;
#pragma unroll
                for (int mh = 0; mh < 4; ++mh) {            // two row tiles of residual in flight (32 registers: the next stage's
                    float4 rv[2][4];                        // fragments, already being read, hold 48)
#pragma unroll
                    for (int mt = 0; mt < 2; ++mt) {
                        const int row = min(row0 + (mh * 2 + mt) * 16 + r, g.M - 1);
#pragma unroll
                        for (int nt = 0; nt < 4; ++nt) rv[mt][nt] = *reinterpret_cast<const float4 *>(resp + (int64_t)row * g.ldr + nt * 16);
                    }
#pragma unroll
                    for (int mt = 0; mt < 2; ++mt) {
                        const int row = row0 + (mh * 2 + mt) * 16 + r;
#pragma unroll
                        for (int nt = 0; nt < 4; ++nt) {
                            const f32x4 a = acc[nt][mh * 2 + mt];
                            float4 o = rv[mt][nt];
                            o.x += a[0] + bv[nt].x;
                            o.y += a[1] + bv[nt].y;
                            o.z += a[2] + bv[nt].z;
                            o.w += a[3] + bv[nt].w;
                            if (row < g.M) *reinterpret_cast<float4 *>(resp + (int64_t)row * g.ldr + nt * 16) = o;
                        }
                    }
                }
            }
        }
    }
    GP_LGKM(0)
#undef GP_ROW
#undef GP_RD4
#undef GP_LGKM
}

// ---------------------------------------------------------------------------------------------
// Skinny GEMM: M <= 128 rows - one text query (77 rows) or one image of a 7 x 7 grid (50 rows).
// The interactive path of the reference (system.py:733 -> core.py:1504) encodes ONE query per call: with 128 x 128 tiles such a
// product has N / 128 workgroups (4 for N = 512) that each walk the whole K alone: 86 us per residual product of the float32 text
// tower.  Here a workgroup owns 16 output columns and a wave one to three 16 x 16 output tiles of them (up to eight waves): N / 16
// workgroups on as many CUs.  The workgroup's weight panel (16 rows x K, the only operand that comes from HBM) is fetched by LDS-DMA, ALL of it
// in flight at once (no registers involved: one memory latency per panel of up to 128 KiB instead of one per few K steps); the
// activation rows come from L2 straight into the MFMA fragment registers, five K steps ahead.  The loop body is branch-free (the
// step index of a load is clamped, the last steps are peeled) so that the compiler's vmcnt counting keeps those loads in flight.
// The accumulation order over K is the tiled kernels' (one fp32 accumulator per output, K steps in ascending order, the same MFMA
// per chunk), so a row's result does not depend on the batch it was encoded in.
// ---------------------------------------------------------------------------------------------
constexpr int SKINNY_MAX_STEPS = 64;                 // K steps (of ROWB bytes) per LDS panel: 64 x 2 KiB = 128 KiB
// Row tiles per wave the launcher uses.  The kernel takes MTL = 1 ... 3 (up to 384 rows), but past 128 rows every one of the N / 16
// workgroups re-reads the whole activation panel from L2 and that traffic takes over: measured with MTL = 2 / 3, one DINO frame (197 rows)
// gained 5 %, one ViT-L/14 image (257 rows) lost 9 % against the 128 x 128 tiled kernel.
constexpr int kSkinnyMaxMtl = 1;

template <typename T, int EPI, int ACT, int MTL>
__global__ __launch_bounds__(512) void gemm_skinny_kernel(GemmArgs g) {
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    constexpr int ES = (int)sizeof(T), DEPTH = MTL == 1 ? 6 : 4;
    const int lane = threadIdx.x & 63, r = lane & 15, kg = lane >> 4;
    const int wv = __builtin_amdgcn_readfirstlane((int)threadIdx.x >> 6), nw = (int)blockDim.x >> 6;
    const int n0 = blockIdx.x * 16;
    const int steps = g.K * ES / ROWB;
    const char *xp[MTL];                              // wave wv owns row tiles wv * MTL .. wv * MTL + MTL - 1 (rows past M read row M - 1)
#pragma unroll
    for (int i = 0; i < MTL; ++i)
        xp[i] = reinterpret_cast<const char *>(g.A) + (int64_t)min((wv * MTL + i) * 16 + r, g.M - 1) * g.lda * ES + kg * 16;
    const auto rsW = __builtin_amdgcn_make_buffer_rsrc(const_cast<void *>(g.W), 0, (int)((int64_t)g.N * g.ldw * ES), 0x00020000);
    // DMA piece = 8 rows x 128 B: lane l lands at 16 l = row (l >> 3), chunk position l & 7, and fetches logical chunk (l & 7) ^ row
    // (the tiled kernels' source-side swizzle); fragment reads apply the same involution
    const unsigned voff0 = (unsigned)((n0 + (lane >> 3)) * g.ldw) * ES + (((lane & 7) ^ (lane >> 3)) << 4);
    const unsigned voff1 = voff0 + (unsigned)(8 * g.ldw) * ES;
    unsigned fo[2];
#pragma unroll
    for (int kk = 0; kk < 2; ++kk) fo[kk] = r * ROWB + ((((kk << 2) + kg) ^ (r & 7)) << 4);
    f32x4 acc[MTL];
#pragma unroll
    for (int i = 0; i < MTL; ++i) acc[i] = f32x4{0.f, 0.f, 0.f, 0.f};
    for (int c0 = 0; c0 < steps; c0 += SKINNY_MAX_STEPS) {
        const int cs = min(SKINNY_MAX_STEPS, steps - c0), last = c0 + cs - 1;
        u32x4 xf[DEPTH][MTL][2];
        auto load = [&](int st, int b) {
            const int64_t off = (int64_t)min(st, last) * ROWB;
#pragma unroll
            for (int i = 0; i < MTL; ++i) {
                xf[b][i][0] = *reinterpret_cast<const u32x4 *>(xp[i] + off);
                xf[b][i][1] = *reinterpret_cast<const u32x4 *>(xp[i] + off + 64);
            }
        };
        auto compute = [&](int st, int b) {
            const unsigned char *p = smem + (st - c0) * 2048;
            const u32x4 w0 = *reinterpret_cast<const u32x4 *>(p + fo[0]), w1 = *reinterpret_cast<const u32x4 *>(p + fo[1]);
#pragma unroll
            for (int i = 0; i < MTL; ++i) mma_chunk<T>(w0, xf[b][i][0], acc[i]);
#pragma unroll
            for (int i = 0; i < MTL; ++i) mma_chunk<T>(w1, xf[b][i][1], acc[i]);
        };
#pragma unroll
        for (int b = 0; b < DEPTH - 1; ++b) load(c0 + b, b);
        if (c0) __syncthreads();                                  // every wave has read the previous panel
        for (int q = wv; q < cs; q += nw) {       // the two pieces of K step c0 + q; per-lane offsets stay in two fixed registers, the step is scalar
            __builtin_amdgcn_raw_ptr_buffer_load_lds(rsW, (__attribute__((address_space(3))) void *)(smem + q * 2048), 16, voff0,
                                                     (unsigned)(c0 + q) * ROWB, 0, 0);
            __builtin_amdgcn_raw_ptr_buffer_load_lds(rsW, (__attribute__((address_space(3))) void *)(smem + q * 2048 + 1024), 16, voff1,
                                                     (unsigned)(c0 + q) * ROWB, 0, 0);
        }
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        __syncthreads();
        int st = c0;
        for (; st + DEPTH <= c0 + cs; st += DEPTH) {
#pragma unroll
            for (int b = 0; b < DEPTH; ++b) {
                load(st + b + DEPTH - 1, (b + DEPTH - 1) % DEPTH);       // buffer of step st + b - 1, consumed just before
                compute(st + b, b);
            }
        }
        const int rem = c0 + cs - st;                                     // the last steps (fewer than DEPTH): buffer b holds step st + b
#pragma unroll
        for (int b = 0; b < DEPTH - 1; ++b)
            if (b < rem) compute(st + b, b);
    }
    // epilogue: lane holds C[m][n .. n+3], m = tile * 16 + (lane & 15), n = n0 + 4 * (lane >> 4)
    const int ncol = n0 + 4 * kg;
    const float4 bv = g.bias ? *reinterpret_cast<const float4 *>(g.bias + ncol) : make_float4(0.f, 0.f, 0.f, 0.f);
#pragma unroll
    for (int i = 0; i < MTL; ++i) {
        const int m = (wv * MTL + i) * 16 + r;
        if (m >= g.M) continue;
        float v[4] = {acc[i][0] + bv.x, acc[i][1] + bv.y, acc[i][2] + bv.z, acc[i][3] + bv.w};
        if (EPI == EPI_RESID) {
            if (g.skip_mod && m % g.skip_mod == 0) continue;
            float4 *p = reinterpret_cast<float4 *>(g.resid + (int64_t)m * g.ldr + ncol);
            const float4 rv = *p;
            *p = make_float4(rv.x + v[0], rv.y + v[1], rv.z + v[2], rv.w + v[3]);      // residual + (accumulator + bias), as in the tiled kernels
        } else if (EPI == EPI_PATCH) {
            const int img = m / g.G2, pch = m % g.G2;
            const float4 pv = *reinterpret_cast<const float4 *>(g.pos + (int64_t)(1 + pch) * g.N + ncol);
            *reinterpret_cast<float4 *>(g.resid + ((int64_t)img * g.T + 1 + pch) * g.ldr + ncol) =
                make_float4(pv.x + v[0], pv.y + v[1], pv.z + v[2], pv.w + v[3]);
        } else {
            if (ACT >= 0) {
#pragma unroll
                for (int j = 0; j < 4; ++j) v[j] = act_fn<sizeof(T) == 2>(v[j], ACT);
            }
            if (EPI == EPI_STORE)
                El<T>::store4(reinterpret_cast<T *>(g.out) + (int64_t)m * g.ldo + ncol, v);
            else
                *reinterpret_cast<float4 *>(reinterpret_cast<float *>(g.out) + (int64_t)m * g.ldo + ncol) = make_float4(v[0], v[1], v[2], v[3]);
        }
    }
}

static int device_cu_count() {
    static std::mutex mu;
    static std::map<int, int> cus;
    int dev = 0;
    (void)hipGetDevice(&dev);
    std::lock_guard<std::mutex> lk(mu);
    int &c = cus[dev];
    if (c == 0) {
        hipDeviceProp_t prop;
        c = hipGetDeviceProperties(&prop, dev) == hipSuccess ? std::max(8, prop.multiProcessorCount) : 256;
    }
    return c;
}

template <typename T, int EPI, int ACT>
int launch_gemm_t(const GemmArgs &g, hipStream_t s) {
    IvrProf prof(g.tag ? g.tag : "gemm", s, 2.0 * g.M * g.N * g.K);
    int mode = gemm_mode();
    {
        // at most 128 rows (one text query, one ViT-B/32 image): see gemm_skinny_kernel
        const auto al16 = [](const void *p) { return reinterpret_cast<uintptr_t>(p) % 16 == 0; };
        const int64_t es = sizeof(T);
        if (g.M <= 128 * kSkinnyMaxMtl && mode < 0 && env_int("IVR_GEMM_SKINNY", 1) && g.N % 16 == 0 && g.lda * es % 16 == 0 && g.ldw * es % 16 == 0 && al16(g.A) &&
            al16(g.W) && al16(g.bias) && (EPI == EPI_STORE || EPI == EPI_F32 ? al16(g.out) && g.ldo % 4 == 0 : al16(g.resid) && g.ldr % 4 == 0) &&
            (EPI != EPI_PATCH || (al16(g.pos) && g.N % 4 == 0))) {
            const int lds = std::min(SKINNY_MAX_STEPS, (int)(g.K * es / ROWB)) * 2048;
            const int tiles = (g.M + 15) / 16, mtl = (tiles + 7) / 8, nw = (tiles + mtl - 1) / mtl;     // 1 .. 3 row tiles per wave, at most 8 waves
            auto go = [&](auto mtl_c) -> int {
                constexpr int MTL = decltype(mtl_c)::value;
                if (int rc = ivr_func_max_lds(reinterpret_cast<const void *>(gemm_skinny_kernel<T, EPI, ACT, MTL>), SKINNY_MAX_STEPS * 2048)) return rc;
                hipLaunchKernelGGL((gemm_skinny_kernel<T, EPI, ACT, MTL>), dim3(g.N / 16), dim3(64 * nw), lds, s, g);
                IVR_LAUNCH_CHECK();
                return IVR_OK;
            };
            static_assert(kSkinnyMaxMtl == 1, "add the MTL instantiations");
            (void)mtl;
            return go(std::integral_constant<int, 1>{});
        }
    }
    // default (-1): the 256 x 256 kernel once it fills the chip, the 128 x 128 kernel for small problems
    // (measured on ViT-B/32: at 150 tiles the large kernel already wins by 8 %, at 117 the small one by 7 %)
    if (mode < 0) mode = ((g.M + LBM - 1) / LBM) * ((g.N + LBN - 1) / LBN) >= 128 ? 4 : 0;
    const int group_env = std::max(0, env_int("IVR_GEMM_GROUP_M", 0)), wide_env = env_int("IVR_GEMM_WIDE_EPI", 1);
    GemmArgs ga = g;
    if (mode == 4) {
        if (int rc = ivr_func_max_lds(reinterpret_cast<const void *>(gemm_big_kernel<T, EPI, ACT>), DEEP_LDS)) return rc;
        const int MT = (g.M + LBM - 1) / LBM, NT = (g.N + LBN - 1) / LBN;
        ga.group_m = group_env ? group_env : (NT <= 3 ? 2 : 4);     // measured (sweep 2..16): within 2 %, narrow outputs want short groups
        // the row-wide epilogue needs whole 64-column wave blocks (N % 64 == 0) and 16-byte aligned rows
        const bool bias_ok = !g.bias || reinterpret_cast<uintptr_t>(g.bias) % 16 == 0;
        if (EPI == EPI_STORE)
            ga.wide_epi = wide_env && bias_ok && g.N % 64 == 0 && g.ldo % 8 == 0 && reinterpret_cast<uintptr_t>(g.out) % 16 == 0;
        else if (EPI == EPI_RESID)
            ga.wide_epi = wide_env && bias_ok && g.N % 64 == 0 && g.ldr % 4 == 0 && reinterpret_cast<uintptr_t>(g.resid) % 16 == 0;
        const int grid = 8 * ((MT + 7) / 8) * NT;
        // persistent form (gemm_pers_kernel): bf16 store / residual epilogues under the row-wide epilogue's alignment conditions,
        // problems of at least two tiles per CU.  IVR_GEMM_PERS=0 keeps the one-tile-per-workgroup kernel (A/B runs, parity tests).
        const int pers = env_int("IVR_GEMM_PERS", 1);          // 2: wherever the shape allows (tests), whatever the size
        if (sizeof(T) == 2 && (EPI == EPI_STORE || EPI == EPI_RESID) && ga.wide_epi && !g.skip_mod && pers &&
            (pers >= 2 || (EPI == EPI_STORE && g.K <= 1024 && MT * NT >= 2 * device_cu_count()))) {
            constexpr int PE = EPI == EPI_RESID ? EPI_RESID : EPI_STORE;      // (the other epilogues never get here)
            if (int rc = ivr_func_max_lds(reinterpret_cast<const void *>(gemm_pers_kernel<PE, ACT>), DEEP_LDS)) return rc;
            // half a tile of head start for the even slots when the epilogue is the HBM-bound one: (K loop + epilogue) / 2 in units
            // of s_sleep(127) = 8,128 cycles; the bf16 store epilogue is bound per CU (activation + packing), no stagger
            const int auto_stagger = EPI == EPI_RESID ? ((g.K / 64) * 2560 + 18000) / 2 / 8128 : 0;
            const int st = env_int("IVR_GEMM_STAGGER", -1);
            ga.stagger = st >= 0 ? st : auto_stagger;
            hipLaunchKernelGGL((gemm_pers_kernel<PE, ACT>), dim3(device_cu_count() / 8 * 8), dim3(512), DEEP_LDS, s, ga);
            IVR_LAUNCH_CHECK();
            return IVR_OK;
        }
        if (EPI == EPI_RESID && g.skip_mod) {
            constexpr bool kSkip = EPI == EPI_RESID;
            if (int rc = ivr_func_max_lds(reinterpret_cast<const void *>(gemm_big_kernel<T, EPI, ACT, kSkip>), DEEP_LDS)) return rc;
            hipLaunchKernelGGL((gemm_big_kernel<T, EPI, ACT, kSkip>), dim3(grid), dim3(512), DEEP_LDS, s, ga);
        } else {
            hipLaunchKernelGGL((gemm_big_kernel<T, EPI, ACT>), dim3(grid), dim3(512), DEEP_LDS, s, ga);
        }
        IVR_LAUNCH_CHECK();
        return IVR_OK;
    }
    const int MT = (g.M + BM - 1) / BM, NT = (g.N + BN - 1) / BN;
    if (int rc = ivr_func_max_lds(reinterpret_cast<const void *>(gemm_kernel<T, EPI, ACT>), GEMM_LDS)) return rc;
    ga.group_m = group_env ? group_env : 8;
    const int grid = 8 * ((MT + 7) / 8) * NT;               // every XCD gets the same number of ids; surplus ones exit
    hipLaunchKernelGGL((gemm_kernel<T, EPI, ACT>), dim3(grid), dim3(256), GEMM_LDS, s, ga);
    IVR_LAUNCH_CHECK();
    return IVR_OK;
}

template <typename T>
int launch_gemm_e(int epi, const GemmArgs &g, hipStream_t s) {
    switch (epi) {
        case EPI_STORE:
            if (g.act == IVR_ACT_QUICK_GELU) return launch_gemm_t<T, EPI_STORE, IVR_ACT_QUICK_GELU>(g, s);
            if (g.act == IVR_ACT_GELU_ERF) return launch_gemm_t<T, EPI_STORE, IVR_ACT_GELU_ERF>(g, s);
            return launch_gemm_t<T, EPI_STORE, -1>(g, s);
        case EPI_RESID: return launch_gemm_t<T, EPI_RESID, -1>(g, s);
        case EPI_PATCH: return launch_gemm_t<T, EPI_PATCH, -1>(g, s);
        default: return launch_gemm_t<T, EPI_F32, -1>(g, s);
    }
}

}  // namespace

// The kernels address the row operand through one buffer resource with 32-bit offsets (2 GiB reach).  Row panels are independent,
// so a taller operand is run as consecutive slabs of whole 256-row tiles (EPI_PATCH: of whole 256-image groups, its epilogue maps
// rows to images), each with the base pointers advanced: ViT-B/32 from ~6.8 k frames per call, ViT-L/14 from ~1 k.
static int64_t gemm_slab_rows(int64_t row_bytes, int64_t quantum) {
    const int64_t most = (int64_t)0x7ffffff0 / std::max<int64_t>(row_bytes, 1);
    return std::max<int64_t>(quantum, most / quantum * quantum);
}

template <typename Launch>
static int gemm_in_slabs(const GemmArgs &g, int epi, int64_t in_elem, int64_t out_elem, Launch launch) {
    int64_t quantum = epi == EPI_PATCH ? (int64_t)256 * g.G2 : 256;
    if (g.skip_mod) {                  // rows r % skip_mod == 0 are skipped by the epilogue: slabs must start on such a row
        int64_t a = quantum, b = g.skip_mod;
        while (b) {
            const int64_t t = a % b;
            a = b;
            b = t;
        }
        quantum = quantum / a * g.skip_mod;
    }
    const int64_t slab = gemm_slab_rows((int64_t)g.lda * in_elem, quantum);
    IVR_REQUIRE(slab * g.lda * in_elem < 0x7fffffff, "gemm: a single slab of %lld rows x lda=%d exceeds the 2 GiB reach of the buffer offsets",
                (long long)slab, g.lda);
    for (int64_t m0 = 0; m0 < g.M; m0 += slab) {
        GemmArgs ga = g;
        ga.M = (int)std::min<int64_t>(slab, g.M - m0);
        ga.A = reinterpret_cast<const char *>(g.A) + m0 * g.lda * in_elem;
        if (epi == EPI_PATCH) {
            if (g.resid) ga.resid = g.resid + (m0 / g.G2) * (int64_t)g.T * g.ldr;      // m0 is a whole number of images
        } else {
            if (g.out) ga.out = reinterpret_cast<char *>(g.out) + m0 * g.ldo * out_elem;
            if (g.resid) ga.resid = g.resid + m0 * g.ldr;
        }
        if (int rc = launch(ga)) return rc;
    }
    return IVR_OK;
}

int ivr_launch_gemm(bool f32, int epi, const GemmArgs &g, hipStream_t s) {
    if (g.M <= 0 || g.N <= 0) return IVR_OK;
    const int epr = f32 ? 32 : 64;
    IVR_REQUIRE(g.K > 0 && g.K % epr == 0, "gemm: K=%d must be a multiple of %d", g.K, epr);
    IVR_REQUIRE(g.N % 4 == 0, "gemm: N=%d must be a multiple of 4", g.N);
    const int64_t es = f32 ? 4 : 2;
    IVR_REQUIRE((int64_t)g.N * g.ldw * es < 0x7fffffff, "gemm: the weight operand exceeds the 2 GiB reach of the 32-bit buffer offsets (N=%d ldw=%d)",
                g.N, g.ldw);
    return gemm_in_slabs(g, epi, es, epi == EPI_F32 ? 4 : es, [&](const GemmArgs &ga) {
        return f32 ? launch_gemm_e<float>(epi, ga, s) : launch_gemm_e<unsigned short>(epi, ga, s);
    });
}

namespace {
template <int EPI, int ACT, bool OUT8>
int launch_gemm8_t(const GemmArgs &g, hipStream_t s) {
    IvrProf prof(g.tag ? g.tag : "gemm_fp8", s, 2.0 * g.M * g.N * g.K);
    if (int rc = ivr_func_max_lds(reinterpret_cast<const void *>(gemm_big8_kernel<EPI, ACT, OUT8>), DEEP_LDS)) return rc;
    const int group_env = std::max(0, env_int("IVR_GEMM_GROUP_M", 0));
    GemmArgs ga = g;
    const int MT = (g.M + LBM - 1) / LBM, NT = (g.N + LBN - 1) / LBN;
    ga.group_m = group_env ? group_env : (NT <= 3 ? 2 : 4);
    ga.wide_epi = 1;
    if (EPI == EPI_RESID && g.skip_mod) {
        constexpr bool kSkip = EPI == EPI_RESID;
        if (int rc = ivr_func_max_lds(reinterpret_cast<const void *>(gemm_big8_kernel<EPI, ACT, OUT8, kSkip>), DEEP_LDS)) return rc;
        hipLaunchKernelGGL((gemm_big8_kernel<EPI, ACT, OUT8, kSkip>), dim3(8 * ((MT + 7) / 8) * NT), dim3(512), DEEP_LDS, s, ga);
    } else {
        hipLaunchKernelGGL((gemm_big8_kernel<EPI, ACT, OUT8>), dim3(8 * ((MT + 7) / 8) * NT), dim3(512), DEEP_LDS, s, ga);
    }
    IVR_LAUNCH_CHECK();
    return IVR_OK;
}
}  // namespace

int ivr_launch_gemm_fp8(int epi, const GemmArgs &g, hipStream_t s) {
    if (g.M <= 0 || g.N <= 0) return IVR_OK;
    IVR_REQUIRE(epi == EPI_STORE || epi == EPI_RESID, "fp8 gemm: epilogue %d", epi);
    IVR_REQUIRE(g.K >= 128 && g.K % 128 == 0 && g.lda % 16 == 0 && g.ldw % 16 == 0, "fp8 gemm: K=%d lda=%d ldw=%d (K %% 128, ld %% 16)", g.K,
                g.lda, g.ldw);
    IVR_REQUIRE(g.N % 64 == 0, "fp8 gemm: N=%d must be a multiple of 64", g.N);
    IVR_REQUIRE((int64_t)g.N * g.ldw < 0x7fffffff, "fp8 gemm: weight operand beyond 2 GiB");
    if ((int64_t)g.M * g.lda >= 0x7fffffff) {        // tall row operand: whole 256-row slabs, see gemm_in_slabs
        return gemm_in_slabs(g, epi, 1, g.out8 ? 1 : 2, [&](const GemmArgs &ga) { return ivr_launch_gemm_fp8(epi, ga, s); });
    }
    const auto al16 = [](const void *p) { return reinterpret_cast<uintptr_t>(p) % 16 == 0; };
    IVR_REQUIRE(al16(g.A) && al16(g.W) && al16(g.bias) && al16(g.colscale), "fp8 gemm: operands must be 16-byte aligned");
    if (epi == EPI_RESID) {
        IVR_REQUIRE(g.resid && g.ldr % 4 == 0 && al16(g.resid), "fp8 gemm: residual [M,%d] must be 16-byte aligned rows", g.ldr);
        return launch_gemm8_t<EPI_RESID, -1, false>(g, s);
    }
    IVR_REQUIRE(g.out && al16(g.out) && g.ldo % (g.out8 ? 16 : 8) == 0, "fp8 gemm: output rows must be 16-byte aligned (ldo=%d)", g.ldo);
    if (g.out8) {
        if (g.act == IVR_ACT_QUICK_GELU) return launch_gemm8_t<EPI_STORE, IVR_ACT_QUICK_GELU, true>(g, s);
        if (g.act == IVR_ACT_GELU_ERF) return launch_gemm8_t<EPI_STORE, IVR_ACT_GELU_ERF, true>(g, s);
        return launch_gemm8_t<EPI_STORE, -1, true>(g, s);
    }
    if (g.act == IVR_ACT_QUICK_GELU) return launch_gemm8_t<EPI_STORE, IVR_ACT_QUICK_GELU, false>(g, s);
    if (g.act == IVR_ACT_GELU_ERF) return launch_gemm8_t<EPI_STORE, IVR_ACT_GELU_ERF, false>(g, s);
    return launch_gemm8_t<EPI_STORE, -1, false>(g, s);
}

bool ivr_fused_qkv_attention_ok(int M, int T, int D, int heads, int causal) {
    // whole images per 256-row tile, at least three K stages, one 64-wide head per column block; large problems only by default
    // (IVR_FUSED_QKV=1 forces it wherever it is valid, 0 switches it off)
    if (T < 1 || T > 64 || causal || D % 64 != 0 || D < 192 || heads * 64 != D || M % T != 0) return false;
    if ((int64_t)M * D * 2 >= 0x7fffffff || (int64_t)3 * D * D * 2 >= 0x7fffffff) return false;
    const int mode = env_int("IVR_FUSED_QKV", -1);
    if (mode == 0) return false;
    if (mode == 1) return true;
    const int RT = (256 / T) * T;
    return (int64_t)((M + RT - 1) / RT) * heads >= 256;       // fills the chip (one workgroup per CU)
}

int ivr_launch_qkv_attention(const void *xn, const void *w, const float *bias, void *att, int n, int T, int D, int heads, bool out_fp8,
                             hipStream_t s, int reverse) {
    if (n <= 0) return IVR_OK;
    const int M = n * T;
    IVR_REQUIRE(ivr_fused_qkv_attention_ok(M, T, D, heads, 0) || env_int("IVR_FUSED_QKV", -1) == 1, "fused qkv+attention: unsupported shape");
    const auto al16 = [](const void *p) { return reinterpret_cast<uintptr_t>(p) % 16 == 0; };
    IVR_REQUIRE(al16(xn) && al16(w) && al16(bias) && al16(att), "fused qkv+attention: operands must be 16-byte aligned");
    QkvAttnArgs g;
    g.X = xn;
    g.W = w;
    g.bias = bias;
    g.att = att;
    g.M = M;
    g.D = D;
    g.T = T;
    g.heads = heads;
    g.G = 256 / T;
    g.group_m = std::max(1, env_int("IVR_QKV_GROUP_M", 4));
    g.reverse = reverse;
    const int RT = g.G * T, MT = (M + RT - 1) / RT;
    const int grid = 8 * ((MT + 7) / 8) * heads;
    // FLOP: the projection (2 M 3D D) and the attention products (4 T^2 64 per image and head)
    IvrProf prof("gemm_qkv_attention", s, 2.0 * M * 3.0 * D * D + 4.0 * n * heads * (double)T * T * 64);
    const int qpers = env_int("IVR_QKV_PERS", 1);           // 0: never, 2: always (tests), default: from two tiles per CU on
    if (qpers == 2 || (qpers == 1 && (int64_t)MT * heads >= 2 * device_cu_count())) {
        const int pgrid = device_cu_count() / 8 * 8;
        if (out_fp8) {
            if (int rc = ivr_func_max_lds(reinterpret_cast<const void *>(qkv_attn_pers_kernel<unsigned char>), QP_LDS)) return rc;
            hipLaunchKernelGGL(qkv_attn_pers_kernel<unsigned char>, dim3(pgrid), dim3(512), QP_LDS, s, g);
        } else {
            if (int rc = ivr_func_max_lds(reinterpret_cast<const void *>(qkv_attn_pers_kernel<unsigned short>), QP_LDS)) return rc;
            hipLaunchKernelGGL(qkv_attn_pers_kernel<unsigned short>, dim3(pgrid), dim3(512), QP_LDS, s, g);
        }
        IVR_LAUNCH_CHECK();
        return IVR_OK;
    }
    if (out_fp8) {
        if (int rc = ivr_func_max_lds(reinterpret_cast<const void *>(qkv_attn_kernel<unsigned char>), QA_LDS)) return rc;
        hipLaunchKernelGGL(qkv_attn_kernel<unsigned char>, dim3(grid), dim3(512), QA_LDS, s, g);
    } else {
        if (int rc = ivr_func_max_lds(reinterpret_cast<const void *>(qkv_attn_kernel<unsigned short>), QA_LDS)) return rc;
        hipLaunchKernelGGL(qkv_attn_kernel<unsigned short>, dim3(grid), dim3(512), QA_LDS, s, g);
    }
    IVR_LAUNCH_CHECK();
    return IVR_OK;
}

int ivr_launch_layernorm(int out_kind, const float *x, int row_mul, const int *offs, const float *g, const float *b, float eps,
                         void *out, int rows, int D, hipStream_t s, int reverse) {
    if (rows <= 0) return IVR_OK;
    IVR_REQUIRE(D % 4 == 0 && D <= 2048, "layernorm: D=%d", D);
    const unsigned grid = (unsigned)ivr_ceil_div(rows, 4);
    IvrProf prof("layernorm", s, (double)rows * D * (4 + (out_kind == OUT_F32 ? 4 : out_kind == OUT_FP8 ? 1 : 2)));
    if (out_kind == OUT_F32)
        hipLaunchKernelGGL(layernorm_kernel<float>, dim3(grid), dim3(256), 0, s, x, row_mul, offs, g, b, eps, (float *)out, rows, D, reverse);
    else if (out_kind == OUT_FP8)
        hipLaunchKernelGGL(layernorm_kernel<unsigned char>, dim3(grid), dim3(256), 0, s, x, row_mul, offs, g, b, eps,
                           (unsigned char *)out, rows, D, reverse);
    else
        hipLaunchKernelGGL(layernorm_kernel<unsigned short>, dim3(grid), dim3(256), 0, s, x, row_mul, offs, g, b, eps,
                           (unsigned short *)out, rows, D, reverse);
    IVR_LAUNCH_CHECK();
    return IVR_OK;
}

int ivr_launch_attention(bool f32, const void *qkv, void *att, int n, int T, int D, int heads, int causal, hipStream_t s, bool out_fp8) {
    if (n <= 0) return IVR_OK;
    // FLOP: QK^T and PV, 2*T*T*64 each per (image, head)
    IvrProf prof("attention", s, 4.0 * n * heads * (double)T * T * 64);
    IVR_REQUIRE(!(f32 && out_fp8), "attention: e4m3 output only from the bf16 kernels");
    if (!f32 && T <= 64) {
        const int64_t items = (int64_t)n * heads;
        if (out_fp8)
            hipLaunchKernelGGL(attention_mfma_short_kernel<unsigned char>, dim3((unsigned)ivr_ceil_div(items, 4)), dim3(256), 4 * 8192,
                               s, (const unsigned short *)qkv, (unsigned char *)att, n, T, D, heads, causal);
        else
            hipLaunchKernelGGL(attention_mfma_short_kernel<unsigned short>, dim3((unsigned)ivr_ceil_div(items, 4)), dim3(256), 4 * 8192,
                               s, (const unsigned short *)qkv, (unsigned short *)att, n, T, D, heads, causal);
        IVR_LAUNCH_CHECK();
        return IVR_OK;
    }
    if (!f32) {
        const int Tp = (int)ivr_round_up(T, 32);
        const int head_env = env_int("IVR_ATTN_HEAD", 1);     // 0: force the generic flash kernel (A/B, tests)
        if (head_env && 2 * Tp * 128 <= 160 * 1024 && (int64_t)T * 3 * D * 2 < 0x7fffffff) {
            // head-resident kernel: K and V of a head in LDS, QC query tiles of 16 per wave, at most 6 waves per workgroup
            const int ntiles = (T + 15) / 16;
            auto go = [&](auto qc_c, auto minw_c) -> int {
                constexpr int QC = decltype(qc_c)::value, MINW = decltype(minw_c)::value;
                const int nsplit = ivr_ceil_div(ntiles, 6 * QC);
                const int tps = ivr_ceil_div(ntiles, nsplit), nw = ivr_ceil_div(tps, QC);
                const int lds = 2 * Tp * 128;
                if (int rc = out_fp8 ? ivr_func_max_lds(reinterpret_cast<const void *>(attention_head_kernel<QC, unsigned char, MINW>), lds)
                                     : ivr_func_max_lds(reinterpret_cast<const void *>(attention_head_kernel<QC, unsigned short, MINW>), lds))
                    return rc;
                const dim3 grid((unsigned)((int64_t)n * heads * nsplit)), block(64 * nw);
                if (out_fp8)
                    hipLaunchKernelGGL((attention_head_kernel<QC, unsigned char, MINW>), grid, block, lds, s, (const unsigned short *)qkv,
                                       (unsigned char *)att, T, D, heads, causal, nsplit, Tp);
                else
                    hipLaunchKernelGGL((attention_head_kernel<QC, unsigned short, MINW>), grid, block, lds, s, (const unsigned short *)qkv,
                                       (unsigned short *)att, T, D, heads, causal, nsplit, Tp);
                IVR_LAUNCH_CHECK();
                return IVR_OK;
            };
            auto run = [&](int qc) -> int {
                return qc == 5   ? go(std::integral_constant<int, 5>{}, std::integral_constant<int, 2>{})
                       : qc == 4 ? go(std::integral_constant<int, 4>{}, std::integral_constant<int, 2>{})
                                 : go(std::integral_constant<int, 3>{}, std::integral_constant<int, 3>{});
            };
            // Three, four or five query tiles per wave: which is faster depends on how the tile count splits into waves and SIMDs
            // (tools/bench_attention.py: 13 tiles 1.6x in favour of four, 17 tiles 1.2x in favour of three), so the first
            // call with a given shape times both once (the kernel is idempotent) and keeps the winner.  Never while the stream
            // is being captured into a graph: that call runs the default and leaves the choice open.
            const int forced = env_int("IVR_ATTN_QC", 0);
            if (forced >= 3 && forced <= 5) return run(forced);
            static std::mutex tune_mu;
            static std::map<uint64_t, int> tuned;
            const uint64_t key = ((uint64_t)T << 32) | ((uint64_t)heads << 8) | ((uint64_t)(causal != 0) << 1) | (uint64_t)out_fp8;
            {
                std::lock_guard<std::mutex> lk(tune_mu);
                auto it = tuned.find(key);
                if (it != tuned.end()) return run(it->second);
            }
            hipStreamCaptureStatus cap = hipStreamCaptureStatusNone;
            (void)hipStreamIsCapturing(s, &cap);
            if (cap != hipStreamCaptureStatusNone || n * heads < 512) return run(3);       // too small to time meaningfully
            hipEvent_t e[2] = {nullptr, nullptr};
            int rc = IVR_OK, best = 3;
            float best_ms = 0.f;
            // no early return between here and the hipEventDestroy below: every failure is carried in rc
            if (hipEventCreate(&e[0]) != hipSuccess || hipEventCreate(&e[1]) != hipSuccess)
                rc = ivr_fail(IVR_ERR_HIP, "attention autotune: hipEventCreate failed");
            for (int qc = 3; qc <= 5 && rc == IVR_OK; ++qc) {
                rc = run(qc);                                 // warm-up (function attributes, caches)
                if (rc == IVR_OK && hipEventRecord(e[0], s) != hipSuccess) rc = ivr_fail(IVR_ERR_HIP, "attention autotune: hipEventRecord");
                if (rc == IVR_OK) rc = run(qc);
                float ms = 0.f;
                if (rc == IVR_OK && (hipEventRecord(e[1], s) != hipSuccess || hipEventSynchronize(e[1]) != hipSuccess ||
                                     hipEventElapsedTime(&ms, e[0], e[1]) != hipSuccess))
                    rc = ivr_fail(IVR_ERR_HIP, "attention autotune: event timing failed");
                if (rc == IVR_OK && (qc == 3 || ms < best_ms)) {
                    best = qc;
                    best_ms = ms;
                }
            }
            for (auto &ev : e)
                if (ev) (void)hipEventDestroy(ev);
            if (rc != IVR_OK) return rc;
            {
                std::lock_guard<std::mutex> lk(tune_mu);
                tuned[key] = best;
            }
            return run(best);                                 // the buffer holds the last candidate's (identical) result anyway
        }
        IVR_REQUIRE(!out_fp8, "attention: T=%d too long for the e4m3-output kernels", T);
        const int nqb = (T + 63) / 64;
        const int64_t items = (int64_t)n * heads * nqb;
        hipLaunchKernelGGL(attention_mfma_kernel, dim3((unsigned)ivr_ceil_div(items, 4)), dim3(256), 4 * VT_BYTES, s,
                           (const unsigned short *)qkv, (unsigned short *)att, n, T, D, heads, causal);
        IVR_LAUNCH_CHECK();
        return IVR_OK;
    }
    // float32 verification / query mode: KS lanes per query row, K/V in LDS, VALU dot products
    IVR_REQUIRE(T <= 512, "attention: T=%d too long for the float32 kernel", T);
    const size_t lds = (size_t)2 * T * AF_STRIDE * 4;
    IVR_REQUIRE(lds <= 160 * 1024, "attention: T=%d needs %zu bytes of LDS", T, lds);
    const int ks = T * 8 <= 512 ? 8 : (T * 4 <= 512 ? 4 : (T * 2 <= 512 ? 2 : 1));          // 192 registers of state per lane: at most 512 lanes
    const int threads = (int)ivr_round_up((int64_t)T * ks, 64);
    auto go = [&](auto ks_c) -> int {
        constexpr int KS = decltype(ks_c)::value;
        if (int rc = ivr_func_max_lds(reinterpret_cast<const void *>(attention_f32_kernel<KS>), (int)lds)) return rc;
        hipLaunchKernelGGL(attention_f32_kernel<KS>, dim3(heads, n), dim3(threads), lds, s, (const float *)qkv, (float *)att, T, D, causal);
        IVR_LAUNCH_CHECK();
        return IVR_OK;
    };
    return ks == 8 ? go(std::integral_constant<int, 8>{}) : ks == 4 ? go(std::integral_constant<int, 4>{})
           : ks == 2 ? go(std::integral_constant<int, 2>{}) : go(std::integral_constant<int, 1>{});
}

int ivr_launch_vision_cls(float *resid, const float *cls, const float *pos, int n, int T, int D, hipStream_t s) {
    if (n <= 0) return IVR_OK;
    hipLaunchKernelGGL(vision_cls_kernel, dim3(n), dim3(256), 0, s, resid, cls, pos, T, D);
    IVR_LAUNCH_CHECK();
    return IVR_OK;
}

int ivr_launch_text_embed(float *resid, const int64_t *ids, const float *tok, const float *pos, int q, int T, int D, int vocab,
                          int eos, int *eos_pos, hipStream_t s) {
    if (q <= 0) return IVR_OK;
    hipLaunchKernelGGL(text_embed_kernel, dim3(q * T), dim3(256), 0, s, resid, ids, tok, pos, T, D, vocab);
    IVR_LAUNCH_CHECK();
    hipLaunchKernelGGL(eos_pos_kernel, dim3((unsigned)ivr_ceil_div(q, 256)), dim3(256), 0, s, ids, q, T, eos, eos_pos);
    IVR_LAUNCH_CHECK();
    return IVR_OK;
}

int ivr_launch_bf16_to_e4m3(const void *src, void *dst, int64_t count, hipStream_t s) {
    if (count <= 0) return IVR_OK;
    IVR_REQUIRE(count % 8 == 0, "bf16_to_e4m3: count=%lld must be a multiple of 8", (long long)count);
    IvrProf prof("bf16_to_e4m3", s, (double)count * 3);
    hipLaunchKernelGGL(bf16_to_e4m3_kernel, dim3((unsigned)ivr_ceil_div(count / 8, 256)), dim3(256), 0, s, (const uint4 *)src, (uint2 *)dst,
                       count / 8);
    IVR_LAUNCH_CHECK();
    return IVR_OK;
}

int ivr_launch_f_normalize(const float *x, float *out, int n, int d, int normalize, hipStream_t s) {
    if (n <= 0) return IVR_OK;
    hipLaunchKernelGGL(f_normalize_kernel, dim3((unsigned)ivr_ceil_div(n, 4)), dim3(256), 0, s, x, out, n, d, normalize);
    IVR_LAUNCH_CHECK();
    return IVR_OK;
}

#ifdef IVR_GEMM_STAMPS
extern "C" int ivr_debug_gemm_stamps(unsigned long long *host_out, int nblocks) {
    return hipMemcpyFromSymbol(host_out, HIP_SYMBOL(ivr_gemm_stamps), sizeof(unsigned long long) * 8 * nblocks) == hipSuccess ? 0 : -2;
}
#endif
