// Flat inner-product index on MI355X: tiled HBM layout, streaming scan with f32 MFMA,
// exact top-k by group maxima + radix select.  (I1, N2/N3, S1 of SURVEY.md section 8a.)
//
// Replaces faiss.IndexFlatIP.add/search as called at unified_index.py:1767-1779,503 and
// core.py:827,891.  Design:
//
//   layout   rows are stored in tiles of 16 rows; inside a tile the order is [d/4][16 rows][4 floats],
//            so lane l of a wave reading float4 number l of a 1 KiB piece holds row (l & 15),
//            floats 16*kc + 4*(l >> 4) .. +3: exactly the A fragment of v_mfma_f32_16x16x4_f32 for four
//            consecutive MFMAs.  Every wave-wide load is 1 KiB contiguous.  Queries use the same layout
//            (they are the B fragment), staged once per workgroup in LDS.
//   pass 1   scan: each wave scores a group of 64 rows x (16*QT) queries, reduces the 64 scores of each
//            query to their maximum (15 v_max + 2 wave shuffles) and stores it.  No data-dependent
//            control flow, the index is read exactly once per 16*QT queries.
//   pass 2   per query, radix-select the k groups with the largest (max, lowest group id).  The k best
//            rows always lie inside those k groups (proof in DESIGN.md).
//   pass 3   re-score the selected groups with the same MFMA sequence (bit-identical scores) and emit
//            64-bit keys (ordered score, ~row).
//   pass 4   per query, radix-select + bitonic sort of the k best keys -> D (float32), I (int64).
#include "ivr_common.h"
#include "search_internal.h"

#include <algorithm>
#include <cfloat>

namespace {

constexpr int kGroupRows = 64;      // rows per scan group (4 MFMA row tiles)
constexpr int kSelThreads = 1024;   // select kernel block size
constexpr int kMaxSort = IVR_MAX_K; // bitonic sort capacity (power of two)

// ---------------------------------------------------------------------------------------------
// tiling: row-major [n,d] -> tiled, with optional L2 normalisation (N2/N3) and non-finite count
// ---------------------------------------------------------------------------------------------
__device__ __forceinline__ float4 load_quad(const float *__restrict__ row, int k0, int d, bool vec) {
    if (vec) return *reinterpret_cast<const float4 *>(row + k0);   // k0 + 3 < d guaranteed by caller when vec
    float4 v;
    v.x = k0 + 0 < d ? row[k0 + 0] : 0.f;
    v.y = k0 + 1 < d ? row[k0 + 1] : 0.f;
    v.z = k0 + 2 < d ? row[k0 + 2] : 0.f;
    v.w = k0 + 3 < d ? row[k0 + 3] : 0.f;
    return v;
}

// one wave per 16-row tile; lane l owns row (l & 15) and quad (l >> 4) of every 16-float chunk.
// Optionally also writes the bf16 scan copy of the tile (dst16; and the rounding remainder dst16lo for query tiles):
// per 32 floats of K one 1 KiB piece, lane l -> 16 bytes = the two quads this lane owns in the pair of 16-float chunks, i.e.
// the operand of one v_mfma_f32_16x16x32_bf16 with K permuted the same way for index rows and queries.
__global__ __launch_bounds__(256) void tile_rows_kernel(const float *__restrict__ src, float *__restrict__ dst,
                                                        int64_t row_start, int64_t n, int d, int dp4,
                                                        int normalize, int32_t *__restrict__ nonfinite,
                                                        const int64_t *__restrict__ start_dev, uint4 *__restrict__ dst16,
                                                        uint4 *__restrict__ dst16lo, unsigned int *__restrict__ maxnorm_bits,
                                                        float *__restrict__ rownorm, int zero_fill, int pstride,
                                                        unsigned int *__restrict__ maxdelta_bits, float *__restrict__ rowdelta) {
    if (start_dev) row_start = *start_dev;          // ring cursor kept in HBM so a captured graph can replay it
    const int lane = threadIdx.x & 63;
    const int64_t tile0 = row_start >> 4;
    const int64_t ntiles = ((row_start + n + 15) >> 4) - tile0;
    const int64_t t = (int64_t)blockIdx.x * 4 + (threadIdx.x >> 6);
    if (t >= ntiles) return;
    const int64_t tile = tile0 + t;
    const int rr = lane & 15, qd = lane >> 4;
    const int64_t row = tile * 16 + rr;
    const bool valid = row >= row_start && row < row_start + n;
    const float *srow = src + (valid ? (row - row_start) : 0) * (int64_t)d;
    const bool vec = (d & 3) == 0 && ((reinterpret_cast<uintptr_t>(src) & 15) == 0);
    const int kchunks = dp4 >> 2;
    float ss = 0.f;
    int bad = 0;
    if (normalize || nonfinite || maxnorm_bits || rownorm) {
        // eight chunks per trip, all loads issued before the first use: a query batch is a handful of rows, so this kernel is
        // a latency chain (32 dependent trips of ~0.4 us at d = 512 before); the sum keeps its ascending-k order
        for (int kc0 = 0; kc0 < kchunks; kc0 += 8) {
            float4 v[8];
#pragma unroll
            for (int u = 0; u < 8; ++u) {
                const int k0 = (kc0 + u) * 16 + qd * 4;
                v[u] = make_float4(0.f, 0.f, 0.f, 0.f);
                if (valid && kc0 + u < kchunks && k0 < d) v[u] = load_quad(srow, k0, d, vec && k0 + 3 < d);
            }
#pragma unroll
            for (int u = 0; u < 8; ++u) {
                ss = fmaf(v[u].x, v[u].x, ss);
                ss = fmaf(v[u].y, v[u].y, ss);
                ss = fmaf(v[u].z, v[u].z, ss);
                ss = fmaf(v[u].w, v[u].w, ss);
                bad += !isfinite(v[u].x) + !isfinite(v[u].y) + !isfinite(v[u].z) + !isfinite(v[u].w);
            }
        }
        ss += __shfl_xor(ss, 16, 64);
        ss += __shfl_xor(ss, 32, 64);
        if (nonfinite) {
            bad += __shfl_xor(bad, 16, 64);
            bad += __shfl_xor(bad, 32, 64);
            if (bad && qd == 0) atomicAdd(nonfinite, bad);
        }
    }
    // core.py:1194-1196: norms[norms == 0] = 1; features / norms
    const float nrm = normalize ? (ss > 0.f ? sqrtf(ss) : 1.f) : 1.f;
    if (maxnorm_bits) {
        // largest stored row norm (an upper bound: overwritten rows keep counting), for the error bound of the bf16 scan;
        // a normalised row is 1 up to rounding, the 1e-6 covers it.  Positive floats order like their bit patterns.
        float stored = valid ? (normalize ? (ss > 0.f ? 1.000001f : 0.f) : sqrtf(ss)) : 0.f;
        if (!(stored == stored)) stored = INFINITY;      // NaN rows: no bound -> the verification always fails over to the exact path
#pragma unroll
        for (int o = 1; o < 16; o <<= 1) stored = fmaxf(stored, __shfl_xor(stored, o, 64));
        const unsigned int bits = __float_as_uint(stored);
        if (lane == 0 && bits > *maxnorm_bits) atomicMax(maxnorm_bits, bits);
    }
    // query tiles: an upper bound of the stored row's norm for the error bound of the bf16 candidate scan (1 up to rounding once
    // normalised, as for the index rows above)
    if (rownorm && valid && qd == 0) rownorm[row - row_start] = normalize ? (ss > 0.f ? 1.000001f : 0.f) : sqrtf(ss);
    float4 *out = reinterpret_cast<float4 *>(dst) + tile * (int64_t)dp4 * 16 + lane;
    const int pieces = (kchunks + 1) >> 1;          // pieces that carry data; the tile's stride is pstride (even, see ivr_index_create)
    const bool store = valid || zero_fill;          // query tiles: the padding rows of the last tile are written as zeros
    float sd = 0.f;                                 // squared norm of this lane's part of  row - bf16(row)
    for (int kb0 = 0; kb0 < pieces; kb0 += 4) {
      float4 vv[4][2];
#pragma unroll
      for (int b4 = 0; b4 < 4; ++b4)
#pragma unroll
        for (int u = 0; u < 2; ++u) {
            const int kc = 2 * (kb0 + b4) + u, k0 = kc * 16 + qd * 4;
            vv[b4][u] = make_float4(0.f, 0.f, 0.f, 0.f);
            if (valid && kc < kchunks && k0 < d) vv[b4][u] = load_quad(srow, k0, d, vec && k0 + 3 < d);
        }
#pragma unroll
      for (int b4 = 0; b4 < 4; ++b4) {
        const int kb = kb0 + b4;
        if (kb >= pieces) break;
        float4 (&v)[2] = vv[b4];
#pragma unroll
        for (int u = 0; u < 2; ++u) {
            const int kc = 2 * kb + u;
            if (store && kc < kchunks) {
                if (valid && normalize) {
                    v[u].x /= nrm;
                    v[u].y /= nrm;
                    v[u].z /= nrm;
                    v[u].w /= nrm;
                }
                out[kc * 64] = v[u];
            }
        }
        if (dst16 && store) {
            uint4 hi;
            hi.x = ivr_pack_bf16x2(v[0].x, v[0].y);
            hi.y = ivr_pack_bf16x2(v[0].z, v[0].w);
            hi.z = ivr_pack_bf16x2(v[1].x, v[1].y);
            hi.w = ivr_pack_bf16x2(v[1].z, v[1].w);
            dst16[(tile * pstride + kb) * 64 + lane] = hi;
            // what rounding to bf16 dropped: exact in float32 (the difference of a float and its own leading bits)
            auto lo2 = [&sd](uint32_t h, float a, float b) {
                const float ra = a - __uint_as_float(h << 16), rb = b - __uint_as_float(h & 0xffff0000u);
                sd = fmaf(ra, ra, sd);
                sd = fmaf(rb, rb, sd);
                return ivr_pack_bf16x2(ra, rb);
            };
            uint4 lo;
            lo.x = lo2(hi.x, v[0].x, v[0].y);
            lo.y = lo2(hi.y, v[0].z, v[0].w);
            lo.z = lo2(hi.z, v[1].x, v[1].y);
            lo.w = lo2(hi.w, v[1].z, v[1].w);
            if (dst16lo) dst16lo[(tile * pstride + kb) * 64 + lane] = lo;
        }
      }
    }
    // |row - bf16(row)| per stored row, for the error bound of the large-batch candidate scan (both operands rounded to nearest):
    // the largest over the index rows, one value per query
    if (dst16 && (maxdelta_bits || rowdelta)) {
        sd += __shfl_xor(sd, 16, 64);
        sd += __shfl_xor(sd, 32, 64);
        float dl = valid ? sqrtf(sd) * 1.0001f : 0.f;
        if (!(dl == dl)) dl = INFINITY;             // NaN rows: no bound, every verification fails over to the exact path
        if (rowdelta && valid && qd == 0) rowdelta[row - row_start] = dl;
        if (maxdelta_bits) {
#pragma unroll
            for (int o = 1; o < 16; o <<= 1) dl = fmaxf(dl, __shfl_xor(dl, o, 64));
            const unsigned int bits = __float_as_uint(dl);
            if (lane == 0 && bits > *maxdelta_bits) atomicMax(maxdelta_bits, bits);
        }
    }
}

__global__ __launch_bounds__(256) void untile_rows_kernel(const float *__restrict__ src, float *__restrict__ dst,
                                                          int64_t row_start, int64_t n, int d, int dp4) {
    const int lane = threadIdx.x & 63;
    const int64_t tile0 = row_start >> 4;
    const int64_t ntiles = ((row_start + n + 15) >> 4) - tile0;
    const int64_t t = (int64_t)blockIdx.x * 4 + (threadIdx.x >> 6);
    if (t >= ntiles) return;
    const int64_t tile = tile0 + t;
    const int rr = lane & 15, qd = lane >> 4;
    const int64_t row = tile * 16 + rr;
    if (row < row_start || row >= row_start + n) return;
    const float4 *in = reinterpret_cast<const float4 *>(src) + tile * (int64_t)dp4 * 16 + lane;
    float *drow = dst + (row - row_start) * (int64_t)d;
    for (int kc = 0; kc < (dp4 >> 2); ++kc) {
        const int k0 = kc * 16 + qd * 4;
        const float4 v = in[kc * 64];
        if (k0 + 0 < d) drow[k0 + 0] = v.x;
        if (k0 + 1 < d) drow[k0 + 1] = v.y;
        if (k0 + 2 < d) drow[k0 + 2] = v.z;
        if (k0 + 3 < d) drow[k0 + 3] = v.w;
    }
}

__global__ void advance_cursor_kernel(int64_t *cursor, int64_t n, int64_t modulo) { *cursor = (*cursor + n) % modulo; }

// in-place row normalisation of a row-major matrix (ivr_l2_normalize): one wave per row
__global__ __launch_bounds__(256) void l2_normalize_kernel(float *__restrict__ x, int64_t n, int d,
                                                           int32_t *__restrict__ nonfinite) {
    const int lane = threadIdx.x & 63;
    const int64_t row = (int64_t)blockIdx.x * 4 + (threadIdx.x >> 6);
    if (row >= n) return;
    float *p = x + row * (int64_t)d;
    float ss = 0.f;
    int bad = 0;
    for (int k = lane; k < d; k += 64) {
        const float v = p[k];
        ss = fmaf(v, v, ss);
        bad += !isfinite(v);
    }
    ss = ivr_wave_sum(ss);
    if (nonfinite) {
        for (int o = 32; o > 0; o >>= 1) bad += __shfl_xor(bad, o, 64);
        if (bad && lane == 0) atomicAdd(nonfinite, bad);
    }
    const float nrm = ss > 0.f ? sqrtf(ss) : 1.f;
    for (int k = lane; k < d; k += 64) p[k] = p[k] / nrm;
}

// ---------------------------------------------------------------------------------------------
// the 64-row x 16-query score tile (shared by pass 1 and pass 3 so the scores are bit-identical)
// ---------------------------------------------------------------------------------------------
// a: this lane's float4 pointer into the group's first tile; tiles are tile_stride float4 apart.
// acc[t][r] = <row 16t + 4(lane>>4) + r , query (lane&15)>
template <int QT, typename BLoad>
__device__ __forceinline__ void score_group(const float4 *__restrict__ a, int64_t tile_stride, int kchunks,
                                            BLoad bload, f32x4 (&acc)[QT][4]) {
#pragma unroll
    for (int q = 0; q < QT; ++q)
#pragma unroll
        for (int t = 0; t < 4; ++t) acc[q][t] = f32x4{0.f, 0.f, 0.f, 0.f};
    // one row tile at a time, eight 16-float chunks per trip: 8 independent 1 KiB loads = 8 KiB CONTIGUOUS in flight per wave
    // before the MFMAs (a tile is [d/4][16 rows][4 floats], so consecutive chunks of one tile are adjacent in memory).
    // Per accumulator the K order is ascending whatever the trip shape, so scan and re-score stay bit-identical.
#pragma unroll
    for (int t = 0; t < 4; ++t) {
        const float4 *at = a + t * tile_stride;
        int kc = 0;
        for (; kc + 8 <= kchunks; kc += 8) {
            float4 av[8];
#pragma unroll
            for (int u = 0; u < 8; ++u) av[u] = at[(kc + u) * 64];
#pragma unroll
            for (int u = 0; u < 8; ++u)
#pragma unroll
                for (int q = 0; q < QT; ++q) {
                    const float4 bv = bload(q, kc + u);
                    acc[q][t] = __builtin_amdgcn_mfma_f32_16x16x4f32(av[u].x, bv.x, acc[q][t], 0, 0, 0);
                    acc[q][t] = __builtin_amdgcn_mfma_f32_16x16x4f32(av[u].y, bv.y, acc[q][t], 0, 0, 0);
                    acc[q][t] = __builtin_amdgcn_mfma_f32_16x16x4f32(av[u].z, bv.z, acc[q][t], 0, 0, 0);
                    acc[q][t] = __builtin_amdgcn_mfma_f32_16x16x4f32(av[u].w, bv.w, acc[q][t], 0, 0, 0);
                }
        }
        for (; kc < kchunks; ++kc) {
            const float4 av = at[kc * 64];
#pragma unroll
            for (int q = 0; q < QT; ++q) {
                const float4 bv = bload(q, kc);
                acc[q][t] = __builtin_amdgcn_mfma_f32_16x16x4f32(av.x, bv.x, acc[q][t], 0, 0, 0);
                acc[q][t] = __builtin_amdgcn_mfma_f32_16x16x4f32(av.y, bv.y, acc[q][t], 0, 0, 0);
                acc[q][t] = __builtin_amdgcn_mfma_f32_16x16x4f32(av.z, bv.z, acc[q][t], 0, 0, 0);
                acc[q][t] = __builtin_amdgcn_mfma_f32_16x16x4f32(av.w, bv.w, acc[q][t], 0, 0, 0);
            }
        }
    }
}

// pass 1.  gmax layout: [16*QT queries][mstride groups]
// the wave loop of the exact scan: every 64-row group against the 16*QT queries staged in qs
template <int QT>
__device__ __forceinline__ void scan_groups_body(const float4 *__restrict__ qs, const float *__restrict__ data, int dp4, int64_t ngroups,
                                                 int64_t ntotal, float *__restrict__ gmax, int64_t mstride) {
    const int per_tile = dp4 * 16;   // float4 per 16-row tile
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6, nw = blockDim.x >> 6;
    const int kchunks = dp4 >> 2;
    auto bload = [&](int q, int kc) { return qs[q * per_tile + kc * 64 + lane]; };
    for (int64_t g = (int64_t)blockIdx.x * nw + wave; g < ngroups; g += (int64_t)gridDim.x * nw) {
        f32x4 acc[QT][4];
        const float4 *a = reinterpret_cast<const float4 *>(data) + g * 4 * (int64_t)per_tile + lane;
        score_group<QT>(a, per_tile, kchunks, bload, acc);
        const bool partial = (g + 1) * kGroupRows > ntotal;   // wave-uniform: only the last group
#pragma unroll
        for (int q = 0; q < QT; ++q) {
            float m = -FLT_MAX;
#pragma unroll
            for (int t = 0; t < 4; ++t)
#pragma unroll
                for (int r = 0; r < 4; ++r) {
                    float s = acc[q][t][r];
                    if (partial && g * kGroupRows + t * 16 + (lane >> 4) * 4 + r >= ntotal) s = -FLT_MAX;
                    m = fmaxf(m, s);
                }
            m = fmaxf(m, __shfl_xor(m, 16, 64));
            m = fmaxf(m, __shfl_xor(m, 32, 64));
            if (lane < 16) gmax[(int64_t)(q * 16 + lane) * mstride + g] = m;
        }
    }
}

template <int QT>
__global__ __launch_bounds__(512) void scan_groupmax_kernel(const float *__restrict__ data,
                                                            const float *__restrict__ qtiled, int dp4,
                                                            int64_t ngroups, int64_t ntotal,
                                                            float *__restrict__ gmax, int64_t mstride,
                                                            const int *__restrict__ tile_flag) {
    extern __shared__ __attribute__((aligned(16))) float4 qs[];
    if (tile_flag) {       // fallback pass behind the bf16 candidate scan: only query tiles that failed their verification
        bool any = false;
#pragma unroll
        for (int i = 0; i < QT; ++i) any |= tile_flag[i] != 0;
        if (!any) return;
    }
    const int per_tile = dp4 * 16;
    for (int i = threadIdx.x; i < QT * per_tile; i += blockDim.x) qs[i] = reinterpret_cast<const float4 *>(qtiled)[i];
    __syncthreads();
    scan_groups_body<QT>(qs, data, dp4, ngroups, ntotal, gmax, mstride);
}

// Exact pass behind the LARGE-batch candidate scan: the queries whose verification failed were appended to `list` by the final
// selection (count = *nlist, known on the device only).  Each chunk of 16*QT listed queries is gathered from the tiled query
// buffer straight into LDS (element (kc, lane) of a staged tile = the float4 of query (lane & 15), quad (lane >> 4), chunk kc)
// and scanned like any other; nothing listed: every workgroup exits at once.  gmax row = position in the list.
template <int QT>
__global__ __launch_bounds__(512) void scan_groupmax_list_kernel(const float *__restrict__ data, const float *__restrict__ qtiled,
                                                                 int dp4, int64_t ngroups, int64_t ntotal, float *__restrict__ gmax,
                                                                 int64_t mstride, const int *__restrict__ nlist,
                                                                 const int *__restrict__ list) {
    extern __shared__ __attribute__((aligned(16))) float4 qs[];
    const int nf = *nlist;
    const int per_tile = dp4 * 16;
    for (int c0 = 0; c0 < nf; c0 += 16 * QT) {
        __syncthreads();                               // the previous chunk's waves are done with qs
        for (int i = threadIdx.x; i < QT * per_tile; i += blockDim.x) {
            const int t = i / per_tile, r = i - t * per_tile, l = r & 63;
            const int pos = c0 + t * 16 + (l & 15);
            float4 v = make_float4(0.f, 0.f, 0.f, 0.f);
            if (pos < nf) {
                const int src = list[pos];
                v = reinterpret_cast<const float4 *>(qtiled)[(int64_t)(src >> 4) * per_tile + (r - l) + (l & 48) + (src & 15)];
            }
            qs[i] = v;
        }
        __syncthreads();
        scan_groups_body<QT>(qs, data, dp4, ngroups, ntotal, gmax + (int64_t)c0 * mstride, mstride);
    }
}

// ---------------------------------------------------------------------------------------------
// bf16 candidate scan (half the index bytes per pass): group maxima of  <bf16(row), q_hi + q_lo>  on v_mfma_f32_16x16x32_bf16.
// The result only RANKS groups; every reported score comes from the exact float32 re-score.  Exactness is kept by a check:
// with e = |approx - exact| <= (2^-8 + 2^-15 + dp 2^-23) |row| |q|  (bf16 rounding of the row, the dropped q remainder, f32
// accumulation), all rows of a group excluded after the kp best approximate maxima score <= (kp+1)-th approximate maximum + e.
// If that is strictly below the k-th exact score found among the kp re-scored groups, no excluded row can enter or tie the
// top k.  Queries that fail the check are redone by the exact float32 scan (same launch sequence, predicated on device).
// ---------------------------------------------------------------------------------------------
template <int QT>
__global__ __launch_bounds__(512) void scan16_groupmax_kernel(const uint4 *__restrict__ data16, const uint4 *__restrict__ qhi,
                                                              const uint4 *__restrict__ qlo, int pieces, int64_t ngroups,
                                                              int64_t ntotal, float *__restrict__ gmax, int64_t mstride) {
    typedef __bf16 bf16x8_t __attribute__((ext_vector_type(8)));
    extern __shared__ __attribute__((aligned(16))) uint4 qs16[];      // [QT][2][pieces][64]
    const int per_q = pieces * 64;
    for (int i = threadIdx.x; i < QT * per_q; i += blockDim.x) {
        const int q = i / per_q, r = i - q * per_q;
        qs16[(q * 2) * per_q + r] = qhi[i];
        qs16[(q * 2 + 1) * per_q + r] = qlo[i];
    }
    __syncthreads();
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6, nw = blockDim.x >> 6;
    for (int64_t g = (int64_t)blockIdx.x * nw + wave; g < ngroups; g += (int64_t)gridDim.x * nw) {
        f32x4 acc[QT][4];
#pragma unroll
        for (int q = 0; q < QT; ++q)
#pragma unroll
            for (int t = 0; t < 4; ++t) acc[q][t] = f32x4{0.f, 0.f, 0.f, 0.f};
        const uint4 *a = data16 + g * 4 * (int64_t)per_q + lane;
        // K outermost: a query fragment pair (hi, lo) is read from LDS once and used for the four row tiles (one LDS read per
        // four MFMAs; tile-outermost it is one per MFMA, which saturates the LDS port from two query tiles on)
        int kb = 0;
        for (; kb + 2 <= pieces; kb += 2) {
            uint4 av[2][4];
#pragma unroll
            for (int u = 0; u < 2; ++u)
#pragma unroll
                for (int t = 0; t < 4; ++t) av[u][t] = a[(t * pieces + kb + u) * 64];
#pragma unroll
            for (int u = 0; u < 2; ++u)
#pragma unroll
                for (int q = 0; q < QT; ++q) {
                    const uint4 bh = qs16[(q * 2) * per_q + (kb + u) * 64 + lane], bl = qs16[(q * 2 + 1) * per_q + (kb + u) * 64 + lane];
#pragma unroll
                    for (int t = 0; t < 4; ++t) {
                        acc[q][t] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(__builtin_bit_cast(bf16x8_t, av[u][t]), __builtin_bit_cast(bf16x8_t, bh), acc[q][t], 0, 0, 0);
                        acc[q][t] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(__builtin_bit_cast(bf16x8_t, av[u][t]), __builtin_bit_cast(bf16x8_t, bl), acc[q][t], 0, 0, 0);
                    }
                }
        }
        for (; kb < pieces; ++kb) {
#pragma unroll
            for (int t = 0; t < 4; ++t) {
                const uint4 av = a[(t * pieces + kb) * 64];
#pragma unroll
                for (int q = 0; q < QT; ++q) {
                    const uint4 bh = qs16[(q * 2) * per_q + kb * 64 + lane], bl = qs16[(q * 2 + 1) * per_q + kb * 64 + lane];
                    acc[q][t] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(__builtin_bit_cast(bf16x8_t, av), __builtin_bit_cast(bf16x8_t, bh), acc[q][t], 0, 0, 0);
                    acc[q][t] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(__builtin_bit_cast(bf16x8_t, av), __builtin_bit_cast(bf16x8_t, bl), acc[q][t], 0, 0, 0);
                }
            }
        }
        const bool partial = (g + 1) * kGroupRows > ntotal;   // wave-uniform: only the last group
#pragma unroll
        for (int q = 0; q < QT; ++q) {
            float m = -FLT_MAX;
#pragma unroll
            for (int t = 0; t < 4; ++t)
#pragma unroll
                for (int r = 0; r < 4; ++r) {
                    float s = acc[q][t][r];
                    if (partial && g * kGroupRows + t * 16 + (lane >> 4) * 4 + r >= ntotal) s = -FLT_MAX;
                    m = fmaxf(m, s);
                }
            m = fmaxf(m, __shfl_xor(m, 16, 64));
            m = fmaxf(m, __shfl_xor(m, 32, 64));
            if (lane < 16) gmax[(int64_t)(q * 16 + lane) * mstride + g] = m;
        }
    }
}

// The same candidate pass for at most 16 queries (QT = 1: the 10-query search of BASELINE configs[1] and of the streaming step), with
// the index streamed through an LDS-DMA ring instead of through registers.  The register version keeps 8 KiB per wave in flight and
// runs at 5.0 TB/s; HBM wants several times that outstanding.  Here every wave owns a ring of PIECES slots of 1 KiB (= one 16-row tile
// of the bf16 copy): piece kb of the next tile is requested (buffer_load ... lds, no registers, no address arithmetic per lane) as soon
// as piece kb of the current tile has been consumed, so a whole tile per wave = PIECES KiB x 8 waves per workgroup stays in flight.
// The query fragments (hi + lo) live in registers (2 x PIECES x 4 VGPRs), the LDS holds nothing but the rings; a wave reads only what
// it requested itself, so its own counted vmcnt orders every read (no workgroup barrier anywhere in the loop).  MFMA sequence per
// accumulator = scan16_groupmax_kernel<1>'s (hi then lo, ascending K), so the group maxima are bit-identical to it.
template <int PIECES>
__global__ __launch_bounds__(512) void scan16_ring_kernel(const uint4 *__restrict__ data16, const uint4 *__restrict__ qhi,
                                                          const uint4 *__restrict__ qlo, int64_t ngroups, int64_t ntotal,
                                                          float *__restrict__ gmax, int64_t mstride) {
    typedef __bf16 bf16x8_t __attribute__((ext_vector_type(8)));
    typedef uint32_t u32x4_t __attribute__((ext_vector_type(4)));
    extern __shared__ __attribute__((aligned(16))) unsigned char ring_all[];
    const int lane = threadIdx.x & 63;
    const int wave = __builtin_amdgcn_readfirstlane((int)(threadIdx.x >> 6));
    const int nw = blockDim.x >> 6;
    unsigned char *ring = ring_all + wave * PIECES * 1024;
    const unsigned ring_lds = (unsigned)(size_t)ring + lane * 16;
    u32x4_t bh[PIECES], bl[PIECES];
#pragma unroll
    for (int kb = 0; kb < PIECES; ++kb) {
        const uint4 h = qhi[kb * 64 + lane], l = qlo[kb * 64 + lane];
        bh[kb] = u32x4_t{h.x, h.y, h.z, h.w};
        bl[kb] = u32x4_t{l.x, l.y, l.z, l.w};
    }
    const int64_t g0 = (int64_t)blockIdx.x * nw + wave, gstep = (int64_t)gridDim.x * nw;
    if (g0 >= ngroups) return;
    constexpr unsigned kGroupBytes = 4u * PIECES * 1024u;
    const unsigned voff = (unsigned)lane * 16u;
    // one buffer descriptor per 64-row group (64-bit base, 32-bit offsets inside the group); no lambda here: a lambda returning the
    // descriptor type inside a kernel TEMPLATE makes hipcc drop the host-side instantiation silently (undefined kernel stub at load)
#define IVR_GROUP_RSRC(G)                                                                                                      \
    __builtin_amdgcn_make_buffer_rsrc(const_cast<char *>(reinterpret_cast<const char *>(data16)) + (G) * (int64_t)kGroupBytes, 0, \
                                      (int)kGroupBytes, 0x00020000)
    // prologue: tile 0 of the first group
    {
        const auto rs0 = IVR_GROUP_RSRC(g0);
#pragma unroll
        for (int kb = 0; kb < PIECES; ++kb)
            __builtin_amdgcn_raw_ptr_buffer_load_lds(rs0, (__attribute__((address_space(3))) void *)(ring + kb * 1024), 16, voff, (unsigned)kb * 1024u, 0,
                                                     0);
    }
    for (int64_t g = g0; g < ngroups; g += gstep) {
        const bool more = g + gstep < ngroups;
        const auto rs = IVR_GROUP_RSRC(g);
        const auto rs_next = IVR_GROUP_RSRC(more ? g + gstep : g);
        f32x4 acc[4];
#pragma unroll
        for (int t = 0; t < 4; ++t) {
            acc[t] = f32x4{0.f, 0.f, 0.f, 0.f};
#pragma unroll
            for (int kb = 0; kb < PIECES; ++kb) {
                // the oldest request in flight is this slot's.  In the steady state PIECES are outstanding (slots kb .. of this tile, slots
                // .. kb-1 of the next), so all but the PIECES - 1 youngest must have landed; in the wave's very last tile nothing is
                // requested any more and the count falls: that tile is waited for as a whole.  (The one gmax store per group counts in
                // vmcnt too; it can only lengthen these waits: loads retire in order among themselves.)
                if (t == 3 && !more) {
                    if (kb == 0) asm volatile("s_waitcnt vmcnt(0)" ::: "memory");      // the whole last tile, once
                } else {
                    asm volatile("s_waitcnt vmcnt(%0)" ::"n"(PIECES - 1) : "memory");
                }
                u32x4_t a;
                asm volatile("ds_read_b128 %0, %1\n\ts_waitcnt lgkmcnt(0)" : "=&v"(a) : "v"(ring_lds + (unsigned)kb * 1024u) : "memory");
                // the slot is free again: request the same piece of the next tile (of this group, or of the wave's next group)
                if (t < 3)
                    __builtin_amdgcn_raw_ptr_buffer_load_lds(rs, (__attribute__((address_space(3))) void *)(ring + kb * 1024), 16, voff,
                                                             (unsigned)((t + 1) * PIECES + kb) * 1024u, 0, 0);
                else if (more)
                    __builtin_amdgcn_raw_ptr_buffer_load_lds(rs_next, (__attribute__((address_space(3))) void *)(ring + kb * 1024), 16, voff,
                                                             (unsigned)kb * 1024u, 0, 0);
                acc[t] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(__builtin_bit_cast(bf16x8_t, a), __builtin_bit_cast(bf16x8_t, bh[kb]), acc[t], 0, 0, 0);
                acc[t] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(__builtin_bit_cast(bf16x8_t, a), __builtin_bit_cast(bf16x8_t, bl[kb]), acc[t], 0, 0, 0);
            }
        }
        const bool partial = (g + 1) * kGroupRows > ntotal;   // wave-uniform: only the last group
        float m = -FLT_MAX;
#pragma unroll
        for (int t = 0; t < 4; ++t)
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                float sc = acc[t][r];
                if (partial && g * kGroupRows + t * 16 + (lane >> 4) * 4 + r >= ntotal) sc = -FLT_MAX;
                m = fmaxf(m, sc);
            }
        m = fmaxf(m, __shfl_xor(m, 16, 64));
        m = fmaxf(m, __shfl_xor(m, 32, 64));
        if (lane < 16) gmax[(int64_t)lane * mstride + g] = m;
    }
#undef IVR_GROUP_RSRC
}

// Verification of the bf16 candidate scan, done by the final selection of each query (select_topk_kernel<SrcKeys, OUT_DI>): does
// the (kp+1)-th approximate group maximum + error bound stay strictly below the k-th exact score?  ok[q] = 1 keeps the fast
// result; otherwise the query's tile is flagged for the exact pass.  tile_flag[0..3] is reset by the group selection launched
// before (same stream), so the blocks of the final selection only ever raise flags.
struct VerifyArgs {
    const float *gmax = nullptr;          // approximate group maxima [query column][mstride]
    int64_t mstride = 0;
    const uint32_t *sel = nullptr;        // [nq][ksel2]: selected groups, entry kp = the first excluded one
    int ksel2 = 0, kp = 0;
    const float *qnorm = nullptr;         // upper bound of each query's stored norm (tile_rows_kernel)
    float rel_eps = 0.f;
    const unsigned int *maxnorm_bits = nullptr;
    int *ok = nullptr;                    // NULL = no verification in this launch
    int *tile_flag = nullptr;
    // large-batch scan (scanq_kernel: both operands rounded to bf16): the bound uses the measured rounding residuals,
    //   |approx - exact| <= (|q| + |dq|) max|dr| + |dq| max|r| + acc_eps |q| max|r|,   dq = q - bf16(q), dr = row - bf16(row);
    // a failed query is appended to fail_list (its exact pass is list-driven, scan_groupmax_list_kernel)
    const float *qdelta = nullptr;        // non-NULL selects this mode
    const unsigned int *maxdelta_bits = nullptr;
    float acc_eps = 0.f;
    int *fail_count = nullptr, *fail_list = nullptr;
};

// list-driven launches (the exact pass behind the large-batch scan): block b works on list position b and exits when
// b >= *count; results go to output row list[b]
struct ListArgs {
    const int *count = nullptr;
    const int *list = nullptr;
};

// pass 3: one wave per (query, selected group, 16-row tile).  cand[q][j*64 + row] = key(score, row id).
// A tile's accumulator sees exactly the MFMA sequence it sees in score_group (ascending K, x y z w per chunk), so the scores are
// bit-identical to pass 1; splitting the group over four waves and keeping 16 KiB of the tile in flight per wave is what makes
// this pass short: it is a latency chain of k x 64 rows per query, not a bandwidth problem.
// TILES: the selection holds 16-row tiles instead of 64-row groups (large-batch scan): one wave per (query, selected tile),
// cand[q][j*16 + row].  qmap (list-driven exact pass): query q of this launch is the list's q-th entry, qmap[q] in the tiled
// query buffer; waves past *qcount exit.
// Pruning of the large-batch re-score (TILES only): the selection is sorted by approximate tile maximum, so with t_k = the k-th
// selected tile's approximate maximum there are k tiles that each hold a row with exact score >= t_k - e; a tile whose approximate
// maximum is below t_k - 2e holds only rows with exact score < t_k - e and cannot reach the top k.  Such tiles are not fetched
// (their candidate slots are written as absent); e is the verification's bound, from the same measured residuals.
struct PruneArgs {
    const float *tmax = nullptr;          // non-NULL enables pruning
    int64_t tstride = 0;
    int k = 0;
    const float *qnorm = nullptr, *qdelta = nullptr;
    const unsigned int *maxnorm_bits = nullptr, *maxdelta_bits = nullptr;
    float acc_eps = 0.f;
};

template <bool TILES>
__global__ __launch_bounds__(256) void rescore_groups_kernel(const float *__restrict__ data,
                                                             const float *__restrict__ qtiled, int dp4,
                                                             int64_t ntotal, const uint32_t *__restrict__ sel,
                                                             int sel_stride, int ksel, int nq, uint64_t *__restrict__ cand,
                                                             const int *__restrict__ skip, ListArgs la, PruneArgs pr = PruneArgs()) {
    constexpr int kRows = TILES ? 16 : kGroupRows, kSplit = TILES ? 1 : 4;
    const int lane = threadIdx.x & 63;
    const int64_t w = (int64_t)blockIdx.x * 4 + (threadIdx.x >> 6);
    if (w >= (int64_t)nq * ksel * kSplit) return;
    const int t = TILES ? 0 : (int)(w & 3);
    const int64_t qj = TILES ? w : w >> 2;
    const int q = (int)(qj / ksel), j = (int)(qj % ksel);
    if (skip && skip[q]) return;
    if (la.count && q >= *la.count) return;
    const int qs = la.list ? la.list[q] : q;       // row of the tiled query buffer
    const uint32_t g = sel[(int64_t)q * sel_stride + j];
    uint64_t *out = cand + ((int64_t)q * ksel + j) * kRows + t * 16;
    if (g == 0xFFFFFFFFu) {   // fewer groups than k
        if (lane < 16) out[lane] = 0;
        return;
    }
    if (TILES && pr.tmax && j >= pr.k) {          // the first k tiles are always re-scored
        const uint32_t gk = sel[(int64_t)q * sel_stride + pr.k - 1];
        if (gk != 0xFFFFFFFFu) {
            const float rmax = __uint_as_float(*pr.maxnorm_bits), qn = pr.qnorm[q], qd = pr.qdelta[q];
            const float e = 1.01f * ((qn + qd) * __uint_as_float(*pr.maxdelta_bits) + qd * rmax + pr.acc_eps * qn * rmax);
            // NaN / inf bounds compare false: nothing is pruned then
            if (pr.tmax[(int64_t)q * pr.tstride + g] < pr.tmax[(int64_t)q * pr.tstride + gk] - 2.f * e) {
                if (lane < 16) out[lane] = 0;
                return;
            }
        }
    }
    const int per_tile = dp4 * 16, kchunks = dp4 >> 2;
    const float4 *b = reinterpret_cast<const float4 *>(qtiled) + (int64_t)(qs >> 4) * per_tile + lane;
    const float4 *a = reinterpret_cast<const float4 *>(data) + ((int64_t)g * kSplit + t) * per_tile + lane;
    f32x4 acc = {0.f, 0.f, 0.f, 0.f};
    int kc = 0;
    for (; kc + 16 <= kchunks; kc += 16) {
        float4 av[16], bv[16];
#pragma unroll
        for (int u = 0; u < 16; ++u) av[u] = a[(kc + u) * 64];
#pragma unroll
        for (int u = 0; u < 16; ++u) bv[u] = b[(kc + u) * 64];
#pragma unroll
        for (int u = 0; u < 16; ++u) {
            acc = __builtin_amdgcn_mfma_f32_16x16x4f32(av[u].x, bv[u].x, acc, 0, 0, 0);
            acc = __builtin_amdgcn_mfma_f32_16x16x4f32(av[u].y, bv[u].y, acc, 0, 0, 0);
            acc = __builtin_amdgcn_mfma_f32_16x16x4f32(av[u].z, bv[u].z, acc, 0, 0, 0);
            acc = __builtin_amdgcn_mfma_f32_16x16x4f32(av[u].w, bv[u].w, acc, 0, 0, 0);
        }
    }
    for (; kc < kchunks; ++kc) {
        const float4 av = a[kc * 64], bv = b[kc * 64];
        acc = __builtin_amdgcn_mfma_f32_16x16x4f32(av.x, bv.x, acc, 0, 0, 0);
        acc = __builtin_amdgcn_mfma_f32_16x16x4f32(av.y, bv.y, acc, 0, 0, 0);
        acc = __builtin_amdgcn_mfma_f32_16x16x4f32(av.z, bv.z, acc, 0, 0, 0);
        acc = __builtin_amdgcn_mfma_f32_16x16x4f32(av.w, bv.w, acc, 0, 0, 0);
    }
    if ((lane & 15) == (qs & 15)) {
#pragma unroll
        for (int r = 0; r < 4; ++r) {
            const int rl = (lane >> 4) * 4 + r;
            const int64_t row = (int64_t)g * kRows + t * 16 + rl;
            uint64_t key = 0;
            if (row < ntotal) key = ((uint64_t)ivr_f2ord(acc[r]) << 32) | (uint32_t)(0xFFFFFFFFu - (uint32_t)row);
            out[rl] = key;
        }
    }
}

// ---------------------------------------------------------------------------------------------
// per-query exact top-k of 64-bit keys: MSB-first radix select (8 x 8 bits) + bitonic sort
// ---------------------------------------------------------------------------------------------
struct SrcGroupMax {   // pass 2: keys from the group-maximum column of query q
    const float *gmax;
    int64_t mstride;
    int64_t n;
    __device__ uint64_t key(int q, int64_t i) const {
        return ((uint64_t)ivr_f2ord(gmax[(int64_t)q * mstride + i]) << 32) |
               (uint32_t)(0xFFFFFFFFu - (uint32_t)i);
    }
};
struct SrcTilesOf {    // large-batch scan, second level: the 16-row tile maxima of the 128-row blocks selected at the first level
    const float *tmax;
    int64_t tstride;
    const uint32_t *selb;      // [nq][kb] selected blocks (0xFFFFFFFF = none)
    int kb;
    int64_t ntiles;            // ceil(ntotal / 16)
    int64_t n;                 // kb * 8
    __device__ uint64_t key(int q, int64_t i) const {
        const uint32_t b = selb[(int64_t)q * kb + (i >> 3)];
        if (b == 0xFFFFFFFFu) return 0;
        const int64_t t = (int64_t)b * 8 + (i & 7);
        if (t >= ntiles) return 0;
        return ((uint64_t)ivr_f2ord(tmax[(int64_t)q * tstride + t]) << 32) | (uint32_t)(0xFFFFFFFFu - (uint32_t)t);
    }
};
struct SrcKeys {       // pass 4: keys already materialised
    const uint64_t *keys;
    int64_t n;
    __device__ uint64_t key(int q, int64_t i) const { return keys[(int64_t)q * n + i]; }
};
struct SrcParts {      // shard merge: candidate p = part*k + j; ties resolve to the lower p = lower global id
    const float *D;
    const int64_t *I;
    int nq, k;
    int64_t n;         // parts * k
    __device__ uint64_t key(int q, int64_t p) const {
        const int64_t part = p / k, j = p % k;
        const int64_t off = (part * nq + q) * k + j;
        if (I[off] < 0) return 0;
        return ((uint64_t)ivr_f2ord(D[off]) << 32) | (uint32_t)(0xFFFFFFFFu - (uint32_t)p);
    }
};

struct SrcPacked {     // shard merge straight from the all-gather buffer: candidate = three int32 words (score bits, id lo, id hi)
    const int32_t *cand;   // [parts][nq][k][3]
    int nq, k;
    int64_t n;             // parts * k
    __device__ const int32_t *at(int q, int64_t p) const { return cand + (((p / k) * nq + q) * k + (p % k)) * 3; }
    __device__ uint64_t key(int q, int64_t p) const {
        const int32_t *c = at(q, p);
        if (c[2] < 0) return 0;                       // id -1: unused slot
        return ((uint64_t)ivr_f2ord(__int_as_float(c[0])) << 32) | (uint32_t)(0xFFFFFFFFu - (uint32_t)p);
    }
};

enum { OUT_GROUPS = 0, OUT_DI = 1, OUT_DI_PARTS = 2, OUT_DI_PACKED = 3 };

__global__ __launch_bounds__(256) void pack_candidates_kernel(const float *__restrict__ D, const int64_t *__restrict__ I, int64_t n,
                                                              int32_t *__restrict__ out) {
    const int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x;
    if (i >= n) return;
    const int64_t id = I[i];
    out[3 * i] = __float_as_int(D[i]);
    out[3 * i + 1] = (int32_t)(uint32_t)(id & 0xffffffffll);
    out[3 * i + 2] = (int32_t)(id >> 32);
}

constexpr int kRegKeys = 16;   // keys cached per thread: n <= 16 * 1024 is selected without re-reading global memory

template <typename Src, int OUT>
__global__ __launch_bounds__(kSelThreads) void select_topk_kernel(Src src, int qcol0, int k, int64_t id_base,
                                                                  uint32_t *__restrict__ out_groups,
                                                                  float *__restrict__ D, int64_t *__restrict__ I,
                                                                  const int64_t *__restrict__ I_parts,
                                                                  const int *__restrict__ skip = nullptr, VerifyArgs vf = VerifyArgs(),
                                                                  int *__restrict__ reset_flags = nullptr, ListArgs la = ListArgs()) {
    if (reset_flags && blockIdx.x == 0 && threadIdx.x < 4) reset_flags[threadIdx.x] = 0;
    if (skip && skip[blockIdx.x]) return;          // whole block: this query kept its fast-path result
    if (la.count && (int)blockIdx.x >= *la.count) return;
    __shared__ unsigned int hist[256];
    __shared__ unsigned long long s_prefix, s_mask;
    __shared__ unsigned int s_kth, s_cnt, s_valid;
    __shared__ uint64_t sorted[kMaxSort];
    const int q = blockIdx.x;
    const int qsrc = qcol0 + q;
    const int tid = threadIdx.x;
    const int nthr = blockDim.x;          // 256 for short candidate lists (cheaper barriers), else 1024
    const int64_t n = src.n;
    // The candidate keys of one query are few (N/64 group maxima, or k*64 rescored rows): keep them in registers so
    // that the eight radix passes cost LDS histogram time only, not eight dependent trips to L2.
    const bool cached = n <= (int64_t)kRegKeys * nthr;
    uint64_t kreg[kRegKeys];
    if (cached) {
#pragma unroll
        for (int j = 0; j < kRegKeys; ++j) {
            const int64_t i = (int64_t)j * nthr + tid;
            kreg[j] = i < n ? src.key(qsrc, i) : 0;
        }
    }
    auto for_each_key = [&](auto &&fn) {
        if (cached) {
#pragma unroll
            for (int j = 0; j < kRegKeys; ++j) fn(kreg[j]);
        } else {
            for (int64_t i = tid; i < n; i += nthr) fn(src.key(qsrc, i));
        }
    };

    // count valid keys (key 0 = absent)
    if (tid == 0) s_valid = 0;
    __syncthreads();
    {
        unsigned int c = 0;
        for_each_key([&](uint64_t key) { c += key != 0; });
        for (int o = 32; o > 0; o >>= 1) c += __shfl_xor(c, o, 64);
        if ((tid & 63) == 0 && c) atomicAdd(&s_valid, c);
    }
    __syncthreads();
    const unsigned int keff = min((unsigned int)k, s_valid);
    __shared__ uint64_t wmax[kSelThreads / 64];
    // Small k (the reference asks for 10..50; here up to 16 keys per wave): no serial extraction rounds.
    //  (1) every thread's largest key; (2) a lower bound T of the keff-th largest key: each wave removes the largest of its
    //  per-thread maxima r = ceil(keff / #waves) times (one DPP wave-max of the 32-bit score per round; equal scores leave
    //  together) and T is the smallest score removed last by any wave - every wave then holds >= r keys >= T, the block >= keff;
    //  (3) the keys >= T are collected, typically a few times keff of them; (4) each survivor counts the survivors above it:
    //  that is its rank (keys are unique).  Four barriers in all; the radix / extraction paths below remain the fallback when
    //  too many keys survive (scores tied in bulk).
    __shared__ uint64_t surv[kSelThreads];
    __shared__ unsigned int s_nsurv;
    __shared__ uint32_t wlow[kSelThreads / 64];
    bool done = false;
    const unsigned int nwv = (unsigned int)nthr >> 6;
    const unsigned int rounds = (keff + nwv - 1) / nwv;
    if (keff >= 1 && rounds <= 16) {
        uint64_t tm = 0;
        for_each_key([&](uint64_t key) { tm = key > tm ? key : tm; });
        uint32_t cur = (uint32_t)(tm >> 32), last = 0;
        for (unsigned int it = 0; it < rounds; ++it) {
            last = ivr_wave_max_u32(cur);
            if (cur == last) cur = 0;
        }
        if ((tid & 63) == 0) wlow[tid >> 6] = last;
        if (tid == 0) s_nsurv = 0;
        __syncthreads();
        uint32_t T = 0xFFFFFFFFu;
        for (unsigned int w = 0; w < nwv; ++w) T = min(T, wlow[w]);
        if (T != 0) {                              // 0: some wave ran out of keys - the fallback handles short lists
            const uint64_t T64 = (uint64_t)T << 32;
            for_each_key([&](uint64_t key) {
                if (key >= T64) {
                    const unsigned int slot = atomicAdd(&s_nsurv, 1u);
                    if (slot < (unsigned int)kSelThreads) surv[slot] = key;
                }
            });
        }
        __syncthreads();
        const unsigned int ns = s_nsurv;
        if (T != 0 && ns <= (unsigned int)nthr) {   // uniform: T and ns come from shared memory; ns >= keff by construction
            if ((unsigned int)tid < ns) {
                const uint64_t mine = surv[tid];
                unsigned int rank = 0;
                for (unsigned int j2 = 0; j2 < ns; ++j2) rank += surv[j2] > mine;
                if (rank < keff) sorted[rank] = mine;
            }
            done = true;
            __syncthreads();
        }
    }
    if (done) {
        // sorted[0 .. keff) is filled
    } else if (cached && keff <= 64) {
        // Small k (the reference asks for 10..50): extract the maximum keff times.  Per round: 16 register compares, a
        // wave max by shuffles, one LDS word per wave, two barriers - a few hundred cycles, against radix passes whose LDS
        // histogram atomics all collide on one bin when the scores share their leading bits.
        for (unsigned int it = 0; it < keff; ++it) {
            uint64_t m = 0;
#pragma unroll
            for (int j = 0; j < kRegKeys; ++j) m = kreg[j] > m ? kreg[j] : m;
            uint64_t wm = m;
#pragma unroll
            for (int o = 32; o > 0; o >>= 1) {
                const uint32_t hi = __shfl_xor((uint32_t)(wm >> 32), o, 64), lo = __shfl_xor((uint32_t)wm, o, 64);
                const uint64_t other = ((uint64_t)hi << 32) | lo;
                wm = other > wm ? other : wm;
            }
            if ((tid & 63) == 0) wmax[tid >> 6] = wm;
            __syncthreads();
            uint64_t gm = 0;
#pragma unroll
            for (int w = 0; w < kSelThreads / 64; ++w) gm = (w < (nthr >> 6) && wmax[w] > gm) ? wmax[w] : gm;
            if (tid == 0) sorted[it] = gm;
            if (m == gm) {                     // keys are unique: exactly one thread owns it
#pragma unroll
                for (int j = 0; j < kRegKeys; ++j)
                    if (kreg[j] == gm) kreg[j] = 0;
            }
            __syncthreads();
        }
    } else {
    uint64_t tau = ~0ull;   // nothing selected when keff == 0
    if (keff > 0) {
        if (tid == 0) {
            s_prefix = 0;
            s_mask = 0;
            s_kth = keff;
        }
        for (int pass = 0; pass < 8; ++pass) {
            const int shift = 56 - 8 * pass;
            if (tid < 256) hist[tid] = 0;
            __syncthreads();
            const unsigned long long prefix = s_prefix, mask = s_mask;
            for_each_key([&](uint64_t key) {
                if (key != 0 && (key & mask) == prefix) atomicAdd(&hist[(key >> shift) & 255], 1u);
            });
            __syncthreads();
            if (tid == 0) {
                unsigned int kth = s_kth, cum = 0;
                int dsel = 0;
                for (int dgt = 255; dgt >= 0; --dgt) {
                    const unsigned int h = hist[dgt];
                    if (cum + h >= kth) {
                        dsel = dgt;
                        break;
                    }
                    cum += h;
                }
                s_kth = kth - cum;
                s_prefix = prefix | ((unsigned long long)dsel << shift);
                s_mask = mask | (0xFFull << shift);
            }
            __syncthreads();
        }
        tau = s_prefix;   // the keff-th largest key (keys are unique)
    }
    // gather keys >= tau, pad, sort descending
    int P = 1;
    while (P < (int)keff) P <<= 1;
    if (tid == 0) s_cnt = 0;
    for (int i = tid; i < P; i += nthr) sorted[i] = 0;
    __syncthreads();
    if (keff > 0) {
        for_each_key([&](uint64_t key) {
            if (key != 0 && key >= tau) {
                const unsigned int slot = atomicAdd(&s_cnt, 1u);
                if (slot < (unsigned int)kMaxSort) sorted[slot] = key;
            }
        });
    }
    __syncthreads();
    for (int size = 2; size <= P; size <<= 1) {
        for (int stride = size >> 1; stride > 0; stride >>= 1) {
            for (int i = tid; i < (P >> 1); i += nthr) {
                const int lo = ((i / stride) * stride * 2) + (i % stride);
                const int hi = lo + stride;
                const bool desc = ((lo & size) == 0);
                const uint64_t a = sorted[lo], b = sorted[hi];
                if ((a < b) == desc) {
                    sorted[lo] = b;
                    sorted[hi] = a;
                }
            }
            __syncthreads();
        }
    }
    }
    for (int j = tid; j < k; j += nthr) {
        const uint64_t key = j < (int)keff ? sorted[j] : 0;
        const uint32_t low = 0xFFFFFFFFu - (uint32_t)key;
        if (OUT == OUT_GROUPS) {
            out_groups[(int64_t)q * k + j] = key ? low : 0xFFFFFFFFu;
        } else {
            const int64_t qo = la.list ? la.list[q] : q;     // output row
            D[qo * k + j] = key ? ivr_ord2f((uint32_t)(key >> 32)) : -FLT_MAX;
            int64_t id = -1;
            if (key) {
                if (OUT == OUT_DI_PARTS) {
                    const int kk = ((const SrcParts *)&src)->k, nq = ((const SrcParts *)&src)->nq;
                    id = I_parts[((int64_t)(low / kk) * nq + q) * kk + (low % kk)];
                } else if (OUT == OUT_DI_PACKED) {
                    const int32_t *c = ((const SrcPacked *)&src)->at(q, low);
                    id = ((int64_t)c[2] << 32) | (uint32_t)c[1];
                } else {
                    id = id_base + (int64_t)low;
                }
            }
            I[qo * k + j] = id;
        }
    }
    if (OUT == OUT_DI && vf.ok && tid == 0) {
        const uint32_t g = vf.sel[(int64_t)q * vf.ksel2 + vf.kp];
        int good = 1;
        if (g != 0xFFFFFFFFu) {                       // there IS an excluded group
            const float rmax = __uint_as_float(*vf.maxnorm_bits);
            float e;
            if (vf.qdelta) {
                const float qn = vf.qnorm[q], qd = vf.qdelta[q];
                e = 1.01f * ((qn + qd) * __uint_as_float(*vf.maxdelta_bits) + qd * rmax + vf.acc_eps * qn * rmax);
            } else {
                e = vf.rel_eps * vf.qnorm[q] * rmax;
            }
            const float bound = vf.gmax[(int64_t)q * vf.mstride + g] + e;
            const float kth = (int)keff >= k ? ivr_ord2f((uint32_t)(sorted[k - 1] >> 32)) : -FLT_MAX;
            good = bound < kth;                        // false for NaN / inf bounds too
        }
        vf.ok[q] = good;
        if (!good) {
            if (vf.fail_list) vf.fail_list[atomicAdd(vf.fail_count, 1)] = q;
            else atomicOr(&vf.tile_flag[q >> 4], 1);
        }
    }
}

}  // namespace

// -------------------------------------------------------------------------------------------------
// host side
// -------------------------------------------------------------------------------------------------
struct ivr_index {
    ivr_ctx *ctx = nullptr;
    int d = 0, dp = 0, dp4 = 0;
    int64_t cap = 0, ntotal = 0;     // cap is a multiple of kGroupRows
    float *data = nullptr;
    std::mutex mu;
    // search workspace (grow-only)
    float *qtiled = nullptr;         // [qtiles][dp4][16][4]
    float *qnorm = nullptr;          // [qtiles*16] upper bound of each tiled query's norm (bf16 candidate scan verification)
    int qtiles_cap = 0;
    float *gmax = nullptr;           // [qcols][mstride]
    int64_t gmax_floats = 0;
    uint32_t *sel = nullptr;         // [nq][ksel]
    uint64_t *cand = nullptr;        // [nq][ksel*64]
    int64_t sel_cap = 0;             // in (nq*ksel) units
    // bf16 candidate scan (scan16_groupmax_kernel): scan copy of the rows, split queries, verification state
    bool scan16 = false;             // IVR_SCAN_BF16 (default on), fixed at creation
    int pieces = 0;                  // 1 KiB pieces of a 16-row tile = ceil(dp / 32)
    uint4 *data16 = nullptr;         // [cap/16][pieces][64]
    uint4 *q16hi = nullptr, *q16lo = nullptr;   // [qtiles][pieces][64]
    unsigned int *maxnorm = nullptr; // DEV: bits of the largest stored row norm
    int *okflag = nullptr;           // DEV [64] per scan chunk + [4] tile flags behind it
    int last_nqc = 0;                // queries of the last chunk that went through the candidate scan
    // large-batch candidate scan (search_scanq.hip): more than 64 queries per call
    unsigned int *maxdelta = nullptr;// DEV: bits of the largest |row - bf16(row)| over the stored rows
    float *qdelta = nullptr;         // DEV [qtiles*16]: |q - bf16(q)| of each tiled query
    float *tmax = nullptr;           // DEV [padded queries of a chunk][tstride]: 16-row tile maxima (also the gmax of its exact pass)
    float *bmax = nullptr;           // DEV [padded queries of a chunk][bstride]: 128-row block maxima
    int64_t tmax_floats = 0, bmax_floats = 0;
    uint32_t *selb = nullptr;        // DEV [queries of a chunk][kp + 1]: selected blocks
    int64_t selb_cap = 0;
    int *okq = nullptr;              // DEV [kBigChunk] verification result per query, [4] failure count, [kBigChunk] failed queries
    bool last_big = false;           // the last search went through the large-batch scan
    bool bigq = true;                // IVR_SCAN_BIGQ=0 keeps every batch on the 64-query chunks (A/B switch, read at creation)
    bool prune = true;               // IVR_SCAN_PRUNE=0: the large-batch re-score fetches all kp selected tiles (A/B switch)
    bool ring = true;                // IVR_SCAN_RING=0: the <= 16-query candidate scan streams the index through registers (A/B switch)
};

namespace {

int64_t tile_bytes(const ivr_index *x, int64_t rows) { return rows * (int64_t)x->dp * 4; }

int64_t tile16_bytes(const ivr_index *x, int64_t rows) { return (rows / 16) * (int64_t)x->pieces * 1024; }

constexpr int kBigChunk = 1024;      // queries per launch chain of the large-batch scan
constexpr int kBigMaxK = 128;        // beyond this k the chunks of 64 queries are used (candidate lists grow with k)

int index_alloc(ivr_index *x, int64_t rows) {
    rows = ivr_round_up(std::max<int64_t>(rows, kGroupRows), kGroupRows);
    float *nd = nullptr;
    IVR_HIP(hipMalloc(&nd, (size_t)tile_bytes(x, rows)));
    IVR_HIP(hipMemset(nd, 0, (size_t)tile_bytes(x, rows)));
    uint4 *nd16 = nullptr;
    if (x->scan16) {
        // padded to whole 256-row blocks: the large-batch scan streams blocks (rows past ntotal are masked, never out of bounds)
        IVR_HIP(hipMalloc(&nd16, (size_t)tile16_bytes(x, ivr_round_up(rows, 256))));
        IVR_HIP(hipMemset(nd16, 0, (size_t)tile16_bytes(x, ivr_round_up(rows, 256))));
    }
    if (x->data) {
        if (x->ntotal > 0) {
            IVR_HIP(hipMemcpy(nd, x->data, (size_t)tile_bytes(x, ivr_round_up(x->ntotal, 16)), hipMemcpyDeviceToDevice));
            if (x->scan16)
                IVR_HIP(hipMemcpy(nd16, x->data16, (size_t)tile16_bytes(x, ivr_round_up(x->ntotal, 16)), hipMemcpyDeviceToDevice));
        }
        IVR_HIP(hipFree(x->data));
        if (x->data16) IVR_HIP(hipFree(x->data16));
    }
    x->data = nd;
    x->data16 = nd16;
    x->cap = rows;
    return IVR_OK;
}

// dst == x->data: index rows (bf16 scan copy + max norm alongside); dst == x->qtiled: queries (bf16 hi / lo split alongside)
int launch_tile_rows(ivr_index *x, float *dst, const float *src, int64_t start, int64_t n, int normalize,
                     int32_t *nonfinite, hipStream_t s, const int64_t *start_dev = nullptr, int64_t max_tiles = 0) {
    if (n <= 0) return IVR_OK;
    const int64_t ntiles = max_tiles ? max_tiles : ((start + n + 15) >> 4) - (start >> 4);
    const unsigned grid = (unsigned)ivr_ceil_div(ntiles, 4);
    const bool rows = dst == x->data;
    IvrProf prof("tile_rows", s, (double)n * (x->d + x->dp) * 4 + (x->scan16 ? (double)n * x->pieces * 64 * (rows ? 1 : 2) : 0.0), true);
    // query tiles: the padding rows of the last tile are zero-filled by the kernel itself (no memset in front of it)
    hipLaunchKernelGGL(tile_rows_kernel, dim3(grid), dim3(256), 0, s, src, dst, start, n, x->d, x->dp4, normalize, nonfinite, start_dev,
                       x->scan16 ? (rows ? x->data16 : x->q16hi) : (uint4 *)nullptr, x->scan16 && !rows ? x->q16lo : (uint4 *)nullptr,
                       x->scan16 && rows ? x->maxnorm : (unsigned int *)nullptr, rows ? (float *)nullptr : x->qnorm, rows ? 0 : 1,
                       x->pieces, x->scan16 && rows ? x->maxdelta : (unsigned int *)nullptr,
                       x->scan16 && !rows ? x->qdelta : (float *)nullptr);
    IVR_LAUNCH_CHECK();
    return IVR_OK;
}

int sel_threads(int64_t n) { return n <= 16 * 256 ? 256 : kSelThreads; }

// choose the query tile width of the scan (queries per index pass = 16*QT)
int pick_qt(int nq) { return nq <= 16 ? 1 : nq <= 32 ? 2 : nq <= 48 ? 3 : 4; }
// groups re-scored exactly behind the bf16 candidate scan: k plus slack for what the approximate ranking may displace
int fast_groups(int k) { return k + std::max(22, k); }

// more than 64 queries: the tiled large-batch scan (search_scanq.hip) instead of chunks of 64 queries past the streamed index
bool use_big(const ivr_index *x, int nq, int k) { return x->scan16 && x->bigq && nq > 64 && k <= kBigMaxK; }

int reserve_search(ivr_index *x, int nq, int k) {
    const bool big = use_big(x, nq, k);
    // the large-batch scan reads whole blocks of 256 queries: the tiled query buffers are padded (zero rows) to that
    const int qtiles = (int)ivr_ceil_div(big ? ivr_round_up(nq, 256) : nq, 16);
    if (qtiles > x->qtiles_cap) {
        if (x->qtiles_cap) IVR_HIP(hipFree(x->qtiled));
        if (x->qnorm) IVR_HIP(hipFree(x->qnorm));
        if (x->qdelta) IVR_HIP(hipFree(x->qdelta));
        x->qtiled = nullptr;
        x->qnorm = nullptr;
        x->qdelta = nullptr;
        x->qtiles_cap = 0;
        const int want = std::max(qtiles, 4);
        IVR_HIP(hipMalloc(&x->qtiled, (size_t)want * 16 * x->dp * 4));
        IVR_HIP(hipMemset(x->qtiled, 0, (size_t)want * 16 * x->dp * 4));
        IVR_HIP(hipMalloc(&x->qnorm, (size_t)want * 16 * 4));
        IVR_HIP(hipMemset(x->qnorm, 0, (size_t)want * 16 * 4));
        if (x->scan16) {
            if (x->q16hi) IVR_HIP(hipFree(x->q16hi));
            if (x->q16lo) IVR_HIP(hipFree(x->q16lo));
            IVR_HIP(hipMalloc(&x->q16hi, (size_t)want * x->pieces * 1024));
            IVR_HIP(hipMalloc(&x->q16lo, (size_t)want * x->pieces * 1024));
            IVR_HIP(hipMemset(x->q16hi, 0, (size_t)want * x->pieces * 1024));
            IVR_HIP(hipMemset(x->q16lo, 0, (size_t)want * x->pieces * 1024));
            IVR_HIP(hipMalloc(&x->qdelta, (size_t)want * 16 * 4));
            IVR_HIP(hipMemset(x->qdelta, 0, (size_t)want * 16 * 4));
        }
        x->qtiles_cap = want;
    }
    const int64_t mstride = ivr_round_up(x->cap / kGroupRows, 64);
    const int64_t need_gmax = (int64_t)64 * mstride;   // one scan chunk = up to 64 query columns
    if (need_gmax > x->gmax_floats) {
        if (x->gmax) IVR_HIP(hipFree(x->gmax));
        x->gmax = nullptr;
        x->gmax_floats = 0;
        IVR_HIP(hipMalloc(&x->gmax, (size_t)need_gmax * 4));
        x->gmax_floats = need_gmax;
    }
    const int chunk_q = std::min(nq, big ? kBigChunk : 64);
    const int64_t need_sel = (int64_t)chunk_q * (x->scan16 ? fast_groups(k) + 1 : k);   // per scan chunk
    if (need_sel > x->sel_cap) {
        if (x->sel) IVR_HIP(hipFree(x->sel));
        if (x->cand) IVR_HIP(hipFree(x->cand));
        x->sel = nullptr;
        x->cand = nullptr;
        x->sel_cap = 0;
        IVR_HIP(hipMalloc(&x->sel, (size_t)need_sel * 4));
        IVR_HIP(hipMalloc(&x->cand, (size_t)need_sel * kGroupRows * 8));
        x->sel_cap = need_sel;
    }
    if (big) {
        const int qpad = (int)ivr_round_up(chunk_q, 256);
        const int64_t cap256 = ivr_round_up(x->cap, 256);
        const int64_t need_t = (int64_t)qpad * ivr_round_up(cap256 / 16, 64), need_b = (int64_t)qpad * ivr_round_up(cap256 / 128, 64);
        if (need_t > x->tmax_floats) {
            if (x->tmax) IVR_HIP(hipFree(x->tmax));
            x->tmax = nullptr;
            x->tmax_floats = 0;
            IVR_HIP(hipMalloc(&x->tmax, (size_t)need_t * 4));
            x->tmax_floats = need_t;
        }
        if (need_b > x->bmax_floats) {
            if (x->bmax) IVR_HIP(hipFree(x->bmax));
            x->bmax = nullptr;
            x->bmax_floats = 0;
            IVR_HIP(hipMalloc(&x->bmax, (size_t)need_b * 4));
            x->bmax_floats = need_b;
        }
        const int64_t need_selb = (int64_t)chunk_q * (fast_groups(k) + 1);
        if (need_selb > x->selb_cap) {
            if (x->selb) IVR_HIP(hipFree(x->selb));
            x->selb = nullptr;
            x->selb_cap = 0;
            IVR_HIP(hipMalloc(&x->selb, (size_t)need_selb * 4));
            x->selb_cap = need_selb;
        }
        if (!x->okq) {
            IVR_HIP(hipMalloc(&x->okq, (size_t)(2 * kBigChunk + 4) * sizeof(int)));
            IVR_HIP(hipMemset(x->okq, 0, (size_t)(2 * kBigChunk + 4) * sizeof(int)));
        }
    }
    return IVR_OK;
}

template <int QT>
void launch_scan(ivr_index *x, const float *qt, int64_t ngroups, int64_t mstride, hipStream_t s, const int *tile_flag = nullptr) {
    const size_t lds = (size_t)QT * 16 * x->dp * 4;
    // 8 waves per workgroup share one staged query tile; size the grid so every CU holds as many
    // workgroups as the LDS allows and let each wave stride over the groups
    const int threads = 512, nw = threads / 64;
    int per_cu = (int)std::max<size_t>(1, std::min<size_t>(4, (160 * 1024) / std::max<size_t>(lds, 1)));
    int64_t grid = std::min<int64_t>(ivr_ceil_div(ngroups, nw), (int64_t)x->ctx->cu_count * per_cu);
    grid = std::max<int64_t>(grid, 1);
    (void)hipFuncSetAttribute(reinterpret_cast<const void *>(scan_groupmax_kernel<QT>), hipFuncAttributeMaxDynamicSharedMemorySize,
                        (int)lds);
    // algorithmic bytes: every stored row once + the query tile + one maximum per (group, query)
    IvrProf prof("scan_groupmax", s, (double)x->ntotal * x->dp * 4 + (double)QT * 16 * x->dp * 4 + (double)ngroups * QT * 16 * 4,
                 tile_flag != nullptr);      // behind the bf16 candidate scan it is the predicated fallback and normally exits at once
    hipLaunchKernelGGL(scan_groupmax_kernel<QT>, dim3((unsigned)grid), dim3(threads), lds, s, x->data, qt, x->dp4, ngroups,
                       x->ntotal, x->gmax, mstride, tile_flag);
}

template <int QT>
void launch_scan16(ivr_index *x, int64_t tile0, int64_t ngroups, int64_t mstride, hipStream_t s) {
    const size_t lds = (size_t)QT * 2 * x->pieces * 1024;
    const int threads = 512, nw = threads / 64;
    int per_cu = (int)std::max<size_t>(1, std::min<size_t>(4, (160 * 1024) / std::max<size_t>(lds, 1)));
    int64_t grid = std::min<int64_t>(ivr_ceil_div(ngroups, nw), (int64_t)x->ctx->cu_count * per_cu);
    grid = std::max<int64_t>(grid, 1);
    (void)hipFuncSetAttribute(reinterpret_cast<const void *>(scan16_groupmax_kernel<QT>), hipFuncAttributeMaxDynamicSharedMemorySize,
                              (int)lds);
    // algorithmic bytes: the bf16 copy of every stored row once + the split query tiles + one maximum per (group, query)
    IvrProf prof("scan16_groupmax", s, (double)x->ntotal * x->pieces * 64 + (double)QT * 2 * x->pieces * 1024 + (double)ngroups * QT * 16 * 4);
    hipLaunchKernelGGL(scan16_groupmax_kernel<QT>, dim3((unsigned)grid), dim3(threads), lds, s, x->data16, x->q16hi + tile0 * x->pieces * 64,
                       x->q16lo + tile0 * x->pieces * 64, x->pieces, ngroups, x->ntotal, x->gmax, mstride);
}

// at most 16 queries and a piece count the ring kernel is built for: stream the index through the LDS-DMA rings
template <int PIECES>
void launch_scan16_ring(ivr_index *x, int64_t tile0, int64_t ngroups, int64_t mstride, hipStream_t s) {
    const int threads = 512, nw = threads / 64;
    const size_t lds = (size_t)nw * PIECES * 1024;
    // one workgroup per CU (its 8 rings already keep PIECES x 8 KiB in flight); groups are dealt round-robin to the waves of the grid
    int64_t grid = std::max<int64_t>(1, std::min<int64_t>(ivr_ceil_div(ngroups, nw), (int64_t)x->ctx->cu_count * (lds <= 64 * 1024 ? 2 : 1)));
    (void)ivr_func_max_lds(reinterpret_cast<const void *>(scan16_ring_kernel<PIECES>), (int)lds);
    IvrProf prof("scan16_groupmax", s, (double)x->ntotal * x->pieces * 64 + (double)2 * x->pieces * 1024 + (double)ngroups * 16 * 4);
    hipLaunchKernelGGL(scan16_ring_kernel<PIECES>, dim3((unsigned)grid), dim3(threads), lds, s, x->data16, x->q16hi + tile0 * x->pieces * 64,
                       x->q16lo + tile0 * x->pieces * 64, ngroups, x->ntotal, x->gmax, mstride);
}

template <int QT>
void launch_scan_list(ivr_index *x, const float *qt, int64_t ngroups, int64_t mstride, float *gmax, const int *nlist, const int *list,
                      hipStream_t s) {
    const size_t lds = (size_t)QT * 16 * x->dp * 4;
    const int threads = 512, nw = threads / 64;
    int per_cu = (int)std::max<size_t>(1, std::min<size_t>(4, (160 * 1024) / std::max<size_t>(lds, 1)));
    int64_t grid = std::max<int64_t>(1, std::min<int64_t>(ivr_ceil_div(ngroups, nw), (int64_t)x->ctx->cu_count * per_cu));
    (void)hipFuncSetAttribute(reinterpret_cast<const void *>(scan_groupmax_list_kernel<QT>), hipFuncAttributeMaxDynamicSharedMemorySize,
                              (int)lds);
    IvrProf prof("scan_groupmax_list", s, 0.0, true);      // normally nothing is listed and every workgroup exits at once
    hipLaunchKernelGGL(scan_groupmax_list_kernel<QT>, dim3((unsigned)grid), dim3(threads), lds, s, x->data, qt, x->dp4, ngroups, x->ntotal,
                       gmax, mstride, nlist, list);
}

// One chunk (<= kBigChunk queries, already tiled at tile q0 / 16) of a large batch:
//   scanq (index read once) -> top kp+1 blocks of 128 rows per query -> top kp+1 tiles of 16 rows among those blocks' tiles (the kp+1
//   best tiles always lie inside the kp+1 best blocks: the argument of DESIGN.md section 4 with tiles for rows) -> exact float32
//   re-score of kp tiles -> final selection, which also verifies the approximate ranking per query and lists the queries that
//   fail -> list-driven exact pass (four launches that exit at once when the list is empty; no host round trip).
int search_big(ivr_index *x, int q0, int nqc, int k, int64_t id_base, float *D, int64_t *I, hipStream_t s) {
    const int kp = fast_groups(k), ksel2 = kp + 1;
    const int qpad = (int)ivr_round_up(nqc, 256);
    const int64_t cap256 = ivr_round_up(x->cap, 256);
    const int64_t tstride = ivr_round_up(cap256 / 16, 64), bstride = ivr_round_up(cap256 / 128, 64);
    const int64_t nblk128 = ivr_ceil_div(x->ntotal, 128), ntiles = ivr_ceil_div(x->ntotal, 16);
    const int64_t ngroups = ivr_ceil_div(x->ntotal, kGroupRows), mstride = ivr_round_up(x->cap / kGroupRows, 64);
    const float *qtile = x->qtiled + (int64_t)(q0 / 16) * 16 * x->dp;
    int *ok = x->okq, *nfail = x->okq + kBigChunk, *flist = x->okq + kBigChunk + 4;
    ScanQArgs a;
    a.data16 = x->data16;
    a.q16 = x->q16hi + (int64_t)(q0 / 16) * x->pieces * 64;
    a.pieces = x->pieces;
    a.qblocks = qpad / 256;
    a.ntotal = x->ntotal;
    a.nblocks = ivr_ceil_div(x->ntotal, 256);
    a.tmax = x->tmax;
    a.tstride = tstride;
    a.bmax = x->bmax;
    a.bstride = bstride;
    int rc = ivr_launch_scanq(x->ctx, a, s);
    if (rc != IVR_OK) return rc;
    {
        SrcGroupMax sb{x->bmax, bstride, nblk128};
        IvrProf prof("select_blocks", s, (double)nqc * nblk128 * 4, true);
        hipLaunchKernelGGL((select_topk_kernel<SrcGroupMax, OUT_GROUPS>), dim3(nqc), dim3(sel_threads(nblk128)), 0, s, sb, 0, ksel2, (int64_t)0,
                           x->selb, (float *)nullptr, (int64_t *)nullptr, (const int64_t *)nullptr, (const int *)nullptr, VerifyArgs(), nfail,
                           ListArgs());
        IVR_LAUNCH_CHECK();
    }
    {
        SrcTilesOf st{x->tmax, tstride, x->selb, ksel2, ntiles, (int64_t)ksel2 * 8};
        IvrProf prof("select_tiles", s, (double)nqc * ksel2 * 8 * 4, true);
        hipLaunchKernelGGL((select_topk_kernel<SrcTilesOf, OUT_GROUPS>), dim3(nqc), dim3(sel_threads((int64_t)ksel2 * 8)), 0, s, st, 0, ksel2,
                           (int64_t)0, x->sel, (float *)nullptr, (int64_t *)nullptr, (const int64_t *)nullptr, (const int *)nullptr,
                           VerifyArgs(), (int *)nullptr, ListArgs());
        IVR_LAUNCH_CHECK();
    }
    {
        const int64_t waves = (int64_t)nqc * kp;
        IvrProf prof("rescore_tiles", s, (double)waves * 16 * x->dp * 4, true);
        PruneArgs pr;
        if (x->prune) {
            pr.tmax = x->tmax;
            pr.tstride = tstride;
            pr.k = k;
            pr.qnorm = x->qnorm + q0;
            pr.qdelta = x->qdelta + q0;
            pr.maxnorm_bits = x->maxnorm;
            pr.maxdelta_bits = x->maxdelta;
            pr.acc_eps = (float)x->dp * 1.2e-7f;
        }
        hipLaunchKernelGGL(rescore_groups_kernel<true>, dim3((unsigned)ivr_ceil_div(waves, 4)), dim3(256), 0, s, x->data, qtile, x->dp4, x->ntotal,
                           x->sel, ksel2, kp, nqc, x->cand, (const int *)nullptr, ListArgs(), pr);
        IVR_LAUNCH_CHECK();
    }
    {
        SrcKeys sk{x->cand, (int64_t)kp * 16};
        VerifyArgs vf;
        vf.gmax = x->tmax;
        vf.mstride = tstride;
        vf.sel = x->sel;
        vf.ksel2 = ksel2;
        vf.kp = kp;
        vf.qnorm = x->qnorm + q0;
        vf.maxnorm_bits = x->maxnorm;
        vf.ok = ok;
        vf.qdelta = x->qdelta + q0;
        vf.maxdelta_bits = x->maxdelta;
        vf.acc_eps = (float)x->dp * 1.2e-7f;
        vf.fail_count = nfail;
        vf.fail_list = flist;
        IvrProf prof("select_final", s, (double)nqc * kp * 16 * 8, true);
        hipLaunchKernelGGL((select_topk_kernel<SrcKeys, OUT_DI>), dim3(nqc), dim3(sel_threads((int64_t)kp * 16)), 0, s, sk, 0, k, id_base,
                           (uint32_t *)nullptr, D + (int64_t)q0 * k, I + (int64_t)q0 * k, (const int64_t *)nullptr, (const int *)nullptr, vf,
                           (int *)nullptr, ListArgs());
        IVR_LAUNCH_CHECK();
    }
    // exact pass over the listed queries; its group maxima reuse the tile-maxima buffer (read for the last time just above)
    const ListArgs la{nfail, flist};
    float *gmax = x->tmax;
    const int qt_max = (int)std::max<int64_t>(1, std::min<int64_t>(4, (128 * 1024) / ((int64_t)16 * x->dp * 4)));
    switch (qt_max) {
        case 1: launch_scan_list<1>(x, qtile, ngroups, mstride, gmax, nfail, flist, s); break;
        case 2: launch_scan_list<2>(x, qtile, ngroups, mstride, gmax, nfail, flist, s); break;
        case 3: launch_scan_list<3>(x, qtile, ngroups, mstride, gmax, nfail, flist, s); break;
        default: launch_scan_list<4>(x, qtile, ngroups, mstride, gmax, nfail, flist, s); break;
    }
    IVR_LAUNCH_CHECK();
    {
        SrcGroupMax sg{gmax, mstride, ngroups};
        IvrProf prof("select_groups", s, 0.0, true);
        hipLaunchKernelGGL((select_topk_kernel<SrcGroupMax, OUT_GROUPS>), dim3(nqc), dim3(sel_threads(ngroups)), 0, s, sg, 0, k, (int64_t)0, x->sel,
                           (float *)nullptr, (int64_t *)nullptr, (const int64_t *)nullptr, (const int *)nullptr, VerifyArgs(), (int *)nullptr, la);
        IVR_LAUNCH_CHECK();
    }
    {
        const int64_t waves = (int64_t)nqc * k;
        IvrProf prof("rescore_groups", s, 0.0, true);
        hipLaunchKernelGGL(rescore_groups_kernel<false>, dim3((unsigned)waves), dim3(256), 0, s, x->data, qtile, x->dp4, x->ntotal, x->sel, k, k, nqc,
                           x->cand, (const int *)nullptr, la);
        IVR_LAUNCH_CHECK();
    }
    {
        SrcKeys sk{x->cand, (int64_t)k * kGroupRows};
        IvrProf prof("select_final", s, 0.0, true);
        hipLaunchKernelGGL((select_topk_kernel<SrcKeys, OUT_DI>), dim3(nqc), dim3(sel_threads((int64_t)k * kGroupRows)), 0, s, sk, 0, k, id_base,
                           (uint32_t *)nullptr, D + (int64_t)q0 * k, I + (int64_t)q0 * k, (const int64_t *)nullptr, (const int *)nullptr, VerifyArgs(),
                           (int *)nullptr, la);
        IVR_LAUNCH_CHECK();
    }
    return IVR_OK;
}

}  // namespace

extern "C" {

int ivr_index_create(ivr_ctx *ctx, int d, int64_t capacity_rows, ivr_index **out) {
    IVR_REQUIRE(ctx && out, "ivr_index_create: NULL argument");
    IVR_REQUIRE(d >= 1 && d <= 2048, "ivr_index_create: d=%d out of range [1,2048]", d);
    IVR_REQUIRE(capacity_rows >= 0 && capacity_rows < (1ll << 32) - 64, "ivr_index_create: capacity %lld out of range",
                (long long)capacity_rows);
    IVR_HIP(hipSetDevice(ctx->device));
    ivr_index *x = new ivr_index();
    x->ctx = ctx;
    x->d = d;
    x->dp = (int)ivr_round_up(d, 16);
    x->dp4 = x->dp / 4;
    x->pieces = (x->dp + 31) / 32;
    {
        const char *e = getenv("IVR_SCAN_BF16");       // A/B switch, read when the index is created
        x->scan16 = !(e && e[0] == '0');   // LDS per 16-query tile (hi + lo) = 64 dp bytes, the same as the float32 scan's
        const char *b = getenv("IVR_SCAN_BIGQ");
        x->bigq = !(b && b[0] == '0');
        const char *pr = getenv("IVR_SCAN_PRUNE");
        x->prune = !(pr && pr[0] == '0');
        const char *rg = getenv("IVR_SCAN_RING");
        x->ring = !(rg && rg[0] == '0');
    }
    // an even number of pieces per tile: the large-batch scan steps K by two pieces; an odd tail piece stays all zero on both sides
    x->pieces = (int)ivr_round_up(x->pieces, 2);
    if (x->scan16) {
        IVR_HIP(hipMalloc(&x->maxnorm, 4));
        IVR_HIP(hipMemset(x->maxnorm, 0, 4));
        IVR_HIP(hipMalloc(&x->okflag, 68 * sizeof(int)));
        IVR_HIP(hipMemset(x->okflag, 0, 68 * sizeof(int)));
        IVR_HIP(hipMalloc(&x->maxdelta, 4));
        IVR_HIP(hipMemset(x->maxdelta, 0, 4));
    }
    int rc = index_alloc(x, capacity_rows);
    if (rc != IVR_OK) {
        delete x;
        return rc;
    }
    *out = x;
    return IVR_OK;
}

int ivr_index_destroy(ivr_index *x) {
    if (!x) return IVR_OK;
    if (x->data) (void)hipFree(x->data);
    if (x->qtiled) (void)hipFree(x->qtiled);
    if (x->qnorm) (void)hipFree(x->qnorm);
    if (x->gmax) (void)hipFree(x->gmax);
    if (x->sel) (void)hipFree(x->sel);
    if (x->cand) (void)hipFree(x->cand);
    for (void *p : {(void *)x->data16, (void *)x->q16hi, (void *)x->q16lo, (void *)x->maxnorm, (void *)x->okflag, (void *)x->maxdelta,
                    (void *)x->qdelta, (void *)x->tmax, (void *)x->bmax, (void *)x->selb, (void *)x->okq})
        if (p) (void)hipFree(p);
    delete x;
    return IVR_OK;
}

int ivr_index_reset(ivr_index *x) {
    IVR_REQUIRE(x, "ivr_index_reset: NULL index");
    std::lock_guard<std::mutex> lk(x->mu);
    IVR_HIP(hipSetDevice(x->ctx->device));
    IVR_HIP(hipMemset(x->data, 0, (size_t)tile_bytes(x, x->cap)));
    if (x->scan16) {
        IVR_HIP(hipMemset(x->data16, 0, (size_t)tile16_bytes(x, ivr_round_up(x->cap, 256))));
        IVR_HIP(hipMemset(x->maxnorm, 0, 4));
        IVR_HIP(hipMemset(x->maxdelta, 0, 4));
    }
    x->ntotal = 0;
    return IVR_OK;
}

int64_t ivr_index_ntotal(ivr_index *x) { return x ? x->ntotal : 0; }
int ivr_index_dim(ivr_index *x) { return x ? x->d : 0; }
int64_t ivr_index_capacity(ivr_index *x) { return x ? x->cap : 0; }

int ivr_index_add(ivr_index *x, const float *rows, int64_t n, int normalize, ivr_stream stream) {
    IVR_REQUIRE(x && (rows || n == 0), "ivr_index_add: NULL argument");
    IVR_REQUIRE(n >= 0, "ivr_index_add: n=%lld", (long long)n);
    std::lock_guard<std::mutex> lk(x->mu);
    IVR_HIP(hipSetDevice(x->ctx->device));
    if (x->ntotal + n > x->cap) {
        IVR_REQUIRE(x->ntotal + n < (1ll << 32) - 64, "ivr_index_add: index would exceed 2^32 rows");
        // growing re-allocates: wait for work that may still read the old buffer
        IVR_HIP(hipDeviceSynchronize());
        int rc = index_alloc(x, std::max<int64_t>(x->ntotal + n, x->cap + x->cap / 2));
        if (rc != IVR_OK) return rc;
    }
    int rc = launch_tile_rows(x, x->data, rows, x->ntotal, n, normalize, nullptr, (hipStream_t)stream);
    if (rc != IVR_OK) return rc;
    x->ntotal += n;
    return IVR_OK;
}

int ivr_index_write(ivr_index *x, int64_t start, const float *rows, int64_t n, int normalize, ivr_stream stream) {
    IVR_REQUIRE(x && (rows || n == 0), "ivr_index_write: NULL argument");
    std::lock_guard<std::mutex> lk(x->mu);
    IVR_REQUIRE(start >= 0 && n >= 0 && start + n <= x->ntotal, "ivr_index_write: rows [%lld,%lld) outside [0,%lld)",
                (long long)start, (long long)(start + n), (long long)x->ntotal);
    IVR_HIP(hipSetDevice(x->ctx->device));
    return launch_tile_rows(x, x->data, rows, start, n, normalize, nullptr, (hipStream_t)stream);
}

int ivr_index_write_ring(ivr_index *x, const float *rows, int64_t n, int normalize, int64_t *cursor, ivr_stream stream) {
    IVR_REQUIRE(x && rows && cursor, "ivr_index_write_ring: NULL argument");
    std::lock_guard<std::mutex> lk(x->mu);
    IVR_REQUIRE(n >= 1 && x->ntotal >= n && x->ntotal % n == 0,
                "ivr_index_write_ring: batch of %lld rows must divide ntotal=%lld (no wrap inside a batch)", (long long)n,
                (long long)x->ntotal);
    IVR_HIP(hipSetDevice(x->ctx->device));
    hipStream_t s = (hipStream_t)stream;
    // the cursor is only known on the device: launch for the worst-case number of touched tiles
    int rc = launch_tile_rows(x, x->data, rows, 0, n, normalize, nullptr, s, cursor, ivr_ceil_div(n, 16) + 1);
    if (rc != IVR_OK) return rc;
    hipLaunchKernelGGL(advance_cursor_kernel, dim3(1), dim3(1), 0, s, cursor, n, x->ntotal);
    IVR_LAUNCH_CHECK();
    return IVR_OK;
}

int ivr_index_reconstruct(ivr_index *x, int64_t start, int64_t n, float *out, ivr_stream stream) {
    IVR_REQUIRE(x && (out || n == 0), "ivr_index_reconstruct: NULL argument");
    std::lock_guard<std::mutex> lk(x->mu);
    IVR_REQUIRE(start >= 0 && n >= 0 && start + n <= x->ntotal, "ivr_index_reconstruct: rows [%lld,%lld) outside [0,%lld)",
                (long long)start, (long long)(start + n), (long long)x->ntotal);
    if (n == 0) return IVR_OK;
    IVR_HIP(hipSetDevice(x->ctx->device));
    const int64_t ntiles = ((start + n + 15) >> 4) - (start >> 4);
    hipLaunchKernelGGL(untile_rows_kernel, dim3((unsigned)ivr_ceil_div(ntiles, 4)), dim3(256), 0, (hipStream_t)stream, x->data,
                       out, start, n, x->d, x->dp4);
    IVR_LAUNCH_CHECK();
    return IVR_OK;
}

int ivr_index_reserve_search(ivr_index *x, int max_nq, int max_k) {
    IVR_REQUIRE(x, "ivr_index_reserve_search: NULL index");
    IVR_REQUIRE(max_nq >= 1 && max_k >= 1 && max_k <= IVR_MAX_K, "ivr_index_reserve_search: nq=%d k=%d", max_nq, max_k);
    std::lock_guard<std::mutex> lk(x->mu);
    IVR_HIP(hipSetDevice(x->ctx->device));
    return reserve_search(x, max_nq, max_k);
}

int ivr_index_search(ivr_index *x, const float *q, int nq, int k, int normalize_q, int64_t id_base, float *D, int64_t *I,
                     ivr_stream stream) {
    IVR_REQUIRE(x && q && D && I, "ivr_index_search: NULL argument");
    IVR_REQUIRE(nq >= 1, "ivr_index_search: nq=%d", nq);
    IVR_REQUIRE(k >= 1 && k <= IVR_MAX_K, "ivr_index_search: k=%d outside [1,%d]", k, IVR_MAX_K);
    std::lock_guard<std::mutex> lk(x->mu);
    IVR_HIP(hipSetDevice(x->ctx->device));
    hipStream_t s = (hipStream_t)stream;
    int rc = reserve_search(x, nq, k);
    if (rc != IVR_OK) return rc;
    const int64_t ngroups = ivr_ceil_div(x->ntotal, kGroupRows);
    const int64_t mstride = ivr_round_up(x->cap / kGroupRows, 64);
    // queries -> tiled layout (normalised on the way when asked: N2 on the query side, core.py:875)
    {
        rc = launch_tile_rows(x, x->qtiled, q, 0, nq, normalize_q, nullptr, s);
        if (rc != IVR_OK) return rc;
    }
    const int ksel = k;
    // queries per index pass: as many 16-query tiles as fit 128 KiB of LDS, at most 4
    const int qt_max = (int)std::max<int64_t>(1, std::min<int64_t>(4, (128 * 1024) / ((int64_t)16 * x->dp * 4)));
    const int chunk = 16 * qt_max;
    // bf16 candidate scan first when it can pay: enough groups that kp of them are a small fraction, k within the selector's
    // range.  Its result is verified per query on the device; failures are redone by the exact pass below (tile_flag / skip).
    const int kp = fast_groups(k);
    const bool fast = x->scan16 && ngroups >= 4 * (int64_t)(kp + 1) && kp + 1 <= IVR_MAX_K;
    x->last_big = false;
    if (fast && use_big(x, nq, k)) {
        // large batch: the index is read once per kBigChunk queries instead of once per 64
        for (int q0 = 0; q0 < nq; q0 += kBigChunk) {
            const int nqc = std::min(kBigChunk, nq - q0);
            rc = search_big(x, q0, nqc, k, id_base, D, I, s);
            if (rc != IVR_OK) return rc;
            x->last_nqc = nqc;
        }
        x->last_big = true;
        return IVR_OK;
    }
    // bf16 keeps 8 significant bits: |row - bf16(row)| <= 2^-8 |row| per element; the query's hi + lo leaves 2^-16; f32 accumulation
    const float rel_eps = (0.00390625f + 0.0000306f + (float)x->dp * 1.2e-7f) * 1.01f;
    int *ok = x->okflag, *tile_flag = x->okflag ? x->okflag + 64 : nullptr;
    auto exact_pass = [&](const float *qtile, int nqc, int qt, int q0, const int *flags, const int *skip) -> int {
        if (ngroups > 0) {
            switch (qt) {
                case 1: launch_scan<1>(x, qtile, ngroups, mstride, s, flags); break;
                case 2: launch_scan<2>(x, qtile, ngroups, mstride, s, flags); break;
                case 3: launch_scan<3>(x, qtile, ngroups, mstride, s, flags); break;
                default: launch_scan<4>(x, qtile, ngroups, mstride, s, flags); break;
            }
            IVR_LAUNCH_CHECK();
        }
        SrcGroupMax sg{x->gmax, mstride, ngroups};
        {
            IvrProf prof("select_groups", s, (double)nqc * ngroups * 4, true);
            hipLaunchKernelGGL((select_topk_kernel<SrcGroupMax, OUT_GROUPS>), dim3(nqc), dim3(sel_threads(ngroups)), 0, s, sg, 0, ksel,
                               (int64_t)0, x->sel, (float *)nullptr, (int64_t *)nullptr, (const int64_t *)nullptr, skip);
        }
        IVR_LAUNCH_CHECK();
        const int64_t waves = (int64_t)nqc * ksel;
        // rescore reads query tile (q >> 4) relative to the chunk's first tile
        {
            IvrProf prof("rescore_groups", s, (double)waves * kGroupRows * x->dp * 4, true);
            hipLaunchKernelGGL(rescore_groups_kernel<false>, dim3((unsigned)waves), dim3(256), 0, s, x->data, qtile, x->dp4,
                               x->ntotal, x->sel, ksel, ksel, nqc, x->cand, skip, ListArgs());
        }
        IVR_LAUNCH_CHECK();
        SrcKeys sk{x->cand, (int64_t)ksel * kGroupRows};
        IvrProf prof("select_final", s, (double)waves * kGroupRows * 8, true);
        hipLaunchKernelGGL((select_topk_kernel<SrcKeys, OUT_DI>), dim3(nqc), dim3(sel_threads((int64_t)ksel * kGroupRows)), 0, s, sk, 0, k,
                           id_base, (uint32_t *)nullptr, D + (int64_t)q0 * k, I + (int64_t)q0 * k, (const int64_t *)nullptr, skip);
        IVR_LAUNCH_CHECK();
        return IVR_OK;
    };
    for (int q0 = 0; q0 < nq; q0 += chunk) {
        const int nqc = std::min(chunk, nq - q0);
        const int qt = pick_qt(nqc);
        const float *qtile = x->qtiled + (int64_t)(q0 / 16) * 16 * x->dp;
        if (!fast) {
            x->last_nqc = 0;
            rc = exact_pass(qtile, nqc, qt, q0, nullptr, nullptr);
            if (rc != IVR_OK) return rc;
            continue;
        }
        x->last_nqc = nqc;
        switch (qt) {
            case 1:
                if (x->ring && x->pieces == 16) launch_scan16_ring<16>(x, q0 / 16, ngroups, mstride, s);
                else if (x->ring && x->pieces == 12) launch_scan16_ring<12>(x, q0 / 16, ngroups, mstride, s);
                else if (x->ring && x->pieces == 8) launch_scan16_ring<8>(x, q0 / 16, ngroups, mstride, s);
                else launch_scan16<1>(x, q0 / 16, ngroups, mstride, s);
                break;
            case 2: launch_scan16<2>(x, q0 / 16, ngroups, mstride, s); break;
            case 3: launch_scan16<3>(x, q0 / 16, ngroups, mstride, s); break;
            default: launch_scan16<4>(x, q0 / 16, ngroups, mstride, s); break;
        }
        IVR_LAUNCH_CHECK();
        const int ksel2 = kp + 1;
        SrcGroupMax sg{x->gmax, mstride, ngroups};
        {
            IvrProf prof("select_groups", s, (double)nqc * ngroups * 4, true);
            hipLaunchKernelGGL((select_topk_kernel<SrcGroupMax, OUT_GROUPS>), dim3(nqc), dim3(sel_threads(ngroups)), 0, s, sg, 0, ksel2,
                               (int64_t)0, x->sel, (float *)nullptr, (int64_t *)nullptr, (const int64_t *)nullptr, (const int *)nullptr,
                               VerifyArgs(), tile_flag);
        }
        IVR_LAUNCH_CHECK();
        const int64_t waves = (int64_t)nqc * kp;
        {
            IvrProf prof("rescore_groups", s, (double)waves * kGroupRows * x->dp * 4, true);
            hipLaunchKernelGGL(rescore_groups_kernel<false>, dim3((unsigned)waves), dim3(256), 0, s, x->data, qtile, x->dp4,
                               x->ntotal, x->sel, ksel2, kp, nqc, x->cand, (const int *)nullptr, ListArgs());
        }
        IVR_LAUNCH_CHECK();
        {
            SrcKeys sk{x->cand, (int64_t)kp * kGroupRows};
            IvrProf prof("select_final", s, (double)waves * kGroupRows * 8, true);
            VerifyArgs vf;
            vf.gmax = x->gmax;
            vf.mstride = mstride;
            vf.sel = x->sel;
            vf.ksel2 = ksel2;
            vf.kp = kp;
            vf.qnorm = x->qnorm + q0;
            vf.rel_eps = rel_eps;
            vf.maxnorm_bits = x->maxnorm;
            vf.ok = ok;
            vf.tile_flag = tile_flag;
            hipLaunchKernelGGL((select_topk_kernel<SrcKeys, OUT_DI>), dim3(nqc), dim3(sel_threads((int64_t)kp * kGroupRows)), 0, s, sk, 0, k,
                               id_base, (uint32_t *)nullptr, D + (int64_t)q0 * k, I + (int64_t)q0 * k, (const int64_t *)nullptr,
                               (const int *)nullptr, vf, (int *)nullptr);
        }
        IVR_LAUNCH_CHECK();
        // exact pass for the queries whose check failed: every kernel below exits at once when nothing is flagged
        rc = exact_pass(qtile, nqc, qt, q0, tile_flag, ok);
        if (rc != IVR_OK) return rc;
    }
    return IVR_OK;
}

int ivr_index_scan_stats(ivr_index *x, int *out) {
    IVR_REQUIRE(x && out, "ivr_index_scan_stats: NULL argument");
    std::lock_guard<std::mutex> lk(x->mu);
    out[0] = x->scan16 ? 1 : 0;
    out[1] = 0;
    if (!x->scan16) return IVR_OK;
    IVR_HIP(hipSetDevice(x->ctx->device));
    IVR_HIP(hipDeviceSynchronize());
    if (x->last_big) {       // large-batch scan: the length of the failure list of the last chunk
        IVR_HIP(hipMemcpy(&out[1], x->okq + kBigChunk, sizeof(int), hipMemcpyDeviceToHost));
        return IVR_OK;
    }
    int ok[64];
    IVR_HIP(hipMemcpy(ok, x->okflag, sizeof(ok), hipMemcpyDeviceToHost));
    for (int i = 0; i < x->last_nqc; ++i) out[1] += ok[i] == 0;
    return IVR_OK;
}

int ivr_topk_merge(ivr_ctx *ctx, const float *D_parts, const int64_t *I_parts, int parts, int nq, int k, float *D,
                   int64_t *I, ivr_stream stream) {
    IVR_REQUIRE(ctx && D_parts && I_parts && D && I, "ivr_topk_merge: NULL argument");
    IVR_REQUIRE(parts >= 1 && nq >= 1 && k >= 1 && k <= IVR_MAX_K, "ivr_topk_merge: parts=%d nq=%d k=%d", parts, nq, k);
    IVR_HIP(hipSetDevice(ctx->device));
    SrcParts sp{D_parts, I_parts, nq, k, (int64_t)parts * k};
    hipLaunchKernelGGL((select_topk_kernel<SrcParts, OUT_DI_PARTS>), dim3(nq), dim3(sel_threads((int64_t)parts * k)), 0, (hipStream_t)stream, sp, 0,
                       k, (int64_t)0, (uint32_t *)nullptr, D, I, I_parts);
    IVR_LAUNCH_CHECK();
    return IVR_OK;
}

int ivr_topk_pack(ivr_ctx *ctx, const float *D, const int64_t *I, int nq, int k, int32_t *packed, ivr_stream stream) {
    IVR_REQUIRE(ctx && D && I && packed, "ivr_topk_pack: NULL argument");
    IVR_REQUIRE(nq >= 1 && k >= 1, "ivr_topk_pack: nq=%d k=%d", nq, k);
    IVR_HIP(hipSetDevice(ctx->device));
    const int64_t n = (int64_t)nq * k;
    hipLaunchKernelGGL(pack_candidates_kernel, dim3((unsigned)ivr_ceil_div(n, 256)), dim3(256), 0, (hipStream_t)stream, D, I, n, packed);
    IVR_LAUNCH_CHECK();
    return IVR_OK;
}

int ivr_topk_merge_packed(ivr_ctx *ctx, const int32_t *packed_parts, int parts, int nq, int k, float *D, int64_t *I, ivr_stream stream) {
    IVR_REQUIRE(ctx && packed_parts && D && I, "ivr_topk_merge_packed: NULL argument");
    IVR_REQUIRE(parts >= 1 && nq >= 1 && k >= 1 && k <= IVR_MAX_K, "ivr_topk_merge_packed: parts=%d nq=%d k=%d", parts, nq, k);
    IVR_HIP(hipSetDevice(ctx->device));
    SrcPacked sp{packed_parts, nq, k, (int64_t)parts * k};
    hipLaunchKernelGGL((select_topk_kernel<SrcPacked, OUT_DI_PACKED>), dim3(nq), dim3(sel_threads((int64_t)parts * k)), 0, (hipStream_t)stream, sp,
                       0, k, (int64_t)0, (uint32_t *)nullptr, D, I, (const int64_t *)nullptr);
    IVR_LAUNCH_CHECK();
    return IVR_OK;
}

int ivr_l2_normalize(ivr_ctx *ctx, float *x, int64_t n, int d, int32_t *nonfinite, ivr_stream stream) {
    IVR_REQUIRE(ctx && (x || n == 0), "ivr_l2_normalize: NULL argument");
    IVR_REQUIRE(n >= 0 && d >= 1, "ivr_l2_normalize: n=%lld d=%d", (long long)n, d);
    if (n == 0) return IVR_OK;
    IVR_HIP(hipSetDevice(ctx->device));
    if (nonfinite) IVR_HIP(hipMemsetAsync(nonfinite, 0, 4, (hipStream_t)stream));
    hipLaunchKernelGGL(l2_normalize_kernel, dim3((unsigned)ivr_ceil_div(n, 4)), dim3(256), 0, (hipStream_t)stream, x, n, d,
                       nonfinite);
    IVR_LAUNCH_CHECK();
    return IVR_OK;
}

}  // extern "C"
