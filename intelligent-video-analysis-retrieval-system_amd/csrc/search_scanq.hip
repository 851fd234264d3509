// Candidate scan for LARGE query batches (S1 of SURVEY.md section 8a at BASELINE configs[2]: 1,000 queries against a
// 1.25M-row shard; replaces faiss.IndexFlatIP.search at unified_index.py:503 / the peer fan-out of system.py:1715-1757).
//
// The small-batch scan (search.hip, scan16_groupmax_kernel) keeps up to 64 queries in LDS and streams the index past them:
// HBM-bound, one index pass per 64 queries.  From a few hundred queries on the work is a dense contraction and belongs on
// the matrix pipe with BOTH operands tiled: this kernel is a 256-row x 256-query MFMA tile machine whose "C" never exists -
// the accumulators are reduced in registers to the maximum of every 16-row tile and of every 128-row block per query.
//
//   operands  rows and queries live in HBM already in MFMA-fragment order ([16-row tile][piece][64 lanes x 16 B], one
//             1 KiB piece = 32 floats of K for 16 rows), so a stage of the K loop is filled by plain contiguous 1 KiB
//             LDS-DMA pieces (buffer_load ... lds, no per-lane address work, no swizzle) and a fragment read is
//             ds_read_b128 at lane*16: conflict-free by construction.
//   tile      8 waves as 2 (rows) x 4 (queries); a wave owns 128 rows x 64 queries = 8 x 4 accumulators of
//             v_mfma_f32_16x16x32_bf16.  K step = 2 pieces (64 values): 32 KiB per operand per stage, three row slots +
//             two query slots = 160 KiB (the row operand comes from HBM and runs two stages ahead; the query operand is
//             L2-resident and runs one ahead) - the stage pipeline of gemm_big_kernel (tower_kernels.hip).
//   persistent  one workgroup per CU walks its work items (row block, query block) as ONE flattened sequence of stages: the
//             DMA of the next item's first stages is in flight while the current item finishes, so only the first item of
//             a workgroup pays a prologue (K = 512 is 8 stages: a per-tile prologue would be > 10 % of the tile).
//   placement workgroups b, b+8, ... share an XCD.  XCD x owns row blocks x, x+8, ...; consecutive slots of an XCD take the
//             query blocks of the SAME row block, so a 256 KiB row block is fetched from HBM once and served to its other
//             query blocks by that XCD's L2 (1,000 queries = 4 query blocks: algorithmic index bytes x 1).
//   epilogue  per query tile: 3 v_max per accumulator, then a transposing butterfly (v_permlane32_swap / v_permlane16_swap)
//             that leaves lane (c, h) with the maxima of row tiles 2h, 2h+1 for query c: one 8-byte store per lane per query
//             tile instead of 8 scattered dword stores, plus the block maximum.  No score is ever written.
//
// The maxima only RANK 16-row tiles (search.hip re-scores the selected tiles from the float32 rows and verifies the ranking
// per query on the device), so the operands are bf16 rounded to nearest on both sides - half the MFMAs of the hi + lo split
// the small-batch scan uses.
#include "search_internal.h"

#include <algorithm>
#include <cfloat>

namespace {

typedef uint32_t u32x4 __attribute__((ext_vector_type(4)));
typedef __bf16 bf16x8_t __attribute__((ext_vector_type(8)));

constexpr int SQ_SLOT = 32 * 1024;        // one operand stage: 16 tiles x 2 pieces x 1 KiB
constexpr int SQ_WBASE = 3 * SQ_SLOT;     // query slots behind the three row slots
constexpr int SQ_LDS = 5 * SQ_SLOT;       // 160 KiB

__device__ __forceinline__ float sq_max(float a, float b) {      // plain v_max_f32 (fmaxf adds canonicalising moves)
    float r;
    asm("v_max_f32 %0, %1, %2" : "=v"(r) : "v"(a), "v"(b));
    return r;
}
__device__ __forceinline__ float sq_max3(float a, float b, float c) {
    float r;
    asm("v_max3_f32 %0, %1, %2, %3" : "=v"(r) : "v"(a), "v"(b), "v"(c));
    return r;
}
// max(x of this lane, x of the lane 32 away) for lanes 0-31 and the same of y for lanes 32-63.  One statement: the two wait
// states a VALU write needs before v_permlane*_swap reads it (gfx950 hazard rule) are inside the string.
__device__ __forceinline__ float sq_fold32(float x, float y) {
    asm("s_nop 1\n\tv_permlane32_swap_b32 %0, %1\n\tv_max_f32 %0, %0, %1" : "+v"(x), "+v"(y));
    return x;
}
// the same across the two 16-lane rows of each half: even rows end with x folded, odd rows with y folded
__device__ __forceinline__ float sq_fold16(float x, float y) {
    asm("s_nop 1\n\tv_permlane16_swap_b32 %0, %1\n\tv_max_f32 %0, %0, %1" : "+v"(x), "+v"(y));
    return x;
}

__global__ __launch_bounds__(512, 2) void scanq_kernel(ScanQArgs g) {
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int wm = wave >> 2, wn = wave & 3;
    const int xcd = blockIdx.x & 7, slot = blockIdx.x >> 3, nslots = gridDim.x >> 3;
    const int QB = g.qblocks;
    const int64_t nrx = g.nblocks > xcd ? (g.nblocks - xcd + 7) >> 3 : 0;      // row blocks of this XCD
    const int64_t nitems = nrx * QB;                                           // (row block, query block) items of this XCD
    if (slot >= nitems) return;
    const int count = (int)((nitems - slot + nslots - 1) / nslots);            // items of this workgroup: slot, slot + nslots, ...
    const int pieces = g.pieces, KT = pieces >> 1;
    const int S = count * KT;                                                  // stages of this workgroup, all items flattened
    const unsigned slab = 16u * (unsigned)pieces * 1024u;                      // bytes of 256 rows (or 256 queries)
    const unsigned lds0 = (unsigned)(size_t)smem;

    // item n of this workgroup is number slot + n * nslots of the XCD's list: row block r = that / QB, query block = that % QB.
    // Each of the three cursors (row DMA, query DMA, compute) walks the list by adding (step_r, step_q): no division in the loop.
    struct Pos {
        int r, qb;
    };
    const int step_r = nslots / QB, step_q = nslots - step_r * QB;
    auto step = [&](Pos &p) {
        p.r += step_r;
        p.qb += step_q;
        if (p.qb >= QB) {
            p.qb -= QB;
            ++p.r;
        }
    };
    Pos px = {slot / QB, slot % QB}, pw = px, pc = px;
    auto rows_rsrc = [&](const Pos &p) {
        return __builtin_amdgcn_make_buffer_rsrc(
            const_cast<char *>(reinterpret_cast<const char *>(g.data16)) + (int64_t)(xcd + 8 * (int64_t)p.r) * (int64_t)slab, 0, (int)slab, 0x00020000);
    };
    auto query_rsrc = [&](const Pos &p) {
        return __builtin_amdgcn_make_buffer_rsrc(const_cast<char *>(reinterpret_cast<const char *>(g.q16)) + (int64_t)p.qb * (int64_t)slab, 0,
                                                 (int)slab, 0x00020000);
    };

    // DMA: 32 pieces of 1 KiB per operand per stage; wave w issues pieces 4w .. 4w+3 of each.  Piece p = tile p >> 1, K half
    // p & 1 of the stage; the LDS image of a slot is the pieces in that order.
    unsigned pbase[4];
#pragma unroll
    for (int j = 0; j < 4; ++j) {
        const int p = wave * 4 + j;
        pbase[j] = (unsigned)((p >> 1) * pieces + (p & 1)) * 1024u;
    }
    const unsigned voff = (unsigned)lane * 16u;
    // cursors of the two DMA streams: the next stage to issue for the row operand (X) and the query operand (W)
    int xn = 0, xkt = 0, wcn = 0, wkt = 0;
    auto rsX = rows_rsrc(px);
    auto rsW = query_rsrc(pw);
    auto pieceX = [&](int xs, int j) {
        __builtin_amdgcn_raw_ptr_buffer_load_lds(rsX, (__attribute__((address_space(3))) void *)(smem + xs * SQ_SLOT + (wave * 4 + j) * 1024), 16,
                                                 voff, pbase[j] + (unsigned)xkt * 2048u, 0, 0);
    };
    auto pieceW = [&](int ws, int j) {
        __builtin_amdgcn_raw_ptr_buffer_load_lds(rsW, (__attribute__((address_space(3))) void *)(smem + SQ_WBASE + ws * SQ_SLOT + (wave * 4 + j) * 1024),
                                                 16, voff, pbase[j] + (unsigned)wkt * 2048u, 0, 0);
    };
    auto advanceX = [&]() {
        if (++xkt == KT) {
            xkt = 0;
            step(px);
            if (++xn < count) rsX = rows_rsrc(px);
        }
    };
    auto advanceW = [&]() {
        if (++wkt == KT) {
            wkt = 0;
            step(pw);
            if (++wcn < count) rsW = query_rsrc(pw);
        }
    };

    unsigned foX[2], foW[2];
#pragma unroll
    for (int kk = 0; kk < 2; ++kk) {
        foX[kk] = lds0 + (unsigned)(wm * 8 * 2048 + kk * 1024 + lane * 16);
        foW[kk] = lds0 + (unsigned)(SQ_WBASE + wn * 4 * 2048 + kk * 1024 + lane * 16);
    }

    f32x4 acc[4][8];   // [query tile][row tile]: lane (c = lane & 15, h = lane >> 4) holds rows 4h .. 4h+3 of the tile for query c
    const f32x4 zero = {0.f, 0.f, 0.f, 0.f};

#define SQ_ROW(XF, WF, MT, FIRST)                                                                                          \
    _Pragma("unroll") for (int nt = 0; nt < 4; ++nt) acc[nt][MT] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(                \
        __builtin_bit_cast(bf16x8_t, XF[MT]), __builtin_bit_cast(bf16x8_t, WF[nt]), FIRST ? zero : acc[nt][MT], 0, 0, 0); \
    __builtin_amdgcn_sched_barrier(0);
#define SQ_RD4(DST, ADDR, O0)                                                                              \
    asm volatile("ds_read_b128 %0, %1 offset:%2" : "=v"(DST[0]) : "v"(ADDR), "n"(O0));                      \
    asm volatile("ds_read_b128 %0, %1 offset:%2" : "=v"(DST[1]) : "v"(ADDR), "n"(O0 + 2048));               \
    asm volatile("ds_read_b128 %0, %1 offset:%2" : "=v"(DST[2]) : "v"(ADDR), "n"(O0 + 4096));               \
    asm volatile("ds_read_b128 %0, %1 offset:%2" : "=v"(DST[3]) : "v"(ADDR), "n"(O0 + 6144));
#define SQ_LGKM(N)                                                                                         \
    asm volatile("s_waitcnt lgkmcnt(" #N ")" ::: "memory");                                                \
    __builtin_amdgcn_sched_barrier(0);

    u32x4 xa0[8], wa0[4], xa1[8], wa1[4];
    u32x4 *x0lo = xa0, *x0hi = xa0 + 4, *x1lo = xa1, *x1hi = xa1 + 4;
    // prologue (once per workgroup): stage 0, then X(1), W(1), X(2) - the order the counted waits rely on
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
    SQ_RD4(wa0, foW[0], 0)
    SQ_RD4(x0lo, foX[0], 0)
    SQ_RD4(x0hi, foX[0], 8192)

    int xs = 0;            // row slot of stage s = s % 3
    int ckt = 0;           // K stage of the item being computed (the item itself: pc)
    for (int s = 0; s < S; ++s) {
        const int xs1 = xs == 2 ? 0 : xs + 1, xs2 = xs1 == 2 ? 0 : xs1 + 1;
        const unsigned xoff = xs * SQ_SLOT, woff = (s & 1) * SQ_SLOT, nxoff = xs1 * SQ_SLOT, nwoff = ((s + 1) & 1) * SQ_SLOT;
        // DMA windows: after the mid-stage barrier of stage s a wave issues W(s+2) x 4, then X(s+3) pieces 0, 1; pieces 2, 3 of
        // X(s+3) go out early in stage s+1 (`tail`; the prologue issued X(2) whole)
        const bool tail = s >= 1 && s + 2 < S;
        const bool morew = s + 2 < S, morex = s + 3 < S, next = s + 1 < S;
        const unsigned wa = foW[1] + woff, xa = foX[1] + xoff, nwa = foW[0] + nwoff, nxa = foX[0] + nxoff;
#define SQ_SET0(FIRST)                                                                                     \
        SQ_LGKM(4)                                                                                         \
        SQ_ROW(xa0, wa0, 0, FIRST)                                                                         \
        if (tail) pieceX(xs2, 2);                                                                          \
        SQ_ROW(xa0, wa0, 1, FIRST)                                                                         \
        SQ_RD4(wa1, wa, 0)                                                                                 \
        SQ_ROW(xa0, wa0, 2, FIRST)                                                                         \
        if (tail) pieceX(xs2, 3);                                                                          \
        SQ_ROW(xa0, wa0, 3, FIRST)                                                                         \
        SQ_RD4(x1lo, xa, 0)                                                                                \
        SQ_LGKM(8)                                                                                         \
        SQ_ROW(xa0, wa0, 4, FIRST)                                                                         \
        SQ_ROW(xa0, wa0, 5, FIRST)                                                                         \
        SQ_RD4(x1hi, xa, 8192)                                                                             \
        SQ_ROW(xa0, wa0, 6, FIRST)                                                                         \
        SQ_ROW(xa0, wa0, 7, FIRST)
        if (ckt == 0) {
            SQ_SET0(true)
        } else {
            SQ_SET0(false)
        }
#undef SQ_SET0
        SQ_LGKM(0)                      // this wave has read everything it needs from the stage's buffers
        if (s >= 1) advanceX();         // the row cursor now names stage s + 3
        if (next) {
            // X(s+1) and W(s+1) must have landed; the four youngest LOADS in flight are X(s+2), which may stay.  The global stores of an
            // item's epilogue count in vmcnt too and are not retired in a common order with the loads, but they can only make this wait
            // longer, never release it early: with at most 4 operations outstanding, an outstanding piece of stage s+1 would imply
            // its four younger loads (in order among themselves) outstanding as well - five.
            if (s + 2 < S) asm volatile("s_waitcnt vmcnt(4)\n\ts_barrier" ::: "memory");
            else asm volatile("s_waitcnt vmcnt(0)\n\ts_barrier" ::: "memory");
        }
        SQ_ROW(xa1, wa1, 0, false)
        if (morew) pieceW(s & 1, 0);
        SQ_ROW(xa1, wa1, 1, false)
        if (next) { SQ_RD4(wa0, nwa, 0) }
        if (morew) pieceW(s & 1, 1);
        SQ_ROW(xa1, wa1, 2, false)
        if (morew) pieceW(s & 1, 2);
        SQ_ROW(xa1, wa1, 3, false)
        if (next) { SQ_RD4(x0lo, nxa, 0) }
        if (morew) pieceW(s & 1, 3);
        SQ_ROW(xa1, wa1, 4, false)
        if (morex) pieceX(xs, 0);
        SQ_ROW(xa1, wa1, 5, false)
        if (next) { SQ_RD4(x0hi, nxa, 8192) }
        SQ_ROW(xa1, wa1, 6, false)
        if (morex) pieceX(xs, 1);
        SQ_ROW(xa1, wa1, 7, false)
        if (morew) advanceW();
        xs = xs1;
        if (++ckt < KT) continue;

        // ---- item finished: reduce the 128 x 64 scores of this wave to tile and block maxima ----
        ckt = 0;
        const int64_t rb = xcd + 8 * (int64_t)pc.r;
        const int qb = pc.qb;
        step(pc);
        const int64_t row0 = rb * 256 + wm * 128;                    // first row of this wave's 8 tiles
        const bool partial = row0 + 128 > g.ntotal;                  // wave-uniform: only the last block of the index
        const int c = lane & 15, h = lane >> 4;
#pragma unroll
        for (int nt = 0; nt < 4; ++nt) {
            float m[8];
#pragma unroll
            for (int mt = 0; mt < 8; ++mt) {
                f32x4 a = acc[nt][mt];
                if (partial) {
#pragma unroll
                    for (int r = 0; r < 4; ++r)
                        if (row0 + mt * 16 + h * 4 + r >= g.ntotal) a[r] = -FLT_MAX;
                }
                m[mt] = sq_max(sq_max3(a[0], a[1], a[2]), a[3]);
            }
            // transposing butterfly: lanes 0-31 keep tiles 0-3, lanes 32-63 tiles 4-7; then even 16-lane rows keep the first
            // of each pair: lane (c, h) ends with tiles 2h (p0) and 2h+1 (p1), each folded over the four h lanes of query c
            const float n0 = sq_fold32(m[0], m[4]), n1 = sq_fold32(m[1], m[5]), n2 = sq_fold32(m[2], m[6]), n3 = sq_fold32(m[3], m[7]);
            const float p0 = sq_fold16(n0, n2), p1 = sq_fold16(n1, n3);
            const int64_t q = (int64_t)qb * 256 + wn * 64 + nt * 16 + c;
            *reinterpret_cast<float2 *>(g.tmax + q * g.tstride + (row0 >> 4) + 2 * h) = make_float2(p0, p1);
            float b = sq_max(p0, p1);
            b = sq_fold16(b, b);
            b = sq_fold32(b, b);
            if (h == 0) g.bmax[q * g.bstride + (row0 >> 7)] = b;
        }
    }
#undef SQ_ROW
#undef SQ_RD4
#undef SQ_LGKM
}

}  // namespace

int ivr_launch_scanq(ivr_ctx *ctx, const ScanQArgs &a, hipStream_t s) {
    IVR_REQUIRE(a.pieces >= 2 && (a.pieces & 1) == 0, "scanq: pieces=%d must be even", a.pieces);
    IVR_REQUIRE(a.qblocks >= 1 && a.nblocks >= 1, "scanq: empty problem");
    int rc = ivr_func_max_lds(reinterpret_cast<const void *>(scanq_kernel), SQ_LDS);
    if (rc != IVR_OK) return rc;
    // one workgroup per CU (160 KiB of LDS each); a multiple of 8 so that slot = blockIdx / 8 is the same on every XCD
    const int grid = std::max(8, ctx->cu_count / 8 * 8);
    // algorithmic work of the launch: 2 * rows * padded queries * K flop
    IvrProf prof("scanq", s, 2.0 * (double)a.ntotal * a.qblocks * 256.0 * a.pieces * 32.0);
    hipLaunchKernelGGL(scanq_kernel, dim3(grid), dim3(512), SQ_LDS, s, a);
    IVR_LAUNCH_CHECK();
    return IVR_OK;
}
