// Frame quality gating of the keyframe filter: the two OpenCV measures the reference computes per
// frame at filter.py:63-92,
//     blur_score   = cv2.Laplacian(gray, cv2.CV_64F).var()                 (calculate_blur_score)
//     edge_density = count(cv2.Canny(gray, 20, 80) > 0) / (H * W) * 100     (calculate_edge_density)
// on a batch of decoded frames resident in HBM.  OpenCV is not installed here, so the operators are restated from their published
// definitions (oracle/quality_ref.py says the same and is "parity unpinned"):
//   gray       cv2.cvtColor(BGR2GRAY) for uint8: (4899 R + 9617 G + 1868 B + 8192) >> 14
//   Laplacian  ksize = 1: the 3x3 aperture [0 1 0; 1 -4 1; 0 1 0], BORDER_REFLECT_101, exact integers; the variance is
//              returned as the two exact integer sums (sum, sum of squares) and finished in float64 on the host
//   Canny      Sobel 3x3 with BORDER_REPLICATE, L1 magnitude |dx| + |dy|, candidates m > low, direction sectors by the fixed-point
//              tan(22.5 deg) test of canny.cpp, the asymmetric > / >= neighbour comparisons, strong m > high, hysteresis over the
//              8-neighbourhood.
//
// ONE tiled kernel does gray + Laplacian sums + Sobel + non-maximum suppression.  A workgroup owns a 64 x 32 pixel tile: the RGB
// bytes of the tile and a 2-pixel rim are fetched with 16-byte buffer loads into LDS; gray (border-replicated) and the packed
// (magnitude, sector) plane live only in LDS and are worked on four pixels per thread through dword LDS accesses; what leaves the
// chip per pixel is two BITS: the candidate plane (local maxima above `low`) and the strong plane (above `high`), one 64-bit word
// per tile row each (a wave ballot): 3 B read + 0.25 B written per pixel.
// The hysteresis is a morphological reconstruction on those bitmaps: S <- the runs of C that touch S or the 3-dilation of S in the
// row above / below.  A wave holds a whole row (one 64-pixel word per lane); a seed fills its run in BOTH
// directions with one multi-word add each (`((C + s) ^ C) & C`: the carry walks the run; carries cross lanes by a carry-lookahead on
// two ballot masks, the leftward fill runs on bit-reversed words and bit-reversed masks).  A frame is cut into up to 16 bands, one
// wave each: down and up sweeps alternate inside a band until one changes nothing, the bands exchange their boundary rows through
// memory until a whole round changes nothing.  A contour of any length is followed in a few sweeps, where round 3's first version (a
// level-synchronous work list) needed one step per pixel of the longest weak chain.  The edge count is the population count of the
// final S.
#include "ivr_common.h"

#include <algorithm>

namespace {

constexpr int TG22 = 13573;          // (int)(0.4142135623730950488 * (1 << 15) + 0.5)
constexpr int QT_W = 64, QT_H = 32;  // core tile
constexpr int QG_H = QT_H + 4, QG_W = QT_W + 4, QG_S = 72;   // gray region (core + 2-pixel rim: 68 columns), LDS row stride
constexpr int QM_H = QT_H + 2, QM_S = 68;                    // magnitude region (1-pixel rim: 66 columns used), u16 row stride
constexpr int QM_GROUPS = QM_S / 4;                          // 17 groups of four pixels per magnitude row
constexpr int QR_CHUNKS = 14, QR_S = QR_CHUNKS * 16;         // raw RGB row in LDS: up to 15 B of misalignment + 3 * 68 B
constexpr int QR_PAD = 16;                                   // the first tile column reads up to 8 B in front of a row
constexpr int kMaxRowWords = 256;                            // frames up to 16,384 pixels wide (4 words per lane)

struct QualityArgs {
    const uint8_t *frames;
    int n, h, w, bgr, low, high, w64;
    unsigned long long *cand, *strong;      // [n][h][w64] bit planes, bit b of word j = pixel 64 j + b
    long long *tile_sums;                   // [n][tiles][2]: every tile's (sum, sum of squares) of the Laplacian, added up by the sweep kernel
};

__device__ __forceinline__ int byte_of(unsigned lo, unsigned hi, int k) { return k < 4 ? (int)((lo >> (8 * k)) & 255u) : (int)((hi >> (8 * (k - 4))) & 255u); }

__global__ __launch_bounds__(256) void quality_tile_kernel(QualityArgs g) {
    __shared__ __attribute__((aligned(16))) uint8_t raw[QR_PAD + QG_H * QR_S];
    __shared__ __attribute__((aligned(16))) uint8_t gray[QG_H * QG_S];
    __shared__ __attribute__((aligned(16))) uint16_t magl[QM_H * QM_S];
    __shared__ int rowmis[QG_H];
    __shared__ int sh1[4];
    __shared__ unsigned sh2[4];
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int img = blockIdx.z, x0 = blockIdx.x * QT_W, y0 = blockIdx.y * QT_H;
    const int h = g.h, w = g.w;
    // in-frame part of the gray region
    const int gx0 = max(x0 - 2, 0), gx1 = min(x0 + QT_W + 2, w), gy0 = max(y0 - 2, 0), gy1 = min(y0 + QT_H + 2, h);
    const int rows = gy1 - gy0, rowbytes = (gx1 - gx0) * 3;
    // The descriptor's length is rounded up to whole 16-byte chunks: its range check works on dwords, so a batch whose byte count is
    // not a multiple of 4 would lose its last bytes.  The base is 16-byte aligned and device allocations are page-granular, so the
    // (at most 15) bytes past the batch are readable; nothing of them is used.
    const auto rs = __builtin_amdgcn_make_buffer_rsrc(const_cast<uint8_t *>(g.frames), 0, (int)((((int64_t)g.n * h * w * 3) + 15) & ~15ll), 0x00020000);
    // stage 1: 16-byte loads of the RGB rows (aligned down)
    for (int i = tid; i < rows * QR_CHUNKS; i += 256) {
        const int r = i / QR_CHUNKS, c = i - r * QR_CHUNKS;
        const unsigned off = (unsigned)(((int64_t)img * h + gy0 + r) * w + gx0) * 3u;
        const unsigned a0 = off & ~15u;
        if (c == 0) rowmis[r] = (int)(off - a0);
        if ((int)(c * 16) < (int)(off - a0) + rowbytes) {
            typedef unsigned u32x4_t __attribute__((ext_vector_type(4)));
            const u32x4_t v = __builtin_amdgcn_raw_buffer_load_b128(rs, a0 + c * 16, 0, 0);
            *reinterpret_cast<u32x4_t *>(raw + QR_PAD + r * QR_S + c * 16) = v;
        }
    }
    __syncthreads();
    // stage 2: gray, four pixels (12 RGB bytes through four aligned dwords and a byte funnel shift) per thread, stored at region
    // coordinates (row y - (y0 - 2), column x - (x0 - 2)).  Columns outside the frame convert whatever bytes lie there (inside the
    // LDS array) and are overwritten below.
    const int cfirst = g.bgr ? 1868 : 4899, clast = g.bgr ? 4899 : 1868;
    const int dxr = (x0 - 2) - gx0;                         // 0, or -2 in the first tile column
    const int ry0 = gy0 - (y0 - 2);
    for (int i = tid; i < rows * QM_GROUPS; i += 256) {
        const int r = i / QM_GROUPS, c = i - r * QM_GROUPS;
        const int b0 = rowmis[r] + (4 * c + dxr) * 3;       // >= -6
        const int a0 = b0 & ~3, m = b0 & 3;
        const unsigned *p = reinterpret_cast<const unsigned *>(raw + QR_PAD + r * QR_S + a0);
        const unsigned d0 = p[0], d1 = p[1], d2 = p[2], d3 = p[3];
        const unsigned e0 = __builtin_amdgcn_alignbyte(d1, d0, m), e1 = __builtin_amdgcn_alignbyte(d2, d1, m), e2 = __builtin_amdgcn_alignbyte(d3, d2, m);
        const int q0 = ((int)(e0 & 255) * cfirst + (int)((e0 >> 8) & 255) * 9617 + (int)((e0 >> 16) & 255) * clast + 8192) >> 14;
        const int q1 = ((int)(e0 >> 24) * cfirst + (int)(e1 & 255) * 9617 + (int)((e1 >> 8) & 255) * clast + 8192) >> 14;
        const int q2 = ((int)((e1 >> 16) & 255) * cfirst + (int)(e1 >> 24) * 9617 + (int)(e2 & 255) * clast + 8192) >> 14;
        const int q3 = ((int)((e2 >> 8) & 255) * cfirst + (int)((e2 >> 16) & 255) * 9617 + (int)(e2 >> 24) * clast + 8192) >> 14;
        *reinterpret_cast<unsigned *>(gray + (ry0 + r) * QG_S + 4 * c) = (unsigned)q0 | ((unsigned)q1 << 8) | ((unsigned)q2 << 16) | ((unsigned)q3 << 24);
    }
    __syncthreads();
    // tiles on the frame's border: BORDER_REPLICATE into the part of the region that lies outside the frame (the Laplacian's
    // BORDER_REFLECT_101 is applied where it is computed)
    if (x0 == 0 || x0 + QT_W + 2 > w || y0 == 0 || y0 + QT_H + 2 > h) {
        for (int i = tid; i < QG_H * QG_W; i += 256) {
            const int ry = i / QG_W, rx = i - ry * QG_W;
            const int y = y0 - 2 + ry, x = x0 - 2 + rx;
            const int cy = min(max(y, 0), h - 1), cx = min(max(x, 0), w - 1);
            if (cy != y || cx != x) gray[ry * QG_S + rx] = gray[(cy - (y0 - 2)) * QG_S + (cx - (x0 - 2))];
        }
        __syncthreads();
    }
    // stage 3: Sobel magnitude + sector on the core and its 1-pixel rim (0 outside the frame), Laplacian sums on the core; a thread
    // takes four consecutive pixels of a row: six dword reads bring the 3 x 6 gray bytes they need, the Sobel is separable over the
    // six column sums / differences
    int s1 = 0;
    unsigned s2 = 0;
    for (int i = tid; i < QM_H * QM_GROUPS; i += 256) {
        const int my = i / QM_GROUPS, gq = i - my * QM_GROUPS;
        const int y = y0 - 1 + my;
        const unsigned *t = reinterpret_cast<const unsigned *>(gray + my * QG_S + 4 * gq);
        const unsigned tl = t[0], th = t[1], ml = t[QG_S / 4], mh = t[QG_S / 4 + 1], bl = t[2 * (QG_S / 4)], bh = t[2 * (QG_S / 4) + 1];
        int vs[6], vd[6], mid[6], top[6], bot[6];
#pragma unroll
        for (int k = 0; k < 6; ++k) {
            top[k] = byte_of(tl, th, k);
            mid[k] = byte_of(ml, mh, k);
            bot[k] = byte_of(bl, bh, k);
            vs[k] = top[k] + 2 * mid[k] + bot[k];
            vd[k] = bot[k] - top[k];
        }
        const bool yin = y >= 0 && y < h, ycore = my >= 1 && my <= QT_H;
        unsigned pk[4];
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            const int mx = 4 * gq + j, x = x0 - 1 + mx;
            unsigned packed = 0;
            if (yin && x >= 0 && x < w) {
                const int dx = vs[j + 2] - vs[j], dy = vd[j] + 2 * vd[j + 1] + vd[j + 2];
                const int ax = abs(dx), ay = abs(dy);
                // canny.cpp: y = |dy| << 15, tg22x = |dx| * TG22: horizontal gradient below 22.5 deg, vertical above 67.5 deg, else the
                // diagonal whose sign is that of dx * dy
                const int yy = ay << 15, tg22x = ax * TG22;
                int d;
                if (yy < tg22x) d = 0;
                else if (yy > tg22x + (ax << 16)) d = 1;
                else d = ((dx ^ dy) < 0) ? 3 : 2;
                packed = (unsigned)((ax + ay) | (d << 12));          // |dx| + |dy| <= 2040: 11 bits
                if (ycore && mx >= 1 && mx <= QT_W) {                // core pixel: BORDER_REFLECT_101 Laplacian
                    int up = top[j + 1], dn = bot[j + 1], lf = mid[j], rt = mid[j + 2];
                    const int up0 = up, lf0 = lf;
                    if (y == 0) up = dn;
                    if (y == h - 1) dn = up0;
                    if (x == 0) lf = rt;
                    if (x == w - 1) rt = lf0;
                    const int lap = up + dn + lf + rt - 4 * mid[j + 1];
                    s1 += lap;
                    s2 += (unsigned)(lap * lap);
                }
            }
            pk[j] = packed;
        }
        uint2 o;
        o.x = pk[0] | (pk[1] << 16);
        o.y = pk[2] | (pk[3] << 16);
        *reinterpret_cast<uint2 *>(magl + my * QM_S + 4 * gq) = o;
    }
    for (int o = 32; o > 0; o >>= 1) {
        s1 += __shfl_xor(s1, o, 64);
        s2 += __shfl_xor(s2, o, 64);          // a tile's sum of squares is below 2048 * 1020^2 < 2^32
    }
    if (lane == 0) {
        sh1[wave] = s1;
        sh2[wave] = s2;
    }
    __syncthreads();
    if (tid == 0) {
        // one slot per tile, summed by the frame's sweep workgroup: atomics on the frame's two sums made every tile of a frame queue
        // on one L2 address (0.45 of the kernel's 0.65 ms at 64 x 1080p)
        long long *slot = g.tile_sums + 2 * (((int64_t)img * gridDim.y + blockIdx.y) * gridDim.x + blockIdx.x);
        slot[0] = (long long)(sh1[0] + sh1[1] + sh1[2] + sh1[3]);
        slot[1] = (long long)((unsigned long long)sh2[0] + sh2[1] + sh2[2] + sh2[3]);
    }
    // stage 4: non-maximum suppression + double threshold on the core: a wave takes a tile row per step (lane = column), the two
    // ballots ARE the row's words of the candidate and strong planes
    unsigned long long *cand = g.cand + ((int64_t)img * h) * g.w64 + blockIdx.x;
    unsigned long long *strong = g.strong + ((int64_t)img * h) * g.w64 + blockIdx.x;
    const int x = x0 + lane;
#pragma unroll 2
    for (int it = 0; it < QT_H / 4; ++it) {
        const int ty = wave * (QT_H / 4) + it, y = y0 + ty;
        const int c = (ty + 1) * QM_S + lane + 1;
        const int pv = magl[c], v = pv & 0x0fff, d = pv >> 12;
        bool isc = false, iss = false;
        if (y < h && x < w && v > g.low) {
            const int off = d == 0 ? 1 : (d == 1 ? QM_S : (d == 2 ? QM_S + 1 : QM_S - 1));
            const int n1 = magl[c - off] & 0x0fff, n2 = magl[c + off] & 0x0fff;
            const bool peak = v > n1 && (d < 2 ? v >= n2 : v > n2);
            isc = peak;
            iss = peak && v > g.high;
        }
        const unsigned long long bc = __ballot(isc), bs = __ballot(iss);
        if (lane == 0 && y < h) {
            cand[(int64_t)y * g.w64] = bc;
            strong[(int64_t)y * g.w64] = bs;
        }
    }
}

// One row of the reconstruction: out = the runs of c that contain a bit of seed (seed is a subset of c).  Lane l holds word
// j * 64 + l of the row in c[j] / seed[j]; lanes past the row hold 0.
template <int KW>
__device__ __forceinline__ void fill_runs(const unsigned long long (&c)[KW], const unsigned long long (&seed)[KW], unsigned long long (&out)[KW], int lane) {
    // towards higher pixels: x = c + seed; the carry of a seed runs to the end of its run and flips exactly the bits it passes
    unsigned long long cin = 0;
#pragma unroll
    for (int j = 0; j < KW; ++j) {
        unsigned long long x = c[j] + seed[j];
        const unsigned long long G = __ballot(x < c[j]), P = __ballot(x == ~0ull);     // word generates / propagates a carry
        const unsigned long long X = G | P, s0 = X + G, s1 = s0 + cin;                 // carry-lookahead over the 64 lanes
        const unsigned long long carries = s1 ^ P;
        cin = (s0 < X || s1 < s0) ? 1 : 0;
        x += (carries >> lane) & 1;
        out[j] = ((x ^ c[j]) & c[j]) | seed[j];
    }
    // towards lower pixels: the same on bit-reversed words, lanes (and words) in reversed order
    cin = 0;
#pragma unroll
    for (int j = KW - 1; j >= 0; --j) {
        const unsigned long long rc = __brevll(c[j]), rsd = __brevll(seed[j]);
        unsigned long long x = rc + rsd;
        const unsigned long long G = __brevll(__ballot(x < rc)), P = __brevll(__ballot(x == ~0ull));
        const unsigned long long X = G | P, s0 = X + G, s1 = s0 + cin;
        const unsigned long long carries = s1 ^ P;
        cin = (s0 < X || s1 < s0) ? 1 : 0;
        x += (carries >> (63 - lane)) & 1;
        out[j] |= __brevll((x ^ rc) & rc);
    }
}

// One sweep over the rows [ya, yb) of a frame, downwards (seeded by row ya - 1) or upwards (seeded by row yb): every row becomes
// the runs of C that touch its own S or the 3-dilation of the neighbour row's S.  Rows are fetched a block (8 words per lane) ahead of their
// use.  Returns whether any word changed; `count` = the population of the band's S after the sweep.
template <int KW>
__device__ __forceinline__ bool sweep_band(const unsigned long long *__restrict__ C, unsigned long long *S, int h, int w64, int ya, int yb, bool down,
                                           const bool (&valid)[KW], int lane, int &count) {
    constexpr int PF = 8 / KW;            // rows in flight: the same number of registers for every row width
    const int nrows = yb - ya;
    unsigned long long prev[KW], cb[PF][KW], sb[PF][KW], cn[PF][KW], sn[PF][KW];
    {
        // the neighbour band's boundary row: another wave of this workgroup may be writing it (bits only ever get set, words are
        // written whole: a stale word costs a round, never a wrong bit); read at agent scope, i.e. from L2, where the write-through stores
        // of the other wave are once the barrier of the round has passed
        const int yn = down ? ya - 1 : yb;
#pragma unroll
        for (int j = 0; j < KW; ++j)
            prev[j] = (yn >= 0 && yn < h && valid[j]) ? __hip_atomic_load(S + (int64_t)yn * w64 + j * 64 + lane, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) : 0;
    }
    auto fetch = [&](int r0, unsigned long long (&cc)[PF][KW], unsigned long long (&ss)[PF][KW]) {
#pragma unroll
        for (int k = 0; k < PF; ++k) {
            const int r = r0 + k, y = down ? ya + r : yb - 1 - r;
#pragma unroll
            for (int j = 0; j < KW; ++j) {
                const bool ok = r < nrows && valid[j];
                cc[k][j] = ok ? C[(int64_t)y * w64 + j * 64 + lane] : 0;
                ss[k][j] = ok ? S[(int64_t)y * w64 + j * 64 + lane] : 0;
            }
        }
    };
    fetch(0, cb, sb);
    bool changed = false;
    count = 0;
    for (int r0 = 0; r0 < nrows; r0 += PF) {
        fetch(r0 + PF, cn, sn);
#pragma unroll
        for (int k = 0; k < PF; ++k) {
            const int r = r0 + k;
            if (r < nrows) {                                       // uniform
                const int y = down ? ya + r : yb - 1 - r;
                // the neighbour row's S, dilated by one pixel to either side (the bit that crosses a word comes by ballot)
                unsigned long long seed[KW], hi[KW], lo[KW], out[KW];
#pragma unroll
                for (int j = 0; j < KW; ++j) {
                    hi[j] = __ballot((prev[j] >> 63) != 0);
                    lo[j] = __ballot((prev[j] & 1) != 0);
                }
#pragma unroll
                for (int j = 0; j < KW; ++j) {
                    const unsigned long long from_lower = lane > 0 ? (hi[j] >> (lane - 1)) & 1 : (j > 0 ? hi[j > 0 ? j - 1 : 0] >> 63 : 0);
                    const unsigned long long from_upper = lane < 63 ? (lo[j] >> (lane + 1)) & 1 : (j + 1 < KW ? lo[j + 1 < KW ? j + 1 : j] & 1 : 0);
                    const unsigned long long dil = prev[j] | (prev[j] << 1) | from_lower | (prev[j] >> 1) | (from_upper << 63);
                    seed[j] = (sb[k][j] | dil) & cb[k][j];
                }
                fill_runs<KW>(cb[k], seed, out, lane);
#pragma unroll
                for (int j = 0; j < KW; ++j) {
                    if (out[j] != sb[k][j]) {
                        changed = true;
                        S[(int64_t)y * w64 + j * 64 + lane] = out[j];
                    }
                    prev[j] = out[j];
                    count += __popcll(out[j]);
                }
            }
        }
#pragma unroll
        for (int k = 0; k < PF; ++k)
#pragma unroll
            for (int j = 0; j < KW; ++j) {
                cb[k][j] = cn[k][j];
                sb[k][j] = sn[k][j];
            }
    }
    return __any(changed) != 0;
}

// Hysteresis by reconstruction.  One workgroup per frame, one wave per horizontal band of it: a wave alternates down / up sweeps over
// its band until one changes nothing (a sweep without change right after one in the other direction leaves every row closed against
// both neighbours), the workgroup repeats such rounds until no band changed in a whole round - in that round nothing was written, so
// every band was checked against final boundary rows.
template <int KW>
__global__ __launch_bounds__(KW == 1 ? 1024 : 512) void quality_sweep_kernel(const unsigned long long *__restrict__ cand_all, unsigned long long *strong_all,
                                                                             int h, int w64, const long long *__restrict__ tile_sums, int tiles,
                                                                             long long *__restrict__ lap_sums, long long *__restrict__ edge_count) {
    __shared__ int flag, counts[16];
    __shared__ long long part[16][2];
    const int img = blockIdx.x, tid = threadIdx.x, lane = tid & 63, wave = tid >> 6, nb = blockDim.x >> 6;
    const unsigned long long *C = cand_all + (int64_t)img * h * w64;
    unsigned long long *S = strong_all + (int64_t)img * h * w64;
    const int ya = (int)((int64_t)h * wave / nb), yb = (int)((int64_t)h * (wave + 1) / nb);
    bool valid[KW];
#pragma unroll
    for (int j = 0; j < KW; ++j) valid[j] = j * 64 + lane < w64;
    if (tid == 0) flag = 0;
    __syncthreads();
    int count = 0;
    for (;;) {
        bool band_changed = false;
        for (int sweep = 0;; ++sweep) {
            const bool ch = sweep_band<KW>(C, S, h, w64, ya, yb, (sweep & 1) == 0, valid, lane, count);
            band_changed |= ch;
            if (!ch && sweep >= 1) break;
        }
        if (band_changed && lane == 0) flag = 1;
        __syncthreads();                     // also orders this round's stores before the next round's boundary loads
        const int f = flag;
        __syncthreads();
        if (!f) break;
        if (tid == 0) flag = 0;
        __syncthreads();
    }
    // the frame's Laplacian sums from its tiles' slots
    long long s1 = 0, s2 = 0;
    for (int i = tid; i < tiles; i += blockDim.x) {
        s1 += tile_sums[2 * ((int64_t)img * tiles + i)];
        s2 += tile_sums[2 * ((int64_t)img * tiles + i) + 1];
    }
    for (int o = 32; o > 0; o >>= 1) {
        count += __shfl_xor(count, o, 64);
        s1 += __shfl_xor(s1, o, 64);
        s2 += __shfl_xor(s2, o, 64);
    }
    if (lane == 0) {
        counts[wave] = count;
        part[wave][0] = s1;
        part[wave][1] = s2;
    }
    __syncthreads();
    if (tid == 0) {
        long long total = 0, t1 = 0, t2 = 0;
        for (int i = 0; i < nb; ++i) {
            total += counts[i];
            t1 += part[i][0];
            t2 += part[i][1];
        }
        edge_count[img] = total;
        lap_sums[2 * img] = t1;
        lap_sums[2 * img + 1] = t2;
    }
}

}  // namespace

extern "C" {

int64_t ivr_frame_quality_scratch_bytes(int n, int h, int w) {
    return (int64_t)n * h * ivr_ceil_div(w, 64) * 16 + (int64_t)n * ivr_ceil_div(h, QT_H) * ivr_ceil_div(w, QT_W) * 16 + 1024;
}

int ivr_frame_quality(ivr_ctx *ctx, const uint8_t *frames, int n, int h, int w, int bgr, int canny_low, int canny_high, int64_t *lap_sums,
                      int64_t *edge_count, ivr_stream stream) {
    IVR_REQUIRE(ctx && (n == 0 || (frames && lap_sums && edge_count)), "ivr_frame_quality: NULL argument");
    IVR_REQUIRE(n >= 0 && h >= 1 && w >= 1 && (int64_t)h * w < (1ll << 29), "ivr_frame_quality: n=%d h=%d w=%d", n, h, w);
    IVR_REQUIRE(w <= kMaxRowWords * 64, "ivr_frame_quality: frames wider than %d pixels are not supported (w=%d)", kMaxRowWords * 64, w);
    IVR_REQUIRE(canny_low >= 0 && canny_high >= canny_low, "ivr_frame_quality: thresholds low=%d high=%d", canny_low, canny_high);
    if (n == 0) return IVR_OK;
    IVR_REQUIRE(reinterpret_cast<uintptr_t>(frames) % 16 == 0, "ivr_frame_quality: frames must be 16-byte aligned");
    IVR_HIP(hipSetDevice(ctx->device));
    hipStream_t s = (hipStream_t)stream;
    // the tile kernel addresses the batch through one buffer descriptor (32-bit byte offsets): at most 2 GiB of pixels per
    // launch chain, longer batches go in slices of whole frames
    const int64_t frame_bytes = (int64_t)h * w * 3;
    // slices start on 16-byte boundaries: a whole number of `align` frames each
    int64_t align = 16, fb = frame_bytes;
    while (align > 1 && fb % 2 == 0) {
        align /= 2;
        fb /= 2;
    }
    const int per_chain = (int)std::max<int64_t>(align, std::min<int64_t>(n, (int64_t)0x7ffffff0 / frame_bytes) / align * align);
    IVR_REQUIRE((int64_t)std::min(per_chain, n) * frame_bytes < 0x7ffffff0, "ivr_frame_quality: %d x %d frames are too large for one launch chain", h, w);
    std::lock_guard<std::mutex> enqueue(ctx->enqueue_mu);        // the launches below share the stream's scratch block
    void *scratch = nullptr;
    int rc = ivr_ctx_scratch(ctx, s, (size_t)ivr_frame_quality_scratch_bytes(per_chain, h, w), &scratch);
    if (rc != IVR_OK) return rc;
    const int w64 = ivr_ceil_div(w, 64);
    for (int f0 = 0; f0 < n; f0 += per_chain) {
        const int nf = std::min(per_chain, n - f0);
        const int64_t npix = (int64_t)nf * h * w;
        QualityArgs a;
        a.frames = frames + (int64_t)f0 * frame_bytes;
        a.n = nf;
        a.h = h;
        a.w = w;
        a.bgr = bgr;
        a.low = canny_low;
        a.high = canny_high;
        a.w64 = w64;
        a.cand = reinterpret_cast<unsigned long long *>(scratch);
        a.strong = a.cand + (int64_t)nf * h * w64;
        a.tile_sums = reinterpret_cast<long long *>(a.strong + (int64_t)nf * h * w64);
        const int tiles = w64 * ivr_ceil_div(h, QT_H);
        long long *laps = reinterpret_cast<long long *>(lap_sums) + 2 * (int64_t)f0;
        {
            // algorithmic bytes: 3 read + 2 bits written per pixel
            IvrProf prof("quality_tile", s, (double)npix * 3.25);
            hipLaunchKernelGGL(quality_tile_kernel, dim3((unsigned)w64, (unsigned)ivr_ceil_div(h, QT_H), (unsigned)nf), dim3(256), 0, s, a);
        }
        IVR_LAUNCH_CHECK();
        {
            IvrProf prof("quality_hysteresis", s, (double)nf * h * w64 * 24);
            long long *out = reinterpret_cast<long long *>(edge_count) + f0;
            // bands of at least 16 rows, at most 16 (8 for the wide-row variants: their rows take 2 - 4 words per lane) waves per frame
            const int nb = std::max(1, std::min(w64 <= 64 ? 16 : 8, h / 16));
            if (w64 <= 64) hipLaunchKernelGGL(quality_sweep_kernel<1>, dim3(nf), dim3(64 * nb), 0, s, a.cand, a.strong, h, w64, a.tile_sums, tiles, laps, out);
            else if (w64 <= 128) hipLaunchKernelGGL(quality_sweep_kernel<2>, dim3(nf), dim3(64 * nb), 0, s, a.cand, a.strong, h, w64, a.tile_sums, tiles, laps, out);
            else hipLaunchKernelGGL(quality_sweep_kernel<4>, dim3(nf), dim3(64 * nb), 0, s, a.cand, a.strong, h, w64, a.tile_sums, tiles, laps, out);
        }
        IVR_LAUNCH_CHECK();
    }
    return IVR_OK;
}

}  // extern "C"
