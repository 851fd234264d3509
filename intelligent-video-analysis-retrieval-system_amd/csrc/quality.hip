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
// Round 3: ONE tiled kernel does gray + Laplacian sums + Sobel + non-maximum suppression.  A workgroup owns a 64 x 32 pixel tile:
// the RGB bytes of the tile and a 2-pixel rim are fetched with 16-byte buffer loads into LDS, gray and the packed (magnitude,
// sector) plane live only in LDS, and what leaves the chip per pixel is one mark byte (+ 4 bytes per strong pixel pushed on the
// frame's work list): 3 B read + 1 B written per pixel instead of the 12 B of the four-kernel chain of round 2.
// The hysteresis is a work list (canny.cpp's stack, level-synchronous here): it starts from the strong pixels the tiles pushed,
// promotes their weak 8-neighbours and pushes those in turn, so its work is proportional to the number of edge pixels, not to
// sweeps x pixels; the final length of the list IS the edge count.
#include "ivr_common.h"

#include <algorithm>

namespace {

constexpr int TG22 = 13573;          // (int)(0.4142135623730950488 * (1 << 15) + 0.5)
constexpr int QT_W = 64, QT_H = 32;  // core tile
constexpr int QG_H = QT_H + 4, QG_S = 72;                    // gray region (core + 2-pixel rim: 68 columns), LDS row stride
constexpr int QM_W = QT_W + 2, QM_H = QT_H + 2, QM_S = 68;   // magnitude region (1-pixel rim)
constexpr int QR_CHUNKS = 14, QR_S = QR_CHUNKS * 16;         // raw RGB row in LDS: up to 15 B of misalignment + 3 * 68 B

struct QualityArgs {
    const uint8_t *frames;
    int n, h, w, bgr, low, high;
    uint8_t *mark;                 // [n][h*w]: 2 = edge, 0 = weak candidate, 1 = not an edge
    int *queue;                    // [n][h*w] work list of edge pixels (y * w + x)
    int *qtail;                    // [n]
    long long *lap_sums;           // [n][2]
};

__global__ __launch_bounds__(256) void quality_tile_kernel(QualityArgs g) {
    __shared__ __attribute__((aligned(16))) uint8_t raw[QG_H * QR_S];
    __shared__ uint8_t gray[QG_H * QG_S];
    __shared__ uint16_t magl[QM_H * QM_S];
    __shared__ int rowmis[QG_H];
    __shared__ long long sh1[4], sh2[4];
    __shared__ int seeds[QT_W * QT_H], nseeds, seed_base;
    const int tid = threadIdx.x, lane = tid & 63;
    const int img = blockIdx.z, x0 = blockIdx.x * QT_W, y0 = blockIdx.y * QT_H;
    const int h = g.h, w = g.w;
    if (tid == 0) nseeds = 0;
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
            *reinterpret_cast<u32x4_t *>(raw + r * QR_S + c * 16) = v;
        }
    }
    __syncthreads();
    // stage 2: gray for the in-frame part of the region, stored at region coordinates (row y - (y0 - 2), column x - (x0 - 2))
    const int cols = gx1 - gx0;
    for (int i = tid; i < rows * cols; i += 256) {
        const int r = i / cols, c = i - r * cols;
        const uint8_t *p = raw + r * QR_S + rowmis[r] + c * 3;
        const int c0 = p[0], c1 = p[1], c2 = p[2];
        const int rr = g.bgr ? c2 : c0, bb = g.bgr ? c0 : c2;
        gray[(gy0 + r - (y0 - 2)) * QG_S + (gx0 + c - (x0 - 2))] = (uint8_t)((rr * 4899 + c1 * 9617 + bb * 1868 + 8192) >> 14);
    }
    __syncthreads();
    auto G = [&](int y, int x) -> int { return gray[(y - (y0 - 2)) * QG_S + (x - (x0 - 2))]; };     // y, x inside the frame
    // stage 3: Sobel magnitude + sector on the core and its 1-pixel rim (0 outside the frame), Laplacian sums on the core
    long long s1 = 0, s2 = 0;
    for (int i = tid; i < QM_H * QM_W; i += 256) {
        const int my = i / QM_W, mx = i - my * QM_W;
        const int y = y0 - 1 + my, x = x0 - 1 + mx;
        uint16_t packed = 0;
        if (y >= 0 && y < h && x >= 0 && x < w) {
            const int ym = max(y - 1, 0), yp = min(y + 1, h - 1), xm = max(x - 1, 0), xp = min(x + 1, w - 1);     // BORDER_REPLICATE
            const int a00 = G(ym, xm), a01 = G(ym, x), a02 = G(ym, xp), a10 = G(y, xm), a12 = G(y, xp);
            const int a20 = G(yp, xm), a21 = G(yp, x), a22 = G(yp, xp);
            const int dx = (a02 + 2 * a12 + a22) - (a00 + 2 * a10 + a20);
            const int dy = (a20 + 2 * a21 + a22) - (a00 + 2 * a01 + a02);
            const int ax = abs(dx), ay = abs(dy);
            // canny.cpp: y = |dy| << 15, tg22x = |dx| * TG22: horizontal gradient below 22.5 deg, vertical above 67.5 deg, else the
            // diagonal whose sign is that of dx * dy
            const int yy = ay << 15, tg22x = ax * TG22;
            int d;
            if (yy < tg22x) d = 0;
            else if (yy > tg22x + (ax << 16)) d = 1;
            else d = ((dx ^ dy) < 0) ? 3 : 2;
            packed = (uint16_t)((ax + ay) | (d << 12));            // |dx| + |dy| <= 2040: 11 bits
            if (my >= 1 && my <= QT_H && mx >= 1 && mx <= QT_W) {  // core pixel: BORDER_REFLECT_101 Laplacian
                const int ymr = y > 0 ? y - 1 : (h > 1 ? 1 : 0), ypr = y + 1 < h ? y + 1 : (h > 1 ? h - 2 : 0);
                const int xmr = x > 0 ? x - 1 : (w > 1 ? 1 : 0), xpr = x + 1 < w ? x + 1 : (w > 1 ? w - 2 : 0);
                const int lap = G(ymr, x) + G(ypr, x) + G(y, xmr) + G(y, xpr) - 4 * G(y, x);
                s1 += lap;
                s2 += (long long)lap * lap;
            }
        }
        magl[my * QM_S + mx] = packed;
    }
    for (int o = 32; o > 0; o >>= 1) {
        s1 += __shfl_xor(s1, o, 64);
        s2 += __shfl_xor(s2, o, 64);
    }
    if (lane == 0) {
        sh1[tid >> 6] = s1;
        sh2[tid >> 6] = s2;
    }
    __syncthreads();
    if (tid == 0) {
        atomicAdd(reinterpret_cast<unsigned long long *>(g.lap_sums + 2 * img), (unsigned long long)(sh1[0] + sh1[1] + sh1[2] + sh1[3]));
        atomicAdd(reinterpret_cast<unsigned long long *>(g.lap_sums + 2 * img + 1), (unsigned long long)(sh2[0] + sh2[1] + sh2[2] + sh2[3]));
    }
    // stage 4: non-maximum suppression + double threshold on the core, four consecutive pixels per thread
    uint8_t *mark = g.mark + (int64_t)img * h * w;
    int *queue = g.queue + (int64_t)img * h * w;
    auto M = [&](int my, int mx) -> int { return magl[my * QM_S + mx] & 0x0fff; };      // region coordinates
    for (int i = tid; i < QT_H * (QT_W / 4); i += 256) {
        const int ty = i / (QT_W / 4), tx = (i - ty * (QT_W / 4)) * 4;
        const int y = y0 + ty;
        uint8_t out[4];
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            const int x = x0 + tx + j, my = ty + 1, mx = tx + j + 1;
            uint8_t o = 1;
            if (y < h && x < w) {
                const int pv = magl[my * QM_S + mx], v = pv & 0x0fff, d = pv >> 12;
                if (v > g.low) {
                    bool peak;
                    if (d == 0) peak = v > M(my, mx - 1) && v >= M(my, mx + 1);
                    else if (d == 1) peak = v > M(my - 1, mx) && v >= M(my + 1, mx);
                    else {
                        const int s = d == 3 ? -1 : 1;
                        peak = v > M(my - 1, mx - s) && v > M(my + 1, mx + s);
                    }
                    if (peak) o = v > g.high ? 2 : 0;
                }
            }
            out[j] = o;
            if (o == 2) seeds[atomicAdd(&nseeds, 1)] = y * w + x;      // strong pixels seed the frame's work list (collected in LDS)
        }
        if (y < h) {
            const int x = x0 + tx;
            if ((w & 3) == 0 && x + 3 < w) {
                *reinterpret_cast<uint32_t *>(mark + (int64_t)y * w + x) = out[0] | (out[1] << 8) | (out[2] << 16) | ((uint32_t)out[3] << 24);
            } else {
#pragma unroll
                for (int j = 0; j < 4; ++j)
                    if (x + j < w) mark[(int64_t)y * w + x + j] = out[j];
            }
        }
    }
    // one global atomic per TILE reserves the tile's stretch of the frame's work list (an atomic per wave made every workgroup of
    // a frame queue on one address: 15 ms per 64 frames of 1080p)
    __syncthreads();
    if (tid == 0) seed_base = nseeds ? atomicAdd(g.qtail + img, nseeds) : 0;
    __syncthreads();
    for (int i = tid; i < nseeds; i += 256) queue[seed_base + i] = seeds[i];
}

// First level of the hysteresis, by the whole chip: the strong pixels the tiles listed ([0, n0) of every frame's list) promote their
// weak neighbours.  No dependency between them, so this level - the bulk of the list - does not have to run on one CU per frame.
__global__ __launch_bounds__(256) void quality_seed_kernel(uint8_t *__restrict__ mark_all, int *__restrict__ queue_all, int *__restrict__ qtail,
                                                           const int *__restrict__ n0_all, int h, int w) {
    __shared__ int found[256 * 8], nfound, base;
    const int img = blockIdx.y, tid = threadIdx.x;
    uint8_t *m = mark_all + (int64_t)img * h * w;
    int *queue = queue_all + (int64_t)img * h * w;
    const int n0 = n0_all[img];
    for (int i0 = blockIdx.x * 256; i0 < n0; i0 += gridDim.x * 256) {
        if (tid == 0) nfound = 0;
        __syncthreads();
        const int idx = i0 + tid;
        if (idx < n0) {
            const int p = queue[idx], y = p / w, x = p - y * w;
#pragma unroll
            for (int nb = 0; nb < 8; ++nb) {
                const int dy = nb < 3 ? -1 : (nb < 5 ? 0 : 1);
                const int dx = nb < 3 ? nb - 1 : (nb == 3 ? -1 : (nb == 4 ? 1 : nb - 6));
                const int yy = y + dy, xx = x + dx;
                if (yy >= 0 && yy < h && xx >= 0 && xx < w && m[(int64_t)yy * w + xx] == 0) {
                    const uintptr_t a = reinterpret_cast<uintptr_t>(m + (int64_t)yy * w + xx);
                    const unsigned sh = (unsigned)(a & 3) * 8;
                    const unsigned old = atomicOr(reinterpret_cast<unsigned *>(a & ~(uintptr_t)3), 2u << sh);
                    if (((old >> sh) & 0xff) == 0) found[atomicAdd(&nfound, 1)] = yy * w + xx;
                }
            }
        }
        __syncthreads();
        if (tid == 0) base = nfound ? atomicAdd(qtail + img, nfound) : 0;
        __syncthreads();
        for (int i = tid; i < nfound; i += 256) queue[base + i] = found[i];
        __syncthreads();
    }
}

// Hysteresis: weak candidates 8-connected to an edge become edges.  One workgroup per frame walks the frame's work list level by
// level: every listed pixel promotes its weak neighbours (an atomic OR on the byte's word decides who promoted it) and appends
// them.  Marks only ever go 0 -> 2, so a stale read of 0 merely costs one failed atomic.  The list ends up holding every edge
// pixel exactly once: its length is the edge count.
__global__ __launch_bounds__(1024) void quality_hysteresis_kernel(uint8_t *__restrict__ mark_all, int *__restrict__ queue_all,
                                                                  int *__restrict__ qtail, const int *__restrict__ n0_all, int h, int w,
                                                                  long long *__restrict__ edge_count) {
    const int img = blockIdx.x;
    uint8_t *m = mark_all + (int64_t)img * h * w;
    int *queue = queue_all + (int64_t)img * h * w;
    __shared__ int s_tail;
    const int tid = threadIdx.x, lane = tid & 63;
    if (tid == 0) s_tail = qtail[img];
    __syncthreads();
    int head = n0_all[img];                               // the first level was expanded by quality_seed_kernel
    for (;;) {
        const int tail = s_tail;
        __syncthreads();                                  // everyone has read this level's end before anyone appends
        if (head >= tail) break;
        for (int i0 = head; i0 < tail; i0 += 1024) {
            const int idx = i0 + tid;
            int p = -1;
            if (idx < tail) p = queue[idx];
            const int y = p >= 0 ? p / w : 0, x = p >= 0 ? p - y * w : 0;
#pragma unroll
            for (int nb = 0; nb < 8; ++nb) {
                const int dy = nb < 3 ? -1 : (nb < 5 ? 0 : 1);
                const int dx = nb < 3 ? nb - 1 : (nb == 3 ? -1 : (nb == 4 ? 1 : nb - 6));
                const int yy = y + dy, xx = x + dx;
                bool won = false;
                if (p >= 0 && yy >= 0 && yy < h && xx >= 0 && xx < w && m[(int64_t)yy * w + xx] == 0) {
                    const uintptr_t a = reinterpret_cast<uintptr_t>(m + (int64_t)yy * w + xx);
                    const unsigned sh = (unsigned)(a & 3) * 8;
                    const unsigned old = atomicOr(reinterpret_cast<unsigned *>(a & ~(uintptr_t)3), 2u << sh);
                    won = ((old >> sh) & 0xff) == 0;
                }
                const unsigned long long b = __ballot(won);
                if (b) {
                    int base = 0;
                    const int leader = __ffsll((long long)b) - 1;
                    if (lane == leader) base = atomicAdd(&s_tail, __popcll(b));
                    base = __shfl(base, leader, 64);
                    if (won) queue[base + __popcll(b & ((1ull << lane) - 1))] = yy * w + xx;
                }
            }
        }
        head = tail;
        __threadfence_block();
        __syncthreads();                                  // this level's appends are visible, s_tail is final
    }
    if (tid == 0) edge_count[img] = s_tail;
}

}  // namespace

extern "C" {

int64_t ivr_frame_quality_scratch_bytes(int n, int h, int w) { return (int64_t)n * h * w * 5 + (int64_t)n * 8 + 1024; }

int ivr_frame_quality(ivr_ctx *ctx, const uint8_t *frames, int n, int h, int w, int bgr, int canny_low, int canny_high, int64_t *lap_sums,
                      int64_t *edge_count, ivr_stream stream) {
    IVR_REQUIRE(ctx && (n == 0 || (frames && lap_sums && edge_count)), "ivr_frame_quality: NULL argument");
    IVR_REQUIRE(n >= 0 && h >= 1 && w >= 1 && (int64_t)h * w < (1ll << 29), "ivr_frame_quality: n=%d h=%d w=%d", n, h, w);
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
    IVR_HIP(hipMemsetAsync(lap_sums, 0, (size_t)n * 16, s));
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
        a.queue = reinterpret_cast<int *>(scratch);
        a.qtail = a.queue + npix;
        int *n0 = a.qtail + nf;                          // snapshot of the list lengths after the tile kernel = the strong pixels
        a.mark = reinterpret_cast<uint8_t *>(n0 + nf);
        a.lap_sums = reinterpret_cast<long long *>(lap_sums) + 2 * (int64_t)f0;
        IVR_HIP(hipMemsetAsync(a.qtail, 0, (size_t)nf * 4, s));
        {
            // algorithmic bytes: 3 read + 1 mark written per pixel
            IvrProf prof("quality_tile", s, (double)npix * 4);
            hipLaunchKernelGGL(quality_tile_kernel, dim3((unsigned)ivr_ceil_div(w, QT_W), (unsigned)ivr_ceil_div(h, QT_H), (unsigned)nf), dim3(256), 0,
                               s, a);
        }
        IVR_LAUNCH_CHECK();
        IVR_HIP(hipMemcpyAsync(n0, a.qtail, (size_t)nf * 4, hipMemcpyDeviceToDevice, s));
        {
            IvrProf prof("quality_hysteresis", s, (double)npix, true);
            const unsigned per_frame = (unsigned)std::max<int64_t>(1, std::min<int64_t>(ivr_ceil_div((int64_t)h * w, 256 * 16), 4096 / std::max(nf, 1) + 1));
            hipLaunchKernelGGL(quality_seed_kernel, dim3(per_frame, (unsigned)nf), dim3(256), 0, s, a.mark, a.queue, a.qtail, n0, h, w);
            hipLaunchKernelGGL(quality_hysteresis_kernel, dim3(nf), dim3(1024), 0, s, a.mark, a.queue, a.qtail, n0, h, w,
                               reinterpret_cast<long long *>(edge_count) + f0);
        }
        IVR_LAUNCH_CHECK();
    }
    return IVR_OK;
}

}  // extern "C"
