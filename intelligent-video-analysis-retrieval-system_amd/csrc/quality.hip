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
// All of it is byte / integer work bound by HBM traffic (3 B read per pixel, then 1-2 B planes), except the hysteresis, which is
// a fixed-point iteration: one workgroup per frame sweeps its mark plane until nothing changes (many frames = many workgroups).
#include "ivr_common.h"

namespace {

constexpr int TG22 = 13573;          // (int)(0.4142135623730950488 * (1 << 15) + 0.5)

__global__ __launch_bounds__(256) void quality_gray_kernel(const uint8_t *__restrict__ src, uint8_t *__restrict__ gray, int64_t npix, int bgr) {
    const int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x;
    if (i >= npix) return;
    const uint8_t *p = src + i * 3;
    const int c0 = p[0], c1 = p[1], c2 = p[2];
    const int r = bgr ? c2 : c0, b = bgr ? c0 : c2;
    gray[i] = (uint8_t)((r * 4899 + c1 * 9617 + b * 1868 + 8192) >> 14);
}

// one thread per pixel: Laplacian sums (block reduction + one 64-bit atomic pair per block) and the Sobel magnitude / sector
__global__ __launch_bounds__(256) void quality_grad_kernel(const uint8_t *__restrict__ gray, int h, int w, uint16_t *__restrict__ mag,
                                                           uint8_t *__restrict__ dir, long long *__restrict__ lap_sums) {
    const int img = blockIdx.y;
    const int64_t base = (int64_t)img * h * w;
    const int idx = blockIdx.x * 256 + threadIdx.x;
    long long s1 = 0, s2 = 0;
    if (idx < h * w) {
        const int y = idx / w, x = idx - y * w;
        const uint8_t *g = gray + base;
        // BORDER_REFLECT_101 (Laplacian) and BORDER_REPLICATE (the Sobel inside Canny) differ only at the frame's rim
        const int ym_r = y > 0 ? y - 1 : (h > 1 ? 1 : 0), yp_r = y + 1 < h ? y + 1 : (h > 1 ? h - 2 : 0);
        const int xm_r = x > 0 ? x - 1 : (w > 1 ? 1 : 0), xp_r = x + 1 < w ? x + 1 : (w > 1 ? w - 2 : 0);
        const int c = g[(int64_t)y * w + x];
        const int lap = g[(int64_t)ym_r * w + x] + g[(int64_t)yp_r * w + x] + g[(int64_t)y * w + xm_r] + g[(int64_t)y * w + xp_r] - 4 * c;
        s1 = lap;
        s2 = (long long)lap * lap;
        const int ym = max(y - 1, 0), yp = min(y + 1, h - 1), xm = max(x - 1, 0), xp = min(x + 1, w - 1);
        const int a00 = g[(int64_t)ym * w + xm], a01 = g[(int64_t)ym * w + x], a02 = g[(int64_t)ym * w + xp];
        const int a10 = g[(int64_t)y * w + xm], a12 = g[(int64_t)y * w + xp];
        const int a20 = g[(int64_t)yp * w + xm], a21 = g[(int64_t)yp * w + x], a22 = g[(int64_t)yp * w + xp];
        const int dx = (a02 + 2 * a12 + a22) - (a00 + 2 * a10 + a20);
        const int dy = (a20 + 2 * a21 + a22) - (a00 + 2 * a01 + a02);
        const int ax = abs(dx), ay = abs(dy);
        mag[base + idx] = (uint16_t)(ax + ay);
        // canny.cpp: y = |dy| << 15, tg22x = |dx| * TG22: horizontal gradient below 22.5 deg, vertical above 67.5 deg, else the
        // diagonal whose sign is that of dx * dy
        const int yy = ay << 15, tg22x = ax * TG22;
        uint8_t d;
        if (yy < tg22x) d = 0;
        else if (yy > tg22x + (ax << 16)) d = 1;
        else d = ((dx ^ dy) < 0) ? 3 : 2;
        dir[base + idx] = d;
    }
    // block reduction of the Laplacian sums
    __shared__ long long sh1[4], sh2[4];
    for (int o = 32; o > 0; o >>= 1) {
        s1 += __shfl_xor(s1, o, 64);
        s2 += __shfl_xor(s2, o, 64);
    }
    if ((threadIdx.x & 63) == 0) {
        sh1[threadIdx.x >> 6] = s1;
        sh2[threadIdx.x >> 6] = s2;
    }
    __syncthreads();
    if (threadIdx.x == 0) {
        atomicAdd(reinterpret_cast<unsigned long long *>(lap_sums + 2 * img), (unsigned long long)(sh1[0] + sh1[1] + sh1[2] + sh1[3]));
        atomicAdd(reinterpret_cast<unsigned long long *>(lap_sums + 2 * img + 1), (unsigned long long)(sh2[0] + sh2[1] + sh2[2] + sh2[3]));
    }
}

// non-maximum suppression + double threshold: mark 2 = strong edge, 0 = weak candidate, 1 = not an edge
__global__ __launch_bounds__(256) void quality_nms_kernel(const uint16_t *__restrict__ mag, const uint8_t *__restrict__ dir, int h, int w, int low,
                                                          int high, uint8_t *__restrict__ mark) {
    const int img = blockIdx.y;
    const int64_t base = (int64_t)img * h * w;
    const int idx = blockIdx.x * 256 + threadIdx.x;
    if (idx >= h * w) return;
    const int y = idx / w, x = idx - y * w;
    const uint16_t *m = mag + base;
    auto at = [&](int yy, int xx) -> int { return (yy < 0 || yy >= h || xx < 0 || xx >= w) ? 0 : (int)m[(int64_t)yy * w + xx]; };
    const int v = m[idx];
    uint8_t out = 1;
    if (v > low) {
        const int d = dir[base + idx];
        bool peak;
        if (d == 0) peak = v > at(y, x - 1) && v >= at(y, x + 1);
        else if (d == 1) peak = v > at(y - 1, x) && v >= at(y + 1, x);
        else {
            const int s = d == 3 ? -1 : 1;
            peak = v > at(y - 1, x - s) && v > at(y + 1, x + s);
        }
        if (peak) out = v > high ? 2 : 0;
    }
    mark[base + idx] = out;
}

// hysteresis: weak candidates 8-connected to a strong edge become edges.  One workgroup per frame; a sweep visits runs of 16
// pixels left-to-right and back, so a chain advances a whole run per sweep horizontally and one row vertically; the sweeps
// repeat until one changes nothing (marks only ever go 0 -> 2, so reading a neighbour's stale 0 merely defers it to the next sweep).
__global__ __launch_bounds__(1024) void quality_hysteresis_kernel(uint8_t *__restrict__ mark, int h, int w, long long *__restrict__ edge_count) {
    const int img = blockIdx.x;
    uint8_t *m = mark + (int64_t)img * h * w;
    __shared__ int changed, total;
    const int runs_per_row = (w + 15) / 16, nruns = h * runs_per_row;
    auto strong_near = [&](int y, int x) -> bool {
        for (int dy = -1; dy <= 1; ++dy) {
            const int yy = y + dy;
            if (yy < 0 || yy >= h) continue;
            for (int dx = -1; dx <= 1; ++dx) {
                const int xx = x + dx;
                if ((dx | dy) == 0 || xx < 0 || xx >= w) continue;
                if (m[(int64_t)yy * w + xx] == 2) return true;
            }
        }
        return false;
    };
    for (int sweep = 0; sweep < h * w; ++sweep) {            // terminates long before: every productive sweep adds an edge pixel
        if (threadIdx.x == 0) changed = 0;
        __syncthreads();
        int mine = 0;
        for (int r = threadIdx.x; r < nruns; r += 1024) {
            const int y = r / runs_per_row, x0 = (r - y * runs_per_row) * 16, x1 = min(x0 + 16, w);
            for (int x = x0; x < x1; ++x)
                if (m[(int64_t)y * w + x] == 0 && strong_near(y, x)) {
                    m[(int64_t)y * w + x] = 2;
                    mine = 1;
                }
            for (int x = x1 - 1; x >= x0; --x)
                if (m[(int64_t)y * w + x] == 0 && strong_near(y, x)) {
                    m[(int64_t)y * w + x] = 2;
                    mine = 1;
                }
        }
        if (mine) atomicOr(&changed, 1);
        __threadfence_block();
        __syncthreads();
        const int any = changed;
        __syncthreads();
        if (!any) break;
    }
    if (threadIdx.x == 0) total = 0;
    __syncthreads();
    int cnt = 0;
    for (int i = threadIdx.x; i < h * w; i += 1024) cnt += m[i] == 2;
    for (int o = 32; o > 0; o >>= 1) cnt += __shfl_xor(cnt, o, 64);
    if ((threadIdx.x & 63) == 0 && cnt) atomicAdd(&total, cnt);
    __syncthreads();
    if (threadIdx.x == 0) edge_count[img] = total;
}

}  // namespace

extern "C" {

int64_t ivr_frame_quality_scratch_bytes(int n, int h, int w) { return (int64_t)n * h * w * 5 + 1024; }

int ivr_frame_quality(ivr_ctx *ctx, const uint8_t *frames, int n, int h, int w, int bgr, int canny_low, int canny_high, int64_t *lap_sums,
                      int64_t *edge_count, ivr_stream stream) {
    IVR_REQUIRE(ctx && (n == 0 || (frames && lap_sums && edge_count)), "ivr_frame_quality: NULL argument");
    IVR_REQUIRE(n >= 0 && h >= 1 && w >= 1 && (int64_t)h * w < (1ll << 30), "ivr_frame_quality: n=%d h=%d w=%d", n, h, w);
    IVR_REQUIRE(canny_low >= 0 && canny_high >= canny_low, "ivr_frame_quality: thresholds low=%d high=%d", canny_low, canny_high);
    if (n == 0) return IVR_OK;
    IVR_HIP(hipSetDevice(ctx->device));
    hipStream_t s = (hipStream_t)stream;
    std::lock_guard<std::mutex> enqueue(ctx->enqueue_mu);        // the launches below share the stream's scratch block
    void *scratch = nullptr;
    int rc = ivr_ctx_scratch(ctx, s, (size_t)ivr_frame_quality_scratch_bytes(n, h, w), &scratch);
    if (rc != IVR_OK) return rc;
    const int64_t npix = (int64_t)n * h * w;
    uint8_t *gray = reinterpret_cast<uint8_t *>(scratch);
    uint16_t *mag = reinterpret_cast<uint16_t *>(gray + ivr_round_up(npix, 256));
    uint8_t *dir = reinterpret_cast<uint8_t *>(mag + npix);
    uint8_t *mark = dir + npix;
    IVR_HIP(hipMemsetAsync(lap_sums, 0, (size_t)n * 16, s));
    {
        IvrProf prof("quality_gray", s, (double)npix * 4);
        hipLaunchKernelGGL(quality_gray_kernel, dim3((unsigned)ivr_ceil_div(npix, 256)), dim3(256), 0, s, frames, gray, npix, bgr);
    }
    IVR_LAUNCH_CHECK();
    const dim3 grid((unsigned)ivr_ceil_div((int64_t)h * w, 256), (unsigned)n);
    {
        IvrProf prof("quality_grad", s, (double)npix * 4);
        hipLaunchKernelGGL(quality_grad_kernel, grid, dim3(256), 0, s, gray, h, w, mag, dir, reinterpret_cast<long long *>(lap_sums));
    }
    IVR_LAUNCH_CHECK();
    {
        IvrProf prof("quality_nms", s, (double)npix * 4);
        hipLaunchKernelGGL(quality_nms_kernel, grid, dim3(256), 0, s, mag, dir, h, w, canny_low, canny_high, mark);
    }
    IVR_LAUNCH_CHECK();
    hipLaunchKernelGGL(quality_hysteresis_kernel, dim3(n), dim3(1024), 0, s, mark, h, w, reinterpret_cast<long long *>(edge_count));
    IVR_LAUNCH_CHECK();
    return IVR_OK;
}

}  // extern "C"
