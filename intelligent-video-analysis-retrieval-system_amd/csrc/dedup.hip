// Near-duplicate frame filter (D1 of SURVEY.md section 8a).
//
// Replaces the per-frame host loop at video_frame_filter.py:63-70:
//     sim = cosine_similarity([emb], [prev_embedding])[0][0]; keep iff sim < SIM_THRESHOLD
// where prev_embedding is the embedding of the last KEPT frame.  The decision chain is sequential,
// so one wave walks the batch in frame order; per frame it needs one dot product and one squared norm
// (the kept frame's norm is carried), reduced with wave shuffles - no host round trip.
#include <algorithm>
#include <type_traits>

#include "ivr_common.h"

namespace {

// The chain is sequential, so its cost is latency per frame, not bandwidth.  Waves 1-3 of the workgroup stream the embeddings
// into a double-buffered LDS chunk (coalesced, one barrier per chunk) while wave 0 walks the previous chunk: its per-frame
// critical path is then an LDS read + two wave reductions (~0.1 us) instead of a dependent trip to L2 / HBM (1.9 us per frame
// in the first version: 8 ms for a 4096-frame batch, a tenth of the DINO embedding time it filters).
// The last kept embedding lives in registers (ceil(d / 64) per lane, templated).
constexpr int kDedupMaxPer = 32;       // d <= 64 * 32
constexpr int kDedupLds = 64 * 1024;   // two chunks

template <int PER>      // floats of a row per lane: d <= 64 * PER
__global__ __launch_bounds__(256) void dedup_kernel(const float *__restrict__ emb, int n, int d, float threshold,
                                                    float *__restrict__ state, uint8_t *__restrict__ keep, int min_distance) {
    extern __shared__ __attribute__((aligned(16))) float buf[];
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int rc = max(1, (kDedupLds / 2) / (d * 4));          // rows per chunk
    const int nchunks = (n + rc - 1) / rc;
    auto load_chunk = [&](int c, int t0, int nthr) {           // threads t0 .. t0+nthr-1 of the block copy chunk c
        const int r0 = c * rc, rows = min(rc, n - r0);
        const int64_t total = (int64_t)rows * d;
        const float *src = emb + (int64_t)r0 * d;
        float *dst = buf + (c & 1) * rc * d;
        for (int64_t i = tid - t0; i < total; i += nthr) dst[i] = src[i];
    };
    load_chunk(0, 0, 256);
    __syncthreads();
    float prev[PER];
    bool has_prev = false;
    float prev_ss = 0.f;
    int last_kept = -1;
    if (wave == 0) {
        has_prev = state && state[0] != 0.f;             // state == NULL: a self-contained sequence (one scene)
#pragma unroll
        for (int j = 0; j < PER; ++j) {
            const int k = lane + 64 * j;
            prev[j] = (has_prev && k < d) ? state[1 + k] : 0.f;
        }
        if (has_prev) {
#pragma unroll
            for (int j = 0; j < PER; ++j)
                if (lane + 64 * j < d) prev_ss = fmaf(prev[j], prev[j], prev_ss);
            prev_ss = ivr_wave_sum(prev_ss);
        }
    }
    for (int c = 0; c < nchunks; ++c) {
        if (wave != 0) {
            if (c + 1 < nchunks) load_chunk(c + 1, 64, 192);
        } else {
            const int r0 = c * rc, rows = min(rc, n - r0);
            const float *cb = buf + (c & 1) * rc * d;
            for (int r = 0; r < rows; ++r) {
                const float *e = cb + r * d;
                float cur[PER];
                float dot = 0.f, ss = 0.f;
#pragma unroll
                for (int j = 0; j < PER; ++j) {
                    const int k = lane + 64 * j;
                    cur[j] = k < d ? e[k] : 0.f;
                    if (k < d) {
                        ss = fmaf(cur[j], cur[j], ss);
                        if (has_prev) dot = fmaf(cur[j], prev[j], dot);
                    }
                }
                dot = ivr_wave_sum(dot);
                ss = ivr_wave_sum(ss);
                bool uniq = true;
                if (has_prev) {
                    // sklearn normalises each row (zero norm -> 1) and takes the dot product
                    const float na = ss > 0.f ? sqrtf(ss) : 1.f, nb = prev_ss > 0.f ? sqrtf(prev_ss) : 1.f;
                    const float sim = dot / (na * nb);
                    if (sim >= threshold) uniq = false;
                    // filter.py:196-198: a frame closer than min_frame_distance to the last kept one is never a candidate
                    if (min_distance > 1 && last_kept >= 0 && r0 + r - last_kept < min_distance) uniq = false;
                }
                if (uniq) {                     // wave-uniform
                    has_prev = true;
                    prev_ss = ss;
                    last_kept = r0 + r;
#pragma unroll
                    for (int j = 0; j < PER; ++j) prev[j] = cur[j];
                }
                if (lane == 0) keep[r0 + r] = uniq ? 1 : 0;
            }
        }
        __syncthreads();
    }
    // carry the last kept embedding to the next batch
    if (wave == 0 && last_kept >= 0 && state) {
#pragma unroll
        for (int j = 0; j < PER; ++j)
            if (lane + 64 * j < d) state[1 + lane + 64 * j] = prev[j];
        if (lane == 0) state[0] = 1.f;
    }
}

// cos(a[i], b[i]) with sklearn's conventions (each row divided by its norm, zero norm -> 1): one wave per row pair
__global__ __launch_bounds__(256) void rowwise_cosine_kernel(const float *__restrict__ a, const float *__restrict__ b, int n, int d,
                                                             float *__restrict__ out) {
    const int lane = threadIdx.x & 63;
    const int row = blockIdx.x * 4 + (threadIdx.x >> 6);
    if (row >= n) return;
    const float *pa = a + (int64_t)row * d, *pb = b + (int64_t)row * d;
    float dot = 0.f, sa = 0.f, sb = 0.f;
    for (int k = lane; k < d; k += 64) {
        const float x = pa[k], y = pb[k];
        dot = fmaf(x, y, dot);
        sa = fmaf(x, x, sa);
        sb = fmaf(y, y, sb);
    }
    dot = ivr_wave_sum(dot);
    sa = ivr_wave_sum(sa);
    sb = ivr_wave_sum(sb);
    const float na = sa > 0.f ? sqrtf(sa) : 1.f, nb = sb > 0.f ? sqrtf(sb) : 1.f;
    if (lane == 0) out[row] = dot / (na * nb);
}

// Window mode of the in-scene filter (filter_similar_frames_advanced, filter.py:224-258): frame i is kept iff no KEPT frame j in
// [i - window, i) has cos(e_i, e_j) >= threshold.  The cosines of every frame with its `window` predecessors do not depend on the
// decisions, so they are computed first by all the waves of the launch (band[i][t-1] = cos(e_i, e_{i-t})); the decision chain is
// then a walk over n x window floats by one thread - no embedding is touched twice by the sequential part.
__global__ __launch_bounds__(256) void band_cosine_kernel(const float *__restrict__ emb, int n, int d, int window, float *__restrict__ band) {
    const int lane = threadIdx.x & 63;
    const int64_t pair = (int64_t)blockIdx.x * 4 + (threadIdx.x >> 6);
    if (pair >= (int64_t)n * window) return;
    const int i = (int)(pair / window), t = (int)(pair % window) + 1;
    if (i - t < 0) return;
    const float *pa = emb + (int64_t)i * d, *pb = emb + (int64_t)(i - t) * d;
    float dot = 0.f, sa = 0.f, sb = 0.f;
    for (int k = lane; k < d; k += 64) {
        const float x = pa[k], y = pb[k];
        dot = fmaf(x, y, dot);
        sa = fmaf(x, x, sa);
        sb = fmaf(y, y, sb);
    }
    dot = ivr_wave_sum(dot);
    sa = ivr_wave_sum(sa);
    sb = ivr_wave_sum(sb);
    const float na = sa > 0.f ? sqrtf(sa) : 1.f, nb = sb > 0.f ? sqrtf(sb) : 1.f;      // sklearn: zero norm -> 1
    if (lane == 0) band[pair] = dot / (na * nb);
}

__global__ void window_chain_kernel(const float *__restrict__ band, int n, int window, float threshold, uint8_t *__restrict__ keep) {
    if (threadIdx.x != 0 || blockIdx.x != 0) return;
    for (int i = 0; i < n; ++i) {
        bool k = true;
        for (int t = 1; t <= window && t <= i && k; ++t)
            if (keep[i - t] && band[(int64_t)i * window + t - 1] >= threshold) k = false;
        keep[i] = k ? 1 : 0;
    }
}

}  // namespace

extern "C" int ivr_scene_keep_mask_window(ivr_ctx *ctx, const float *emb, int n, int d, float threshold, int window, uint8_t *keep,
                                          ivr_stream stream) {
    IVR_REQUIRE(ctx && (n == 0 || (emb && keep)), "ivr_scene_keep_mask_window: NULL argument");
    IVR_REQUIRE(n >= 0 && d >= 1 && window >= 1, "ivr_scene_keep_mask_window: n=%d d=%d window=%d", n, d, window);
    if (n == 0) return IVR_OK;
    IVR_HIP(hipSetDevice(ctx->device));
    hipStream_t s = (hipStream_t)stream;
    window = std::min(window, n);                                   // filter.py:233
    std::lock_guard<std::mutex> enqueue(ctx->enqueue_mu);           // the two launches share the stream's scratch block
    void *scratch = nullptr;
    if (int rc = ivr_ctx_scratch(ctx, s, (size_t)n * window * sizeof(float), &scratch)) return rc;
    float *band = reinterpret_cast<float *>(scratch);
    hipLaunchKernelGGL(band_cosine_kernel, dim3((unsigned)ivr_ceil_div((int64_t)n * window, 4)), dim3(256), 0, s, emb, n, d, window, band);
    IVR_LAUNCH_CHECK();
    hipLaunchKernelGGL(window_chain_kernel, dim3(1), dim3(64), 0, s, band, n, window, threshold, keep);
    IVR_LAUNCH_CHECK();
    return IVR_OK;
}

extern "C" int ivr_rowwise_cosine(ivr_ctx *ctx, const float *a, const float *b, int n, int d, float *out, ivr_stream stream) {
    IVR_REQUIRE(ctx && (n == 0 || (a && b && out)), "ivr_rowwise_cosine: NULL argument");
    IVR_REQUIRE(n >= 0 && d >= 1, "ivr_rowwise_cosine: n=%d d=%d", n, d);
    if (n == 0) return IVR_OK;
    IVR_HIP(hipSetDevice(ctx->device));
    hipLaunchKernelGGL(rowwise_cosine_kernel, dim3((unsigned)ivr_ceil_div(n, 4)), dim3(256), 0, (hipStream_t)stream, a, b, n, d, out);
    IVR_LAUNCH_CHECK();
    return IVR_OK;
}

static int keep_chain(ivr_ctx *ctx, const float *emb, int n, int d, float threshold, float *state, uint8_t *keep, int min_distance,
                      ivr_stream stream) {
    IVR_REQUIRE(n >= 0 && d >= 1 && d <= 64 * kDedupMaxPer, "keep mask: n=%d d=%d (d <= %d)", n, d, 64 * kDedupMaxPer);
    if (n == 0) return IVR_OK;
    IVR_HIP(hipSetDevice(ctx->device));
    auto launch = [&](auto per) -> int {
        constexpr int PER = decltype(per)::value;
        if (int rc = ivr_func_max_lds(reinterpret_cast<const void *>(dedup_kernel<PER>), kDedupLds)) return rc;
        hipLaunchKernelGGL(dedup_kernel<PER>, dim3(1), dim3(256), kDedupLds, (hipStream_t)stream, emb, n, d, threshold, state, keep,
                           min_distance);
        return IVR_OK;
    };
    int rc;
    if (d <= 64 * 6) rc = launch(std::integral_constant<int, 6>{});            // DINO ViT-S: 384
    else if (d <= 64 * 8) rc = launch(std::integral_constant<int, 8>{});       // CLIP ViT-B: 512
    else if (d <= 64 * 12) rc = launch(std::integral_constant<int, 12>{});     // 768
    else if (d <= 64 * 16) rc = launch(std::integral_constant<int, 16>{});     // 1024
    else rc = launch(std::integral_constant<int, kDedupMaxPer>{});
    if (rc != IVR_OK) return rc;
    IVR_LAUNCH_CHECK();
    return IVR_OK;
}

extern "C" int ivr_dedup_keep_mask(ivr_ctx *ctx, const float *emb, int n, int d, float threshold, float *state,
                                   uint8_t *keep, ivr_stream stream) {
    IVR_REQUIRE(ctx && state && (n == 0 || (emb && keep)), "ivr_dedup_keep_mask: NULL argument");
    return keep_chain(ctx, emb, n, d, threshold, state, keep, 1, stream);
}

extern "C" int ivr_scene_keep_mask(ivr_ctx *ctx, const float *emb, int n, int d, float threshold, int min_distance, uint8_t *keep,
                                   ivr_stream stream) {
    IVR_REQUIRE(ctx && (n == 0 || (emb && keep)), "ivr_scene_keep_mask: NULL argument");
    IVR_REQUIRE(min_distance >= 1, "ivr_scene_keep_mask: min_distance=%d", min_distance);
    return keep_chain(ctx, emb, n, d, threshold, nullptr, keep, min_distance, stream);
}
