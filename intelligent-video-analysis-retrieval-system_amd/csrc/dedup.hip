// Near-duplicate frame filter (D1 of SURVEY.md section 8a).
//
// Replaces the per-frame host loop at video_frame_filter.py:63-70:
//     sim = cosine_similarity([emb], [prev_embedding])[0][0]; keep iff sim < SIM_THRESHOLD
// where prev_embedding is the embedding of the last KEPT frame.  The decision chain is sequential,
// so one wave walks the batch in frame order; per frame it needs one dot product and one squared norm
// (the kept frame's norm is carried), reduced with wave shuffles - no barriers, no host round trip.
#include "ivr_common.h"

namespace {

__global__ __launch_bounds__(64) void dedup_kernel(const float *__restrict__ emb, int n, int d, float threshold,
                                                   float *__restrict__ state, uint8_t *__restrict__ keep) {
    const int lane = threadIdx.x;
    bool has_prev = state[0] != 0.f;
    int prev_row = -1;          // -1: previous kept embedding lives in state[1..d]
    float prev_ss = 0.f;
    if (has_prev) {
        for (int k = lane; k < d; k += 64) prev_ss = fmaf(state[1 + k], state[1 + k], prev_ss);
        prev_ss = ivr_wave_sum(prev_ss);
    }
    for (int t = 0; t < n; ++t) {
        const float *e = emb + (int64_t)t * d;
        const float *p = prev_row < 0 ? state + 1 : emb + (int64_t)prev_row * d;
        float dot = 0.f, ss = 0.f;
        for (int k = lane; k < d; k += 64) {
            const float v = e[k];
            ss = fmaf(v, v, ss);
            if (has_prev) dot = fmaf(v, p[k], dot);
        }
        dot = ivr_wave_sum(dot);
        ss = ivr_wave_sum(ss);
        bool uniq = true;
        if (has_prev) {
            // sklearn normalises each row (zero norm -> 1) and takes the dot product
            const float na = ss > 0.f ? sqrtf(ss) : 1.f, nb = prev_ss > 0.f ? sqrtf(prev_ss) : 1.f;
            const float sim = dot / (na * nb);
            if (sim >= threshold) uniq = false;
        }
        if (uniq) {
            has_prev = true;
            prev_row = t;
            prev_ss = ss;
        }
        if (lane == 0) keep[t] = uniq ? 1 : 0;
    }
    // carry the last kept embedding to the next batch
    if (prev_row >= 0) {
        for (int k = lane; k < d; k += 64) state[1 + k] = emb[(int64_t)prev_row * d + k];
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

}  // namespace

extern "C" int ivr_rowwise_cosine(ivr_ctx *ctx, const float *a, const float *b, int n, int d, float *out, ivr_stream stream) {
    IVR_REQUIRE(ctx && (n == 0 || (a && b && out)), "ivr_rowwise_cosine: NULL argument");
    IVR_REQUIRE(n >= 0 && d >= 1, "ivr_rowwise_cosine: n=%d d=%d", n, d);
    if (n == 0) return IVR_OK;
    IVR_HIP(hipSetDevice(ctx->device));
    hipLaunchKernelGGL(rowwise_cosine_kernel, dim3((unsigned)ivr_ceil_div(n, 4)), dim3(256), 0, (hipStream_t)stream, a, b, n, d, out);
    IVR_LAUNCH_CHECK();
    return IVR_OK;
}

extern "C" int ivr_dedup_keep_mask(ivr_ctx *ctx, const float *emb, int n, int d, float threshold, float *state,
                                   uint8_t *keep, ivr_stream stream) {
    IVR_REQUIRE(ctx && state && (n == 0 || (emb && keep)), "ivr_dedup_keep_mask: NULL argument");
    IVR_REQUIRE(n >= 0 && d >= 1, "ivr_dedup_keep_mask: n=%d d=%d", n, d);
    if (n == 0) return IVR_OK;
    IVR_HIP(hipSetDevice(ctx->device));
    hipLaunchKernelGGL(dedup_kernel, dim3(1), dim3(64), 0, (hipStream_t)stream, emb, n, d, threshold, state, keep);
    IVR_LAUNCH_CHECK();
    return IVR_OK;
}
