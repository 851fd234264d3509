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

}  // namespace

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
