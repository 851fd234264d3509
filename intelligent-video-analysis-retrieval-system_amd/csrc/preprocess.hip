// Frame preprocessing on MI355X (P1/P2 of SURVEY.md section 8a).
//
// Replaces HFCLIPProcessor(images=...) (core.py:1613: resize shortest edge BICUBIC, centre crop,
// rescale 1/255, normalise) and cv2.cvtColor + Image.resize + processor (video_frame_filter.py:58-59,29).
//
// Geometry is PIL's resampler restated in integer arithmetic (22-bit fixed-point taps, int32
// accumulate from 1<<21, >>22, clamp, uint8 image between the horizontal and the vertical pass), so
// the uint8 result is bit-identical to Pillow's.  The value map u8 -> float is the HF one:
// float32(float64(u) * (1/255)), then (v - mean) / std in float32; the host tabulates its 3x256
// possible results and the emit kernel either evaluates one fma whose rounded result was checked
// against all 768 table entries, or falls back to the table in LDS.
//
// Kernels are HBM-bound byte movers: 16-byte coalesced loads of whole input lines into LDS, 16-byte
// stores of 8 consecutive output elements; the patch-major layout makes every (patch, channel) run
// 2 KiB contiguous for P = 32.
#include "ivr_common.h"

#include <cmath>
#include <cstring>
#include <map>
#include <tuple>
#include <vector>

namespace {

constexpr int kPrecisionBits = 32 - 8 - 2;

struct Taps {          // one resampling axis, for the surviving output positions only
    int out_len = 0;   // positions computed
    int ksize = 0;
    int in_lo = 0, in_hi = 0;   // union of input positions touched
    std::vector<int32_t> xmin, xcnt, kk;   // kk is [ksize][out_len] (tap-major: coalesced across positions)
};

double filt_bicubic(double x) {
    const double a = -0.5;
    if (x < 0.0) x = -x;
    if (x < 1.0) return ((a + 2.0) * x - (a + 3.0)) * x * x + 1;
    if (x < 2.0) return (((x - 5) * x + 8) * x - 4) * a;
    return 0.0;
}
double filt_bilinear(double x) {
    if (x < 0.0) x = -x;
    return x < 1.0 ? 1.0 - x : 0.0;
}

// Pillow Resample.c precompute_coeffs + normalize_coeffs_8bpc for output positions [first, first+count)
// of a full-axis resize in_size -> out_size.
Taps make_taps(int in_size, int out_size, int first, int count, bool bilinear) {
    Taps t;
    double (*fn)(double) = bilinear ? filt_bilinear : filt_bicubic;
    const double fsupport = bilinear ? 1.0 : 2.0;
    const double scale = (double)in_size / out_size;
    const double filterscale = scale < 1.0 ? 1.0 : scale;
    const double support = fsupport * filterscale;
    t.ksize = (int)std::ceil(support) * 2 + 1;
    t.out_len = count;
    t.xmin.resize(count);
    t.xcnt.resize(count);
    t.kk.assign((size_t)t.ksize * count, 0);
    t.in_lo = in_size;
    t.in_hi = 0;
    const double ss = 1.0 / filterscale;
    std::vector<double> k(t.ksize);
    for (int i = 0; i < count; ++i) {
        const int xx = first + i;
        const double center = (xx + 0.5) * scale;
        int lo = (int)(center - support + 0.5);
        if (lo < 0) lo = 0;
        int hi = (int)(center + support + 0.5);
        if (hi > in_size) hi = in_size;
        const int n = hi - lo;
        double ww = 0.0;
        for (int x = 0; x < n; ++x) {
            const double w = fn((x + lo - center + 0.5) * ss);
            k[x] = w;
            ww += w;
        }
        for (int x = 0; x < n; ++x) {
            if (ww != 0.0) k[x] /= ww;
            const double v = k[x] * (1 << kPrecisionBits);
            t.kk[(size_t)x * count + i] = k[x] < 0 ? (int)(-0.5 + v) : (int)(0.5 + v);
        }
        t.xmin[i] = lo;
        t.xcnt[i] = n;
        if (lo < t.in_lo) t.in_lo = lo;
        if (hi > t.in_hi) t.in_hi = hi;
    }
    return t;
}

struct Plan {            // geometry of one (h, w, mode) on the device
    bool need_h = false, need_v = false;
    int out_w = 0, out_h = 0;     // region actually produced inside the SxS canvas
    int off_x = 0, off_y = 0;     // its placement (letterbox), else 0
    int rows_lo = 0, rows_n = 0;  // input rows the horizontal pass must produce
    int cols_lo = 0;              // first input column used when there is no horizontal pass (crop)
    Taps th, tv;
    int32_t *d_h = nullptr, *d_v = nullptr;   // device tables: xmin[out], xcnt[out], kk[ksize][out]
};

struct PlanCache {
    std::mutex mu;
    std::map<std::tuple<int, int, int, int, int, int>, Plan> plans;
};
PlanCache &plan_cache() {
    static PlanCache c;
    return c;
}

int upload_taps(const Taps &t, int32_t **out) {
    const size_t n = (size_t)t.out_len * (2 + t.ksize);
    std::vector<int32_t> host(n);
    memcpy(host.data(), t.xmin.data(), (size_t)t.out_len * 4);
    memcpy(host.data() + t.out_len, t.xcnt.data(), (size_t)t.out_len * 4);
    memcpy(host.data() + 2 * (size_t)t.out_len, t.kk.data(), t.kk.size() * 4);
    IVR_HIP(hipMalloc(out, n * 4));
    IVR_HIP(hipMemcpy(*out, host.data(), n * 4, hipMemcpyHostToDevice));
    return IVR_OK;
}

// image_transforms.py:296-310
void shortest_edge_size(int h, int w, int size, int *nh, int *nw) {
    const int sh = w <= h ? w : h, lg = w <= h ? h : w;
    const int new_long = (int)((double)size * lg / sh);
    if (w <= h) {
        *nh = new_long;
        *nw = size;
    } else {
        *nh = size;
        *nw = new_long;
    }
}

int get_plan(int device, int h, int w, int mode, bool bilinear, int S, const Plan **out) {
    PlanCache &pc = plan_cache();
    std::lock_guard<std::mutex> lk(pc.mu);
    auto key = std::make_tuple(device, h, w, mode, (int)bilinear, S);
    auto it = pc.plans.find(key);
    if (it != pc.plans.end()) {
        *out = &it->second;
        return IVR_OK;
    }
    Plan p;
    int rw = S, rh = S;                 // full resize target
    int first_x = 0, first_y = 0;       // surviving window inside the resize target
    p.out_w = S;
    p.out_h = S;
    if (mode == IVR_PP_MODE_SHORTEST_EDGE_CROP) {
        shortest_edge_size(h, w, S, &rh, &rw);
        first_y = (rh - S) / 2;         // image_transforms.py:493-497
        first_x = (rw - S) / 2;
    } else if (mode == IVR_PP_MODE_LETTERBOX) {
        if (h >= w) {
            rh = S;
            rw = (int)((double)S * w / h);
            if (rw < 1) rw = 1;
        } else {
            rw = S;
            rh = (int)((double)S * h / w);
            if (rh < 1) rh = 1;
        }
        p.out_w = rw;
        p.out_h = rh;
        p.off_x = (S - rw) / 2;
        p.off_y = (S - rh) / 2;
    }
    p.need_h = rw != w;
    p.need_v = rh != h;
    if (p.need_v) {
        p.tv = make_taps(h, rh, first_y, p.out_h, bilinear);
        p.rows_lo = p.tv.in_lo;
        p.rows_n = p.tv.in_hi - p.tv.in_lo;
        for (auto &m : p.tv.xmin) m -= p.rows_lo;   // relative to the intermediate image
        int rc = upload_taps(p.tv, &p.d_v);
        if (rc != IVR_OK) return rc;
    } else {
        p.rows_lo = first_y;
        p.rows_n = p.out_h;
    }
    if (p.need_h) {
        p.th = make_taps(w, rw, first_x, p.out_w, bilinear);
        int rc = upload_taps(p.th, &p.d_h);
        if (rc != IVR_OK) return rc;
    } else {
        p.cols_lo = first_x;
    }
    auto ins = pc.plans.emplace(key, std::move(p));
    *out = &ins.first->second;
    return IVR_OK;
}

// ---------------------------------------------------------------------------------------------
// 1-D resample along one axis of a uint8 [n, rows, cols, 3] image (Pillow ImagingResampleHorizontal /
// Vertical_8bpc).  Generic strides: the axis being resampled has stride `sa`, the other one `so`.
// One thread per (frame, other, position, channel); adjacent threads share taps through L1/L2.
// ---------------------------------------------------------------------------------------------
__global__ __launch_bounds__(256) void resample_axis_kernel(const uint8_t *__restrict__ src, uint8_t *__restrict__ dst,
                                                            int n, int other_len, int out_len,
                                                            int64_t s_frame, int64_t s_other, int64_t s_axis,
                                                            int64_t d_frame, int64_t d_other, int64_t d_axis,
                                                            const int32_t *__restrict__ tab, int ksize) {
    const int64_t per_frame = (int64_t)other_len * out_len * 3;
    const int64_t gid = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (gid >= per_frame * n) return;
    const int f = (int)(gid / per_frame);
    int64_t r = gid % per_frame;
    // order (other, position, channel) when the axis is the fast one (horizontal), else (position, other, channel)
    int o, p, c;
    c = (int)(r % 3);
    r /= 3;
    if (s_axis < s_other) {
        p = (int)(r % out_len);
        o = (int)(r / out_len);
    } else {
        o = (int)(r % other_len);
        p = (int)(r / other_len);
    }
    const int lo = tab[p], cnt = tab[out_len + p];
    const int32_t *kk = tab + 2 * (int64_t)out_len + p;
    const uint8_t *s = src + f * s_frame + o * s_other + lo * s_axis + c;
    int32_t acc = 1 << (kPrecisionBits - 1);
    for (int t = 0; t < cnt; ++t) acc += (int32_t)s[t * s_axis] * kk[(int64_t)t * out_len];
    acc >>= kPrecisionBits;
    acc = acc < 0 ? 0 : (acc > 255 ? 255 : acc);
    dst[f * d_frame + o * d_other + p * d_axis + c] = (uint8_t)acc;
}

// crop-only copy (geometry needs no resampling on either axis but is not the whole frame)
__global__ __launch_bounds__(256) void crop_kernel(const uint8_t *__restrict__ src, uint8_t *__restrict__ dst, int n,
                                                   int h, int w, int y0, int x0, int oh, int ow, int S, int offy, int offx) {
    const int64_t per = (int64_t)oh * ow * 3;
    const int64_t gid = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (gid >= per * n) return;
    const int f = (int)(gid / per);
    int64_t r = gid % per;
    const int c = (int)(r % 3);
    r /= 3;
    const int x = (int)(r % ow), y = (int)(r / ow);
    dst[((int64_t)f * S * S + (int64_t)(y + offy) * S + (x + offx)) * 3 + c] =
        src[((int64_t)f * h * w + (int64_t)(y + y0) * w + (x + x0)) * 3 + c];
}

// ---------------------------------------------------------------------------------------------
// emit: uint8 [n,S,S,3] -> normalised bf16/f32 in patch-major layout [n*(S/P)^2, Kpad]
// (NCHW is the P = S case).  One workgroup stages R whole image lines in LDS.
// ---------------------------------------------------------------------------------------------
struct EmitParams {
    float a[3], b[3];     // fma path: v = u * a[c] + b[c]
    int use_lut, bgr;
    int S, P, R, Kpad;
};

template <typename T>
struct Out8;
template <>
struct Out8<unsigned short> {
    static __device__ __forceinline__ void store(unsigned short *p, const float (&v)[8]) {
        uint4 o;
        o.x = ivr_pack_bf16x2(v[0], v[1]);
        o.y = ivr_pack_bf16x2(v[2], v[3]);
        o.z = ivr_pack_bf16x2(v[4], v[5]);
        o.w = ivr_pack_bf16x2(v[6], v[7]);
        *reinterpret_cast<uint4 *>(p) = o;
    }
    static __device__ __forceinline__ void store1(unsigned short *p, float v) { *p = ivr_f32_to_bf16(v); }
};
template <>
struct Out8<float> {
    static __device__ __forceinline__ void store(float *p, const float (&v)[8]) {
        reinterpret_cast<float4 *>(p)[0] = make_float4(v[0], v[1], v[2], v[3]);
        reinterpret_cast<float4 *>(p)[1] = make_float4(v[4], v[5], v[6], v[7]);
    }
    static __device__ __forceinline__ void store1(float *p, float v) { *p = v; }
};

// vector path: P % 8 == 0, R rows per workgroup with R | P or P | R.  PC > 0 fixes P = R = PC at compile time so that the
// (patch column, channel, row, 8-pixel group) decomposition of a work item is shifts and constant divisions: with run-time
// P the integer divisions cost more VALU time than the bytes cost HBM time (measured 34 % -> see profiles/).
template <typename T, int PC>
__global__ __launch_bounds__(256) void emit_vec_kernel(const uint8_t *__restrict__ src, T *__restrict__ dst,
                                                       const float *__restrict__ lut, EmitParams ep) {
    extern __shared__ __attribute__((aligned(16))) uint8_t lines[];
    const int S = ep.S, P = PC > 0 ? PC : ep.P, R = PC > 0 ? PC : ep.R;
    float *slut = reinterpret_cast<float *>(lines + (((size_t)R * S * 3 + 15) & ~(size_t)15));
    const int blocks_per_frame = S / R;
    const int f = blockIdx.x / blocks_per_frame, rb = blockIdx.x % blocks_per_frame;
    const int y0 = rb * R;
    const int nbytes = R * S * 3;
    const uint8_t *in = src + ((int64_t)f * S + y0) * S * 3;
    for (int i = threadIdx.x * 16; i < nbytes; i += blockDim.x * 16)
        *reinterpret_cast<uint4 *>(lines + i) = *reinterpret_cast<const uint4 *>(in + i);
    if (ep.use_lut)
        for (int i = threadIdx.x; i < 768; i += blockDim.x) slut[i] = lut[i];
    __syncthreads();
    const int G = S / P;                 // patches per side
    const int xs_n = P / 8;              // 8-element groups per patch line
    const int ngroups = G * 3 * R * xs_n;
    const float a0 = ep.a[0], a1 = ep.a[1], a2 = ep.a[2], b0 = ep.b[0], b1 = ep.b[1], b2 = ep.b[2];
    for (int g = threadIdx.x; g < ngroups; g += blockDim.x) {
        int r = g;
        const int xs = r % xs_n;
        r /= xs_n;
        const int yy = r % R;
        r /= R;
        const int c = r % 3;
        const int px = r / 3;
        const int y = y0 + yy;
        const int cc = ep.bgr ? 2 - c : c;
        const uint8_t *lp = lines + ((yy * S) + px * P + xs * 8) * 3 + cc;
        const float sa = c == 0 ? a0 : (c == 1 ? a1 : a2), sb = c == 0 ? b0 : (c == 1 ? b1 : b2);
        float v[8];
#pragma unroll
        for (int i = 0; i < 8; ++i) {
            const int u = lp[i * 3];
            v[i] = ep.use_lut ? slut[c * 256 + u] : fmaf((float)u, sa, sb);
        }
        const int64_t patch = ((int64_t)f * G + y / P) * G + px;
        Out8<T>::store(dst + patch * ep.Kpad + (c * P + (y % P)) * P + xs * 8, v);
    }
}

// scalar path for patch sizes that are not a multiple of 8 (ViT-L/14): one thread per output element,
// including the zero padding of K up to Kpad
template <typename T>
__global__ __launch_bounds__(256) void emit_scalar_kernel(const uint8_t *__restrict__ src, T *__restrict__ dst,
                                                          const float *__restrict__ lut, EmitParams ep, int64_t total) {
    const int64_t gid = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (gid >= total) return;
    const int S = ep.S, P = ep.P, G = S / P;
    const int k = (int)(gid % ep.Kpad);
    const int64_t patch = gid / ep.Kpad;
    float v = 0.f;
    if (k < 3 * P * P) {
        const int c = k / (P * P), yy = (k / P) % P, xx = k % P;
        const int px = (int)(patch % G), py = (int)((patch / G) % G);
        const int64_t f = patch / (G * G);
        const int cc = ep.bgr ? 2 - c : c;
        const int u = src[((f * S + py * P + yy) * S + px * P + xx) * 3 + cc];
        v = ep.use_lut ? lut[c * 256 + u] : fmaf((float)u, ep.a[c], ep.b[c]);
    }
    Out8<T>::store1(dst + gid, v);
}

float bf16_round(float f) {
    uint32_t u;
    memcpy(&u, &f, 4);
    u = (u + 0x7fffu + ((u >> 16) & 1u)) & 0xffff0000u;
    memcpy(&f, &u, 4);
    return f;
}

}  // namespace

extern "C" {

int64_t ivr_preprocess_scratch_bytes(int n, int h, int w, int flags, int out_size) {
    const int mode = flags & IVR_PP_MODE_MASK;
    if (mode == IVR_PP_MODE_IDENTITY) return 0;
    // worst case: horizontal intermediate [n, h, S, 3] + SxS canvas [n, S, S, 3]
    return (int64_t)n * h * out_size * 3 + ivr_round_up((int64_t)n * out_size * out_size * 3, 256) + 4096;
}

int ivr_preprocess(ivr_ctx *ctx, const uint8_t *src, int n, int h, int w, int flags, const float mean[3], const float std[3],
                   int out_size, int patch, void *dst, ivr_stream stream) {
    IVR_REQUIRE(ctx && mean && std, "ivr_preprocess: NULL argument");
    IVR_REQUIRE(n >= 0 && h >= 1 && w >= 1, "ivr_preprocess: n=%d h=%d w=%d", n, h, w);
    IVR_REQUIRE(n == 0 || (src && dst), "ivr_preprocess: NULL frame buffer");
    const int S = out_size;
    const int mode = flags & IVR_PP_MODE_MASK;
    const bool patch_major = flags & IVR_PP_OUT_PATCH_MAJOR;
    const int P = patch_major ? patch : S;
    IVR_REQUIRE(S >= 16 && S % 16 == 0 && S <= 1024, "ivr_preprocess: out_size=%d must be a multiple of 16 in [16,1024]", S);
    IVR_REQUIRE(P >= 1 && S % P == 0, "ivr_preprocess: patch=%d does not divide out_size=%d", P, S);
    IVR_REQUIRE(mode <= IVR_PP_MODE_LETTERBOX, "ivr_preprocess: unknown mode %d", mode);
    IVR_REQUIRE(mode != IVR_PP_MODE_IDENTITY || (h == S && w == S), "ivr_preprocess: identity mode needs %dx%d frames, got %dx%d",
                S, S, h, w);
    for (int c = 0; c < 3; ++c) IVR_REQUIRE(std[c] != 0.f, "ivr_preprocess: std[%d] is zero", c);
    if (n == 0) return IVR_OK;
    IVR_HIP(hipSetDevice(ctx->device));
    hipStream_t s = (hipStream_t)stream;

    // ---- geometry -> uint8 [n,S,S,3]
    const uint8_t *canvas = src;
    const bool trivial = h == S && w == S;
    // the resample / emit launches of one call go out back to back: two host threads on the same stream must not interleave
    // them on the stream's scratch block (distinct streams have distinct blocks)
    std::unique_lock<std::mutex> enqueue(ctx->enqueue_mu, std::defer_lock);
    if (!trivial) {
        enqueue.lock();
        const Plan *p = nullptr;
        int rc = get_plan(ctx->device, h, w, mode, (flags & IVR_PP_BILINEAR) != 0, S, &p);
        if (rc != IVR_OK) return rc;
        void *scratch = nullptr;
        rc = ivr_ctx_scratch(ctx, s, (size_t)ivr_preprocess_scratch_bytes(n, h, w, flags, S), &scratch);
        if (rc != IVR_OK) return rc;
        uint8_t *cv = reinterpret_cast<uint8_t *>(scratch);
        uint8_t *tmp = cv + ivr_round_up((int64_t)n * S * S * 3, 256);
        const int64_t cv_frame = (int64_t)S * S * 3;
        if (p->out_w != S || p->out_h != S) IVR_HIP(hipMemsetAsync(cv, 0, (size_t)n * cv_frame, s));
        uint8_t *cv_org = cv + ((int64_t)p->off_y * S + p->off_x) * 3;
        if (p->need_h && p->need_v) {
            // horizontal over the rows the vertical pass needs, then vertical
            const int64_t t_frame = (int64_t)p->rows_n * p->out_w * 3;
            int64_t total = (int64_t)n * t_frame;
            hipLaunchKernelGGL(resample_axis_kernel, dim3((unsigned)ivr_ceil_div(total, 256)), dim3(256), 0, s,
                               src + (int64_t)p->rows_lo * w * 3, tmp, n, p->rows_n, p->out_w, (int64_t)h * w * 3,
                               (int64_t)w * 3, (int64_t)3, t_frame, (int64_t)p->out_w * 3, (int64_t)3, p->d_h, p->th.ksize);
            IVR_LAUNCH_CHECK();
            total = (int64_t)n * p->out_h * p->out_w * 3;
            hipLaunchKernelGGL(resample_axis_kernel, dim3((unsigned)ivr_ceil_div(total, 256)), dim3(256), 0, s, tmp, cv_org, n,
                               p->out_w, p->out_h, t_frame, (int64_t)3, (int64_t)p->out_w * 3, cv_frame, (int64_t)3,
                               (int64_t)S * 3, p->d_v, p->tv.ksize);
            IVR_LAUNCH_CHECK();
        } else if (p->need_h) {
            const int64_t total = (int64_t)n * p->out_h * p->out_w * 3;
            hipLaunchKernelGGL(resample_axis_kernel, dim3((unsigned)ivr_ceil_div(total, 256)), dim3(256), 0, s,
                               src + (int64_t)p->rows_lo * w * 3, cv_org, n, p->out_h, p->out_w, (int64_t)h * w * 3,
                               (int64_t)w * 3, (int64_t)3, cv_frame, (int64_t)S * 3, (int64_t)3, p->d_h, p->th.ksize);
            IVR_LAUNCH_CHECK();
        } else if (p->need_v) {
            const int64_t total = (int64_t)n * p->out_h * p->out_w * 3;
            hipLaunchKernelGGL(resample_axis_kernel, dim3((unsigned)ivr_ceil_div(total, 256)), dim3(256), 0, s,
                               src + (int64_t)p->rows_lo * w * 3 + (int64_t)p->cols_lo * 3, cv_org, n, p->out_w, p->out_h,
                               (int64_t)h * w * 3, (int64_t)3, (int64_t)w * 3, cv_frame, (int64_t)3, (int64_t)S * 3, p->d_v,
                               p->tv.ksize);
            IVR_LAUNCH_CHECK();
        } else {
            const int64_t total = (int64_t)n * p->out_h * p->out_w * 3;
            hipLaunchKernelGGL(crop_kernel, dim3((unsigned)ivr_ceil_div(total, 256)), dim3(256), 0, s, src, cv, n, h, w,
                               p->rows_lo, p->cols_lo, p->out_h, p->out_w, S, p->off_y, p->off_x);
            IVR_LAUNCH_CHECK();
        }
        canvas = cv;
    }

    // ---- value map (image_transforms.py:118-122 then :417-439), tabulated exactly on the host
    const bool f32_out = flags & IVR_PP_OUT_F32;
    float lut[768];
    EmitParams ep;
    memset(&ep, 0, sizeof(ep));
    bool fma_ok = true;
    for (int c = 0; c < 3; ++c) {
        ep.a[c] = (float)(1.0 / (255.0 * (double)std[c]));
        ep.b[c] = (float)(-(double)mean[c] / (double)std[c]);
        for (int u = 0; u < 256; ++u) {
            const float v = (float)((double)u * (1.0 / 255.0));
            const float exact = (v - mean[c]) / std[c];
            lut[c * 256 + u] = exact;
            const float fast = fmaf((float)u, ep.a[c], ep.b[c]);
            if (f32_out ? (fast != exact) : (bf16_round(fast) != bf16_round(exact))) fma_ok = false;
        }
    }
    ep.use_lut = !fma_ok;
    ep.bgr = (flags & IVR_PP_BGR) ? 1 : 0;
    ep.S = S;
    ep.P = P;
    ep.Kpad = patch_major ? (int)ivr_round_up(3 * P * P, 64) : 3 * S * S;
    float *d_lut = nullptr;
    if (ep.use_lut) {
        // device copy of the table, cached per (mean, std) in the context (uploaded once, synchronously)
        std::string key(reinterpret_cast<const char *>(mean), 12);
        key.append(reinterpret_cast<const char *>(std), 12);
        std::lock_guard<std::mutex> lk(ctx->mu);
        auto it = ctx->luts.find(key);
        if (it == ctx->luts.end()) {
            float *d = nullptr;
            IVR_HIP(hipMalloc(&d, sizeof(lut)));
            IVR_HIP(hipMemcpy(d, lut, sizeof(lut), hipMemcpyHostToDevice));
            it = ctx->luts.emplace(key, d).first;
        }
        d_lut = it->second;
    }
    // algorithmic bytes: uint8 canvas in, normalised elements out
    IvrProf prof("preprocess_emit", s, (double)n * S * S * 3 * (1 + (f32_out ? 4 : 2)));
    if (P % 8 == 0) {
        ep.R = P <= 32 ? P : 8;
        IVR_REQUIRE(S % ep.R == 0, "ivr_preprocess: out_size %d not divisible by row block %d", S, ep.R);
        const size_t lds = (((size_t)ep.R * S * 3 + 15) & ~(size_t)15) + 768 * 4;
        const unsigned grid = (unsigned)((int64_t)n * (S / ep.R));
#define IVR_EMIT(TT, PCV) \
    hipLaunchKernelGGL((emit_vec_kernel<TT, PCV>), dim3(grid), dim3(256), lds, s, canvas, (TT *)dst, d_lut, ep)
        if (f32_out) {
            if (P == 32) IVR_EMIT(float, 32); else if (P == 16) IVR_EMIT(float, 16); else IVR_EMIT(float, 0);
        } else {
            if (P == 32) IVR_EMIT(unsigned short, 32); else if (P == 16) IVR_EMIT(unsigned short, 16); else IVR_EMIT(unsigned short, 0);
        }
#undef IVR_EMIT
    } else {
        const int G = S / P;
        const int64_t total = (int64_t)n * G * G * ep.Kpad;
        const unsigned grid = (unsigned)ivr_ceil_div(total, 256);
        if (f32_out)
            hipLaunchKernelGGL(emit_scalar_kernel<float>, dim3(grid), dim3(256), 0, s, canvas, (float *)dst, d_lut, ep, total);
        else
            hipLaunchKernelGGL(emit_scalar_kernel<unsigned short>, dim3(grid), dim3(256), 0, s, canvas, (unsigned short *)dst,
                               d_lut, ep, total);
    }
    IVR_LAUNCH_CHECK();
    return IVR_OK;
}

}  // extern "C"
