/*
 * ivr_api.h - C ABI of libivr_hip.so: the MI355X (gfx950) implementation of the
 * frame -> embedding -> cosine top-k hot path of
 * DMDung2k3/Intelligent-Video-Analysis-Retrieval-System.
 *
 * The reference has no FFI of its own: the path sits behind duck-typed Python
 * objects.  Each entry point below names the reference call
 * it stands in for; paths are relative to the reference checkout.  The Python
 * mirror of the reference classes (ivr_amd/compat.py) binds these through ctypes
 * (ivr_amd/_ffi.py); INTEGRATION.md shows the stub a maintainer would add.
 *
 * Conventions
 *  - extern "C", plain pointers and sizes; no torch / numpy types.
 *  - Every bulk pointer marked DEV is a device (HBM) pointer owned by the caller;
 *    HOST pointers are small parameter blocks or one-time weight uploads.
 *  - `stream` is a hipStream_t passed as void* (NULL = the null stream).  Calls
 *    only enqueue work; nothing synchronises unless stated.
 *  - Every function returns IVR_OK (0) or a negative ivr_status and never throws
 *    or aborts; ivr_last_error() returns the calling thread's last message.
 *  - Handles may be used from several host threads (the reference calls
 *    encode_images from a 4-thread pool, unified_index.py:773): the host side of calls
 *    on one handle is serialised by a per-handle mutex, distinct handles are independent.
 *    A tower / index handle owns device workspaces (activations, search candidates), so
 *    all calls on ONE handle must be enqueued on ONE stream at a time: to move a handle
 *    to another stream, make the new stream wait for the old one first.  ivr_preprocess
 *    keeps its scratch per stream and may be called concurrently on different streams.
 */
#ifndef IVR_API_H
#define IVR_API_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define IVR_API_VERSION 4
#define IVR_MAX_K 2048          /* reference: k=50 default, SearchOptions.limit <= 1000 (system.py:91) */

typedef enum ivr_status {
    IVR_OK = 0,
    IVR_ERR_INVALID = -1,       /* bad argument / shape: shim raises ValueError (core.py:1178-1191) */
    IVR_ERR_HIP = -2,           /* HIP runtime failure: shim raises RuntimeError (core.py:894-896) */
    IVR_ERR_OOM = -3,
    IVR_ERR_STATE = -4,         /* handle not ready (e.g. tower not finalized, index empty where rows are needed) */
    IVR_ERR_UNSUPPORTED = -5
} ivr_status;

typedef struct ivr_ctx ivr_ctx;
typedef struct ivr_index ivr_index;
typedef struct ivr_tower ivr_tower;
typedef void *ivr_stream;

/* ---- context ------------------------------------------------------------------------------- */
int ivr_api_version(void);
int ivr_init(int device, ivr_ctx **out);
int ivr_destroy(ivr_ctx *ctx);
const char *ivr_last_error(ivr_ctx *ctx);       /* ctx may be NULL */
int ivr_device_info(ivr_ctx *ctx, int *cu_count, int64_t *hbm_bytes, char *arch, int arch_len);
/* Device scratch (ivr_preprocess intermediates, ivr_frame_quality planes, ...) is kept per stream and only grows.  A caller that
 * retires a stream (and every graph captured on it) hands its block back with this call: it waits for the stream, frees the block
 * and forgets the stream, so a recycled stream handle starts clean.  No-op for a stream that never used scratch. */
int ivr_release_stream_scratch(ivr_ctx *ctx, ivr_stream stream);

/* ---- measurement hooks (bench.py): per-kernel HIP-event timing on the launch stream ---------------
 * No counterpart in the reference (its only profiler is the wall-clock PerformanceMonitor.timer,
 * utils.py:2481).  on = 1 brackets every launch of the kernels that carry a step (GEMMs, LayerNorm,
 * attention, index scans, preprocess emit) with two events; on = 2 also the short launches of the search
 * tail and the index append (an event pair costs microseconds on the stream: too much to leave around
 * 5-microsecond kernels inside a timed region).  ivr_profile_json synchronises on the events and writes
 * {"kernel": {"launches", "ms", "work"}} where work is the launch's algorithmic bytes (HBM-bound kernels)
 * or FLOP (MFMA-bound kernels). */
int ivr_profile_enable(ivr_ctx *ctx, int on);
int ivr_profile_reset(ivr_ctx *ctx);
int ivr_profile_json(ivr_ctx *ctx, char *buf /*HOST*/, int len);

/* ---- P1 / P2: frame preprocessing -------------------------------------------------------------
 * Replaces HFCLIPProcessor(images=...) at core.py:1613 and
 * cv2.cvtColor + Image.resize + processor(...) at video_frame_filter.py:58-59,29.
 * src: DEV uint8 [n, h, w, 3] (NHWC, dense).  Output: [n,3,S,S] (NCHW) or the patch-major
 * im2col matrix [n*(S/P)^2, 3*P*P] (column order c,py,px = the conv weight's) that feeds the
 * patch-embed GEMM directly.  Geometry is PIL's two-pass fixed-point resampler, bit-exact.
 */
enum {
    IVR_PP_MODE_IDENTITY = 0,           /* h == w == S */
    IVR_PP_MODE_SHORTEST_EDGE_CROP = 1, /* CLIP processor: shortest edge -> S, centre crop SxS */
    IVR_PP_MODE_STRETCH = 2,            /* video_frame_filter.py:59 */
    IVR_PP_MODE_LETTERBOX = 3,          /* extra (BASELINE.json wording); long edge -> S, zero canvas */
    IVR_PP_MODE_MASK = 0xF,
    IVR_PP_BGR = 1 << 4,                /* src is BGR (cv2) - swapped while loading */
    IVR_PP_OUT_F32 = 1 << 5,            /* default output dtype is bf16 */
    IVR_PP_OUT_PATCH_MAJOR = 1 << 6,    /* default layout is NCHW */
    IVR_PP_BILINEAR = 1 << 7            /* default filter is BICUBIC (PIL default, CLIP processor) */
};
int ivr_preprocess(ivr_ctx *ctx, const uint8_t *src /*DEV*/, int n, int h, int w, int flags,
                   const float mean[3] /*HOST*/, const float std[3] /*HOST*/, int out_size, int patch,
                   void *dst /*DEV*/, ivr_stream stream);
/* bytes of DEV scratch ivr_preprocess needs for (n,h,w,flags); 0 for identity geometry.  The scratch
 * lives in the context, one block per stream, and grows on demand (outside of stream capture); outgrown
 * blocks stay allocated until ivr_destroy because kernels in flight or a captured graph may still use them. */
int64_t ivr_preprocess_scratch_bytes(int n, int h, int w, int flags, int out_size);

/* ---- E1 / E2 / E3 + N1: encoder towers --------------------------------------------------------
 * Replaces CLIPModel.get_image_features + F.normalize (core.py:1619-1620),
 * CLIPModel.get_text_features + F.normalize (core.py:1541-1542) and
 * ViTModel(...).last_hidden_state[:,0,:] (video_frame_filter.py:31-32).
 */
enum { IVR_ACT_QUICK_GELU = 0, IVR_ACT_GELU_ERF = 1 };
enum { IVR_POOL_CLS_POSTLN_PROJ = 0, IVR_POOL_LN_ALL_CLS = 1, IVR_POOL_EOS_LN_PROJ = 2 };
enum { IVR_KIND_VISION = 0, IVR_KIND_TEXT = 1 };
enum {
    IVR_COMPUTE_BF16 = 0,
    IVR_COMPUTE_F32 = 1, /* verification mode: f32 MFMA, f32 activations */
    IVR_COMPUTE_FP8 = 2  /* BASELINE config 5: the linear sites of every block named by desc.fp8_sites on the CDNA4 fp8 MFMA
                          * (e4m3 operands, per-output-channel weight scales, f32 accumulate), the other sites on the bf16 MFMA;
                          * patch embedding, attention products, LN, residual stream and projection as in the bf16 mode */
};
/* the four linear sites of a transformer block (modeling_clip.py:259-277 q/k/v/out_proj, :338-350 fc1/fc2) */
enum { IVR_FP8_SITE_QKV = 1, IVR_FP8_SITE_ATTN_OUT = 2, IVR_FP8_SITE_FC1 = 4, IVR_FP8_SITE_FC2 = 8, IVR_FP8_SITE_ALL = 15 };

typedef struct ivr_tower_desc {
    int kind, width, layers, heads, mlp, tokens, out_dim, act, pool;
    int image, patch, pre_ln, patch_bias;   /* vision */
    int vocab, eos_id, causal;               /* text */
    int compute;                             /* IVR_COMPUTE_* */
    float ln_eps;
    /* IVR_COMPUTE_FP8 only.  fp8_sites: mask of IVR_FP8_SITE_* that run in e4m3 (0 = all four).  fp8_mlp_cls_bf16 = 1:
     * the rows of token 0 (CLS, the only row the vision pooling reads) go through fc1 / fc2 in bf16 on a side path
     * (n rows per launch instead of n*tokens), the e4m3 GEMMs leave those residual rows alone.  Which assignment stays
     * inside which tolerance: DESIGN.md section 4, profiles/r02_fp8_error_budget.json. */
    int fp8_sites, fp8_mlp_cls_bf16;
    /* IVR_COMPUTE_FP8 only: the mask applies to blocks fp8_first_layer .. layers-1, earlier blocks run in bf16 (the early
     * blocks are the sensitive ones: their error passes through every later attention).  0 = every block. */
    int fp8_first_layer;
} ivr_tower_desc;

int ivr_tower_create(ivr_ctx *ctx, const ivr_tower_desc *desc, ivr_tower **out);
/* name = canonical tensor name (ivr_amd/weights.py); data = HOST float32, nn.Linear layout W[out,in]. */
int ivr_tower_set_weight(ivr_tower *t, const char *name, const float *data /*HOST*/, int64_t count);
/* checks that every tensor was set, packs QKV, allocates the activation workspace for max_batch. */
int ivr_tower_finalize(ivr_tower *t, int max_batch);
int ivr_tower_destroy(ivr_tower *t);
/* patches: DEV patch-major pixels from ivr_preprocess (bf16, or f32 for IVR_COMPUTE_F32),
 * out: DEV float32 [n, embed_dim]; normalize=1 applies x / max(||x||, 1e-12) (F.normalize). */
int ivr_tower_encode_image(ivr_tower *t, const void *patches /*DEV*/, int n, int normalize,
                           float *out /*DEV*/, ivr_stream stream);
/* ids: DEV int64 [q, T] (T <= desc.tokens); pooled at the first eos_id per row. */
int ivr_tower_encode_text(ivr_tower *t, const int64_t *ids /*DEV*/, int q, int T, int normalize,
                          float *out /*DEV*/, ivr_stream stream);
/* bring-up / parity: arm a one-shot capture - the NEXT encode call copies the residual stream after
 * `layer` blocks (0 = embeddings) into out as f32 [n,T,width]. */
int ivr_tower_debug_hidden(ivr_tower *t, int layer, int n, float *out /*DEV*/, ivr_stream stream);
int64_t ivr_tower_workspace_bytes(ivr_tower *t);

/* Building block of the towers, exposed for parity tests and kernel benchmarks: y = x W^T (+ bias), i.e. the
 * nn.Linear calls inside the HF modules (modeling_clip.py:259-277, 338-350).  x: DEV [M,K], w: DEV [N,K], both
 * bf16 (f32_mode = 0) or float32 (f32_mode = 1); bias: DEV float32 [N] or NULL.  K must be a multiple of 64 (32 in f32 mode),
 * N of 4.  epilogue 0: out = act(y) in the operand dtype (act = -1 none, else IVR_ACT_*); 1: resid (DEV float32 [M,N]) += y;
 * 3: out = y as float32. */
int ivr_linear(ivr_ctx *ctx, int f32_mode, int epilogue, const void *x /*DEV*/, const void *w /*DEV*/,
               const float *bias /*DEV*/, int M, int N, int K, int act, void *out /*DEV*/, float *resid /*DEV*/,
               ivr_stream stream);

/* fp8 variant of ivr_linear (the GEMM of IVR_COMPUTE_FP8): x DEV [M,K] and w DEV [N,K] are OCP e4m3 bytes, colscale DEV
 * float32 [N] multiplies column n of x w^T before the bias (the weight's dequantisation scale; NULL = 1).  K % 128 == 0,
 * N % 64 == 0.  epilogue 0: out = act(y) as bf16 (out_fp8 = 0) or saturated e4m3 (out_fp8 = 1); 1: resid += y. */
int ivr_linear_fp8(ivr_ctx *ctx, int epilogue, const void *x /*DEV*/, const void *w /*DEV*/, const float *colscale /*DEV*/,
                   const float *bias /*DEV*/, int M, int N, int K, int act, void *out /*DEV*/, int out_fp8,
                   float *resid /*DEV*/, ivr_stream stream);

/* The weight quantiser of IVR_COMPUTE_FP8, exposed so that it can be checked without a GPU: HOST float32 -> HOST OCP e4m3 bytes
 * (bias 7, no infinity, max 448), round to nearest even, saturating; NaN -> 0x7f | sign. */
int ivr_quantize_e4m3_host(const float *src /*HOST*/, uint8_t *dst /*HOST*/, int64_t n);

/* ---- N2 / N3: row L2 normalisation -------------------------------------------------------------
 * Replaces FAISSRetriever._normalize_and_validate_features (core.py:1176-1196) and
 * faiss.normalize_L2 (unified_index.py:1776): x /= ||x||, all-zero rows stay zero.  In place.
 * nonfinite (DEV int32, may be NULL) receives the count of NaN/Inf input elements (the shim turns
 * a non-zero count into the ValueError of core.py:1190-1191).
 */
int ivr_l2_normalize(ivr_ctx *ctx, float *x /*DEV*/, int64_t n, int d, int32_t *nonfinite /*DEV*/,
                     ivr_stream stream);

/* ---- I1 + S1: flat inner-product index ---------------------------------------------------------
 * Replaces faiss.IndexFlatIP(d) / .add / .search (unified_index.py:1767,1779,503; core.py:1208,827,891).
 * Rows live in HBM as float32 in a 16-row x 4-float interleaved tile layout chosen so that one
 * wave-wide 16-byte load is an MFMA operand fragment and 1 KiB contiguous.
 */
int ivr_index_create(ivr_ctx *ctx, int d, int64_t capacity_rows, ivr_index **out);
int ivr_index_destroy(ivr_index *idx);
int ivr_index_reset(ivr_index *idx);                        /* ntotal = 0 */
int64_t ivr_index_ntotal(ivr_index *idx);
int ivr_index_dim(ivr_index *idx);
int64_t ivr_index_capacity(ivr_index *idx);
/* append n rows (DEV float32 [n,d] row-major); grows the allocation when capacity is exceeded. */
int ivr_index_add(ivr_index *idx, const float *rows /*DEV*/, int64_t n, int normalize, ivr_stream stream);
/* overwrite rows [start, start+n) (ring-buffer use, BASELINE config 4); start+n <= ntotal. */
int ivr_index_write(ivr_index *idx, int64_t start, const float *rows /*DEV*/, int64_t n, int normalize,
                    ivr_stream stream);
/* rolling-window variant for a captured (hipGraph) streaming step: overwrite the n rows at *cursor (DEV int64, a multiple of
 * n; n must divide ntotal) and advance the cursor by n modulo ntotal, all on the stream. */
int ivr_index_write_ring(ivr_index *idx, const float *rows /*DEV*/, int64_t n, int normalize, int64_t *cursor /*DEV*/,
                         ivr_stream stream);
/* copy rows [start, start+n) back to row-major float32 (faiss reconstruct_n). */
int ivr_index_reconstruct(ivr_index *idx, int64_t start, int64_t n, float *out /*DEV*/, ivr_stream stream);
/* pre-size the search workspace so that ivr_index_search allocates nothing (hipGraph capture). */
int ivr_index_reserve_search(ivr_index *idx, int max_nq, int max_k);
/* Exact top-k by inner product.  q: DEV float32 [nq,d]; D: DEV float32 [nq,k] descending;
 * I: DEV int64 [nq,k] = id_base + row, ties broken by lower id; unused slots (-FLT_MAX, -1). */
int ivr_index_search(ivr_index *idx, const float *q /*DEV*/, int nq, int k, int normalize_q,
                     int64_t id_base, float *D /*DEV*/, int64_t *I /*DEV*/, ivr_stream stream);
/* Diagnostics of the bf16 candidate scan behind ivr_index_search: HOST out[0] = 1 when the index keeps a bf16
 * scan copy, out[1] = number of queries of the LAST scan chunk (<= 64 queries) whose verification failed and which were redone by
 * the exact float32 scan.  Synchronises the device. */
int ivr_index_scan_stats(ivr_index *index, int *out /*HOST [2]*/);

/* Merge per-shard candidate lists (the reference's concat + sort of peer results, system.py:1744-1746):
 * D_parts/I_parts DEV [parts, nq, k] with global ids, parts ordered by ascending id range. */
int ivr_topk_merge(ivr_ctx *ctx, const float *D_parts /*DEV*/, const int64_t *I_parts /*DEV*/, int parts,
                   int nq, int k, float *D /*DEV*/, int64_t *I /*DEV*/, ivr_stream stream);

/* The same merge straight from the buffer of the ONE all-gather of the sharded search: a candidate travels as three int32 words
 * (score bits, id low, id high), so the exchange is pack (one launch) -> all_gather_into_tensor -> merge (one launch).
 * packed: DEV int32 [nq,k,3]; packed_parts: DEV int32 [parts,nq,k,3], parts ordered by ascending id range. */
int ivr_topk_pack(ivr_ctx *ctx, const float *D /*DEV*/, const int64_t *I /*DEV*/, int nq, int k, int32_t *packed /*DEV*/,
                  ivr_stream stream);
int ivr_topk_merge_packed(ivr_ctx *ctx, const int32_t *packed_parts /*DEV*/, int parts, int nq, int k, float *D /*DEV*/,
                          int64_t *I /*DEV*/, ivr_stream stream);

/* ---- D1: near-duplicate frame filter ------------------------------------------------------------
 * Replaces the cosine_similarity(...) >= SIM_THRESHOLD loop at video_frame_filter.py:63-70.
 * emb: DEV float32 [n,d] in frame order.  keep[t] = 1 iff cos(emb[t], last kept) < threshold.
 * state: DEV float32 [d+1]: state[0] = 1 if a previous kept embedding is stored in state[1..d];
 * updated in place so consecutive batches of one video continue the same sequence.
 */
int ivr_dedup_keep_mask(ivr_ctx *ctx, const float *emb /*DEV*/, int n, int d, float threshold,
                        float *state /*DEV*/, uint8_t *keep /*DEV*/, ivr_stream stream);

/* In-scene similarity filter of the keyframe pipeline, filter_similar_frames_in_scene at filter.py:178-222, as ONE launch per
 * scene: emb DEV float32 [n,d] = the scene's frames in order; keep[0] = 1; keep[i] = 1 iff i is at least min_distance after the
 * last kept frame and cos(emb[i], emb[last kept]) < threshold (the caller appends the scene's last frame, filter.py:218-220). */
int ivr_scene_keep_mask(ivr_ctx *ctx, const float *emb /*DEV*/, int n, int d, float threshold, int min_distance,
                        uint8_t *keep /*DEV*/, ivr_stream stream);

/* Window variant, filter_similar_frames_advanced at filter.py:224-258 (selected by use_advanced_similarity_filtering, filter.py:292-295):
 * keep[0] = 1; keep[i] = 1 iff no KEPT frame j in [i - min(window, n), i) has cos(emb[i], emb[j]) >= threshold.  Two launches: the
 * banded cosines in parallel, then the decision chain over n x window floats. */
int ivr_scene_keep_mask_window(ivr_ctx *ctx, const float *emb /*DEV*/, int n, int d, float threshold, int window,
                               uint8_t *keep /*DEV*/, ivr_stream stream);

/* cos(a[i], b[i]) for i < n (DEV float32 [n,d] each), sklearn cosine_similarity conventions.  Replaces the per-pair
 * cosine_similarity([x],[y])[0][0] calls of the keyframe filter (filter.py:147, filter.py:208). */
int ivr_rowwise_cosine(ivr_ctx *ctx, const float *a /*DEV*/, const float *b /*DEV*/, int n, int d, float *out /*DEV*/,
                       ivr_stream stream);

/* ---- frame quality gating of the keyframe filter ---------------------------------------------------
 * Replaces calculate_blur_score / calculate_edge_density at filter.py:63-92 (cv2.Laplacian(gray, CV_64F).var() and the share of
 * cv2.Canny(gray, low, high) edge pixels) for a batch of decoded frames.  frames: DEV uint8 [n,h,w,3] (bgr = 1: cv2.imread
 * order, 0: RGB).  lap_sums: DEV int64 [n,2] = (sum, sum of squares) of the integer Laplacian response over the frame - the
 * variance is (s2 - s1*s1/N) / N with N = h*w, finished in float64 by the caller; edge_count: DEV int64 [n] = number of Canny
 * edge pixels (edge density = 100 * count / N).  w <= 16384.  Scratch (two bit planes = 2 bits per pixel, rows padded to 64 pixels, + 16 B per
 * 64 x 32 tile) lives in the context, per stream. */
int64_t ivr_frame_quality_scratch_bytes(int n, int h, int w);
int ivr_frame_quality(ivr_ctx *ctx, const uint8_t *frames /*DEV*/, int n, int h, int w, int bgr, int canny_low, int canny_high,
                      int64_t *lap_sums /*DEV*/, int64_t *edge_count /*DEV*/, ivr_stream stream);

#ifdef __cplusplus
}
#endif
#endif /* IVR_API_H */
