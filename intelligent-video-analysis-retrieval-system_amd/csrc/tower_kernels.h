// Launchers implemented in tower_kernels.hip.
#pragma once
#include "ivr_common.h"

enum { EPI_STORE = 0, EPI_RESID = 1, EPI_PATCH = 2, EPI_F32 = 3 };

struct GemmArgs {
    const void *A = nullptr;   // X [M,K] row-major, compute dtype
    int lda = 0;
    const void *W = nullptr;   // W [N,K] row-major (nn.Linear), compute dtype
    int ldw = 0;
    int M = 0, N = 0, K = 0;
    const float *bias = nullptr;   // [N] or NULL
    void *out = nullptr;           // EPI_STORE: compute dtype [M,ldo]; EPI_F32: float [M,ldo]
    int ldo = 0;
    float *resid = nullptr;        // EPI_RESID: += in place; EPI_PATCH: token rows written
    int ldr = 0;
    const float *pos = nullptr;    // EPI_PATCH: position embedding [T,N]
    int T = 0, G2 = 0;             // tokens per image, patches per image
    int act = -1;                  // EPI_STORE: -1 none, else IVR_ACT_*
    const char *tag = nullptr;     // profiler name of this call site
    int group_m = 8;               // row panels per L2-resident group (tile order of gemm_kernel)
    const float *colscale = nullptr;   // fp8 GEMM: per-output-column dequantisation scale [N] (NULL = 1)
    int out8 = 0;                  // fp8 GEMM, EPI_STORE: write saturated e4m3 instead of bf16
    int wide_epi = 0;              // set by the launcher: 256 x 256 kernel may use the row-wide LDS-staged epilogue
    int reverse_m = 0;             // walk the row panels from the last to the first (see run_layers: producer / consumer order)
    int stagger = 0;               // persistent kernel: odd slots start this many s_sleep(127) later (half a tile); set by the launcher
    int skip_mod = 0;              // EPI_RESID: rows r with r % skip_mod == 0 are left untouched (0 = none); the fp8 mode's
                                   // token-0 rows are updated by a bf16 side GEMM instead
};

int ivr_launch_gemm(bool f32, int epi, const GemmArgs &g, hipStream_t s);
int ivr_launch_gemm_fp8(int epi, const GemmArgs &g, hipStream_t s);   // A, W: e4m3 bytes; 256 x 256 kernel only
enum { OUT_BF16 = 0, OUT_F32 = 1, OUT_FP8 = 2 };
int ivr_launch_layernorm(int out_kind, const float *x, int row_mul, const int *offs, const float *g, const float *b, float eps,
                         void *out, int rows, int D, hipStream_t s, int reverse = 0);
int ivr_launch_attention(bool f32, const void *qkv, void *att, int n, int T, int D, int heads, int causal, hipStream_t s,
                         bool out_fp8 = false);
// fused QKV projection + attention (bf16, T <= 64, not causal): xn [n*T, D] bf16, w [3D, D] bf16, bias [3D] -> att [n*T, D]
bool ivr_fused_qkv_attention_ok(int M, int T, int D, int heads, int causal);
int ivr_launch_qkv_attention(const void *xn, const void *w, const float *bias, void *att, int n, int T, int D, int heads, bool out_fp8,
                             hipStream_t s, int reverse = 0);
int ivr_launch_vision_cls(float *resid, const float *cls, const float *pos, int n, int T, int D, hipStream_t s);
int ivr_launch_text_embed(float *resid, const int64_t *ids, const float *tok, const float *pos, int q, int T, int D, int vocab,
                          int eos, int *eos_pos, hipStream_t s);
int ivr_launch_bf16_to_e4m3(const void *src, void *dst, int64_t count, hipStream_t s);   // saturating RNE, count % 8 == 0
int ivr_launch_f_normalize(const float *x, float *out, int n, int d, int normalize, hipStream_t s);
