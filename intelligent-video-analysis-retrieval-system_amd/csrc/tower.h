// Shared declarations of the tower host code (tower.hip) and kernels (tower_kernels.hip).
#pragma once
#include <map>
#include <string>
#include <vector>

#include "ivr_common.h"

struct TowerTensor {
    void *ptr = nullptr;      // device
    int64_t count = 0;        // elements
    int bytes_per = 4;        // 4 = f32, 2 = bf16
};

struct ivr_tower {
    ivr_ctx *ctx = nullptr;
    ivr_tower_desc d{};
    std::mutex mu;
    std::map<std::string, TowerTensor> w;   // canonical name -> device tensor (GEMM weights in compute dtype)
    std::map<std::string, std::vector<float>> host;   // staged float32 masters until finalize
    bool finalized = false;
    int debug_layer = -1;
    float *debug_out = nullptr;
    int max_batch = 0;
    int kpad = 0;            // vision: 3*P*P rounded up to 64
    // activation workspace (one allocation, carved)
    void *ws = nullptr;
    size_t ws_bytes = 0;
    float *resid = nullptr;  // [rows, D] f32 residual stream
    void *xn = nullptr;      // [rows, D]   LN output (compute dtype)
    void *qkv = nullptr;     // [rows, 3D]
    void *att = nullptr;     // [rows, D]
    void *hid = nullptr;     // [rows, mlp]
    void *xn_cls = nullptr;  // [max_batch, D]   bf16 LN rows of token 0 (fp8 mode, fp8_mlp_cls_bf16)
    void *hid_cls = nullptr; // [max_batch, mlp] bf16 MLP hidden rows of token 0
    int sites = 0;           // fp8 mode: effective IVR_FP8_SITE_* mask (blocks >= d.fp8_first_layer)
    bool mlp_cls = false;    // fp8 mode: token-0 rows take fc1 / fc2 in bf16
    void *pool = nullptr;    // [max_batch, D] pooled + LN rows (compute dtype)
    float *pooled_f32 = nullptr;   // [max_batch, D] (POOL_LN_ALL_CLS output before normalise)
    int *eos_pos = nullptr;  // [max_batch]
    int last_n = 0, last_T = 0;
};
