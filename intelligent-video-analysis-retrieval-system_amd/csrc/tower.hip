// Encoder towers (E1/E2/E3 + N1 of SURVEY.md section 8a): host side of ivr_tower_*.
// Kernels live in tower_kernels.hip; this file owns descriptors, weight upload/packing and the
// per-layer launch sequence.
#include "tower.h"
#include "tower_kernels.h"

#include <cmath>
#include <cstring>
#include <vector>

extern "C" {

int ivr_tower_create(ivr_ctx *ctx, const ivr_tower_desc *d, ivr_tower **out) {
    IVR_REQUIRE(ctx && d && out, "ivr_tower_create: NULL argument");
    IVR_REQUIRE(d->kind == IVR_KIND_VISION || d->kind == IVR_KIND_TEXT, "ivr_tower_create: kind=%d", d->kind);
    IVR_REQUIRE(d->width >= 64 && d->width % 64 == 0 && d->width <= 2048, "ivr_tower_create: width=%d must be a multiple of 64 in [64,2048]",
                d->width);
    IVR_REQUIRE(d->heads >= 1 && d->width == d->heads * 64, "ivr_tower_create: head_dim must be 64 (width=%d heads=%d)", d->width,
                d->heads);
    IVR_REQUIRE(d->mlp >= 64 && d->mlp % 64 == 0, "ivr_tower_create: mlp=%d must be a multiple of 64", d->mlp);
    IVR_REQUIRE(d->layers >= 1 && d->layers <= 64, "ivr_tower_create: layers=%d", d->layers);
    IVR_REQUIRE(d->tokens >= 1 && d->tokens <= 1024, "ivr_tower_create: tokens=%d", d->tokens);
    IVR_REQUIRE(d->out_dim >= 0 && d->out_dim % 16 == 0 && d->out_dim <= 2048, "ivr_tower_create: out_dim=%d", d->out_dim);
    IVR_REQUIRE(d->act == IVR_ACT_QUICK_GELU || d->act == IVR_ACT_GELU_ERF, "ivr_tower_create: act=%d", d->act);
    IVR_REQUIRE(d->compute == IVR_COMPUTE_BF16 || d->compute == IVR_COMPUTE_F32 || d->compute == IVR_COMPUTE_FP8,
                "ivr_tower_create: compute=%d", d->compute);
    IVR_REQUIRE(d->compute != IVR_COMPUTE_FP8 || (d->width % 128 == 0 && d->mlp % 128 == 0),
                "ivr_tower_create: the fp8 mode needs width and mlp to be multiples of 128 (width=%d mlp=%d)", d->width, d->mlp);
    IVR_REQUIRE(d->fp8_sites >= 0 && d->fp8_sites <= IVR_FP8_SITE_ALL && (d->fp8_mlp_cls_bf16 == 0 || d->fp8_mlp_cls_bf16 == 1),
                "ivr_tower_create: fp8_sites=%d fp8_mlp_cls_bf16=%d", d->fp8_sites, d->fp8_mlp_cls_bf16);
    IVR_REQUIRE(d->compute == IVR_COMPUTE_FP8 || (d->fp8_sites == 0 && d->fp8_mlp_cls_bf16 == 0 && d->fp8_first_layer == 0),
                "ivr_tower_create: fp8_sites / fp8_mlp_cls_bf16 / fp8_first_layer only apply to IVR_COMPUTE_FP8");
    IVR_REQUIRE(d->fp8_first_layer >= 0 && d->fp8_first_layer < d->layers, "ivr_tower_create: fp8_first_layer=%d outside [0,%d)",
                d->fp8_first_layer, d->layers);
    if (d->kind == IVR_KIND_VISION) {
        IVR_REQUIRE(d->patch >= 1 && d->image % d->patch == 0, "ivr_tower_create: patch=%d image=%d", d->patch, d->image);
        const int g = d->image / d->patch;
        IVR_REQUIRE(d->tokens == g * g + 1, "ivr_tower_create: tokens=%d != 1 + (%d/%d)^2", d->tokens, d->image, d->patch);
        IVR_REQUIRE(d->pool == IVR_POOL_CLS_POSTLN_PROJ || d->pool == IVR_POOL_LN_ALL_CLS, "ivr_tower_create: pool=%d", d->pool);
        IVR_REQUIRE(d->pool != IVR_POOL_CLS_POSTLN_PROJ || d->out_dim > 0, "ivr_tower_create: CLIP pooling needs out_dim");
    } else {
        IVR_REQUIRE(d->vocab >= 1 && d->eos_id >= 0 && d->eos_id < d->vocab, "ivr_tower_create: vocab=%d eos=%d", d->vocab, d->eos_id);
        IVR_REQUIRE(d->pool == IVR_POOL_EOS_LN_PROJ && d->out_dim > 0, "ivr_tower_create: text towers pool at EOS and project");
    }
    IVR_HIP(hipSetDevice(ctx->device));
    ivr_tower *t = new ivr_tower();
    t->ctx = ctx;
    t->d = *d;
    if (d->compute == IVR_COMPUTE_FP8) {
        t->sites = d->fp8_sites ? d->fp8_sites : IVR_FP8_SITE_ALL;
        // the side path only exists where the pooling reads token 0 and at least one MLP site is e4m3
        t->mlp_cls = d->fp8_mlp_cls_bf16 && d->kind == IVR_KIND_VISION && (t->sites & (IVR_FP8_SITE_FC1 | IVR_FP8_SITE_FC2));
    }
    *out = t;
    return IVR_OK;
}

int ivr_tower_destroy(ivr_tower *t) {
    if (!t) return IVR_OK;
    for (auto &kv : t->w) (void)hipFree(kv.second.ptr);
    if (t->ws) (void)hipFree(t->ws);
    delete t;
    return IVR_OK;
}

int64_t ivr_tower_workspace_bytes(ivr_tower *t) { return t ? (int64_t)t->ws_bytes : 0; }

}  // extern "C"


namespace {

struct Spec {
    std::string name;
    int64_t count;
};

// the tensors a tower needs (mirror of ivr_amd/weights.py tensor_specs)
std::vector<Spec> tower_specs(const ivr_tower_desc &d) {
    std::vector<Spec> v;
    const int64_t D = d.width, M = d.mlp;
    if (d.kind == IVR_KIND_VISION) {
        v.push_back({"patch_w", D * 3 * d.patch * d.patch});
        if (d.patch_bias) v.push_back({"patch_b", D});
        v.push_back({"cls", D});
        v.push_back({"pos", (int64_t)d.tokens * D});
        if (d.pre_ln) {
            v.push_back({"pre_ln_g", D});
            v.push_back({"pre_ln_b", D});
        }
    } else {
        v.push_back({"tok", (int64_t)d.vocab * D});
        v.push_back({"pos", (int64_t)d.tokens * D});
    }
    for (int i = 0; i < d.layers; ++i) {
        const std::string p = "l" + std::to_string(i) + ".";
        v.push_back({p + "ln1_g", D});
        v.push_back({p + "ln1_b", D});
        for (const char *n : {"q", "k", "v", "o"}) {
            v.push_back({p + n + "_w", D * D});
            v.push_back({p + n + "_b", D});
        }
        v.push_back({p + "ln2_g", D});
        v.push_back({p + "ln2_b", D});
        v.push_back({p + "fc1_w", M * D});
        v.push_back({p + "fc1_b", M});
        v.push_back({p + "fc2_w", D * M});
        v.push_back({p + "fc2_b", D});
    }
    v.push_back({"post_ln_g", D});
    v.push_back({"post_ln_b", D});
    if (d.out_dim) v.push_back({"proj_w", (int64_t)d.out_dim * D});
    return v;
}

unsigned short host_bf16(float f) {
    uint32_t u;
    memcpy(&u, &f, 4);
    if ((u & 0x7fffffffu) > 0x7f800000u) return (unsigned short)((u >> 16) | 0x40);
    return (unsigned short)((u + 0x7fffu + ((u >> 16) & 1u)) >> 16);
}

// upload `count` floats either as f32 or converted to bf16 (GEMM operands in bf16 mode)
int upload(ivr_tower *t, const std::string &name, const float *host, int64_t count, bool as_compute) {
    const bool bf16 = as_compute && t->d.compute != IVR_COMPUTE_F32;
    TowerTensor tt;
    tt.count = count;
    tt.bytes_per = bf16 ? 2 : 4;
    IVR_HIP(hipMalloc(&tt.ptr, (size_t)count * tt.bytes_per));
    if (bf16) {
        std::vector<unsigned short> tmp((size_t)count);
        for (int64_t i = 0; i < count; ++i) tmp[(size_t)i] = host_bf16(host[i]);
        IVR_HIP(hipMemcpy(tt.ptr, tmp.data(), (size_t)count * 2, hipMemcpyHostToDevice));
    } else {
        IVR_HIP(hipMemcpy(tt.ptr, host, (size_t)count * 4, hipMemcpyHostToDevice));
    }
    auto it = t->w.find(name);
    if (it != t->w.end()) {
        (void)hipFree(it->second.ptr);
        t->w.erase(it);
    }
    t->w[name] = tt;
    return IVR_OK;
}

// float32 -> OCP e4m3 (bias 7, no infinity, max 448), round to nearest even, saturating
unsigned char host_e4m3(float f) {
    uint32_t u;
    memcpy(&u, &f, 4);
    const unsigned char sign = (unsigned char)((u >> 24) & 0x80);
    const float a = fabsf(f);
    if (a != a) return (unsigned char)(sign | 0x7f);
    if (a >= 448.f) return (unsigned char)(sign | 0x7e);
    if (a < 0.015625f) return (unsigned char)(sign | (unsigned char)nearbyintf(a * 512.f));   // subnormals: multiples of 2^-9 (8 -> 2^-6)
    int e;
    const float m = frexpf(a, &e);                       // a = m * 2^e, m in [0.5, 1)
    int E = e - 1, mant = (int)nearbyintf((m * 2.f - 1.f) * 8.f);
    if (mant == 8) {
        mant = 0;
        ++E;
    }
    if (E > 8 || (E == 8 && mant == 7)) return (unsigned char)(sign | 0x7e);
    return (unsigned char)(sign | ((E + 7) << 3) | mant);
}

// fp8 mode: weight [N, K] -> e4m3 bytes with one scale per output row (absmax / 448), stored as `name`@e4m3 and `name`@scale
int upload_fp8(ivr_tower *t, const std::string &name, const float *host, int64_t N, int64_t K) {
    std::vector<unsigned char> q((size_t)(N * K));
    std::vector<float> sc((size_t)N);
    for (int64_t n = 0; n < N; ++n) {
        float amax = 0.f;
        for (int64_t k = 0; k < K; ++k) amax = std::max(amax, fabsf(host[n * K + k]));
        const float s = amax > 0.f ? amax / 448.f : 1.f, inv = 1.f / s;
        sc[(size_t)n] = s;
        for (int64_t k = 0; k < K; ++k) q[(size_t)(n * K + k)] = host_e4m3(host[n * K + k] * inv);
    }
    TowerTensor tt;
    tt.count = N * K;
    tt.bytes_per = 1;
    IVR_HIP(hipMalloc(&tt.ptr, q.size()));
    IVR_HIP(hipMemcpy(tt.ptr, q.data(), q.size(), hipMemcpyHostToDevice));
    t->w[name + "@e4m3"] = tt;
    TowerTensor ts;
    ts.count = N;
    ts.bytes_per = 4;
    IVR_HIP(hipMalloc(&ts.ptr, (size_t)N * 4));
    IVR_HIP(hipMemcpy(ts.ptr, sc.data(), (size_t)N * 4, hipMemcpyHostToDevice));
    t->w[name + "@scale"] = ts;
    return IVR_OK;
}

template <typename T>
T *wptr(ivr_tower *t, const std::string &name) {
    auto it = t->w.find(name);
    return it == t->w.end() ? nullptr : reinterpret_cast<T *>(it->second.ptr);
}

// y = x W^T of one linear site: e4m3 operands with the weight's per-row scale when the site is in the fp8 mask, else bf16 / f32
int layer_gemm(ivr_tower *t, bool site_fp8, int epi, GemmArgs &g, const std::string &wname, hipStream_t s) {
    if (site_fp8) {
        g.W = wptr<void>(t, wname + "@e4m3");
        g.colscale = wptr<float>(t, wname + "@scale");
        return ivr_launch_gemm_fp8(epi, g, s);
    }
    g.W = wptr<void>(t, wname);
    return ivr_launch_gemm(t->d.compute == IVR_COMPUTE_F32, epi, g, s);
}

int env_zigzag() {
    static const int v = [] {
        const char *e = getenv("IVR_ZIGZAG");
        return e ? (atoi(e) != 0) : 1;
    }();
    return v;
}

// one transformer stack over `rows` = n*T residual rows already in t->resid
int run_layers(ivr_tower *t, int n, int T, hipStream_t s) {
    const ivr_tower_desc &d = t->d;
    const bool f32 = d.compute == IVR_COMPUTE_F32;

    // dtype of each site's A operand (LN output / attention output / MLP hidden): e4m3 where the consuming site is
    auto kind = [&](bool site8) { return f32 ? OUT_F32 : site8 ? OUT_FP8 : OUT_BF16; };
    const int D = d.width, rows = n * T;
    // Producer / consumer order.  Every kernel of a block streams a few hundred MB that the next one reads back; the Infinity
    // Cache (256 MiB) still holds what a kernel wrote LAST.  So consecutive kernels walk the rows in opposite directions: the
    // consumer starts where its producer just finished and takes that part from the cache instead of HBM (IVR_ZIGZAG=0: all
    // ascending, as before).  fc2 ascending -> LN1 descending -> qkv / attention ascending -> attn-out descending -> LN2
    // ascending -> fc1 descending -> fc2 ascending.
    const int zz = env_zigzag();
    int rc;
    for (int i = 0; i < d.layers; ++i) {
        const std::string p = "l" + std::to_string(i) + ".";
        const int sites = i >= d.fp8_first_layer ? t->sites : 0;      // blocks in front of fp8_first_layer run in bf16
        const bool q8 = sites & IVR_FP8_SITE_QKV, o8 = sites & IVR_FP8_SITE_ATTN_OUT, f18 = sites & IVR_FP8_SITE_FC1, f28 = sites & IVR_FP8_SITE_FC2;
        const bool mlp_cls = t->mlp_cls && (f18 || f28);
        if (t->debug_out && t->debug_layer == i)
            IVR_HIP(hipMemcpyAsync(t->debug_out, t->resid, (size_t)rows * D * 4, hipMemcpyDeviceToDevice, s));
        rc = ivr_launch_layernorm(kind(q8), t->resid, 1, nullptr, wptr<float>(t, p + "ln1_g"), wptr<float>(t, p + "ln1_b"), d.ln_eps,
                                  t->xn, rows, D, s, zz);
        if (rc) return rc;
        GemmArgs g;
        if (!f32 && !q8 && ivr_fused_qkv_attention_ok(rows, T, D, d.heads, d.causal)) {
            // short sequences, bf16: projection and attention in one kernel, the QKV activations stay in LDS
            rc = ivr_launch_qkv_attention(t->xn, wptr<void>(t, p + "qkv_w"), wptr<float>(t, p + "qkv_b"), t->att, n, T, D, d.heads, o8, s);
            if (rc) return rc;
        } else {
            g.A = t->xn;
            g.lda = D;
            g.ldw = D;
            g.M = rows;
            g.N = 3 * D;
            g.K = D;
            g.bias = wptr<float>(t, p + "qkv_b");
            g.out = t->qkv;                  // bf16 in the fp8 mode too: the attention products stay on the bf16 MFMA
            g.ldo = 3 * D;
            g.tag = "gemm_qkv";
            rc = layer_gemm(t, q8, EPI_STORE, g, p + "qkv_w", s);                  // ascending, and so is the attention kernel
            if (rc) return rc;
            rc = ivr_launch_attention(f32, t->qkv, t->att, n, T, D, d.heads, d.causal, s, o8);
            if (rc) return rc;
        }
        g = GemmArgs();
        g.A = t->att;
        g.lda = D;
        g.ldw = D;
        g.M = rows;
        g.N = D;
        g.K = D;
        g.bias = wptr<float>(t, p + "o_b");
        g.resid = t->resid;
        g.ldr = D;
        g.tag = "gemm_attn_out";
        g.reverse_m = zz;
        rc = layer_gemm(t, o8, EPI_RESID, g, p + "o_w", s);
        if (rc) return rc;
        rc = ivr_launch_layernorm(kind(f18), t->resid, 1, nullptr, wptr<float>(t, p + "ln2_g"), wptr<float>(t, p + "ln2_b"), d.ln_eps,
                                  t->xn, rows, D, s);
        if (rc) return rc;
        if (mlp_cls) {                   // token-0 rows of the same LayerNorm in bf16 for the side path
            rc = ivr_launch_layernorm(OUT_BF16, t->resid, T, nullptr, wptr<float>(t, p + "ln2_g"), wptr<float>(t, p + "ln2_b"), d.ln_eps,
                                      t->xn_cls, n, D, s);
            if (rc) return rc;
        }
        g = GemmArgs();
        g.A = t->xn;
        g.lda = D;
        g.ldw = D;
        g.M = rows;
        g.N = d.mlp;
        g.K = D;
        g.bias = wptr<float>(t, p + "fc1_b");
        g.out = t->hid;
        g.ldo = d.mlp;
        g.act = d.act;
        g.out8 = f18 && f28;             // the e4m3 GEMM writes fc2's e4m3 operand itself
        g.reverse_m = zz;
        g.tag = "gemm_fc1";
        rc = layer_gemm(t, f18, EPI_STORE, g, p + "fc1_w", s);
        if (rc) return rc;
        const void *hid_a = t->hid;
        if (!f18 && f28) {               // bf16 fc1 feeding an e4m3 fc2 (error-budget configurations only): converted copy behind it
            unsigned char *h8 = reinterpret_cast<unsigned char *>(t->hid) + (size_t)t->max_batch * d.tokens * d.mlp * 2;
            rc = ivr_launch_bf16_to_e4m3(t->hid, h8, (int64_t)rows * d.mlp, s);
            if (rc) return rc;
            hid_a = h8;
        }
        g = GemmArgs();
        g.A = hid_a;
        g.lda = d.mlp;
        g.ldw = d.mlp;
        g.M = rows;
        g.N = D;
        g.K = d.mlp;
        g.bias = wptr<float>(t, p + "fc2_b");
        g.resid = t->resid;
        g.ldr = D;
        g.skip_mod = mlp_cls ? T : 0;
        g.tag = "gemm_fc2";
        rc = layer_gemm(t, f28, EPI_RESID, g, p + "fc2_w", s);
        if (rc) return rc;
        if (mlp_cls) {
            // side path: the n token-0 rows through fc1 / fc2 in bf16 (rows of image i sit T*D apart in the residual stream)
            g = GemmArgs();
            g.A = t->xn_cls;
            g.lda = D;
            g.ldw = D;
            g.M = n;
            g.N = d.mlp;
            g.K = D;
            g.bias = wptr<float>(t, p + "fc1_b");
            g.out = t->hid_cls;
            g.ldo = d.mlp;
            g.act = d.act;
            g.tag = "gemm_fc1_cls";
            rc = layer_gemm(t, false, EPI_STORE, g, p + "fc1_w", s);
            if (rc) return rc;
            g = GemmArgs();
            g.A = t->hid_cls;
            g.lda = d.mlp;
            g.ldw = d.mlp;
            g.M = n;
            g.N = D;
            g.K = d.mlp;
            g.bias = wptr<float>(t, p + "fc2_b");
            g.resid = t->resid;
            g.ldr = T * D;
            g.tag = "gemm_fc2_cls";
            rc = layer_gemm(t, false, EPI_RESID, g, p + "fc2_w", s);
            if (rc) return rc;
        }
    }
    if (t->debug_out && t->debug_layer == d.layers)
        IVR_HIP(hipMemcpyAsync(t->debug_out, t->resid, (size_t)rows * D * 4, hipMemcpyDeviceToDevice, s));
    t->debug_out = nullptr;
    return IVR_OK;
}

// pooled rows -> LN -> (projection) -> optional L2 normalise -> out
int run_pool(ivr_tower *t, int n, int T, const int *offs, int normalize, float *out, hipStream_t s) {
    const ivr_tower_desc &d = t->d;
    const bool f32 = d.compute == IVR_COMPUTE_F32;
    const int D = d.width;
    int rc;
    if (d.pool == IVR_POOL_LN_ALL_CLS) {
        rc = ivr_launch_layernorm(OUT_F32, t->resid, T, offs, wptr<float>(t, "post_ln_g"), wptr<float>(t, "post_ln_b"), d.ln_eps,
                                  t->pooled_f32, n, D, s);
        if (rc) return rc;
        return ivr_launch_f_normalize(t->pooled_f32, out, n, D, normalize, s);
    }
    rc = ivr_launch_layernorm(f32 ? OUT_F32 : OUT_BF16, t->resid, T, offs, wptr<float>(t, "post_ln_g"), wptr<float>(t, "post_ln_b"), d.ln_eps, t->pool, n,
                              D, s);
    if (rc) return rc;
    GemmArgs g;
    g.A = t->pool;
    g.lda = D;
    g.W = wptr<void>(t, "proj_w");
    g.ldw = D;
    g.M = n;
    g.N = d.out_dim;
    g.K = D;
    g.out = t->pooled_f32;
    g.ldo = d.out_dim;
    g.tag = "gemm_proj";
    rc = ivr_launch_gemm(f32, EPI_F32, g, s);
    if (rc) return rc;
    return ivr_launch_f_normalize(t->pooled_f32, out, n, d.out_dim, normalize, s);
}

}  // namespace

extern "C" {

int ivr_tower_set_weight(ivr_tower *t, const char *name, const float *data, int64_t count) {
    IVR_REQUIRE(t && name && data, "ivr_tower_set_weight: NULL argument");
    std::lock_guard<std::mutex> lk(t->mu);
    IVR_REQUIRE(!t->finalized, "ivr_tower_set_weight: tower already finalized");
    int64_t want = -1;
    for (const Spec &sp : tower_specs(t->d))
        if (sp.name == name) want = sp.count;
    IVR_REQUIRE(want >= 0, "ivr_tower_set_weight: unknown tensor '%s' for this tower", name);
    IVR_REQUIRE(want == count, "ivr_tower_set_weight: '%s' has %lld elements, expected %lld", name, (long long)count, (long long)want);
    t->host[name].assign(data, data + count);
    return IVR_OK;
}

int ivr_tower_finalize(ivr_tower *t, int max_batch) {
    IVR_REQUIRE(t, "ivr_tower_finalize: NULL tower");
    IVR_REQUIRE(max_batch >= 1 && (int64_t)max_batch * t->d.tokens < (1ll << 24), "ivr_tower_finalize: max_batch=%d", max_batch);
    std::lock_guard<std::mutex> lk(t->mu);
    IVR_REQUIRE(!t->finalized, "ivr_tower_finalize: already finalized");
    const ivr_tower_desc &d = t->d;
    for (const Spec &sp : tower_specs(d))
        if (!t->host.count(sp.name)) return ivr_fail(IVR_ERR_STATE, "ivr_tower_finalize: tensor '%s' was never set", sp.name.c_str());
    IVR_HIP(hipSetDevice(t->ctx->device));
    const int D = d.width;
    int rc;
    auto H = [&](const std::string &n) -> std::vector<float> & { return t->host[n]; };
    if (d.kind == IVR_KIND_VISION) {
        // conv weight [D, 3*P*P] -> K padded to a multiple of 64 (zeros)
        const int K = 3 * d.patch * d.patch;
        t->kpad = (int)ivr_round_up(K, 64);
        std::vector<float> pw((size_t)D * t->kpad, 0.f);
        for (int o = 0; o < D; ++o) memcpy(&pw[(size_t)o * t->kpad], &H("patch_w")[(size_t)o * K], (size_t)K * 4);
        if ((rc = upload(t, "patch_w", pw.data(), (int64_t)pw.size(), true))) return rc;
        if (d.patch_bias && (rc = upload(t, "patch_b", H("patch_b").data(), D, false))) return rc;
        if ((rc = upload(t, "cls", H("cls").data(), D, false))) return rc;
        if (d.pre_ln) {
            if ((rc = upload(t, "pre_ln_g", H("pre_ln_g").data(), D, false))) return rc;
            if ((rc = upload(t, "pre_ln_b", H("pre_ln_b").data(), D, false))) return rc;
        }
    } else {
        if ((rc = upload(t, "tok", H("tok").data(), (int64_t)d.vocab * D, false))) return rc;
    }
    if ((rc = upload(t, "pos", H("pos").data(), (int64_t)d.tokens * D, false))) return rc;
    const float scale = 1.0f / sqrtf((float)(D / d.heads));   // head_dim^-0.5 = 0.125: exact power of two
    for (int i = 0; i < d.layers; ++i) {
        const std::string p = "l" + std::to_string(i) + ".";
        // fused QKV weight [3D, D] and bias [3D]; the attention scale is folded into the Q rows
        std::vector<float> qkv((size_t)3 * D * D), qb((size_t)3 * D);
        for (size_t j = 0; j < (size_t)D * D; ++j) qkv[j] = H(p + "q_w")[j] * scale;
        memcpy(&qkv[(size_t)D * D], H(p + "k_w").data(), (size_t)D * D * 4);
        memcpy(&qkv[(size_t)2 * D * D], H(p + "v_w").data(), (size_t)D * D * 4);
        for (int j = 0; j < D; ++j) qb[j] = H(p + "q_b")[j] * scale;
        memcpy(&qb[D], H(p + "k_b").data(), (size_t)D * 4);
        memcpy(&qb[2 * D], H(p + "v_b").data(), (size_t)D * 4);
        // every site gets the operand dtype it runs in; the MLP sites keep a bf16 copy too when token 0 takes the side path
        const struct {
            const char *name;
            const float *host;
            int64_t N, K;
            int bit;
            bool also_bf16;
        } site[4] = {{"qkv_w", qkv.data(), 3 * D, D, IVR_FP8_SITE_QKV, false},
                     {"o_w", H(p + "o_w").data(), D, D, IVR_FP8_SITE_ATTN_OUT, false},
                     {"fc1_w", H(p + "fc1_w").data(), d.mlp, D, IVR_FP8_SITE_FC1, t->mlp_cls},
                     {"fc2_w", H(p + "fc2_w").data(), D, d.mlp, IVR_FP8_SITE_FC2, t->mlp_cls}};
        for (const auto &st : site) {
            const bool s8 = i >= d.fp8_first_layer && (t->sites & st.bit);
            if (s8 && (rc = upload_fp8(t, p + st.name, st.host, st.N, st.K))) return rc;
            if ((!s8 || st.also_bf16) && (rc = upload(t, p + st.name, st.host, st.N * st.K, true))) return rc;
        }
        if ((rc = upload(t, p + "qkv_b", qb.data(), 3 * D, false))) return rc;
        for (const char *n : {"ln1_g", "ln1_b", "ln2_g", "ln2_b", "o_b", "fc2_b"})
            if ((rc = upload(t, p + n, H(p + n).data(), D, false))) return rc;
        if ((rc = upload(t, p + "fc1_b", H(p + "fc1_b").data(), d.mlp, false))) return rc;
    }
    if ((rc = upload(t, "post_ln_g", H("post_ln_g").data(), D, false))) return rc;
    if ((rc = upload(t, "post_ln_b", H("post_ln_b").data(), D, false))) return rc;
    if (d.out_dim && (rc = upload(t, "proj_w", H("proj_w").data(), (int64_t)d.out_dim * D, true))) return rc;
    t->host.clear();

    // activation workspace, one allocation
    const size_t es = d.compute == IVR_COMPUTE_F32 ? 4 : 2;                       // qkv, pooled rows
    const size_t ea = es;        // GEMM A operands (LN out, attention out, MLP hidden): sized for bf16, e4m3 sites use half of it
    const size_t rows = (size_t)max_batch * d.tokens;
    auto al = [](size_t b) { return (b + 255) & ~(size_t)255; };
    const size_t b_resid = al(rows * D * 4), b_xn = al(rows * D * ea), b_qkv = al(rows * 3 * D * es), b_att = al(rows * D * ea),
                 b_hid = al(rows * d.mlp * (ea + (((t->sites & IVR_FP8_SITE_FC2) && !(t->sites & IVR_FP8_SITE_FC1)) ? 1 : 0))), b_pool = al((size_t)max_batch * D * es),
                 b_pf = al((size_t)max_batch * std::max(D, d.out_dim) * 4), b_eos = al((size_t)max_batch * 4),
                 b_xc = t->mlp_cls ? al((size_t)max_batch * D * 2) : 0, b_hc = t->mlp_cls ? al((size_t)max_batch * d.mlp * 2) : 0;
    t->ws_bytes = b_resid + b_xn + b_qkv + b_att + b_hid + b_pool + b_pf + b_eos + b_xc + b_hc;
    IVR_HIP(hipMalloc(&t->ws, t->ws_bytes));
    IVR_HIP(hipMemset(t->ws, 0, t->ws_bytes));
    unsigned char *p = reinterpret_cast<unsigned char *>(t->ws);
    t->resid = reinterpret_cast<float *>(p);
    p += b_resid;
    t->xn = p;
    p += b_xn;
    t->qkv = p;
    p += b_qkv;
    t->att = p;
    p += b_att;
    t->hid = p;
    p += b_hid;
    t->pool = p;
    p += b_pool;
    t->pooled_f32 = reinterpret_cast<float *>(p);
    p += b_pf;
    t->eos_pos = reinterpret_cast<int *>(p);
    p += b_eos;
    t->xn_cls = p;
    p += b_xc;
    t->hid_cls = p;
    t->max_batch = max_batch;
    t->finalized = true;
    return IVR_OK;
}

int ivr_tower_encode_image(ivr_tower *t, const void *patches, int n, int normalize, float *out, ivr_stream stream) {
    IVR_REQUIRE(t && (n == 0 || (patches && out)), "ivr_tower_encode_image: NULL argument");
    std::lock_guard<std::mutex> lk(t->mu);
    if (!t->finalized) return ivr_fail(IVR_ERR_STATE, "ivr_tower_encode_image: tower not finalized");
    const ivr_tower_desc &d = t->d;
    IVR_REQUIRE(d.kind == IVR_KIND_VISION, "ivr_tower_encode_image: not a vision tower");
    IVR_REQUIRE(n >= 0 && n <= t->max_batch, "ivr_tower_encode_image: n=%d exceeds max_batch=%d", n, t->max_batch);
    if (n == 0) return IVR_OK;
    IVR_HIP(hipSetDevice(t->ctx->device));
    hipStream_t s = (hipStream_t)stream;
    const bool f32 = d.compute == IVR_COMPUTE_F32;
    const int D = d.width, T = d.tokens, G2 = T - 1;
    int rc = ivr_launch_vision_cls(t->resid, wptr<float>(t, "cls"), wptr<float>(t, "pos"), n, T, D, s);
    if (rc) return rc;
    GemmArgs g;
    g.A = patches;
    g.lda = t->kpad;
    g.W = wptr<void>(t, "patch_w");
    g.ldw = t->kpad;
    g.M = n * G2;
    g.N = D;
    g.K = t->kpad;
    g.bias = d.patch_bias ? wptr<float>(t, "patch_b") : nullptr;
    g.resid = t->resid;
    g.ldr = D;
    g.pos = wptr<float>(t, "pos");
    g.T = T;
    g.G2 = G2;
    g.tag = "gemm_patch";
    rc = ivr_launch_gemm(f32, EPI_PATCH, g, s);
    if (rc) return rc;
    if (d.pre_ln) {
        rc = ivr_launch_layernorm(OUT_F32, t->resid, 1, nullptr, wptr<float>(t, "pre_ln_g"), wptr<float>(t, "pre_ln_b"), d.ln_eps, t->resid,
                                  n * T, D, s);
        if (rc) return rc;
    }
    rc = run_layers(t, n, T, s);
    if (rc) return rc;
    return run_pool(t, n, T, nullptr, normalize, out, s);
}

int ivr_tower_encode_text(ivr_tower *t, const int64_t *ids, int q, int T, int normalize, float *out, ivr_stream stream) {
    IVR_REQUIRE(t && (q == 0 || (ids && out)), "ivr_tower_encode_text: NULL argument");
    std::lock_guard<std::mutex> lk(t->mu);
    if (!t->finalized) return ivr_fail(IVR_ERR_STATE, "ivr_tower_encode_text: tower not finalized");
    const ivr_tower_desc &d = t->d;
    IVR_REQUIRE(d.kind == IVR_KIND_TEXT, "ivr_tower_encode_text: not a text tower");
    IVR_REQUIRE(T >= 1 && T <= d.tokens, "ivr_tower_encode_text: T=%d outside [1,%d]", T, d.tokens);
    IVR_REQUIRE(q >= 0 && (int64_t)q * T <= (int64_t)t->max_batch * d.tokens && q <= t->max_batch,
                "ivr_tower_encode_text: q=%d exceeds max_batch=%d", q, t->max_batch);
    if (q == 0) return IVR_OK;
    IVR_HIP(hipSetDevice(t->ctx->device));
    hipStream_t s = (hipStream_t)stream;
    int rc = ivr_launch_text_embed(t->resid, ids, wptr<float>(t, "tok"), wptr<float>(t, "pos"), q, T, d.width, d.vocab, d.eos_id,
                                   t->eos_pos, s);
    if (rc) return rc;
    rc = run_layers(t, q, T, s);
    if (rc) return rc;
    return run_pool(t, q, T, t->eos_pos, normalize, out, s);
}

int ivr_linear(ivr_ctx *ctx, int f32_mode, int epilogue, const void *x, const void *w, const float *bias, int M, int N, int K, int act,
               void *out, float *resid, ivr_stream stream) {
    IVR_REQUIRE(ctx && x && w, "ivr_linear: NULL argument");
    IVR_REQUIRE(epilogue == EPI_STORE || epilogue == EPI_RESID || epilogue == EPI_F32, "ivr_linear: epilogue=%d", epilogue);
    IVR_REQUIRE(M >= 0 && N >= 4 && K >= 1, "ivr_linear: M=%d N=%d K=%d", M, N, K);
    IVR_REQUIRE(epilogue == EPI_RESID ? resid != nullptr : out != nullptr, "ivr_linear: NULL output");
    IVR_HIP(hipSetDevice(ctx->device));
    GemmArgs g;
    g.A = x;
    g.lda = K;
    g.W = w;
    g.ldw = K;
    g.M = M;
    g.N = N;
    g.K = K;
    g.bias = bias;
    g.out = out;
    g.ldo = N;
    g.resid = resid;
    g.ldr = N;
    g.act = act;
    g.tag = "linear";
    return ivr_launch_gemm(f32_mode != 0, epilogue, g, (hipStream_t)stream);
}

int ivr_linear_fp8(ivr_ctx *ctx, int epilogue, const void *x, const void *w, const float *colscale, const float *bias, int M, int N, int K,
                   int act, void *out, int out_fp8, float *resid, ivr_stream stream) {
    IVR_REQUIRE(ctx && x && w, "ivr_linear_fp8: NULL argument");
    IVR_REQUIRE(epilogue == EPI_STORE || epilogue == EPI_RESID, "ivr_linear_fp8: epilogue=%d", epilogue);
    IVR_REQUIRE(M >= 0 && N >= 64 && K >= 128, "ivr_linear_fp8: M=%d N=%d K=%d", M, N, K);
    IVR_REQUIRE(epilogue == EPI_RESID ? resid != nullptr : out != nullptr, "ivr_linear_fp8: NULL output");
    IVR_HIP(hipSetDevice(ctx->device));
    GemmArgs g;
    g.A = x;
    g.lda = K;
    g.W = w;
    g.ldw = K;
    g.M = M;
    g.N = N;
    g.K = K;
    g.bias = bias;
    g.colscale = colscale;
    g.out = out;
    g.ldo = N;
    g.out8 = out_fp8;
    g.resid = resid;
    g.ldr = N;
    g.act = act;
    g.tag = "linear_fp8";
    return ivr_launch_gemm_fp8(epilogue, g, (hipStream_t)stream);
}

int ivr_quantize_e4m3_host(const float *src, uint8_t *dst, int64_t n) {
    IVR_REQUIRE(n >= 0 && (n == 0 || (src && dst)), "ivr_quantize_e4m3_host: NULL argument");
    for (int64_t i = 0; i < n; ++i) dst[i] = host_e4m3(src[i]);
    return IVR_OK;
}

#ifdef IVR_GEMM_STAMPS
// diagnostic build only: launch the fused QKV + attention kernel of layer 0 once more on the workspace of the last call
int ivr_debug_last_qkv_attention(ivr_tower *t, int n, ivr_stream stream) {
    const ivr_tower_desc &d = t->d;
    return ivr_launch_qkv_attention(t->xn, wptr<void>(t, "l0.qkv_w"), wptr<float>(t, "l0.qkv_b"), t->att, n, d.tokens, d.width, d.heads, false,
                                    (hipStream_t)stream);
}
#endif

int ivr_tower_debug_hidden(ivr_tower *t, int layer, int n, float *out, ivr_stream) {
    IVR_REQUIRE(t && out, "ivr_tower_debug_hidden: NULL argument");
    std::lock_guard<std::mutex> lk(t->mu);
    IVR_REQUIRE(layer >= 0 && layer <= t->d.layers, "ivr_tower_debug_hidden: layer=%d outside [0,%d]", layer, t->d.layers);
    (void)n;
    // one-shot: the NEXT encode call copies the residual stream after `layer` blocks into `out`
    t->debug_layer = layer;
    t->debug_out = out;
    return IVR_OK;
}

}  // extern "C"
