// Context, error slot and scratch management for libivr_hip.so.
#include "ivr_common.h"

#include <algorithm>
#include <array>
#include <atomic>
#include <cstring>
#include <vector>

std::string &ivr_err_slot() {
    static thread_local std::string s;
    return s;
}

int ivr_fail(int code, const char *fmt, ...) {
    char buf[1024];
    va_list ap;
    va_start(ap, fmt);
    vsnprintf(buf, sizeof(buf), fmt, ap);
    va_end(ap);
    ivr_err_slot() = buf;
    return code;
}

int ivr_ctx_scratch(ivr_ctx *ctx, hipStream_t stream, size_t bytes, void **out) {
    std::lock_guard<std::mutex> lk(ctx->mu);
    ivr_ctx::Scratch &b = ctx->scratch[stream];
    if (bytes > b.bytes) {
        hipStreamCaptureStatus cap = hipStreamCaptureStatusNone;
        // the legacy null stream cannot be captured - and querying it while ANOTHER thread captures in global mode would
        // invalidate that capture
        if (stream != nullptr) (void)hipStreamIsCapturing(stream, &cap);
        if (cap != hipStreamCaptureStatusNone)
            return ivr_fail(IVR_ERR_STATE, "scratch of %zu bytes for a stream under graph capture: run the same call once on THIS stream "
                                           "before capturing (the block is per stream and cannot be allocated during capture)", bytes);
        // at least 1.5x the old block, so the retired list stays logarithmic in the final size
        const size_t want = (size_t)ivr_round_up((int64_t)std::max(bytes, b.bytes + b.bytes / 2), 1 << 20);
        void *p = nullptr;
        IVR_HIP(hipMalloc(&p, want));
        if (b.ptr) ctx->retired.push_back(b.ptr);
        b.ptr = p;
        b.bytes = want;
    }
    *out = b.ptr;
    return IVR_OK;
}

int ivr_func_max_lds(const void *fn, int bytes) {
    static std::mutex mu;
    static std::map<std::pair<int, const void *>, int> done;
    int dev = 0;
    IVR_HIP(hipGetDevice(&dev));
    std::lock_guard<std::mutex> lk(mu);
    int &have = done[{dev, fn}];
    if (bytes > have) {
        IVR_HIP(hipFuncSetAttribute(fn, hipFuncAttributeMaxDynamicSharedMemorySize, bytes));
        have = bytes;
    }
    return IVR_OK;
}

namespace {
struct ProfEntry {
    std::string name;
    double work;
    hipEvent_t a, b;
};
struct Prof {
    std::mutex mu;
    std::atomic<int> level{0};
    std::vector<ProfEntry> entries;
    std::vector<hipEvent_t> pool;
};
Prof &prof() {
    static Prof p;
    return p;
}
hipEvent_t prof_event(Prof &p) {
    if (!p.pool.empty()) {
        hipEvent_t e = p.pool.back();
        p.pool.pop_back();
        return e;
    }
    hipEvent_t e = nullptr;
    (void)hipEventCreate(&e);
    return e;
}
}  // namespace

int ivr_prof_level() { return prof().level.load(std::memory_order_relaxed); }

void ivr_prof_begin(const char *name, hipStream_t s, double work) {
    Prof &p = prof();
    std::lock_guard<std::mutex> lk(p.mu);
    ProfEntry e{name, work, prof_event(p), prof_event(p)};
    (void)hipEventRecord(e.a, s);
    p.entries.push_back(e);
}

void ivr_prof_end(hipStream_t s) {
    Prof &p = prof();
    std::lock_guard<std::mutex> lk(p.mu);
    if (!p.entries.empty()) (void)hipEventRecord(p.entries.back().b, s);
}

extern "C" {

int ivr_profile_enable(ivr_ctx *, int on) {
    prof().level.store(on < 0 ? 0 : on);
    return IVR_OK;
}

int ivr_profile_reset(ivr_ctx *) {
    Prof &p = prof();
    std::lock_guard<std::mutex> lk(p.mu);
    for (auto &e : p.entries) {
        p.pool.push_back(e.a);
        p.pool.push_back(e.b);
    }
    p.entries.clear();
    return IVR_OK;
}

// JSON: {"name": {"launches": n, "ms": total, "work": total}, ...}; synchronises on the recorded events
int ivr_profile_json(ivr_ctx *, char *buf, int len) {
    IVR_REQUIRE(buf && len > 2, "ivr_profile_json: bad buffer");
    Prof &p = prof();
    std::lock_guard<std::mutex> lk(p.mu);
    std::map<std::string, std::array<double, 3>> agg;
    for (auto &e : p.entries) {
        float ms = 0.f;
        if (hipEventSynchronize(e.b) != hipSuccess || hipEventElapsedTime(&ms, e.a, e.b) != hipSuccess) continue;
        auto &a = agg[e.name];
        a[0] += 1;
        a[1] += ms;
        a[2] += e.work;
    }
    std::string out = "{";
    bool first = true;
    for (auto &kv : agg) {
        char tmp[256];
        snprintf(tmp, sizeof(tmp), "%s\"%s\": {\"launches\": %.0f, \"ms\": %.6f, \"work\": %.6e}", first ? "" : ", ",
                 kv.first.c_str(), kv.second[0], kv.second[1], kv.second[2]);
        out += tmp;
        first = false;
    }
    out += "}";
    IVR_REQUIRE((int)out.size() + 1 <= len, "ivr_profile_json: buffer of %d bytes too small for %zu", len, out.size() + 1);
    memcpy(buf, out.c_str(), out.size() + 1);
    return IVR_OK;
}

int ivr_api_version(void) { return IVR_API_VERSION; }

int ivr_init(int device, ivr_ctx **out) {
    IVR_REQUIRE(out != nullptr, "ivr_init: out is NULL");
    int count = 0;
    IVR_HIP(hipGetDeviceCount(&count));
    IVR_REQUIRE(device >= 0 && device < count, "ivr_init: device %d out of range (%d visible)", device, count);
    IVR_HIP(hipSetDevice(device));
    hipDeviceProp_t prop;
    IVR_HIP(hipGetDeviceProperties(&prop, device));
    ivr_ctx *c = new ivr_ctx();
    c->device = device;
    c->cu_count = prop.multiProcessorCount;
    c->hbm_bytes = (int64_t)prop.totalGlobalMem;
    strncpy(c->arch, prop.gcnArchName, sizeof(c->arch) - 1);
    if (strncmp(c->arch, "gfx950", 6) != 0) {
        std::string a = c->arch;
        delete c;
        return ivr_fail(IVR_ERR_UNSUPPORTED, "ivr_init: built for gfx950 (MI355X), device reports %s", a.c_str());
    }
    *out = c;
    return IVR_OK;
}

int ivr_destroy(ivr_ctx *ctx) {
    if (!ctx) return IVR_OK;
    for (auto &kv : ctx->scratch)
        if (kv.second.ptr) (void)hipFree(kv.second.ptr);
    for (void *p : ctx->retired) (void)hipFree(p);
    for (auto &kv : ctx->luts) (void)hipFree(kv.second);
    delete ctx;
    return IVR_OK;
}

int ivr_release_stream_scratch(ivr_ctx *ctx, ivr_stream stream) {
    IVR_REQUIRE(ctx != nullptr, "ivr_release_stream_scratch: ctx is NULL");
    hipStream_t s = (hipStream_t)stream;
    void *p = nullptr;
    {
        std::lock_guard<std::mutex> lk(ctx->mu);
        auto it = ctx->scratch.find(s);
        if (it == ctx->scratch.end()) return IVR_OK;
        p = it->second.ptr;
        ctx->scratch.erase(it);
    }
    if (p) {
        IVR_HIP(hipSetDevice(ctx->device));
        IVR_HIP(hipStreamSynchronize(s));        // kernels of this stream may still use the block
        IVR_HIP(hipFree(p));
    }
    return IVR_OK;
}

const char *ivr_last_error(ivr_ctx *) { return ivr_err_slot().c_str(); }

int ivr_device_info(ivr_ctx *ctx, int *cu_count, int64_t *hbm_bytes, char *arch, int arch_len) {
    IVR_REQUIRE(ctx != nullptr, "ivr_device_info: ctx is NULL");
    if (cu_count) *cu_count = ctx->cu_count;
    if (hbm_bytes) *hbm_bytes = ctx->hbm_bytes;
    if (arch && arch_len > 0) {
        strncpy(arch, ctx->arch, arch_len - 1);
        arch[arch_len - 1] = 0;
    }
    return IVR_OK;
}

}  // extern "C"
