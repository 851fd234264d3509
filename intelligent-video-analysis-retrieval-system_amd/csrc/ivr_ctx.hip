// Context, error slot and scratch management for libivr_hip.so.
#include "ivr_common.h"

#include <cstring>

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

int ivr_ctx_scratch(ivr_ctx *ctx, size_t bytes, void **out) {
    std::lock_guard<std::mutex> lk(ctx->mu);
    if (bytes > ctx->scratch_bytes) {
        if (ctx->scratch) IVR_HIP(hipFree(ctx->scratch));
        ctx->scratch = nullptr;
        ctx->scratch_bytes = 0;
        size_t want = (size_t)ivr_round_up((int64_t)bytes, 1 << 20);
        IVR_HIP(hipMalloc(&ctx->scratch, want));
        ctx->scratch_bytes = want;
    }
    *out = ctx->scratch;
    return IVR_OK;
}

extern "C" {

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
    if (ctx->scratch) (void)hipFree(ctx->scratch);
    delete ctx;
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
