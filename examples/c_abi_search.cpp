// The drop-in boundary without Python or torch: a plain C++ host program that uses nothing but include/ivr_api.h, the HIP
// runtime for device memory and libivr_hip.so.  Builds an index of unit rows, searches it, and checks ids and scores against a
// double-precision brute force on the host.  (What a cgo / JNI / N-API binding of the same path would do.)
//
//   hipcc -O2 -Iinclude examples/c_abi_search.cpp -Lintelligent-video-analysis-retrieval-system_amd/lib -livr_hip \
//         -Wl,-rpath,$PWD/intelligent-video-analysis-retrieval-system_amd/lib -o /tmp/c_abi_search && /tmp/c_abi_search
#include <hip/hip_runtime.h>

#include <algorithm>
#include <cmath>
#include <cstdint>
#include <cstdio>
#include <random>
#include <vector>

#include "ivr_api.h"

#define CHECK_IVR(call)                                                              \
    do {                                                                             \
        const int rc_ = (call);                                                      \
        if (rc_ != 0) {                                                              \
            std::fprintf(stderr, "%s -> %d: %s\n", #call, rc_, ivr_last_error(nullptr)); \
            return 1;                                                                \
        }                                                                            \
    } while (0)
#define CHECK_HIP(call)                                                   \
    do {                                                                  \
        const hipError_t e_ = (call);                                     \
        if (e_ != hipSuccess) {                                           \
            std::fprintf(stderr, "%s: %s\n", #call, hipGetErrorString(e_)); \
            return 1;                                                     \
        }                                                                 \
    } while (0)

int main() {
    const int d = 512, nq = 7, k = 10;
    const int64_t n = 200000;
    std::mt19937 gen(1234);
    std::normal_distribution<float> nd(0.f, 1.f);
    std::vector<float> rows((size_t)n * d), q((size_t)nq * d);
    for (float &v : rows) v = nd(gen);
    for (float &v : q) v = nd(gen);

    ivr_ctx *ctx = nullptr;
    ivr_index *index = nullptr;
    CHECK_IVR(ivr_init(0, &ctx));
    CHECK_IVR(ivr_index_create(ctx, d, n, &index));
    float *d_rows = nullptr, *d_q = nullptr, *d_D = nullptr, *d_back = nullptr;
    int64_t *d_I = nullptr;
    CHECK_HIP(hipMalloc(&d_rows, rows.size() * 4));
    CHECK_HIP(hipMalloc(&d_back, rows.size() * 4));
    CHECK_HIP(hipMalloc(&d_q, q.size() * 4));
    CHECK_HIP(hipMalloc(&d_D, (size_t)nq * k * 4));
    CHECK_HIP(hipMalloc(&d_I, (size_t)nq * k * 8));
    CHECK_HIP(hipMemcpy(d_rows, rows.data(), rows.size() * 4, hipMemcpyHostToDevice));
    CHECK_HIP(hipMemcpy(d_q, q.data(), q.size() * 4, hipMemcpyHostToDevice));
    CHECK_IVR(ivr_index_add(index, d_rows, n, /*normalize=*/1, nullptr));
    CHECK_IVR(ivr_index_search(index, d_q, nq, k, /*normalize_q=*/1, /*id_base=*/0, d_D, d_I, nullptr));
    CHECK_IVR(ivr_index_reconstruct(index, 0, n, d_back, nullptr));          // the rows as stored (normalised)
    CHECK_HIP(hipDeviceSynchronize());
    std::vector<float> D((size_t)nq * k), stored(rows.size());
    std::vector<int64_t> I((size_t)nq * k);
    CHECK_HIP(hipMemcpy(D.data(), d_D, D.size() * 4, hipMemcpyDeviceToHost));
    CHECK_HIP(hipMemcpy(I.data(), d_I, I.size() * 8, hipMemcpyDeviceToHost));
    CHECK_HIP(hipMemcpy(stored.data(), d_back, stored.size() * 4, hipMemcpyDeviceToHost));

    int bad = 0;
    double worst = 0.0;
    for (int qi = 0; qi < nq; ++qi) {
        double qn = 0.0;
        for (int j = 0; j < d; ++j) qn += (double)q[(size_t)qi * d + j] * q[(size_t)qi * d + j];
        qn = std::sqrt(qn);
        std::vector<std::pair<double, int64_t>> s((size_t)n);
        for (int64_t r = 0; r < n; ++r) {
            double acc = 0.0;
            for (int j = 0; j < d; ++j) acc += (double)stored[(size_t)r * d + j] * (q[(size_t)qi * d + j] / qn);
            s[(size_t)r] = {-acc, r};                                         // descending score, ascending id
        }
        std::partial_sort(s.begin(), s.begin() + k, s.end());
        for (int j = 0; j < k; ++j) {
            const bool near_tie = j + 1 < k && std::fabs(s[j].first - s[j + 1].first) < 1e-6;
            if (I[(size_t)qi * k + j] != s[(size_t)j].second && !near_tie && !(j > 0 && std::fabs(s[j].first - s[j - 1].first) < 1e-6)) ++bad;
            worst = std::max(worst, std::fabs((double)D[(size_t)qi * k + j] + s[(size_t)j].first));
        }
    }
    int stats[2] = {0, 0};
    CHECK_IVR(ivr_index_scan_stats(index, stats));
    std::printf("C ABI search: %d queries x %lld rows, top-%d: %d id mismatches, max |score - f64| = %.2e, bf16 scan copy %d, redone %d\n", nq,
                (long long)n, k, bad, worst, stats[0], stats[1]);
    CHECK_IVR(ivr_index_destroy(index));
    CHECK_IVR(ivr_destroy(ctx));
    return bad == 0 && worst < 1e-5 ? 0 : 2;
}
