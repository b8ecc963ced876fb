// Shared helpers of the C++ API tests (mirror of the reference's tests/unit_tests drivers, but with
// exit codes: any mismatch -> non-zero).  The oracle (oracle/llmie_oracle.h) is linked as the checker.
#pragma once
#include <cmath>
#include <cstdio>
#include <cstring>
#include <random>
#include <vector>

#include "../../oracle/llmie_oracle.h"
#include "../src/utils/tensor.h"

static int g_failures = 0;

template <typename T> struct DeviceArray {
    T *d = nullptr;
    size_t n = 0;
    explicit DeviceArray(size_t n_) : n(n_) { CHECK(hipMalloc(reinterpret_cast<void **>(&d), sizeof(T) * (n ? n : 1))); }
    DeviceArray(const std::vector<T> &h) : DeviceArray(h.size()) { upload(h); }
    ~DeviceArray() { (void)hipFree(d); }
    DeviceArray(const DeviceArray &) = delete;
    void upload(const std::vector<T> &h) { CHECK(hipMemcpy(d, h.data(), sizeof(T) * h.size(), hipMemcpyHostToDevice)); }
    std::vector<T> download() const {
        std::vector<T> h(n);
        CHECK(hipMemcpy(h.data(), d, sizeof(T) * n, hipMemcpyDeviceToHost));
        return h;
    }
};

inline std::vector<half> to_half(const std::vector<float> &v) {
    std::vector<half> h(v.size());
    for (size_t i = 0; i < v.size(); ++i) h[i] = __float2half(v[i]);
    return h;
}
inline std::vector<float> to_float(const std::vector<half> &v) {
    std::vector<float> f(v.size());
    for (size_t i = 0; i < v.size(); ++i) f[i] = __half2float(v[i]);
    return f;
}
inline std::vector<float> to_float(const std::vector<float> &v) { return v; }
template <typename T> std::vector<T> cast_vec(const std::vector<float> &v);
template <> inline std::vector<float> cast_vec<float>(const std::vector<float> &v) { return v; }
template <> inline std::vector<half> cast_vec<half>(const std::vector<float> &v) { return to_half(v); }
// round-trip through the storage type so the oracle sees exactly the device inputs
template <typename T> std::vector<float> storage_round(const std::vector<float> &v) { return to_float(cast_vec<T>(v)); }

inline bool check_close(const char *what, const std::vector<float> &got, const std::vector<float> &exp, float rtol, float atol) {
    if (got.size() != exp.size()) {
        std::printf("FAIL %s: size %zu vs %zu\n", what, got.size(), exp.size());
        ++g_failures;
        return false;
    }
    for (size_t i = 0; i < got.size(); ++i) {
        const float err = std::fabs(got[i] - exp[i]);
        if (!(err <= atol + rtol * std::fabs(exp[i]))) {
            std::printf("FAIL %s: index %zu expected %g got %g\n", what, i, exp[i], got[i]);
            ++g_failures;
            return false;
        }
    }
    std::printf("%s passed\n", what);
    return true;
}
template <typename U> inline bool check_equal(const char *what, const std::vector<U> &got, const std::vector<U> &exp) {
    if (got.size() != exp.size() || std::memcmp(got.data(), exp.data(), sizeof(U) * got.size()) != 0) {
        std::printf("FAIL %s: not bit-identical\n", what);
        ++g_failures;
        return false;
    }
    std::printf("%s passed\n", what);
    return true;
}
inline std::vector<float> randn(std::mt19937_64 &rng, size_t n, float scale) {
    std::normal_distribution<float> d(0.f, scale);
    std::vector<float> v(n);
    for (auto &x : v) x = d(rng);
    return v;
}
inline std::vector<float> randu(std::mt19937_64 &rng, size_t n, float a) {
    std::uniform_real_distribution<float> d(-a, a);
    std::vector<float> v(n);
    for (auto &x : v) x = d(rng);
    return v;
}
