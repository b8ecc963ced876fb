// C++ API mirror, part 1: runtime glue (HIP types, error macros, string helpers).
// Mirrors the names callers of the reference use from src/utils/macro.h:11-94,
// src/utils/string_utils.h and src/memory/memory_deleter.cuh, on HIP types.
#pragma once
#include <hip/hip_runtime.h>
#include <hip/hip_fp16.h>

#include <cstdio>
#include <cstdlib>
#include <iostream>
#include <memory>
#include <sstream>
#include <stdexcept>
#include <string>
#include <vector>

#include "../../include/llmie.h"

// ---- printf-style std::string and vector printers (string_utils.h) ----
template <typename... Args> inline std::string fmtstr(const std::string &format, Args... args) {
    const int n = std::snprintf(nullptr, 0, format.c_str(), args...);
    if (n < 0) throw std::runtime_error("fmtstr: formatting error");
    std::string out(static_cast<size_t>(n) + 1, '\0');
    std::snprintf(&out[0], out.size(), format.c_str(), args...);
    out.resize(static_cast<size_t>(n));
    return out;
}
template <typename T> inline std::string vec2str(const std::vector<T> &v) {
    std::ostringstream ss;
    ss << "(";
    for (size_t i = 0; i < v.size(); ++i) ss << (i ? ", " : "") << v[i];
    ss << ")";
    return ss.str();
}
template <typename T> inline std::string arr2str(const T *arr, size_t n) {
    std::ostringstream ss;
    ss << "(";
    for (size_t i = 0; i < n; ++i) ss << (i ? ", " : "") << arr[i];
    ss << ")";
    return ss.str();
}

// ---- CHECK(hip call): print + exit(1)   (macro.h:11-22) ----
inline void llmieCheckHip(hipError_t result, const char *file, int line) {
    if (result != hipSuccess) {
        std::cerr << "HIP Error:\n    File:       " << file << "\n    Line:       " << line
                  << "\n    Error code: " << static_cast<int>(result) << "\n    Error text: "
                  << hipGetErrorString(result) << '\n';
        std::exit(1);
    }
}
#define CHECK(call) llmieCheckHip((call), __FILE__, __LINE__)

// ---- LLM_CHECK / LLM_CHECK_WITH_INFO: throw std::runtime_error   (macro.h:74-94) ----
[[noreturn]] inline void throwRuntimeError(const char *file, int line, const std::string &info = "") {
    throw std::runtime_error("[oneLLM][ERROR] " + info + " Assertion fail: " + file + ":" + std::to_string(line) + " \n");
}
inline void llmAssert(bool ok, const char *file, int line, const std::string &info = "") {
    if (!ok) throwRuntimeError(file, line, info);
}
#define LLM_CHECK(val) llmAssert((val), __FILE__, __LINE__)
#define LLM_CHECK_WITH_INFO(val, info) llmAssert((val), __FILE__, __LINE__, (info))

// ---- device sync + last-error check: throw   (macro.h:60-71; name kept for callers) ----
inline void syncAndCheck(const char *file, int line) {
    (void)hipDeviceSynchronize();
    const hipError_t e = hipGetLastError();
    if (e != hipSuccess)
        throw std::runtime_error(std::string("[TM][ERROR] HIP runtime error: ") + hipGetErrorString(e) + " " + file +
                                 ":" + std::to_string(line) + " \n");
}
#define DeviceSyncAndCheckCudaError() syncAndCheck(__FILE__, __LINE__)

// status of a C-ABI call -> the reference's LLM_CHECK behaviour
inline void llmieCheckStatus(int rc, const char *file, int line) {
    if (rc != 0) throwRuntimeError(file, line, std::string(llmie_last_error()));
}
#define LLMIE_CALL(expr) llmieCheckStatus((expr), __FILE__, __LINE__)

// ---- deallocate(ptr, kind)   (memory_deleter.cuh) ----
template <typename T> void deallocate(T *ptr, const std::string &alloc_type) {
    if (!ptr) return;
    if (alloc_type == "new") delete ptr;
    else if (alloc_type == "new[]") delete[] ptr;
    else if (alloc_type == "cudaMalloc" || alloc_type == "hipMalloc") (void)hipFree(ptr);
    else if (alloc_type == "malloc") std::free(ptr);
    else std::cerr << "Unknown allocation type for deallocation." << std::endl;
}

namespace llmie_api {
// Stream every launch* adaptor uses.  Default = the null stream, as in the reference (its
// layers carry a `stream` member that is never initialised nor used: model_utils.h:45).
inline hipStream_t &launch_stream() {
    static thread_local hipStream_t s = nullptr;
    return s;
}
inline void set_launch_stream(hipStream_t s) { launch_stream() = s; }
template <typename T> inline llmie_dtype dtype_of();
template <> inline llmie_dtype dtype_of<float>() { return LLMIE_F32; }
template <> inline llmie_dtype dtype_of<half>() { return LLMIE_F16; }
}  // namespace llmie_api
