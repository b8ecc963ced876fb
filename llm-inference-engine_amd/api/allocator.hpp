// C++ API mirror, part 5: allocators (src/memory/allocator/base_allocator.h, cuda_allocator.h).
// Kept only so layer/model constructors keep their BaseAllocator* argument; the hot path does no
// allocation per token, so the reference's block pool is not rebuilt.
#pragma once
#include "runtime.hpp"

class BaseAllocator {
public:
    BaseAllocator() = default;
    virtual ~BaseAllocator() = default;
    template <typename T> void malloc(T **ptr, size_t size, bool is_host = false) {
        unifyMalloc(reinterpret_cast<void **>(ptr), size, is_host);
    }
    template <typename T> T *malloc(T *ptr, size_t size, bool is_host = false) {
        (void)ptr;
        void *p = nullptr;
        unifyMalloc(&p, size, is_host);
        return static_cast<T *>(p);
    }
    template <typename T> void free(T *ptr, bool is_host = false) {
        if (ptr) unifyFree(static_cast<void *>(ptr), is_host);
    }
    virtual void unifyMalloc(void **ptr, size_t size, bool is_host = false) = 0;
    virtual void unifyFree(void *ptr, bool is_host = false) = 0;
};

class CudaAllocator : public BaseAllocator {
public:
    void unifyMalloc(void **ptr, size_t size, bool is_host = false) override {
        if (is_host) {
            *ptr = std::malloc(size);
            LLM_CHECK_WITH_INFO(*ptr != nullptr, "host allocation failed");
        } else {
            CHECK(hipMalloc(ptr, size));
        }
    }
    void unifyFree(void *ptr, bool is_host = false) override {
        if (is_host) std::free(ptr);
        else CHECK(hipFree(ptr));
    }
};
