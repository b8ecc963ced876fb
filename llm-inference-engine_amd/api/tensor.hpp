// C++ API mirror, part 2: Tensor / TensorWrapper<T> / TensorMap  (src/utils/tensor.h:18-295).
// Same names, members and TensorMap keys as the reference; ownership is fixed: wrappers and
// maps are NON-owning views (the reference's destructors free `data` and delete every mapped
// Tensor, which double-frees as soon as two maps share a tensor: self_decoder.cpp:101-102).
#pragma once
#include <algorithm>
#include <cstdint>
#include <functional>
#include <numeric>
#include <type_traits>
#include <unordered_map>

#include "runtime.hpp"

enum class Device { CPU_PINNED, CPU, GPU };
enum class DataType { FP32, FP16, INT8, INT32, BOOL, BYTES, UNSUPPORTED };

template <typename T> inline DataType getTensorType() {
    using U = typename std::remove_const<T>::type;
    if (std::is_same<U, float>::value) return DataType::FP32;
    if (std::is_same<U, half>::value) return DataType::FP16;
    if (std::is_same<U, int8_t>::value) return DataType::INT8;
    if (std::is_same<U, int>::value) return DataType::INT32;
    if (std::is_same<U, bool>::value) return DataType::BOOL;
    if (std::is_same<U, char>::value) return DataType::BYTES;
    return DataType::UNSUPPORTED;
}

template <typename T> class TensorWrapper;

class Tensor {
public:
    Device device = Device::GPU;
    DataType dtype = DataType::UNSUPPORTED;
    std::vector<int> shape;

    Tensor() = default;
    Tensor(const Device &device, const DataType &dtype, const std::vector<int> &shape)
        : device(device), dtype(dtype), shape(shape) {}
    virtual ~Tensor() = default;

    virtual int size() const {
        if (shape.empty()) return 0;
        return std::accumulate(shape.begin(), shape.end(), 1, std::multiplies<int>());
    }
    template <typename T> TensorWrapper<T> *wrap() { return static_cast<TensorWrapper<T> *>(this); }

    std::string deviceString() const {
        switch (device) {
            case Device::CPU: return "CPU";
            case Device::CPU_PINNED: return "CPU_PINNED";
            default: return "GPU";
        }
    }
    static const char *typeString(DataType t) {
        switch (t) {
            case DataType::FP32: return "FP32";
            case DataType::FP16: return "FP16";
            case DataType::INT8: return "INT8";
            case DataType::INT32: return "INT32";
            case DataType::BOOL: return "BOOL";
            case DataType::BYTES: return "BYTES";
            default: return "UNSUPPORTED";
        }
    }
    virtual std::string toString() const {
        return fmtstr("Tensor[device = %s, type = %s, shape = %s]", deviceString().c_str(), typeString(dtype),
                      vec2str(shape).c_str());
    }
};

template <typename T> class TensorWrapper : public Tensor {
public:
    T *data = nullptr;

    TensorWrapper(const Device &device, const DataType &dtype, const std::vector<int> &shape)
        : Tensor(device, dtype, shape) {}
    TensorWrapper(const Device &device, const DataType &dtype, const std::vector<int> &shape, T *const &data)
        : Tensor(device, dtype, shape), data(data) {
        LLM_CHECK_WITH_INFO(getTensorType<T>() == dtype, "Passed in data type should be same as dtype in params");
    }
    int size() const override { return (data == nullptr) ? 0 : Tensor::size(); }
    inline T getVal(const int &id) const {
        LLM_CHECK(device == Device::CPU);
        return data[id];
    }
    inline T getVal() const { return getVal(0); }
    inline T *getPtr() const { return data; }
    inline T *getPtrByOffset(const int &offset) const { return data + offset; }
    std::string toString() const override {
        return fmtstr("Tensor[device = %s, type = %s, shape = %s, data = %p]", deviceString().c_str(),
                      typeString(dtype), vec2str(shape).c_str(), static_cast<const void *>(data));
    }
};

class TensorMap {
public:
    std::unordered_map<std::string, Tensor *> tensor_map;

    TensorMap() = default;
    TensorMap(std::initializer_list<std::pair<std::string, Tensor *>> init) {
        for (const auto &kv : init) {
            LLM_CHECK_WITH_INFO(kv.second != nullptr && isValid(kv.second),
                                fmtstr("%s is not a valid tensor, skipping insert into TensorMap", kv.first.c_str()));
            insert(kv.first, kv.second);
        }
    }
    TensorMap(const std::unordered_map<std::string, Tensor *> &m) {
        for (const auto &kv : m)
            if (kv.second && isValid(kv.second)) insert(kv.first, kv.second);
    }
    virtual ~TensorMap() = default;  // non-owning

    inline size_t size() const { return tensor_map.size(); }
    inline bool isExist(const std::string &key) const { return tensor_map.find(key) != tensor_map.end(); }
    inline bool isValid(const Tensor *tensor) const { return tensor->size() > 0; }
    inline void insert(const std::string &key, Tensor *value) { tensor_map[key] = value; }
    inline void insert(const std::pair<std::string, Tensor *> &kv) { tensor_map[kv.first] = kv.second; }
    inline Tensor *at(const std::string &key) const {
        LLM_CHECK_WITH_INFO(isExist(key), fmtstr("Cannot find a tensor of name %s in the tensor map (keys: %s)",
                                                 key.c_str(), vec2str(keys()).c_str()));
        return tensor_map.at(key);
    }
    inline Tensor *operator[](const std::string &key) const { return at(key); }
    std::vector<std::string> keys() const {
        std::vector<std::string> names;
        for (const auto &kv : tensor_map) names.push_back(kv.first);
        return names;
    }
    std::string toString() const {
        std::ostringstream ss;
        ss << "{";
        size_t i = 0;
        for (const auto &kv : tensor_map) ss << (i++ ? ", " : "") << kv.first << ": " << kv.second->toString();
        ss << "}";
        return ss.str();
    }
};
