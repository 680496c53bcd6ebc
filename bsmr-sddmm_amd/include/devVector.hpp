#pragma once
// RAII device buffer with the interface subset of the reference's dev::vector
// (include/devVector.cuh:9-52, 54-126) and its h2d / d2h helpers (:170-229).
// Storage comes from the C ABI (bsmr_dev_alloc / bsmr_dev_free), so host code
// never includes HIP headers.

#include <cstddef>
#include <cstdio>
#include <vector>

#include "bsmr_hip.h"

namespace dev {

template <typename T>
class vector {
public:
    vector() = default;
    explicit vector(size_t n, int device = 0) : device_(device) { allocate(n); }
    vector(size_t n, int fillByte, int device) : device_(device) {
        allocate(n);
        if (data_) bsmr_dev_memset(data_, fillByte, n * sizeof(T));
    }
    explicit vector(const std::vector<T>& src, int device = 0) : device_(device) {
        allocate(src.size());
        if (data_ && !src.empty()) bsmr_memcpy_h2d(data_, src.data(), src.size() * sizeof(T));
    }
    ~vector() { clear(); }
    vector(const vector&) = delete;
    vector& operator=(const vector&) = delete;
    vector(vector&& o) noexcept : size_(o.size_), data_(o.data_), device_(o.device_) {
        o.size_ = 0;
        o.data_ = nullptr;
    }
    vector& operator=(vector&& o) noexcept {
        if (this != &o) {
            clear();
            size_ = o.size_;
            data_ = o.data_;
            device_ = o.device_;
            o.size_ = 0;
            o.data_ = nullptr;
        }
        return *this;
    }

    // Contents are not preserved (same as the reference's resize).
    void resize(size_t n) {
        clear();
        allocate(n);
    }
    void clear() {
        if (data_) bsmr_dev_free(data_);
        data_ = nullptr;
        size_ = 0;
    }
    size_t size() const { return size_; }
    const T* data() const { return data_; }
    T* data() { return data_; }
    bool ok() const { return size_ == 0 || data_ != nullptr; }

private:
    void allocate(size_t n) {
        size_ = n;
        data_ = nullptr;
        if (n == 0) return;
        void* p = nullptr;
        const int st = bsmr_dev_alloc(device_, n * sizeof(T), &p);
        if (st != BSMR_OK) {
            fprintf(stderr, "dev::vector: allocation of %zu bytes failed: %s\n", n * sizeof(T),
                    bsmr_strerror(st));
            size_ = 0;
            return;
        }
        data_ = static_cast<T*>(p);
    }
    size_t size_ = 0;
    T* data_ = nullptr;
    int device_ = 0;
};

}  // namespace dev

template <typename T>
inline void h2d(dev::vector<T>& dst, const std::vector<T>& src) {
    dst.resize(src.size());
    if (dst.data() && !src.empty()) bsmr_memcpy_h2d(dst.data(), src.data(), src.size() * sizeof(T));
}

template <typename T>
inline void d2h(std::vector<T>& dst, const dev::vector<T>& src) {
    dst.resize(src.size());
    if (src.data() && src.size()) bsmr_memcpy_d2h(dst.data(), src.data(), src.size() * sizeof(T));
}

template <typename T>
inline std::vector<T> d2h(const dev::vector<T>& src) {
    std::vector<T> out;
    d2h(out, src);
    return out;
}
