// TEST INFRASTRUCTURE (never shipped): the two hipCUB device algorithms the product calls, restated for the CPU stand-in for HIP
// (tests/hip_emu/hip/hip_runtime.h: device memory is host memory, streams are synchronous). With tests/hip_emu first on the include path
// `#include <hipcub/hipcub.hpp>` lands here, so the product's host code around sorts and scans compiles and runs as it lies.
// Same calling convention as hipCUB: a first call with d_temp_storage == nullptr only reports the scratch size.
#pragma once
#include <algorithm>
#include <cstddef>
#include <numeric>
#include <vector>
#include <hip/hip_runtime.h>

namespace hipcub {
struct DeviceRadixSort {
    // stable sort of (key, value) pairs by the key bits [begin_bit, end_bit)
    template <typename K, typename V>
    static hipError_t SortPairs(void* d_temp_storage, size_t& temp_storage_bytes, const K* keys_in, K* keys_out, const V* values_in, V* values_out, int n, int begin_bit = 0,
                                int end_bit = int(sizeof(K) * 8), hipStream_t = nullptr) {
        if (!d_temp_storage) { temp_storage_bytes = 16; return hipSuccess; }
        const K mask = end_bit - begin_bit >= int(sizeof(K) * 8) ? ~K(0) : K(((K(1) << (end_bit - begin_bit)) - 1) << begin_bit);
        std::vector<int> order(size_t(n > 0 ? n : 0));
        std::iota(order.begin(), order.end(), 0);
        std::stable_sort(order.begin(), order.end(), [&](int a, int b) { return (keys_in[a] & mask) < (keys_in[b] & mask); });
        for (int i = 0; i < n; ++i) { keys_out[i] = keys_in[order[size_t(i)]]; values_out[i] = values_in[order[size_t(i)]]; }
        return hipSuccess;
    }
};
struct Equality { template <typename A, typename B> bool operator()(const A& a, const B& b) const { return a == b; } };
struct DeviceScan {
    // inclusive scan with `op` restarting wherever the key changes
    template <typename K, typename In, typename Out, typename Op, typename N, typename Eq = Equality>
    static hipError_t InclusiveScanByKey(void* d_temp_storage, size_t& temp_storage_bytes, const K* keys, const In* in, Out* out, Op op, N n, Eq eq = Eq(), hipStream_t = nullptr) {
        if (!d_temp_storage) { temp_storage_bytes = 16; return hipSuccess; }
        for (size_t i = 0; i < size_t(n); ++i) {
            if (i == 0 || !eq(keys[i - 1], keys[i])) out[i] = Out(in[i]);
            else { const Out prev = out[i - 1]; out[i] = op(prev, Out(in[i])); }
        }
        return hipSuccess;
    }
    template <typename In, typename Out>
    static hipError_t ExclusiveSum(void* d_temp_storage, size_t& temp_storage_bytes, const In* in, Out* out, int n, hipStream_t = nullptr) {
        if (!d_temp_storage) { temp_storage_bytes = 16; return hipSuccess; }
        Out acc = Out(0);
        for (int i = 0; i < n; ++i) { const Out v = Out(in[i]); out[i] = acc; acc = Out(acc + v); }      // in == out is allowed, as in hipCUB
        return hipSuccess;
    }
};
}  // namespace hipcub
