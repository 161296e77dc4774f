// TEST INFRASTRUCTURE (never shipped, never linked into libkajiya_amd.so): a CPU stand-in for the small slice of the HIP language and
// runtime that kajiya_amd/csrc/post.hip uses. tests/post_emu.cpp includes the product's kernel SOURCE with this directory first on the
// include path, so `#include <hip/hip_runtime.h>` lands here and the kernels run on the build machine: every workgroup is executed by
// blockDim host threads with a real barrier behind __syncthreads(), `__shared__` becomes one static array (workgroups run one after the
// other), device memory is host memory. It checks the kernels' arithmetic, indexing and host sequencing against the oracle where no GPU
// is available; it says nothing about code generation, and the -m gpu parity tests remain the statement about the real thing.
// Not emulated (post.hip does not use them): wave intrinsics, textures, dynamic shared memory, streams that overlap.
#pragma once
#include <algorithm>
#include <barrier>
#include <cmath>
#include <cstdint>
#include <cstdlib>
#include <cstring>
#include <functional>
#include <thread>
#include <vector>

#define __HIPCC__ 1
#define __host__
#define __device__
#define __global__
#define __forceinline__ inline
#define __launch_bounds__(...)
#define __shared__ static

struct uint2 { unsigned x, y; };
struct uint4 { unsigned x, y, z, w; };
struct float4 { float x, y, z, w; };
inline uint2 make_uint2(unsigned x, unsigned y) { return uint2{x, y}; }
inline uint4 make_uint4(unsigned x, unsigned y, unsigned z, unsigned w) { return uint4{x, y, z, w}; }
struct dim3 {
    unsigned x, y, z;
    dim3(unsigned x_ = 1, unsigned y_ = 1, unsigned z_ = 1) : x(x_), y(y_), z(z_) {}
};
inline thread_local dim3 threadIdx, blockIdx;
inline dim3 gridDim, blockDim;

inline int min(int a, int b) { return a < b ? a : b; }
inline int max(int a, int b) { return a > b ? a : b; }
inline unsigned min(unsigned a, unsigned b) { return a < b ? a : b; }
inline unsigned max(unsigned a, unsigned b) { return a > b ? a : b; }

namespace hip_emu {
inline std::barrier<>* g_barrier = nullptr;
inline void launch(dim3 grid, dim3 block, const std::function<void()>& body) {
    const unsigned n = block.x * block.y * block.z;
    std::barrier<> bar(n);
    g_barrier = &bar;
    gridDim = grid;
    blockDim = block;
    std::vector<std::thread> threads;
    for (unsigned t = 0; t < n; ++t)
        threads.emplace_back([&, t]() {
            threadIdx = dim3(t % block.x, (t / block.x) % block.y, t / (block.x * block.y));
            for (unsigned bz = 0; bz < grid.z; ++bz)
                for (unsigned by = 0; by < grid.y; ++by)
                    for (unsigned bx = 0; bx < grid.x; ++bx) {
                        blockIdx = dim3(bx, by, bz);
                        body();
                        bar.arrive_and_wait();      // the next workgroup reuses the static "LDS"
                    }
        });
    for (auto& th : threads) th.join();
    g_barrier = nullptr;
}
}  // namespace hip_emu
inline void __syncthreads() { hip_emu::g_barrier->arrive_and_wait(); }
#define hipLaunchKernelGGL(kernel, grid, block, shmem, stream, ...) hip_emu::launch(grid, block, [=]() { kernel(__VA_ARGS__); })

inline unsigned atomicAdd(unsigned* p, unsigned v) { return __atomic_fetch_add(p, v, __ATOMIC_RELAXED); }
inline float __uint_as_float(unsigned u) { float f; memcpy(&f, &u, 4); return f; }
inline unsigned __float_as_uint(float f) { unsigned u; memcpy(&u, &f, 4); return u; }

// ---- runtime: device memory is host memory, streams are synchronous. `static`: these must never resolve to libamdhip64's symbols
// when the real library happens to be loaded in the same process.
typedef int hipError_t;
typedef void* hipStream_t;
enum { hipSuccess = 0, hipErrorOutOfMemory = 2 };
enum hipMemcpyKind { hipMemcpyHostToDevice = 1, hipMemcpyDeviceToHost = 2, hipMemcpyDeviceToDevice = 3 };
enum { hipHostMallocDefault = 0 };
static inline hipError_t hipMalloc(void** p, size_t n) { *p = malloc(n); return *p ? hipSuccess : hipErrorOutOfMemory; }
static inline hipError_t hipFree(void* p) { free(p); return hipSuccess; }
static inline hipError_t hipHostMalloc(void** p, size_t n, unsigned) { return hipMalloc(p, n); }
static inline hipError_t hipHostFree(void* p) { return hipFree(p); }
static inline hipError_t hipMemsetAsync(void* p, int v, size_t n, hipStream_t = nullptr) { memset(p, v, n); return hipSuccess; }
static inline hipError_t hipMemcpyAsync(void* d, const void* s, size_t n, hipMemcpyKind, hipStream_t = nullptr) { memcpy(d, s, n); return hipSuccess; }
static inline hipError_t hipGetLastError() { return hipSuccess; }
static inline hipError_t hipDeviceSynchronize() { return hipSuccess; }
static inline const char* hipGetErrorString(hipError_t) { return "hip_emu"; }
