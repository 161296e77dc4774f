// TEST INFRASTRUCTURE (never shipped, never linked into libkajiya_amd.so): a CPU stand-in for the part of the HIP language and runtime that
// kajiya_amd/csrc uses. With this directory first on the include path `#include <hip/hip_runtime.h>` lands here and the product's kernel
// SOURCE compiles as host C++ (tests/hip_emu/build_emu.py: the whole library; tests/post_emu.cpp: post.hip alone, with g++):
//   * threaded mode (default): every workgroup is executed by blockDim host threads with a real barrier behind __syncthreads(),
//     `__shared__` is one static array (workgroups run one after the other) — slow, but AddressSanitizer / UBSan understand it;
//   * fiber mode (-DHIP_EMU_FIBERS): the lanes of a workgroup are ucontext fibers on one host thread, workgroups are spread over the cores,
//     `__shared__` is per host thread — ~30x faster, no sanitizers.
// Device memory is host memory, streams and events are no-ops, atomics are the host's. Wave intrinsics (__shfl_xor, __ballot, __lane_id)
// work for workgroups of ONE wave (<= 64 threads) whose unfinished lanes all reach the call, which is how the kernels here use them; a lane
// that has returned drops out of the exchange as it drops out of the exec mask. Dynamic LDS is handled by a source rewrite (see below).
// It checks the kernels' arithmetic, indexing, LDS staging, wave votes and host sequencing against the oracle where no GPU is available; it
// says nothing about hipcc's code generation or the hardware, and the -m gpu parity tests on an MI355X remain the statement about those.
// Not emulated: divergent wave operations, textures, streams that actually overlap.
#pragma once
#include <algorithm>
#include <barrier>
#include <cmath>
#include <cstdint>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <functional>
#include <memory>
#include <mutex>
#include <thread>
#include <vector>

#define __HIPCC__ 1
#define __host__
#define __device__
#define __global__
#define __forceinline__ inline
#define __launch_bounds__(...)
struct uint2 { unsigned x, y; };
struct uint4 { unsigned x, y, z, w; };
struct float4 { float x, y, z, w; };
struct float2 { float x, y; };
struct int2 { int x, y; };
struct int4 { int x, y, z, w; };
inline int4 make_int4(int x, int y, int z, int w) { return int4{x, y, z, w}; }
inline int2 make_int2(int x, int y) { return int2{x, y}; }
inline float4 make_float4(float x, float y, float z, float w) { return float4{x, y, z, w}; }
inline float2 make_float2(float x, float y) { return float2{x, y}; }
inline uint2 make_uint2(unsigned x, unsigned y) { return uint2{x, y}; }
inline uint4 make_uint4(unsigned x, unsigned y, unsigned z, unsigned w) { return uint4{x, y, z, w}; }
struct dim3 {
    unsigned x, y, z;
    dim3(unsigned x_ = 1, unsigned y_ = 1, unsigned z_ = 1) : x(x_), y(y_), z(z_) {}
};
inline thread_local dim3 threadIdx, blockIdx;
inline thread_local dim3 gridDim, blockDim;     // per host thread: launches may come from several host threads at once (every worker of a launch sets its own copy)

inline int min(int a, int b) { return a < b ? a : b; }
inline int max(int a, int b) { return a > b ? a : b; }
inline unsigned min(unsigned a, unsigned b) { return a < b ? a : b; }
inline unsigned max(unsigned a, unsigned b) { return a > b ? a : b; }

#ifndef HIP_EMU_FIBERS
#define __shared__ static
namespace hip_emu {
// ---- threaded mode (default; what the sanitizer builds use): one host thread per lane.
// g_barrier: the barrier behind __syncthreads() and the wave intrinsics of the workgroup that is running. A lane that has returned from
// the kernel drops out of it (as an exited lane drops out of the exec mask), so the remaining lanes can still vote / exchange.
inline std::barrier<>* g_barrier = nullptr;
inline bool g_lane_active[1024];
inline unsigned long long g_exchange[64];
inline void barrier_wait() { g_barrier->arrive_and_wait(); }
inline void launch(dim3 grid, dim3 block, const std::function<void()>& body) {
    const unsigned n = block.x * block.y * block.z;
    const size_t blocks = size_t(grid.x) * grid.y * grid.z;
    if (blocks == 0 || n == 0) return;
    std::barrier<> block_end(n);
    std::unique_ptr<std::barrier<>> current;
    static std::mutex one_launch_at_a_time;      // this mode keeps the running workgroup's barrier and "LDS" in process-wide statics
    std::lock_guard<std::mutex> lock(one_launch_at_a_time);
    gridDim = grid;
    blockDim = block;
    for (unsigned t = 0; t < n; ++t) g_lane_active[t] = true;
    current.reset(new std::barrier<>(n));
    g_barrier = current.get();
    std::vector<std::thread> threads;
    for (unsigned t = 0; t < n; ++t)
        threads.emplace_back([&, t]() {
            gridDim = grid; blockDim = block;
            threadIdx = dim3(t % block.x, (t / block.x) % block.y, t / (block.x * block.y));
            for (unsigned bz = 0; bz < grid.z; ++bz)
                for (unsigned by = 0; by < grid.y; ++by)
                    for (unsigned bx = 0; bx < grid.x; ++bx) {
                        blockIdx = dim3(bx, by, bz);
                        body();
                        g_lane_active[t] = false;
                        g_barrier->arrive_and_drop();
                        block_end.arrive_and_wait();          // everyone is out of the kernel: the static "LDS" may be reused
                        if (t == 0) {                          // one lane re-arms the workgroup barrier for the next workgroup
                            for (unsigned i = 0; i < n; ++i) g_lane_active[i] = true;
                            current.reset(new std::barrier<>(n));
                            g_barrier = current.get();
                        }
                        block_end.arrive_and_wait();
                    }
        });
    for (auto& th : threads) th.join();
    g_barrier = nullptr;
}
}  // namespace hip_emu
#else
#define __shared__ static thread_local
#include <ucontext.h>
#include <atomic>
namespace hip_emu {
// ---- fiber mode (-DHIP_EMU_FIBERS; fast, not for sanitizer builds): HIP_EMU_WORKERS host threads take workgroups off a shared counter; each
// runs the lanes of its workgroup as ucontext fibers on that one thread. A lane runs until it finishes or reaches a barrier (__syncthreads or a
// wave exchange); when every unfinished lane of the workgroup waits, all are released. `__shared__` is per host thread, so workgroups really
// do run concurrently, and the atomics between them are real.
enum { READY = 0, WAITING = 1, DONE = 2 };
struct Fiber { ucontext_t ctx; char* stack = nullptr; int state = DONE; dim3 tid; };
struct Worker {
    ucontext_t scheduler;
    std::vector<Fiber> fibers;
    Fiber* running = nullptr;
    const std::function<void()>* body = nullptr;
    ~Worker() { for (auto& f : fibers) free(f.stack); }
};
inline thread_local Worker* g_worker = nullptr;
inline thread_local bool g_lane_active[1024];
inline thread_local unsigned long long g_exchange[64];
inline void barrier_wait() {
    Fiber* f = g_worker->running;
    f->state = WAITING;
    swapcontext(&f->ctx, &g_worker->scheduler);
}
inline void fiber_entry() {
    Worker* w = g_worker;
    (*w->body)();
    w->running->state = DONE;
    swapcontext(&w->running->ctx, &w->scheduler);
}
inline void run_block(Worker& w, unsigned n, dim3 block) {
    static const size_t STACK = 512 * 1024;
    if (w.fibers.size() < n) { w.fibers.resize(n); for (auto& f : w.fibers) if (!f.stack) f.stack = (char*)malloc(STACK); }
    for (unsigned t = 0; t < n; ++t) {
        Fiber& f = w.fibers[t];
        getcontext(&f.ctx);
        f.ctx.uc_stack.ss_sp = f.stack;
        f.ctx.uc_stack.ss_size = STACK;
        f.ctx.uc_link = &w.scheduler;
        makecontext(&f.ctx, (void (*)())fiber_entry, 0);
        f.state = READY;
        f.tid = dim3(t % block.x, (t / block.x) % block.y, t / (block.x * block.y));
        g_lane_active[t] = true;
    }
    unsigned alive = n;
    while (alive) {
        for (unsigned t = 0; t < n; ++t) {
            Fiber& f = w.fibers[t];
            if (f.state != READY) continue;
            w.running = &f;
            threadIdx = f.tid;
            swapcontext(&w.scheduler, &f.ctx);
            if (f.state == DONE) { --alive; g_lane_active[t] = false; }
        }
        for (unsigned t = 0; t < n; ++t) if (w.fibers[t].state == WAITING) w.fibers[t].state = READY;   // every unfinished lane has arrived
    }
}
inline void launch(dim3 grid, dim3 block, const std::function<void()>& body) {
    const unsigned n = block.x * block.y * block.z;
    const size_t blocks = size_t(grid.x) * grid.y * grid.z;
    if (blocks == 0 || n == 0) return;
    gridDim = grid;
    blockDim = block;
    static const unsigned workers = [] { const char* e = getenv("HIP_EMU_WORKERS"); unsigned v = e ? unsigned(atoi(e)) : std::thread::hardware_concurrency(); return v ? v : 1u; }();
    std::atomic<size_t> next{0};
    auto work = [&]() {
        static thread_local Worker w;
        gridDim = grid; blockDim = block;
        g_worker = &w;
        w.body = &body;
        for (;;) {
            const size_t b = next.fetch_add(1);
            if (b >= blocks) break;
            blockIdx = dim3(unsigned(b % grid.x), unsigned((b / grid.x) % grid.y), unsigned(b / (size_t(grid.x) * grid.y)));
            run_block(w, n, block);
        }
    };
    const unsigned nt = unsigned(std::min<size_t>(workers, blocks));
    std::vector<std::thread> threads;
    for (unsigned i = 1; i < nt; ++i) threads.emplace_back(work);
    work();
    for (auto& th : threads) th.join();
}
}  // namespace hip_emu
#endif

// dynamic LDS (`extern __shared__ T name[];`): tests/hip_emu/build_emu.py rewrites that one declaration form to
// `T* name = (T*)hip_emu::dynamic_lds();` in a scratch copy of the source; everything else is compiled as it lies.
#ifndef HIP_EMU_FIBERS
namespace hip_emu { inline unsigned long long g_dynamic_lds[64 * 1024 / 8]; inline void* dynamic_lds() { return g_dynamic_lds; } }
#else
namespace hip_emu { inline thread_local unsigned long long g_dynamic_lds[64 * 1024 / 8]; inline void* dynamic_lds() { return g_dynamic_lds; } }
#endif
inline void __syncthreads() { hip_emu::barrier_wait(); }
#define hipLaunchKernelGGL(kernel, grid, block, shmem, stream, ...) hip_emu::launch(grid, block, [=]() { (kernel)(__VA_ARGS__); })

// ---- wave intrinsics for one-wave workgroups with all lanes converged at the call
namespace hip_emu {
inline unsigned lane() { return threadIdx.x + threadIdx.y * blockDim.x + threadIdx.z * blockDim.x * blockDim.y; }
inline void require_one_wave() { if (blockDim.x * blockDim.y * blockDim.z > 64) { fprintf(stderr, "hip_emu: wave intrinsic in a workgroup of more than 64 threads\n"); abort(); } }
}  // namespace hip_emu
inline unsigned __lane_id() { return hip_emu::lane(); }
template <typename T> inline T __shfl_xor(T v, int lane_mask) {
    static_assert(sizeof(T) <= 8, "hip_emu: __shfl_xor of a type wider than 8 bytes");
    hip_emu::require_one_wave();
    const unsigned l = hip_emu::lane(), n = blockDim.x * blockDim.y * blockDim.z;
    memcpy(&hip_emu::g_exchange[l], &v, sizeof(T));
    hip_emu::barrier_wait();
    unsigned src = (l ^ unsigned(lane_mask)) < n ? (l ^ unsigned(lane_mask)) : l;
    if (!hip_emu::g_lane_active[src]) src = l;         // reading an exited lane is undefined on hardware; keep it harmless here
    T r;
    memcpy(&r, &hip_emu::g_exchange[src], sizeof(T));
    hip_emu::barrier_wait();
    return r;
}
template <typename T> inline T __shfl(T v, int src_lane) {
    static_assert(sizeof(T) <= 8, "hip_emu: __shfl of a type wider than 8 bytes");
    hip_emu::require_one_wave();
    const unsigned l = hip_emu::lane(), n = blockDim.x * blockDim.y * blockDim.z;
    memcpy(&hip_emu::g_exchange[l], &v, sizeof(T));
    hip_emu::barrier_wait();
    unsigned src = unsigned(src_lane) < n ? unsigned(src_lane) : l;
    if (!hip_emu::g_lane_active[src]) src = l;
    T r;
    memcpy(&r, &hip_emu::g_exchange[src], sizeof(T));
    hip_emu::barrier_wait();
    return r;
}
inline unsigned long long __ballot(int predicate) {
    hip_emu::require_one_wave();
    const unsigned l = hip_emu::lane(), n = blockDim.x * blockDim.y * blockDim.z;
    hip_emu::g_exchange[l] = predicate ? 1ull : 0ull;
    hip_emu::barrier_wait();
    unsigned long long m = 0;
    for (unsigned i = 0; i < n; ++i) if (hip_emu::g_lane_active[i]) m |= hip_emu::g_exchange[i] << i;
    hip_emu::barrier_wait();
    return m;
}
inline int __popcll(unsigned long long v) { return __builtin_popcountll(v); }
inline int __popc(unsigned v) { return __builtin_popcount(v); }
inline int __ffsll(long long v) { return __builtin_ffsll(v); }
inline int __ffs(int v) { return __builtin_ffs(v); }
inline int __clz(int v) { return v ? __builtin_clz(unsigned(v)) : 32; }
inline int __clzll(long long v) { return v ? __builtin_clzll((unsigned long long)v) : 64; }

inline void __threadfence() { __atomic_thread_fence(__ATOMIC_SEQ_CST); }
inline void __threadfence_block() { __atomic_thread_fence(__ATOMIC_SEQ_CST); }
inline unsigned atomicAdd(unsigned* p, unsigned v) { return __atomic_fetch_add(p, v, __ATOMIC_RELAXED); }
inline unsigned long long atomicAdd(unsigned long long* p, unsigned long long v) { return __atomic_fetch_add(p, v, __ATOMIC_RELAXED); }
inline int atomicAdd(int* p, int v) { return __atomic_fetch_add(p, v, __ATOMIC_RELAXED); }
inline float atomicAdd(float* p, float v) { float o, n; do { o = *p; n = o + v; } while (!__atomic_compare_exchange(p, &o, &n, false, __ATOMIC_RELAXED, __ATOMIC_RELAXED)); return o; }
inline unsigned atomicOr(unsigned* p, unsigned v) { return __atomic_fetch_or(p, v, __ATOMIC_RELAXED); }
inline unsigned atomicAnd(unsigned* p, unsigned v) { return __atomic_fetch_and(p, v, __ATOMIC_RELAXED); }
inline unsigned atomicMax(unsigned* p, unsigned v) { unsigned o = *p; while (o < v && !__atomic_compare_exchange_n(p, &o, v, false, __ATOMIC_RELAXED, __ATOMIC_RELAXED)) {} return o; }
inline unsigned atomicMin(unsigned* p, unsigned v) { unsigned o = *p; while (o > v && !__atomic_compare_exchange_n(p, &o, v, false, __ATOMIC_RELAXED, __ATOMIC_RELAXED)) {} return o; }
inline unsigned long long atomicMin(unsigned long long* p, unsigned long long v) { unsigned long long o = *p; while (o > v && !__atomic_compare_exchange_n(p, &o, v, false, __ATOMIC_RELAXED, __ATOMIC_RELAXED)) {} return o; }
inline unsigned atomicExch(unsigned* p, unsigned v) { return __atomic_exchange_n(p, v, __ATOMIC_RELAXED); }
inline unsigned atomicCAS(unsigned* p, unsigned cmp, unsigned v) { __atomic_compare_exchange_n(p, &cmp, v, false, __ATOMIC_RELAXED, __ATOMIC_RELAXED); return cmp; }
inline float __fmul_rn(float a, float b) { return a * b; }     // the emulation builds with -ffp-contract=off: every product rounds once anyway
inline float __fadd_rn(float a, float b) { return a + b; }
inline float __uint_as_float(unsigned u) { float f; memcpy(&f, &u, 4); return f; }
inline unsigned __float_as_uint(float f) { unsigned u; memcpy(&u, &f, 4); return u; }

// ---- runtime: device memory is host memory, streams are synchronous. `static`: these must never resolve to libamdhip64's symbols
// when the real library happens to be loaded in the same process.
typedef int hipError_t;
typedef void* hipStream_t;
enum { hipSuccess = 0, hipErrorInvalidValue = 1, hipErrorOutOfMemory = 2, hipErrorUnknown = 999 };
enum hipMemcpyKind { hipMemcpyHostToDevice = 1, hipMemcpyDeviceToHost = 2, hipMemcpyDeviceToDevice = 3 };
enum { hipHostMallocDefault = 0 };
static inline hipError_t hipMalloc(void** p, size_t n) { *p = malloc(n); return *p ? hipSuccess : hipErrorOutOfMemory; }
static inline hipError_t hipFree(void* p) { free(p); return hipSuccess; }
static inline hipError_t hipHostMalloc(void** p, size_t n, unsigned) { return hipMalloc(p, n); }
static inline hipError_t hipHostFree(void* p) { return hipFree(p); }
static inline hipError_t hipMemsetAsync(void* p, int v, size_t n, hipStream_t = nullptr) { memset(p, v, n); return hipSuccess; }
static inline hipError_t hipMemcpyAsync(void* d, const void* s, size_t n, hipMemcpyKind, hipStream_t = nullptr) { memcpy(d, s, n); return hipSuccess; }
static inline hipError_t hipMemcpy(void* d, const void* s, size_t n, hipMemcpyKind) { memcpy(d, s, n); return hipSuccess; }
static inline hipError_t hipMemset(void* p, int v, size_t n) { memset(p, v, n); return hipSuccess; }
static inline hipError_t hipStreamSynchronize(hipStream_t) { return hipSuccess; }
static inline hipError_t hipStreamCreateWithFlags(hipStream_t* s, unsigned) { *s = nullptr; return hipSuccess; }
static inline hipError_t hipStreamCreate(hipStream_t* s) { *s = nullptr; return hipSuccess; }
static inline hipError_t hipStreamDestroy(hipStream_t) { return hipSuccess; }
enum { hipStreamNonBlocking = 1, hipEventDisableTiming = 2, hipEventDefault = 0 };
typedef struct hip_emu_event* hipEvent_t;
static inline hipError_t hipEventCreate(hipEvent_t* e) { *e = nullptr; return hipSuccess; }
static inline hipError_t hipEventCreateWithFlags(hipEvent_t* e, unsigned) { *e = nullptr; return hipSuccess; }
static inline hipError_t hipEventDestroy(hipEvent_t) { return hipSuccess; }
static inline hipError_t hipEventRecord(hipEvent_t, hipStream_t = nullptr) { return hipSuccess; }
static inline hipError_t hipEventSynchronize(hipEvent_t) { return hipSuccess; }
static inline hipError_t hipEventElapsedTime(float* ms, hipEvent_t, hipEvent_t) { *ms = 0.0f; return hipSuccess; }
static inline hipError_t hipStreamWaitEvent(hipStream_t, hipEvent_t, unsigned = 0) { return hipSuccess; }
struct hipDeviceProp_t { int multiProcessorCount = 256; char name[64] = "hip_emu"; };
static inline hipError_t hipGetDeviceCount(int* n) { *n = 1; return hipSuccess; }
static inline hipError_t hipSetDevice(int) { return hipSuccess; }
static inline hipError_t hipGetDevice(int* d) { *d = 0; return hipSuccess; }
static inline hipError_t hipGetDeviceProperties(hipDeviceProp_t* p, int) { *p = hipDeviceProp_t(); return hipSuccess; }
static inline hipError_t hipGetLastError() { return hipSuccess; }
static inline hipError_t hipDeviceSynchronize() { return hipSuccess; }
static inline const char* hipGetErrorString(hipError_t) { return "hip_emu"; }
