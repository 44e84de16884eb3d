// TEST INFRASTRUCTURE -- not part of the product, never linked into libmpcx.so.
//
// A lock-step interpreter for HIP workgroup kernels on the host, so that the device code of include/mpcx/*.hpp can be stepped through
// and checked in a container without a GPU (tests/emu/run_nlmpc.cpp, tests/test_emu_nlmpc.py).  This header stands in for
// <hip/hip_runtime.h> when a test program is compiled with g++ -Itests/emu: the kernel sources are compiled unchanged, every thread of a
// workgroup becomes a fibre, and everything that exchanges data between the lanes of a wavefront (DPP, v_readlane, ds_bpermute shuffles,
// ballots) or synchronises (s_barrier, wave barriers) is a rendezvous of the fibres involved.  Between two rendezvous the fibres of a
// workgroup run one after another -- in ascending or (HIPEMU_ORDER=reverse) descending thread order, so that an exchange through LDS
// that lacks its barrier shows up as a different result in at least one of the two orders.  A rendezvous that not every lane of its scope
// reaches (a wave-level operation inside divergent control flow) is reported as a deadlock with the position of every fibre.
//
// It says nothing about performance, register allocation or code generation; results agree with the GPU's up to the compiler's choice
// of fused multiply-adds.  Nothing in libmpc_amd/, pympcxx/, bench.py or __graft_entry__.py may include or load it.
#pragma once

#include <algorithm>
#include <cmath>
#include <cstdint>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <functional>
#include <vector>
#include <execinfo.h>

#define __device__
#define __host__
#define __global__
#define __shared__
#define __forceinline__ inline
#define __noinline__ __attribute__((noinline))
#define __launch_bounds__(...)

struct dim3 {
    unsigned x, y, z;
    constexpr dim3(unsigned a = 1, unsigned b = 1, unsigned c = 1) : x(a), y(b), z(c) {}
};
typedef struct ihipStream_t *hipStream_t;
typedef int hipError_t;
constexpr hipError_t hipSuccess = 0;
inline hipError_t hipGetLastError() { return hipSuccess; }
enum hipFuncAttribute { hipFuncAttributeMaxDynamicSharedMemorySize = 8 };
inline hipError_t hipFuncSetAttribute(const void *, hipFuncAttribute, int) { return hipSuccess; }
template <class K> inline hipError_t hipOccupancyMaxActiveBlocksPerMultiprocessor(int *n, K, int, size_t) { *n = 0; return hipSuccess; }

extern "C" void hipemu_switch(void **save_sp, void *new_sp);

namespace hipemu {

constexpr int kMaxThreads = 1024, kWave = 64;

struct Bar { int count = 0; unsigned gen = 0; };

struct State {
    int nthreads = 0, cur = 0, block = 0, nblocks = 0, alive = 0;
    bool reverse = false;
    void *sp[kMaxThreads] = {};
    void *main_sp = nullptr;
    char *stacks = nullptr;
    size_t stack_bytes = 2048 * 1024;
    bool done[kMaxThreads] = {};
    const char *where[kMaxThreads] = {};
    void *trace[kMaxThreads][8] = {};
    int trace_n[kMaxThreads] = {};
    bool tracing = false;
    long idle_switches = 0;
    Bar block_bar, wave_bar[kMaxThreads / kWave];
    uint64_t xbuf[kMaxThreads / kWave][2][kWave];
    std::function<void()> body;
    const void *kernarg = nullptr;          // the first kernel argument of the launch in progress
    size_t kernarg_bytes = 0;
    long n_block_syncs = 0, n_wave_syncs = 0;
};
inline State &st() { static State s; return s; }

inline void report_deadlock()
{
    State &s = st();
    fprintf(stderr, "hipemu: deadlock in block %d -- a rendezvous was not reached by every thread of its scope\n", s.block);
    for (int t = 0; t < s.nthreads; ++t)
        if (!s.done[t]) {
            fprintf(stderr, "  thread %3d waits at %s", t, s.where[t] ? s.where[t] : "?");
            for (int i = 2; i < s.trace_n[t]; ++i) fprintf(stderr, " %p", s.trace[t][i]);      // HIPEMU_TRACE=1: return addresses (addr2line -e <binary>)
            fprintf(stderr, "\n");
        }
        else fprintf(stderr, "  thread %3d finished\n", t);
    abort();
}

inline void yield()
{
    State &s = st();
    if (++s.idle_switches > 64L * s.nthreads + 1024) report_deadlock();
    const int from = s.cur;
    int nxt = from;
    for (int step = 0; step < s.nthreads; ++step) {
        nxt = s.reverse ? (nxt + s.nthreads - 1) % s.nthreads : (nxt + 1) % s.nthreads;
        if (!s.done[nxt]) break;
    }
    if (nxt == from) return;
    s.cur = nxt;
    hipemu_switch(&s.sp[from], s.sp[nxt]);
}

inline void rendezvous(Bar &b, int size, const char *what)
{
    State &s = st();
    s.where[s.cur] = what;
    if (s.tracing) s.trace_n[s.cur] = backtrace(s.trace[s.cur], 8);
    const unsigned gen = b.gen;
    if (++b.count == size) { b.count = 0; ++b.gen; s.idle_switches = 0; }
    else while (b.gen == gen) yield();
    s.where[s.cur] = "running";
}

inline int tid() { return st().cur; }
inline int wave_of(int t) { return t / kWave; }
inline int wave_size(int w) { const int n = st().nthreads - w * kWave; return n < kWave ? n : kWave; }

inline void sync_block() { ++st().n_block_syncs; rendezvous(st().block_bar, st().nthreads, "__syncthreads"); }
inline void sync_wave(const char *what = "wave barrier")
{
    const int w = wave_of(tid());
    ++st().n_wave_syncs;
    rendezvous(st().wave_bar[w], wave_size(w), what);
}

// every lane of the wavefront contributes v and receives the contribution of lane src (of the same wavefront)
inline uint64_t exchange(uint64_t v, int src, const char *what)
{
    State &s = st();
    const int t = tid(), w = wave_of(t), lane = t % kWave;
    const unsigned par = s.wave_bar[w].gen & 1u;
    s.xbuf[w][par][lane] = v;
    sync_wave(what);
    return s.xbuf[w][par][((src % kWave) + kWave) % kWave];
}
inline uint64_t ballot(bool p)
{
    State &s = st();
    const int t = tid(), w = wave_of(t), lane = t % kWave;
    const unsigned par = s.wave_bar[w].gen & 1u;
    s.xbuf[w][par][lane] = p ? 1u : 0u;
    sync_wave("ballot");
    uint64_t m = 0;
    for (int l = 0; l < wave_size(w); ++l) m |= (uint64_t)(s.xbuf[w][par][l] & 1u) << l;
    return m;
}

extern "C" inline void hipemu_entry()
{
    State &s = st();
    s.where[s.cur] = "running";
    s.body();
    s.done[s.cur] = true;
    s.where[s.cur] = "finished";
    --s.alive;
    s.idle_switches = 0;
    // hand over to any fibre that is still alive, or back to the launcher
    const int from = s.cur;
    if (s.alive > 0) {
        int nxt = from;
        for (int step = 0; step < s.nthreads; ++step) {
            nxt = s.reverse ? (nxt + s.nthreads - 1) % s.nthreads : (nxt + 1) % s.nthreads;
            if (!s.done[nxt]) break;
        }
        s.cur = nxt;
        hipemu_switch(&s.sp[from], s.sp[nxt]);
    } else {
        hipemu_switch(&s.sp[from], s.main_sp);
    }
    abort();        // a finished fibre is never resumed
}

inline void run_block(int block, int nblocks, int nthreads, const std::function<void()> &body)
{
    State &s = st();
    if (nthreads > kMaxThreads) { fprintf(stderr, "hipemu: %d threads per block\n", nthreads); abort(); }
    if (const char *e = getenv("HIPEMU_ORDER")) s.reverse = !strcmp(e, "reverse");
    s.tracing = getenv("HIPEMU_TRACE") != nullptr;
    s.nthreads = nthreads; s.block = block; s.nblocks = nblocks; s.alive = nthreads; s.body = body;
    s.block_bar = Bar{};
    for (auto &b : s.wave_bar) b = Bar{};
    if (!s.stacks) s.stacks = static_cast<char *>(aligned_alloc(64, s.stack_bytes * kMaxThreads));
    for (int t = 0; t < nthreads; ++t) {
        s.done[t] = false; s.where[t] = "not started";
        char *top = s.stacks + s.stack_bytes * (t + 1);
        void **sp = reinterpret_cast<void **>(top) - 8;      // six callee-saved registers, the entry point, a null return address
        for (int i = 0; i < 6; ++i) sp[i] = nullptr;
        sp[6] = reinterpret_cast<void *>(&hipemu_entry);
        sp[7] = nullptr;
        s.sp[t] = sp;
    }
    s.idle_switches = 0;
    s.cur = s.reverse ? nthreads - 1 : 0;
    hipemu_switch(&s.main_sp, s.sp[s.cur]);
}

struct Idx { unsigned x, y, z; };
inline const void *first_arg_ptr() { return nullptr; }
template <class A0, class... R> inline const void *first_arg_ptr(const A0 &a, const R &...) { return &a; }
inline size_t first_arg_bytes() { return 0; }
template <class A0, class... R> inline size_t first_arg_bytes(const A0 &, const R &...) { return sizeof(A0); }

}  // namespace hipemu

#define threadIdx (hipemu::Idx{(unsigned)hipemu::tid(), 0u, 0u})
#define blockIdx (hipemu::Idx{(unsigned)hipemu::st().block, 0u, 0u})
#define blockDim (hipemu::Idx{(unsigned)hipemu::st().nthreads, 1u, 1u})
#define gridDim (hipemu::Idx{(unsigned)hipemu::st().nblocks, 1u, 1u})

// dynamic shared memory: the kernels declare `extern __shared__ double name[]` at block scope, i.e. a namespace-scope array of the
// enclosing namespace; the test program defines it (HIPEMU_DEFINE_LDS)
#define HIPEMU_DEFINE_LDS(ns_open, ns_close, name, doubles) ns_open alignas(64) double name[doubles]; ns_close

#define hipLaunchKernelGGL(kern, grid, block, lds, stream, ...)                                                   \
    do {                                                                                                          \
        const dim3 g_ = (grid), b_ = (block);                                                                     \
        (void)(lds); (void)(stream);                                                                              \
        hipemu::st().kernarg = hipemu::first_arg_ptr(__VA_ARGS__);                                                \
        hipemu::st().kernarg_bytes = hipemu::first_arg_bytes(__VA_ARGS__);                                        \
        for (unsigned blk_ = 0; blk_ < g_.x; ++blk_) hipemu::run_block((int)blk_, (int)g_.x, (int)b_.x, [&]() { kern(__VA_ARGS__); }); \
    } while (0)

inline void __syncthreads() { hipemu::sync_block(); }
inline double atomicAdd(double *p, double v) { const double o = *p; *p = o + v; return o; }     // (fibres run one at a time)

using std::max;
using std::min;

inline int __double2hiint(double v) { int64_t b; memcpy(&b, &v, 8); return (int)(b >> 32); }
inline int __double2loint(double v) { int64_t b; memcpy(&b, &v, 8); return (int)(b & 0xffffffff); }
inline double __hiloint2double(int hi, int lo)
{
    const uint64_t b = ((uint64_t)(uint32_t)hi << 32) | (uint32_t)lo;
    double v; memcpy(&v, &b, 8); return v;
}

#define __ATOMIC_ACQ_REL_EMU 0
inline void __builtin_amdgcn_fence(int, const char *) {}
inline void __builtin_amdgcn_wave_barrier() { hipemu::sync_wave(); }
inline void __builtin_amdgcn_sched_barrier(int) {}
inline void __builtin_amdgcn_s_setprio(int) {}
inline void __builtin_amdgcn_s_barrier() { hipemu::sync_block(); }
inline unsigned long long __builtin_amdgcn_read_exec() { return ~0ull; }
inline void *__builtin_amdgcn_kernarg_segment_ptr() { return const_cast<void *>(hipemu::st().kernarg); }
// the implicit arguments follow the explicit ones (here: one struct) at the next multiple of eight bytes
inline void *__builtin_amdgcn_implicitarg_ptr() { return (char *)const_cast<void *>(hipemu::st().kernarg) + ((hipemu::st().kernarg_bytes + 7) & ~(size_t)7); }
inline long long __builtin_readcyclecounter() { return 0; }
// v_rsq_f64: about 2^-26 relative -- the emulation truncates to float precision so that the Newton steps after it have something to do
inline double __builtin_amdgcn_rsq(double x) { return (double)(float)(1.0 / std::sqrt(x)); }
inline int __builtin_amdgcn_readlane(int v, int l) { return (int)(uint32_t)hipemu::exchange((uint32_t)v, l, "v_readlane"); }
inline int __builtin_amdgcn_readfirstlane(int v) { return (int)(uint32_t)hipemu::exchange((uint32_t)v, 0, "v_readfirstlane"); }
inline int __builtin_amdgcn_update_dpp(int, int src, int ctrl, int, int, bool)
{
    const int lane = hipemu::tid() % hipemu::kWave;
    int from;
    if (ctrl >= 0 && ctrl <= 0xFF) from = (lane & ~3) | ((ctrl >> (2 * (lane & 3))) & 3);                 // quad_perm
    else if (ctrl == 0x140) from = (lane & ~15) | (15 - (lane & 15));                                     // row_mirror
    else if (ctrl == 0x141) from = (lane & ~7) | (7 - (lane & 7));                                        // row_half_mirror
    else if (ctrl >= 0x150 && ctrl <= 0x15F) from = (lane & ~15) | (ctrl - 0x150);                        // row_share
    else { fprintf(stderr, "hipemu: DPP control 0x%x is not modelled\n", ctrl); abort(); }
    return (int)(uint32_t)hipemu::exchange((uint32_t)src, from, "DPP");
}
inline unsigned long long __ballot(int p) { return hipemu::ballot(p != 0); }
// v_mfma_f64_16x16x4_f64: D = C + A B with A[m = lane & 15][k = lane >> 4], B[k = lane >> 4][n = lane & 15] one double per lane, C / D
// four per lane: row (lane >> 4) + 4 r, column lane & 15 in element r.  (clang's ext_vector_type spelled for g++.)
#define ext_vector_type(N) vector_size(8 * (N))
typedef double hipemu_v4d __attribute__((vector_size(32)));
inline hipemu_v4d __builtin_amdgcn_mfma_f64_16x16x4f64(double a, double b, hipemu_v4d c, int, int, int)
{
    const int lane = hipemu::tid() % hipemu::kWave, col = lane & 15, g = lane >> 4;
    uint64_t ab, bb; memcpy(&ab, &a, 8); memcpy(&bb, &b, 8);
    double bk[4];
    for (int k = 0; k < 4; ++k) { const uint64_t x = hipemu::exchange(bb, 16 * k + col, "v_mfma (B)"); memcpy(&bk[k], &x, 8); }
    hipemu_v4d d = c;
    for (int r = 0; r < 4; ++r) {
        const int row = g + 4 * r;
        double acc = c[r];
        for (int k = 0; k < 4; ++k) { const uint64_t x = hipemu::exchange(ab, 16 * k + row, "v_mfma (A)"); double ak; memcpy(&ak, &x, 8); acc = std::fma(ak, bk[k], acc); }
        d[r] = acc;
    }
    return d;
}
template <class T> inline T __shfl(T v, int src)
{
    static_assert(sizeof(T) <= 8, "");
    uint64_t b = 0; memcpy(&b, &v, sizeof(T));
    b = hipemu::exchange(b, src, "__shfl");
    T r; memcpy(&r, &b, sizeof(T)); return r;
}
template <class T> inline T __shfl_xor(T v, int mask) { return __shfl(v, (hipemu::tid() % hipemu::kWave) ^ mask); }
