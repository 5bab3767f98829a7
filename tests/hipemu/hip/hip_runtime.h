// TEST INFRASTRUCTURE ONLY -- a single-thread wave64 emulator that lets the product's kernel
// headers (cuda-image-and-video-codec_amd/csrc/*_kernels.hpp) be compiled with g++ and executed
// on the CPU, lane by lane, so kernel logic can be checked against the oracle where no GPU
// exists.  It shadows <hip/hip_runtime.h> ONLY for the test driver (tests/hipemu/emu_driver.cpp,
// built with -I tests/hipemu); the product is always built by hipcc against the real header and
// contains no emulation hooks.
//
// Model: every lane of a workgroup is a coroutine (hand-rolled x86-64 context switch).  A lane
// runs until it reaches a cross-lane operation (DPP move, ballot, shuffle, barrier), parks its
// operand and yields; the last lane of the wave (or block, for __syncthreads) to arrive performs
// the exchange for everybody.  Cross-lane builtins must therefore be reached in wave-uniform
// control flow -- the same requirement the hardware has.
#pragma once
#include <cmath>
#include <cstdint>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <functional>
#include <vector>

#define __global__
#define __device__
#define __host__
#define __forceinline__ inline __attribute__((always_inline))
#define __launch_bounds__(...)
#define __shared__ static

struct dim3 { unsigned x, y, z; dim3(unsigned a = 1, unsigned b = 1, unsigned c = 1) : x(a), y(b), z(c) {} };
struct int2 { int x, y; };
struct uint2 { unsigned x, y; };
struct float2 { float x, y; };
struct uint4 { unsigned x, y, z, w; };
struct float4 { float x, y, z, w; };
static inline float4 make_float4(float x, float y, float z, float w) { float4 r = { x, y, z, w }; return r; }
static inline int2 make_int2(int a, int b) { int2 r = { a, b }; return r; }

namespace emu {

struct Lane {
    void *sp = nullptr;
    char *stack = nullptr;
    bool done = false;
    unsigned tid = 0;
    // collective parking
    uint64_t in = 0, out = 0;
    unsigned aux = 0;
    uint64_t gen_seen = 0;
};

struct Wave { unsigned arrived = 0, alive = 0; uint64_t gen = 0; };

struct State {
    std::vector<Lane> lanes;
    std::vector<Wave> waves;
    unsigned bar_arrived = 0, alive = 0;
    uint64_t bar_gen = 0;
    Lane *cur = nullptr;
    void *sched_sp = nullptr;
    std::function<void()> body;
};

extern State g;
extern "C" void emu_switch(void **from_sp, void *to_sp);
void yield();

}  // namespace emu

extern dim3 threadIdx, blockIdx, blockDim, gridDim;

namespace emu {

enum Op { OP_DPP_SHR, OP_DPP_SHL, OP_BALLOT, OP_SHFL_XOR, OP_SHFL_UP, OP_READFIRST };

// generic wave collective: park (in, aux), last arriver resolves all lanes of the wave
uint64_t collective(Op op, uint64_t in, unsigned aux, uint64_t old);
void barrier();
void launch(dim3 grid, dim3 block, const std::function<void()> &body);

}  // namespace emu

static inline unsigned __builtin_amdgcn_update_dpp(unsigned old, unsigned src, int ctrl, int, int, bool)
{
    if (ctrl == 0x138) return (unsigned)emu::collective(emu::OP_DPP_SHR, src, 0, old);
    if (ctrl == 0x130) return (unsigned)emu::collective(emu::OP_DPP_SHL, src, 0, old);
    fprintf(stderr, "hipemu: unsupported dpp ctrl %x\n", ctrl);
    abort();
}
static inline unsigned long long __builtin_amdgcn_ballot_w64(bool p)
{
    return emu::collective(emu::OP_BALLOT, p ? 1 : 0, 0, 0);
}
static inline bool __builtin_amdgcn_inverse_ballot_w64(unsigned long long m)
{
    return ((m >> (threadIdx.x & 63u)) & 1ull) != 0ull;
}
static inline unsigned __builtin_amdgcn_mbcnt_lo(unsigned m, unsigned base)
{
    unsigned lane = threadIdx.x & 63u;
    unsigned tm = lane >= 32 ? 0xFFFFFFFFu : ((1u << lane) - 1u);
    return (unsigned)__builtin_popcount(m & tm) + base;
}
static inline unsigned __builtin_amdgcn_mbcnt_hi(unsigned m, unsigned base)
{
    unsigned lane = threadIdx.x & 63u;
    unsigned tm = lane < 32 ? 0u : ((lane - 32u) == 0 ? 0u : ((1u << (lane - 32u)) - 1u));
    return (unsigned)__builtin_popcount(m & tm) + base;
}
static inline unsigned __builtin_amdgcn_perm(unsigned s0, unsigned s1, unsigned sel)
{
    uint64_t both = ((uint64_t)s0 << 32) | s1;
    unsigned r = 0;
    for (int i = 0; i < 4; i++) {
        unsigned v = (sel >> (8 * i)) & 0xFF, b;
        if (v < 8) b = (unsigned)((both >> (8 * v)) & 0xFF);
        else if (v == 0x0C) b = 0;
        else if (v >= 0x0D) b = 0xFF;
        else b = 0;   // 8..11 replicate a sign bit on hardware; callers here discard that byte
        r |= b << (8 * i);
    }
    return r;
}
static inline unsigned __builtin_amdgcn_alignbit(unsigned hi, unsigned lo, unsigned sh)
{
    uint64_t both = ((uint64_t)hi << 32) | lo;
    return (unsigned)(both >> (sh & 31u));
}
static inline unsigned __builtin_amdgcn_ubfe(unsigned v, unsigned off, unsigned w)
{
    off &= 31u; w &= 31u;
    return w ? (v >> off) & ((1u << w) - 1u) : 0u;
}
template <typename T> static inline T __shfl_xor(T v, int mask)
{
    uint64_t in = 0;
    memcpy(&in, &v, sizeof v);
    uint64_t o = emu::collective(emu::OP_SHFL_XOR, in, (unsigned)mask, 0);
    T r;
    memcpy(&r, &o, sizeof r);
    return r;
}
template <typename T> static inline T __shfl_up(T v, unsigned d)
{
    uint64_t in = 0;
    memcpy(&in, &v, sizeof v);
    uint64_t o = emu::collective(emu::OP_SHFL_UP, in, d, 0);
    T r;
    memcpy(&r, &o, sizeof r);
    return r;
}
static inline unsigned __builtin_amdgcn_readfirstlane(unsigned v)
{
    return (unsigned)emu::collective(emu::OP_READFIRST, v, 0, 0);
}
static inline unsigned __umul24(unsigned a, unsigned b) { return (a & 0xFFFFFFu) * (b & 0xFFFFFFu); }
static inline int emu_sext24(int v) { return (int)((((unsigned)v & 0xFFFFFFu) ^ 0x800000u)) - 0x800000; }
static inline int __mul24(int a, int b) { return (int)(unsigned)((long long)emu_sext24(a) * (long long)emu_sext24(b)); }
// the emulator runs one lane at a time: atomics are plain read-modify-writes
static inline int atomicOr(int *p, int v) { int o = *p; *p = o | v; return o; }
static inline unsigned atomicOr(unsigned *p, unsigned v) { unsigned o = *p; *p = o | v; return o; }
static inline int atomicAdd(int *p, int v) { int o = *p; *p = o + v; return o; }
static inline unsigned atomicAdd(unsigned *p, unsigned v) { unsigned o = *p; *p = o + v; return o; }
static inline int atomicMax(int *p, int v) { int o = *p; if (v > o) *p = v; return o; }
#define __ATOMIC_RELAXED_EMU 0
#define __HIP_MEMORY_SCOPE_WAVEFRONT 2
#define __HIP_MEMORY_SCOPE_WORKGROUP 3
template <typename T> static inline T __hip_atomic_fetch_add(T *p, T v, int, int) { T o = *p; *p = o + v; return o; }
static inline void __threadfence() {}
static inline void __builtin_amdgcn_s_waitcnt(int) {}
static inline void __builtin_amdgcn_s_setprio(int) {}
static inline void __syncthreads() { emu::barrier(); }
static inline float __uint_as_float(unsigned u) { float f; memcpy(&f, &u, 4); return f; }
static inline unsigned __float_as_uint(float f) { unsigned u; memcpy(&u, &f, 4); return u; }
