// Mock HIP runtime for HOST-SIDE KERNEL TESTS ONLY (tests/emu).
//
// The product (onssen_amd/csrc/onssen_hip.hip) is plain HIP for gfx950 and has
// no conditional compilation.  To exercise its index arithmetic, MFMA
// fragment bookkeeping and predication on a box without a GPU, the test
// suite compiles that same source with g++ against THIS header instead of
// ROCm's <hip/hip_runtime.h>.  Every workgroup runs as real OS threads
// (one per work-item, workgroups one after another); __syncthreads() is a
// pthread barrier, wave-level collectives (shuffles, the f32 MFMA) exchange
// operands through a per-wave buffer following the gfx950 lane layouts
// documented in cdna_hip_programming.md section 3.  Slow by design; sizes in
// tests/test_emu_*.py are tiny.  Never shipped, never timed.
#pragma once
#define ONSSEN_HOST_EMULATION 1
#include <pthread.h>
#include <sched.h>
#include <sys/wait.h>
#include <unistd.h>
#include <cmath>
#include <cstdint>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <memory>
#include <thread>
#include <vector>

struct dim3 {
  unsigned x, y, z;
  constexpr dim3(unsigned x_ = 1, unsigned y_ = 1, unsigned z_ = 1) : x(x_), y(y_), z(z_) {}
};
struct float2 { float x, y; };
struct float4 { float x, y, z, w; };
struct double2 { double x, y; };
struct uint2 { unsigned x, y; };
static inline uint2 make_uint2(unsigned x, unsigned y) { return {x, y}; }
static inline float2 make_float2(float x, float y) { return {x, y}; }
static inline float4 make_float4(float x, float y, float z, float w) { return {x, y, z, w}; }
static inline double2 make_double2(double x, double y) { return {x, y}; }

typedef void* hipStream_t;
typedef int hipError_t;
enum { hipSuccess = 0, hipErrorInvalidValue = 1 };
static inline hipError_t hipGetLastError() { return hipSuccess; }
static inline const char* hipGetErrorString(hipError_t) { return "emu"; }
static inline hipError_t hipMemsetAsync(void* p, int v, size_t n, hipStream_t) { memset(p, v, n); return hipSuccess; }
enum hipDeviceAttribute_t { hipDeviceAttributeMultiprocessorCount = 0 };
static inline hipError_t hipGetDevice(int* d) { *d = 0; return hipSuccess; }
static inline hipError_t hipDeviceGetAttribute(int* v, hipDeviceAttribute_t, int) { *v = 256; return hipSuccess; }

#define __global__
#define __device__
#define __host__
#define __forceinline__ inline __attribute__((always_inline))
#define __shared__ static
#define __launch_bounds__(...)

namespace emu {
struct WaveBuf { float a[64]; float b[64]; };
struct BlockCtx {
  pthread_barrier_t block_bar;
  std::vector<pthread_barrier_t> wave_bar;
  std::vector<WaveBuf> wbuf;
  explicit BlockCtx(int nthr) : wave_bar((nthr + 63) / 64), wbuf((nthr + 63) / 64) {
    pthread_barrier_init(&block_bar, nullptr, nthr);
    for (size_t w = 0; w < wave_bar.size(); ++w) {
      int n = nthr - int(w) * 64;
      pthread_barrier_init(&wave_bar[w], nullptr, n > 64 ? 64 : n);
    }
  }
  ~BlockCtx() {
    pthread_barrier_destroy(&block_bar);
    for (auto& b : wave_bar) pthread_barrier_destroy(&b);
  }
};
inline thread_local BlockCtx* ctx = nullptr;
inline thread_local int lane = 0, wave = 0;
inline void wave_sync() { pthread_barrier_wait(&ctx->wave_bar[wave]); }

template <class F>
void launch(dim3 grid, dim3 block, F body);
}  // namespace emu

inline thread_local dim3 threadIdx, blockIdx, blockDim, gridDim;

// Runs the workgroups [b_lo, b_hi) of the grid one after another, each as `nthr` OS threads.
template <class F>
static void emu_run_blocks(dim3 grid, dim3 block, F& body, unsigned b_lo, unsigned b_hi) {
  const int nthr = int(block.x * block.y * block.z);
  emu::BlockCtx c(nthr);
  std::vector<std::thread> th;
  th.reserve(nthr);
  for (int tid = 0; tid < nthr; ++tid) {
    th.emplace_back([&, tid]() {
      emu::ctx = &c;
      emu::lane = tid & 63;
      emu::wave = tid >> 6;
      threadIdx = dim3(tid % block.x, (tid / block.x) % block.y, tid / (block.x * block.y));
      blockDim = block;
      gridDim = grid;
      for (unsigned b = b_lo; b < b_hi; ++b) {
        blockIdx = dim3(b % grid.x, (b / grid.x) % grid.y, b / (grid.x * grid.y));
        body();
        pthread_barrier_wait(&c.block_bar);
      }
    });
  }
  for (auto& t : th) t.join();
}

// Default: workgroups run one after another in this process (any host memory works).
// ONSSEN_EMU_FORK=1: every workgroup runs CONCURRENTLY in its own forked process (its own copy of the
// `__shared__` statics); kernels whose workgroups talk to each other inside a launch need this, and the
// test must then keep all "device" buffers in MAP_SHARED memory.
template <class F>
void emu::launch(dim3 grid, dim3 block, F body) {
  const int nthr = int(block.x * block.y * block.z);
  if (nthr % 64 != 0) { fprintf(stderr, "emu: block size must be a multiple of 64\n"); abort(); }
  const unsigned nblk = grid.x * grid.y * grid.z;
  const char* fk = getenv("ONSSEN_EMU_FORK");
  if (!(fk && fk[0] == '1') || nblk == 1) {
    emu_run_blocks(grid, block, body, 0, nblk);
    return;
  }
  std::vector<pid_t> kids;
  for (unsigned b = 0; b < nblk; ++b) {
    pid_t pid = fork();
    if (pid == 0) {
      emu_run_blocks(grid, block, body, b, b + 1);
      _exit(0);
    }
    kids.push_back(pid);
  }
  for (pid_t k : kids) {
    int st = 0;
    waitpid(k, &st, 0);
    if (!WIFEXITED(st) || WEXITSTATUS(st) != 0) { fprintf(stderr, "emu: workgroup process failed\n"); abort(); }
  }
}

#define hipLaunchKernelGGL(kern, grid, block, shmem, stream, ...) \
  emu::launch((grid), (block), [&]() { (kern)(__VA_ARGS__); })

static inline void __syncthreads() { pthread_barrier_wait(&emu::ctx->block_bar); }

static inline float __shfl(float v, int src) {
  auto& w = emu::ctx->wbuf[emu::wave];
  w.a[emu::lane] = v;
  emu::wave_sync();
  float r = w.a[src & 63];
  emu::wave_sync();
  return r;
}
static inline float __shfl_xor(float v, int mask) { return __shfl(v, emu::lane ^ mask); }
// ds_read_b64_tr_b16: lane c of a 16-lane group receives element c % 4 of the four 16-bit elements addressed by lanes
// 4 j + c / 4 (j = 0..3) of its group (measured on gfx950: tools/micro/tr16_probe.hip)
typedef short emu_s16x4 __attribute__((vector_size(8)));
static inline emu_s16x4 emu_ds_read_tr16_b64(const unsigned short* p) {
  auto& w = emu::ctx->wbuf[emu::wave];
  static_assert(sizeof(w.a[0]) == 4, "wave buffer of 32-bit slots");
  uintptr_t u = reinterpret_cast<uintptr_t>(p);
  uint32_t lo32 = uint32_t(u), hi32 = uint32_t(uint64_t(u) >> 32);
  memcpy(&w.a[emu::lane], &lo32, 4);
  memcpy(&w.b[emu::lane], &hi32, 4);
  emu::wave_sync();
  emu_s16x4 r;
  const int g = emu::lane & ~15, c = emu::lane & 15;
  for (int j = 0; j < 4; ++j) {
    uint32_t l2, h2;
    memcpy(&l2, &w.a[g + 4 * j + c / 4], 4);
    memcpy(&h2, &w.b[g + 4 * j + c / 4], 4);
    const unsigned short* q = reinterpret_cast<const unsigned short*>(uintptr_t((uint64_t(h2) << 32) | l2));
    r[j] = (short)q[c % 4];
  }
  emu::wave_sync();
  return r;
}
static inline double __shfl_xor(double v, int mask) {
  // two 32-bit halves, like the hardware
  uint64_t u; memcpy(&u, &v, 8);
  float lo, hi; uint32_t l = uint32_t(u), h = uint32_t(u >> 32);
  memcpy(&lo, &l, 4); memcpy(&hi, &h, 4);
  lo = __shfl_xor(lo, mask); hi = __shfl_xor(hi, mask);
  memcpy(&l, &lo, 4); memcpy(&h, &hi, 4);
  u = (uint64_t(h) << 32) | l; memcpy(&v, &u, 8);
  return v;
}

// v_mfma_f32_16x16x4_f32: lane l supplies A[i=l&15][k=l>>4] and B[k=l>>4][j=l&15];
// it receives D[row=4*(l>>4)+r][col=l&15], r=0..3; each output is a k-ordered
// fmaf chain (cdna_hip_programming.md section 3).
typedef float emu_f32x4 __attribute__((vector_size(16)));
static inline emu_f32x4 __builtin_amdgcn_mfma_f32_16x16x4f32(float a, float b, emu_f32x4 c, int, int, int) {
  auto& w = emu::ctx->wbuf[emu::wave];
  w.a[emu::lane] = a;
  w.b[emu::lane] = b;
  emu::wave_sync();
  const int col = emu::lane & 15, rg = emu::lane >> 4;
  for (int r = 0; r < 4; ++r) {
    const int row = 4 * rg + r;
    float acc = c[r];
    for (int k = 0; k < 4; ++k) acc = fmaf(w.a[row + 16 * k], w.b[col + 16 * k], acc);
    c[r] = acc;
  }
  emu::wave_sync();
  return c;
}

// ---- agent-scope atomics, waits, buffer resources (persistent-kernel hand-offs) -----------------
#define __HIP_MEMORY_SCOPE_AGENT 4
#define __hip_atomic_load(p, order, scope) __atomic_load_n((p), __ATOMIC_SEQ_CST)
#define __hip_atomic_fetch_add(p, v, order, scope) __atomic_fetch_add((p), (v), __ATOMIC_SEQ_CST)
#define __hip_atomic_fetch_or(p, v, order, scope) __atomic_fetch_or((p), (v), __ATOMIC_SEQ_CST)
template <class T, class V>
static inline void emu_atomic_store(T* p, V val) {
  const T v = static_cast<T>(val);
  if constexpr (sizeof(T) == 8) {
    uint64_t u; memcpy(&u, &v, 8);
    __atomic_store_n(reinterpret_cast<uint64_t*>(p), u, __ATOMIC_SEQ_CST);
  } else {
    static_assert(sizeof(T) == 4, "4- or 8-byte stores only");
    uint32_t u; memcpy(&u, &v, 4);
    __atomic_store_n(reinterpret_cast<uint32_t*>(p), u, __ATOMIC_SEQ_CST);
  }
}
#define __hip_atomic_store(p, v, order, scope) emu_atomic_store((p), (v))
static inline void __builtin_amdgcn_s_sleep(int) { sched_yield(); }
static inline void __builtin_amdgcn_s_setprio(int) {}
static inline void __builtin_amdgcn_sched_barrier(int) {}
// a hardware wave runs its lanes in lockstep; here the lanes are threads, so the scheduling barrier is a real one
static inline void __builtin_amdgcn_wave_barrier() { emu::wave_sync(); }
// a wave executes s_waitcnt once for all of its lanes: here the lanes are threads, so "every earlier store of
// this WAVE is visible" needs a wave rendezvous (call it from wave-uniform control flow only)
static inline void __builtin_amdgcn_s_waitcnt(int) {
  __atomic_thread_fence(__ATOMIC_SEQ_CST);
  emu::wave_sync();
  __atomic_thread_fence(__ATOMIC_SEQ_CST);
}
struct __amdgpu_buffer_rsrc_t { char* base; int bytes; };
static inline __amdgpu_buffer_rsrc_t __builtin_amdgcn_make_buffer_rsrc(void* p, short, int bytes, int) {
  return {static_cast<char*>(p), bytes};
}
typedef unsigned int emu_u32x4 __attribute__((vector_size(16)));
static inline emu_u32x4 __builtin_amdgcn_raw_buffer_load_b128(__amdgpu_buffer_rsrc_t r, int voff, int soff, int) {
  emu_u32x4 v = {0, 0, 0, 0};
  // raw-buffer range check like the hardware: on the (unsigned) per-lane offset only -- the scalar offset is
  // added to the address afterwards and is NOT checked
  if ((unsigned long)(unsigned)voff + 16 <= (unsigned long)r.bytes) {
    __atomic_thread_fence(__ATOMIC_SEQ_CST);
    memcpy(&v, r.base + voff + soff, 16);
  }
  return v;
}
static inline float __builtin_amdgcn_rcpf(float x) { return 1.0f / x; }
static inline float __builtin_amdgcn_exp2f(float x) { return exp2f(x); }

// ---- bf16: __bf16 is only ever used by the product as the element type of an 8-wide MFMA operand vector
#define __bf16 short
typedef short emu_s16x8 __attribute__((vector_size(16)));
static inline float emu_bf16_to_f32(short b) {
  uint32_t u = uint32_t(uint16_t(b)) << 16; float f; memcpy(&f, &u, 4); return f;
}
// v_mfma_f32_16x16x32_bf16: lane l supplies A[i=l&15][k=8*(l>>4)+j] and B[k=8*(l>>4)+j][col=l&15], j=0..7;
// products are exact in fp32, accumulation in fp32
static inline emu_f32x4 __builtin_amdgcn_mfma_f32_16x16x32_bf16(emu_s16x8 a, emu_s16x8 b, emu_f32x4 c, int, int, int) {
  static thread_local int dummy; (void)dummy;
  struct Buf { short a[64][8]; short b[64][8]; };
  static Buf bufs[16];                      // one per wave of the running workgroup (statics are per process)
  Buf& w = bufs[emu::wave];
  for (int j = 0; j < 8; ++j) { w.a[emu::lane][j] = a[j]; w.b[emu::lane][j] = b[j]; }
  emu::wave_sync();
  const int col = emu::lane & 15, rg = emu::lane >> 4;
  for (int r = 0; r < 4; ++r) {
    const int row = 4 * rg + r;
    float acc = c[r];
    for (int g = 0; g < 4; ++g)
      for (int j = 0; j < 8; ++j)
        acc = fmaf(emu_bf16_to_f32(w.a[row + 16 * g][j]), emu_bf16_to_f32(w.b[col + 16 * g][j]), acc);
    c[r] = acc;
  }
  emu::wave_sync();
  return c;
}
// v_mfma_f32_32x32x16_bf16: lane l supplies A[i=l&31][k=8*(l>>5)+j] and B[k=8*(l>>5)+j][col=l&31], j=0..7; the 16 accumulators of a
// lane are column l&31, rows (reg&3) + 8*(reg>>2) + 4*(l>>5)
typedef float emu_f32x16 __attribute__((vector_size(64)));
static inline emu_f32x16 __builtin_amdgcn_mfma_f32_32x32x16_bf16(emu_s16x8 a, emu_s16x8 b, emu_f32x16 c, int, int, int) {
  struct Buf { short a[64][8]; short b[64][8]; };
  static Buf bufs[16];
  Buf& w = bufs[emu::wave];
  for (int j = 0; j < 8; ++j) { w.a[emu::lane][j] = a[j]; w.b[emu::lane][j] = b[j]; }
  emu::wave_sync();
  const int col = emu::lane & 31, half = emu::lane >> 5;
  for (int r = 0; r < 16; ++r) {
    const int row = (r & 3) + 8 * (r >> 2) + 4 * half;
    float acc = c[r];
    for (int g = 0; g < 2; ++g)
      for (int j = 0; j < 8; ++j)
        acc = fmaf(emu_bf16_to_f32(w.a[row + 32 * g][j]), emu_bf16_to_f32(w.b[col + 32 * g][j]), acc);
    c[r] = acc;
  }
  emu::wave_sync();
  return c;
}
static inline long long clock64() { return 0; }
static inline long long wall_clock64() { return 0; }

// ---- XCD-local persistent kernel support
#define __HIP_MEMORY_SCOPE_WORKGROUP 2
template <class T> static inline T emu_fetch_max(T* p, T v) {
  T old = __atomic_load_n(p, __ATOMIC_SEQ_CST);
  while (old < v && !__atomic_compare_exchange_n(p, &old, v, false, __ATOMIC_SEQ_CST, __ATOMIC_SEQ_CST)) {}
  return old;
}
#define __hip_atomic_fetch_max(p, v, order, scope) emu_fetch_max((p), (v))
static inline void __builtin_amdgcn_fence(int, const char*) { __atomic_thread_fence(__ATOMIC_SEQ_CST); }
// XCC id: 0 for every workgroup, or scrambled per workgroup when ONSSEN_EMU_SCRAMBLE_XCC=1 (exercises the
// placement-independent protocol)
static inline unsigned __builtin_amdgcn_s_getreg(int) {
  const char* e = getenv("ONSSEN_EMU_SCRAMBLE_XCC");
  return (e && e[0] == '1') ? (blockIdx.x / 8u + blockIdx.x) & 7u : 0u;
}
static inline void __builtin_amdgcn_raw_buffer_store_b16(short v, __amdgpu_buffer_rsrc_t r, int voff, int soff, int) {
  if ((unsigned long)(unsigned)voff + 2 <= (unsigned long)r.bytes) {
    __atomic_store_n(reinterpret_cast<short*>(r.base + voff + soff), v, __ATOMIC_SEQ_CST);
  }
}
static inline unsigned __builtin_amdgcn_raw_buffer_load_b32(__amdgpu_buffer_rsrc_t r, int voff, int soff, int) {
  unsigned v = 0;
  if ((unsigned long)(unsigned)voff + 4 <= (unsigned long)r.bytes)
    v = __atomic_load_n(reinterpret_cast<unsigned*>(r.base + voff + soff), __ATOMIC_SEQ_CST);
  return v;
}
static inline void __builtin_amdgcn_raw_buffer_store_b32(unsigned v, __amdgpu_buffer_rsrc_t r, int voff, int soff, int) {
  if ((unsigned long)(unsigned)voff + 4 <= (unsigned long)r.bytes)
    __atomic_store_n(reinterpret_cast<unsigned*>(r.base + voff + soff), v, __ATOMIC_SEQ_CST);
}
static inline void __builtin_amdgcn_raw_buffer_store_b128(emu_u32x4 v, __amdgpu_buffer_rsrc_t r, int voff, int soff, int) {
  if ((unsigned long)(unsigned)voff + 16 <= (unsigned long)r.bytes) {
    memcpy(r.base + voff + soff, &v, 16);
    __atomic_thread_fence(__ATOMIC_SEQ_CST);
  }
}
typedef unsigned int emu_u32x2 __attribute__((vector_size(8)));
static inline void __builtin_amdgcn_raw_buffer_store_b64(emu_u32x2 v, __amdgpu_buffer_rsrc_t r, int voff, int soff, int) {
  if ((unsigned long)(unsigned)voff + 8 <= (unsigned long)r.bytes) {
    uint64_t u; memcpy(&u, &v, 8);
    __atomic_store_n(reinterpret_cast<uint64_t*>(r.base + voff + soff), u, __ATOMIC_SEQ_CST);
  }
}
// DPP quad_perm (ctrl 0..255: two selector bits per lane of a quad), full row / bank masks: lane l receives the source
// value of lane (l & ~3) + sel.  Every lane of the wave must call it (lanes are threads here).
static inline int __builtin_amdgcn_update_dpp(int old, int src, int ctrl, int, int, bool bound_ctrl) {
  // quad_perm (ctrl 0..255: two selector bits per lane of a quad) and row_shl:n (0x101..0x10f: lane l receives lane l+n of
  // its row of 16; beyond the row: 0 with bound_ctrl, else `old`), full row / bank masks.  Every lane of the wave must call it.
  if (ctrl < 0 || (ctrl > 255 && (ctrl < 0x101 || ctrl > 0x10f) && (ctrl < 0x121 || ctrl > 0x12f))) { fprintf(stderr, "emu: DPP control not emulated\n"); abort(); }
  auto& w = emu::ctx->wbuf[emu::wave];
  memcpy(&w.a[emu::lane], &src, 4);
  emu::wave_sync();
  int r = old;
  if (ctrl <= 255) {
    memcpy(&r, &w.a[(emu::lane & ~3) + ((ctrl >> (2 * (emu::lane & 3))) & 3)], 4);
  } else if (ctrl >= 0x121) {       // row_ror:n -- lane l receives lane (l - n) mod 16 of its row
    memcpy(&r, &w.a[(emu::lane & ~15) + (((emu::lane & 15) - (ctrl - 0x120)) & 15)], 4);
  } else {
    const int srcl = (emu::lane & 15) + (ctrl - 0x100);
    if (srcl < 16) memcpy(&r, &w.a[(emu::lane & ~15) + srcl], 4);
    else if (bound_ctrl) r = 0;
  }
  emu::wave_sync();
  return r;
}
// v_perm_b32: result byte i = byte sel[i] of {s0 (bytes 4-7), s1 (bytes 0-3)}; selectors 0x0c -> 0x00, >= 0x0d -> 0xff
static inline unsigned __builtin_amdgcn_perm(unsigned s0, unsigned s1, unsigned sel) {
  const unsigned long long src = ((unsigned long long)s0 << 32) | s1;
  unsigned r = 0;
  for (int i = 0; i < 4; ++i) {
    const unsigned c = (sel >> (8 * i)) & 0xffu;
    unsigned b;
    if (c < 8) b = (unsigned)(src >> (8 * c)) & 0xffu;
    else if (c == 0x0c) b = 0;
    else if (c >= 0x0d) b = 0xffu;
    else { fprintf(stderr, "emu: v_perm_b32 selector %u not emulated\n", c); abort(); }
    r |= b << (8 * i);
  }
  return r;
}
// the lanes of an emulated wave are threads with their own copy of every wave-uniform value
static inline int __builtin_amdgcn_readfirstlane(int v) { return v; }
// v_readlane_b32: every lane of the wave receives lane `l`'s value (all lanes must call it: lanes are threads here)
static inline int __builtin_amdgcn_readlane(int v, int l) {
  auto& w = emu::ctx->wbuf[emu::wave];
  memcpy(&w.a[emu::lane], &v, 4);
  emu::wave_sync();
  int r;
  memcpy(&r, &w.a[l & 63], 4);
  emu::wave_sync();
  return r;
}
// buffer_load_dwordx4 ... lds: every lane copies `size` bytes (zeros when out of range) to lds + lane * size
static inline void __builtin_amdgcn_raw_ptr_buffer_load_lds(__amdgpu_buffer_rsrc_t r, void* lds, int size, int voff, int soff, int off, int) {
  char* dst = static_cast<char*>(lds) + emu::lane * size;
  if ((unsigned long)(unsigned)voff + size <= (unsigned long)r.bytes) memcpy(dst, r.base + voff + soff + off, size);
  else memset(dst, 0, size);
  __atomic_thread_fence(__ATOMIC_SEQ_CST);
}
static inline bool __all(bool p) {
  auto& w = emu::ctx->wbuf[emu::wave];
  w.a[emu::lane] = p ? 1.0f : 0.0f;
  emu::wave_sync();
  bool r = true;
  for (int i = 0; i < 64; ++i) r = r && (w.a[i] != 0.0f);
  emu::wave_sync();
  return r;
}
