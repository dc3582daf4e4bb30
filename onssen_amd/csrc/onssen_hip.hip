// libonssen_hip.so -- hand-written HIP kernels for gfx950 (MI355X / CDNA4) behind the C ABI of
// include/onssen_hip.h.  See DESIGN.md for the data layouts and the per-kernel rooflines.
//
// Kernel inventory (one translation unit, the parts are the .inc files next to this one):
//   fft.inc             stft_logmag_kernel (K1+K2: fp64 radix-4 FFT in LDS, one wavefront per PAIR of frames, fused log10(|X|+eps)),
//                       mask_istft_kernel (K10: mask-apply + fp64 inverse FFT, two speakers per transform, gather overlap-add)
//   gemm.inc            linear_x3q_kernel (K3/K7/K8/K9 on pre-split bf16 images: 256|128 x 320|256 tiles staged by LDS-DMA into a swizzled
//                       layout, 8 waves, register epilogues), linear_x3p_kernel (round-1 form: 256x160 tiles; batched weight gradients;
//                       bias | sigmoid | grouped L2-norm), linear_x3_kernel / linear_kernel (fp32-A forms, exact-fp32 MFMA),
//                       x3_image(_t)_kernel (fp32 -> split-bf16 operand images)
//   lstm.inc            lstm_xcd_kernel (K4, default: XCD-local persistent recurrence, ONE launch per layer, data-tagged h
//                       exchange through the XCD's L2; split-bf16 / bf16 / exact-fp32 instantiations), lstm_step_kernel (one
//                       launch per time step: H > 640 and the re-run of an aborted call)
//   lstm_bwd.inc        lstm_xcd_bwd_kernel / lstm_bwd_step_kernel (training: backward recurrence)
//   labels_cluster.inc  labels_kernel (training labels), kmeans2_* (deep-clustering back end: compaction of the active bins, all
//                       Lloyd iterations in one persistent launch with register-resident rows; launch-per-iteration fallback)
//   loss_sdr.inc        loss_dc_* (value and gradient) / loss_mask_* (chimera mask term: value, winning assignment, gradient),
//                       sdr_* (batch SI-SDR with best permutation, fp64 sums)
//   pack.inc            one-off weight re-layout (gate permutation, MFMA fragment order, BatchNorm fold); training glue: dropout,
//                       row-wise L2 normalisation and train-mode BatchNorm with their backward passes
#include <hip/hip_runtime.h>
#include <hip/hip_bf16.h>

#include <math.h>
#include <stdint.h>
#include <stdlib.h>

#include <type_traits>

#include "../../include/onssen_hip.h"

typedef float f32x4 __attribute__((vector_size(16)));
typedef unsigned int u32x4 __attribute__((vector_size(16)));
typedef unsigned int u32x2 __attribute__((vector_size(8)));
typedef short s16x8 __attribute__((vector_size(16)));       // 8 bf16 bit patterns = one MFMA A/B fragment
typedef __bf16 bf16x8_t __attribute__((vector_size(16)));

// Opaque to the optimiser: the first use of `x` cannot be scheduled (or folded into another block) before `after` exists.
// (The host-side emulation of tests/emu defines ONSSEN_HOST_EMULATION: no such reordering there, and no VGPR constraints.)
#ifdef ONSSEN_HOST_EMULATION
#define ONSSEN_USE_AFTER(x, after) ((void)0)
#define ONSSEN_LDS_PTR(p) ((void*)(p))
#else
#define ONSSEN_USE_AFTER(x, after) asm volatile("" : "+v"(x) : "v"(after))
#define ONSSEN_LDS_PTR(p) ((__attribute__((address_space(3))) void*)(p))     // LDS pointer for buffer_load ... lds
#endif

// ---- split-bf16 ("bf16x3") arithmetic ---------------------------------------------------------------
// An fp32 value x is carried as hi = bf16(x), lo = bf16(x - hi) (|x - hi - lo| <= 2^-17 |x|).  A product
// a*b is evaluated as a_hi*b_hi + a_hi*b_lo + a_lo*b_hi on the bf16 MFMA pipe (each bf16 x bf16 product
// is exact in fp32; accumulation is fp32), dropping only a_lo*b_lo ~ 2^-16 |ab|.  That keeps dot products
// at ~1e-5 relative -- inside the 1e-4 parity budget -- at 3/16 of the exact-fp32 MFMA time.
__device__ __forceinline__ void split_bf16(float x, unsigned short& hi, unsigned short& lo) {
  const __hip_bfloat16 h = __float2bfloat16(x);
  const __hip_bfloat16 l = __float2bfloat16(x - __bfloat162float(h));
  hi = __builtin_bit_cast(unsigned short, h);
  lo = __builtin_bit_cast(unsigned short, l);
}
__device__ __forceinline__ f32x4 mfma_bf16(s16x8 a, s16x8 b, f32x4 c) {
  return __builtin_amdgcn_mfma_f32_16x16x32_bf16(__builtin_bit_cast(bf16x8_t, a), __builtin_bit_cast(bf16x8_t, b), c, 0,
                                                 0, 0);
}

// gfx950's transposing LDS read (ds_read_b64_tr_b16): every lane supplies the address of FOUR consecutive 16-bit elements; inside a
// 16-lane group, lane c receives element c % 4 of the rows addressed by lanes 4 j + c / 4, j = 0..3 (measured:
// tools/micro/tr16_probe.hip).  With the lanes of a group addressing a [4 k][16 m] block row by row, lane c gets column c of
// the block: four consecutive k of one m -- half an MFMA fragment from a tile that was stored as it came from memory.
typedef short s16x4 __attribute__((vector_size(8)));
#ifdef ONSSEN_HOST_EMULATION
__device__ __forceinline__ s16x4 lds_read_tr16(const unsigned short* p) { return emu_ds_read_tr16_b64(p); }
#else
__device__ __forceinline__ s16x4 lds_read_tr16(const unsigned short* p) {
  return __builtin_amdgcn_ds_read_tr16_b64_v4i16((s16x4 __attribute__((address_space(3)))*)p);
}
#endif

// hipGetLastError() is sticky per thread: other libraries' failed probes (e.g. a device query before
// the runtime is initialised) linger.  Every ABI entry clears it first, then checks its own launches.
#define ONSSEN_CLEAR_ERROR() ((void)hipGetLastError())
#define ONSSEN_LAUNCH_CHECK()                   \
  do {                                          \
    hipError_t e__ = hipGetLastError();         \
    if (e__ != hipSuccess) return (int)e__;     \
  } while (0)

static inline int ceil_div(int a, int b) { return (a + b - 1) / b; }
static inline size_t align256(size_t x) { return (x + 255) & ~(size_t)255; }
// Experiment / ablation switches exist only in builds made with -DONSSEN_DEBUG_KNOBS (tools/ab_variants.py): the product
// library reads no environment variable for them and carries their defaults as constants.
#ifdef ONSSEN_DEBUG_KNOBS
#define ONSSEN_KNOB_INT(name, dflt) (getenv(name) ? atoi(getenv(name)) : (dflt))
#else
#define ONSSEN_KNOB_INT(name, dflt) (dflt)
#endif
static inline bool aligned16(const void* p) { return (reinterpret_cast<uintptr_t>(p) & 15u) == 0; }
#ifndef ONSSEN_X3Q_SMALL_DEFAULT
#define ONSSEN_X3Q_SMALL_DEFAULT 0   // onssen_linear_x3p, bias mode: KB (32-k blocks) up to which 128 x 128 tiles run (0 = never); ONSSEN_X3Q_SMALL overrides
#endif
#ifndef ONSSEN_X3R_DEFAULT
#define ONSSEN_X3R_DEFAULT 0     // onssen_linear_x3p, bias mode: 1 = the 32x32x16-MFMA kernel (linear_x3r_kernel) unless ONSSEN_X3R=0 says otherwise
#endif

// ---- device code, one translation unit (the host-side emulation compiles exactly this file too)
#include "pack.inc"
#include "gemm.inc"
#include "labels_cluster.inc"
#include "lstm.inc"
#include "lstm_bwd.inc"
#include "fft.inc"
#include "loss_sdr.inc"
#include "wav_io.inc"      // host code: the batch RIFF reader of the file loader
#include "optim.inc"

// Bounded waits of the persistent kernels: ~0.2 s of polling on the GPU by default.  A run-time setting of the library
// (onssen_xcd_spin_limit), initialised from ONSSEN_XCD_SPIN_LIMIT: the host-side emulation -- where a 'workgroup' is a
// process at the mercy of the OS scheduler -- and data-parallel training -- where a co-tenant RCCL kernel may hold CUs
// while it waits for a slower rank -- raise it; 0 makes every wait give up at once (abort-path tests).
static unsigned& xcd_spin_limit() {
  static unsigned v = getenv("ONSSEN_XCD_SPIN_LIMIT") ? (unsigned)strtoul(getenv("ONSSEN_XCD_SPIN_LIMIT"), nullptr, 10) : 400000u;
  return v;
}

// Co-tenant probe (tools/cotenant_probe.py): `workgroups` workgroups of `threads` threads that do nothing but hold their
// CU for `ticks` ticks of the 100 MHz wall clock -- a stand-in for RCCL's channel kernels next to the persistent recurrences.
// NV live VGPRs per lane and LDS bytes decide whether such a workgroup can share a CU with a recurrence
// workgroup (2 waves x ~216 VGPRs per SIMD and ~114 KB of LDS leave ~80 VGPRs per SIMD and ~46 KB): the light form can,
// the heavy form (>= 100 VGPRs, like a collective's unrolled copy loops) needs a CU of its own.
template <int NV, int LDS>
__global__ __launch_bounds__(1024) void cotenant_spin_kernel(long long ticks, float* sink) {
  __shared__ char cotenant_lds[LDS];
  float v[NV];
#pragma unroll
  for (int i = 0; i < NV; ++i) v[i] = (float)(threadIdx.x + i);
  const long long t0 = wall_clock64();
  for (long long it = 0; it < ticks && wall_clock64() - t0 < ticks; ++it) {   // (bounded by count too)
    __builtin_amdgcn_s_sleep(32);
#ifndef ONSSEN_HOST_EMULATION
#pragma unroll
    for (int i = 0; i < NV; ++i) asm volatile("" : "+v"(v[i]));                // keeps every value in a register across the loop
#endif
  }
  if (sink) {                                                                    // never taken by the probe: keeps v and the LDS alive
    float acc = 0.f;
#pragma unroll
    for (int i = 0; i < NV; ++i) acc += v[i];
    cotenant_lds[threadIdx.x * 31 % LDS] = (char)acc;
    sink[threadIdx.x] = acc + (float)cotenant_lds[(threadIdx.x + 1) * 31 % LDS];
  }
}

// =================================================================================================
// C ABI
// =================================================================================================
extern "C" {

long long onssen_xcd_spin_limit(long long new_limit) {
  const long long old = (long long)xcd_spin_limit();
  if (new_limit >= 0) xcd_spin_limit() = new_limit > 0xffffffffLL ? 0xffffffffu : (unsigned)new_limit;
  return old;
}

int onssen_debug_cotenant_spin(int workgroups, int threads, long long ticks, int heavy, void* stream) {
  if (workgroups <= 0 || threads <= 0 || threads > 1024 || ticks < 0) return ONSSEN_E_ARG;
  ONSSEN_CLEAR_ERROR();
  if (heavy)
    hipLaunchKernelGGL((cotenant_spin_kernel<112, 32768>), dim3((unsigned)workgroups), dim3((unsigned)threads), 0,
                       (hipStream_t)stream, ticks, (float*)nullptr);
  else
    hipLaunchKernelGGL((cotenant_spin_kernel<2, 64>), dim3((unsigned)workgroups), dim3((unsigned)threads), 0,
                       (hipStream_t)stream, ticks, (float*)nullptr);
  ONSSEN_LAUNCH_CHECK();
  return ONSSEN_OK;
}

int onssen_debug_launch_chain(float* scratch, int n, int workgroups, void* stream) {
  if (!scratch || n <= 0 || workgroups <= 0) return ONSSEN_E_ARG;
  ONSSEN_CLEAR_ERROR();
  for (int i = 0; i < n; ++i)
    hipLaunchKernelGGL(probe_kernel, dim3((unsigned)workgroups), dim3(256), 0, (hipStream_t)stream, scratch);
  ONSSEN_LAUNCH_CHECK();
  return ONSSEN_OK;
}

int onssen_abi_version(void) { return ONSSEN_ABI_VERSION; }

const char* onssen_error_string(int code) {
  switch (code) {
    case ONSSEN_OK: return "ok";
    case ONSSEN_E_ARG: return "onssen: invalid argument or unsupported shape";
    case ONSSEN_E_WORKSPACE: return "onssen: workspace too small";
    case ONSSEN_E_ALIGN: return "onssen: pointer or stride alignment requirement violated";
    case ONSSEN_WAV_E_OPEN: return "onssen: wav file could not be opened";
    case ONSSEN_WAV_E_FORMAT: return "onssen: not a RIFF/WAVE file, or malformed";
    case ONSSEN_WAV_E_UNSUPPORTED: return "onssen: wav sample format not supported (PCM 8/16/24/32-bit, IEEE float 32/64)";
    case ONSSEN_WAV_E_SOME_FAILED: return "onssen: at least one file of the batch failed (see the per-file status)";
    default: return code > 0 ? hipGetErrorString((hipError_t)code) : "onssen: unknown error";
  }
}

static int stft_logmag_impl(const float* wav, int B, int n_samples, int64_t wav_stride, int n_fft, int hop, float eps,
                            float* logmag, float* stft_ri, void* stream, const int32_t* n_per_utt) {
  if (!wav || !logmag || B <= 0 || hop <= 0 || n_samples <= n_fft / 2) return ONSSEN_E_ARG;
  ONSSEN_CLEAR_ERROR();
  const int T = 1 + n_samples / hop;
  if (stft_ri && (reinterpret_cast<uintptr_t>(stft_ri) & 7u)) return ONSSEN_E_ALIGN;     // (re, im) pairs are stored as one 8-byte word
  // one wave per PAIR of consecutive frames of an utterance (one complex transform), `ppw` consecutive pairs per wave: the
  // per-lane constants of the transform are built once per wave and the next pair's samples are fetched under this pair's
  // butterflies; fewer pairs per wave when that leaves CUs without a workgroup
  static const int ppw_knob = ONSSEN_KNOB_INT("ONSSEN_STFT_PPW", 4);      // debug builds: tools/ab_variants.py
  const int npair = (T + 1) / 2;
  int ppw = ppw_knob < 1 ? 1 : ppw_knob;
  while (ppw > 1 && (long)B * ceil_div(npair, 4 * ppw) < 512) --ppw;
  const dim3 grid((unsigned)((long)B * ceil_div(npair, 4 * ppw))), block(256);
  hipStream_t st = (hipStream_t)stream;
  if (n_fft == 256)
    hipLaunchKernelGGL((stft_logmag_kernel<256>), grid, block, 0, st, wav, n_samples, (long)wav_stride, hop, T, eps, ppw,
                       logmag, stft_ri, n_per_utt);
  else if (n_fft == 512)
    hipLaunchKernelGGL((stft_logmag_kernel<512>), grid, block, 0, st, wav, n_samples, (long)wav_stride, hop, T, eps, ppw,
                       logmag, stft_ri, n_per_utt);
  else if (n_fft == 1024)
    hipLaunchKernelGGL((stft_logmag_kernel<1024>), grid, block, 0, st, wav, n_samples, (long)wav_stride, hop, T, eps, ppw,
                       logmag, stft_ri, n_per_utt);
  else
    return ONSSEN_E_ARG;
  ONSSEN_LAUNCH_CHECK();
  return ONSSEN_OK;
}

int onssen_stft_logmag_f32(const float* wav, int B, int n_samples, int64_t wav_stride, int n_fft, int hop, float eps,
                           float* logmag, float* stft_ri, void* stream) {
  return stft_logmag_impl(wav, B, n_samples, wav_stride, n_fft, hop, eps, logmag, stft_ri, stream, nullptr);
}

int onssen_stft_logmag_ragged_f32(const float* wav, int B, int n_max, int64_t wav_stride, const int32_t* n_per_utt, int n_fft,
                                  int hop, float eps, float* logmag, float* stft_ri, void* stream) {
  if (!n_per_utt) return ONSSEN_E_ARG;
  return stft_logmag_impl(wav, B, n_max, wav_stride, n_fft, hop, eps, logmag, stft_ri, stream, n_per_utt);
}

int onssen_lstm_geometry(int H, int ug, int* Hp, int* NP, int* KQ, int64_t* whh_elems) {
  if (H <= 0 || ug < 4 || ug > 24 || (ug % 4) != 0) return ONSSEN_E_ARG;
  const int hp = ceil_div(H, ug) * ug, kq = ceil_div(hp, 16);
  if (Hp) *Hp = hp;
  if (NP) *NP = 4 * hp;
  if (KQ) *KQ = kq;
  if (whh_elems) *whh_elems = (int64_t)(hp / ug) * kq * (ug / 4) * 256;
  return ONSSEN_OK;
}

int onssen_lstm_geometry_x3(int H, int ug, int* KQ2, int* Hs, int64_t* whh_x3_elems) {
  int Hp;
  if (onssen_lstm_geometry(H, ug, &Hp, nullptr, nullptr, nullptr) != ONSSEN_OK) return ONSSEN_E_ARG;
  const int kq2 = ceil_div(Hp, 32);
  if (KQ2) *KQ2 = kq2;
  if (Hs) *Hs = 32 * kq2;
  if (whh_x3_elems) *whh_x3_elems = (int64_t)(Hp / ug) * kq2 * (ug / 4) * 1024;
  return ONSSEN_OK;
}

int onssen_lstm_pack_whh_bf16x3(const float* w_hh, int H, int ug, uint16_t* whh_x3, void* stream) {
  int Hp, KQ2;
  int64_t we;
  if (!w_hh || !whh_x3 || onssen_lstm_geometry(H, ug, &Hp, nullptr, nullptr, nullptr) != ONSSEN_OK ||
      onssen_lstm_geometry_x3(H, ug, &KQ2, nullptr, &we) != ONSSEN_OK)
    return ONSSEN_E_ARG;
  if (!aligned16(whh_x3)) return ONSSEN_E_ALIGN;      // 16-byte stores (one fragment lane's 8 values per half)
  ONSSEN_CLEAR_ERROR();
  const long n = we / 16;
  hipLaunchKernelGGL(pack_whh_bf16x3_kernel, dim3((unsigned)((n + 255) / 256 > 4096 ? 4096 : (n + 255) / 256)),
                     dim3(256), 0, (hipStream_t)stream, w_hh, H, Hp, ug, KQ2, H, whh_x3);
  ONSSEN_LAUNCH_CHECK();
  return ONSSEN_OK;
}
int onssen_lstm_pack_wih_bf16x3(const float* w_ih, int in_dim, int H, int ug, uint16_t* wih_x3, void* stream) {
  int Hp;
  if (!w_ih || !wih_x3 || in_dim <= 0 || onssen_lstm_geometry(H, ug, &Hp, nullptr, nullptr, nullptr) != ONSSEN_OK)
    return ONSSEN_E_ARG;
  if (!aligned16(wih_x3)) return ONSSEN_E_ALIGN;
  ONSSEN_CLEAR_ERROR();
  const int KC = ceil_div(in_dim, 32);
  const long n = (long)(Hp / ug) * KC * (ug / 4) * 64;
  hipLaunchKernelGGL(pack_whh_bf16x3_kernel, dim3((unsigned)((n + 255) / 256 > 4096 ? 4096 : (n + 255) / 256)),
                     dim3(256), 0, (hipStream_t)stream, w_ih, H, Hp, ug, KC, in_dim, wih_x3);
  ONSSEN_LAUNCH_CHECK();
  return ONSSEN_OK;
}

int onssen_lstm_pack_f32(const float* w_ih, const float* w_hh, const float* b_ih, const float* b_hh, int in_dim,
                         int bidir_in, int H, int ug, float* wih_p, float* whh_p, float* bias_p, void* stream) {
  int Hp, NP, KQ;
  int64_t we;
  if (onssen_lstm_geometry(H, ug, &Hp, &NP, &KQ, &we) != ONSSEN_OK) return ONSSEN_E_ARG;
  if (!w_ih || !w_hh || !b_ih || !b_hh || !wih_p || !whh_p || !bias_p || in_dim <= 0) return ONSSEN_E_ARG;
  if (bidir_in && in_dim != 2 * H) return ONSSEN_E_ARG;
  const int Kp = bidir_in ? 2 * Hp : ceil_div(in_dim, 4) * 4;
  hipStream_t st = (hipStream_t)stream;
  ONSSEN_CLEAR_ERROR();
  const long n1 = (long)NP * Kp;
  hipLaunchKernelGGL(pack_wih_kernel, dim3((unsigned)((n1 + 255) / 256 > 4096 ? 4096 : (n1 + 255) / 256)), dim3(256),
                     0, st, w_ih, b_ih, b_hh, in_dim, bidir_in, H, Hp, ug, Kp, wih_p, bias_p);
  ONSSEN_LAUNCH_CHECK();
  hipLaunchKernelGGL(pack_whh_kernel, dim3((unsigned)((we + 255) / 256 > 4096 ? 4096 : (we + 255) / 256)), dim3(256),
                     0, st, w_hh, H, Hp, ug, KQ, whh_p);
  ONSSEN_LAUNCH_CHECK();
  return ONSSEN_OK;
}

int onssen_lstm_pack_wih_image_f32(const float* w_ih, const float* b_ih, const float* b_hh, int in_dim, int bidir_in, int H, int ug,
                                   float* wih_p, float* bias_p, uint16_t* wih_img, void* stream) {
  int Hp, NP;
  if (onssen_lstm_geometry(H, ug, &Hp, &NP, nullptr, nullptr) != ONSSEN_OK) return ONSSEN_E_ARG;
  if (!w_ih || !b_ih || !b_hh || !wih_p || !bias_p || !wih_img || in_dim <= 0) return ONSSEN_E_ARG;
  if (bidir_in && in_dim != 2 * H) return ONSSEN_E_ARG;
  if (!aligned16(wih_img)) return ONSSEN_E_ALIGN;
  const int Kp = bidir_in ? 2 * Hp : ceil_div(in_dim, 4) * 4, K = bidir_in ? 2 * Hp : in_dim, KB = ceil_div(K, 32);
  ONSSEN_CLEAR_ERROR();
  const long n = (long)NP * KB * 4;
  hipLaunchKernelGGL(pack_wih_image_kernel, dim3((unsigned)((n + 255) / 256 > 16384 ? 16384 : (n + 255) / 256)), dim3(256), 0,
                     (hipStream_t)stream, w_ih, b_ih, b_hh, in_dim, bidir_in, H, Hp, ug, Kp, K, KB, wih_p, bias_p, wih_img);
  ONSSEN_LAUNCH_CHECK();
  return ONSSEN_OK;
}

int onssen_head_pack_f32(const float* w, const float* b, int N, int H, int Hp, const float* bn_gamma,
                         const float* bn_beta, const float* bn_mean, const float* bn_var, float bn_eps, float* w_p,
                         float* b_p, void* stream) {
  if (!w || !b || !w_p || !b_p || N <= 0 || H <= 0 || Hp < H || (Hp % 4) != 0) return ONSSEN_E_ARG;
  if (bn_gamma && (!bn_beta || !bn_mean || !bn_var)) return ONSSEN_E_ARG;
  ONSSEN_CLEAR_ERROR();
  hipLaunchKernelGGL(pack_head_kernel, dim3((unsigned)N), dim3(256), 0, (hipStream_t)stream, w, b, N, H, Hp,
                     bn_gamma, bn_beta, bn_mean, bn_var, bn_eps, w_p, b_p);
  ONSSEN_LAUNCH_CHECK();
  return ONSSEN_OK;
}

int onssen_linear_f32(const float* A, int64_t a_s0, int64_t a_s1, int R, int M, int K, const float* W, int ldw,
                      const float* bias, int N, int mode, int group, float eps, const float* resid, float* C,
                      int64_t c_s0, int64_t c_s1, void* stream) {
  if (!A || !W || !bias || !C || R <= 0 || M <= 0 || K <= 0 || N <= 0 || ldw < K) return ONSSEN_E_ARG;
  if ((ldw % 4) != 0 || !aligned16(W)) return ONSSEN_E_ALIGN;
  if (mode == ONSSEN_EPI_L2NORM) {
    if (group <= 0 || (lin::BN % group) != 0 || (N % group) != 0) return ONSSEN_E_ARG;
  } else if (resid && mode != ONSSEN_EPI_RELU) {
    return ONSSEN_E_ARG;
  }
  ONSSEN_CLEAR_ERROR();
  LinearArgs p;
  p.A = A; p.a_s0 = (long)a_s0; p.a_s1 = (long)a_s1; p.W = W; p.bias = bias; p.resid = resid; p.C = C;
  p.c_s0 = (long)c_s0; p.c_s1 = (long)c_s1; p.R = R; p.M = M; p.N = N; p.K = K; p.ldw = ldw; p.group = group;
  p.eps = eps;
  const bool a_vec = aligned16(A) && (a_s0 % 4) == 0 && (a_s1 % 4) == 0 && (K % 4) == 0;
  const dim3 grid((unsigned)ceil_div(N, lin::BN), (unsigned)ceil_div(M, lin::BM)), block(256);
  hipStream_t st = (hipStream_t)stream;
#define ONSSEN_LIN(VEC, MODE_) hipLaunchKernelGGL((linear_kernel<VEC, MODE_>), grid, block, 0, st, p)
  if (mode == ONSSEN_EPI_BIAS) {
    if (a_vec) ONSSEN_LIN(true, ONSSEN_EPI_BIAS); else ONSSEN_LIN(false, ONSSEN_EPI_BIAS);
  } else if (mode == ONSSEN_EPI_L2NORM) {
    if (a_vec) ONSSEN_LIN(true, ONSSEN_EPI_L2NORM); else ONSSEN_LIN(false, ONSSEN_EPI_L2NORM);
  } else if (mode == ONSSEN_EPI_SIGMOID) {
    if (a_vec) ONSSEN_LIN(true, ONSSEN_EPI_SIGMOID); else ONSSEN_LIN(false, ONSSEN_EPI_SIGMOID);
  } else if (mode == ONSSEN_EPI_RELU) {
    if (a_vec) ONSSEN_LIN(true, ONSSEN_EPI_RELU); else ONSSEN_LIN(false, ONSSEN_EPI_RELU);
  } else {
    return ONSSEN_E_ARG;
  }
#undef ONSSEN_LIN
  ONSSEN_LAUNCH_CHECK();
  return ONSSEN_OK;
}

int onssen_linear_pack_bf16x3(const float* w, int N, int K, int ld_in, int ld_out, uint16_t* planes, void* stream) {
  if (!w || !planes || N <= 0 || K <= 0 || ld_in < K || ld_out < K || (ld_out % 32) != 0) return ONSSEN_E_ARG;
  ONSSEN_CLEAR_ERROR();
  const long n = (long)N * ld_out;
  hipLaunchKernelGGL(pack_w_bf16x3_kernel, dim3((unsigned)((n + 255) / 256 > 8192 ? 8192 : (n + 255) / 256)), dim3(256),
                     0, (hipStream_t)stream, w, N, K, ld_in, ld_out, planes, planes + n);
  ONSSEN_LAUNCH_CHECK();
  return ONSSEN_OK;
}

int onssen_linear_bf16x3(const float* A, int64_t a_s0, int64_t a_s1, int R, int M, int K, const uint16_t* w_planes,
                         int ldw, const float* bias, int N, int mode, int group, float eps, const float* resid,
                         float* C, int64_t c_s0, int64_t c_s1, void* stream) {
  if (!A || !w_planes || !bias || !C || R <= 0 || M <= 0 || K <= 0 || N <= 0 || ldw < K) return ONSSEN_E_ARG;
  if ((ldw % 32) != 0 || !aligned16(w_planes)) return ONSSEN_E_ALIGN;
  if (mode == ONSSEN_EPI_L2NORM) {
    if (group <= 0 || (lx3::BN % group) != 0 || (N % group) != 0) return ONSSEN_E_ARG;
  } else if (resid) {
    return ONSSEN_E_ARG;
  }
  ONSSEN_CLEAR_ERROR();
  LinearX3Args p;
  p.A = A; p.a_s0 = (long)a_s0; p.a_s1 = (long)a_s1; p.Whi = w_planes; p.Wlo = w_planes + (size_t)N * ldw;
  p.bias = bias; p.resid = resid; p.C = C; p.c_s0 = (long)c_s0; p.c_s1 = (long)c_s1; p.R = R; p.M = M; p.N = N;
  p.K = K; p.ldw = ldw; p.group = group; p.eps = eps;
  static const int x3_ablate = ONSSEN_KNOB_INT("ONSSEN_X3_ABLATE", 0);
  p.ablate = x3_ablate;
  static const int x3_gn = ONSSEN_KNOB_INT("ONSSEN_X3_GN", 4);
  p.tile_group = x3_gn < 1 ? 1 : x3_gn;
  p.c_vec = aligned16(C) && (N % 4) == 0 && (c_s0 % 4) == 0 && (c_s1 % 4) == 0;
  const bool a_vec = aligned16(A) && (a_s0 % 4) == 0 && (a_s1 % 4) == 0 && (K % 4) == 0;
  // tile height: 256 rows x 1 workgroup per CU (default), or 128 rows x 2 co-resident (ONSSEN_X3_WM=2)
  static const int wm_env = ONSSEN_KNOB_INT("ONSSEN_X3_WM", 0);
  const int wmv = wm_env == 2 ? 2 : 4;   // measured: the 256-row tile re-reads W half as often and wins end to end
  const dim3 grid((unsigned)ceil_div(N, lx3::BN), (unsigned)ceil_div(M, 64 * wmv)), block(128 * wmv);
  hipStream_t st = (hipStream_t)stream;
#define ONSSEN_LINX3(VEC, MODE_)                                                              \
  do {                                                                                        \
    if (wmv == 4) hipLaunchKernelGGL((linear_x3_kernel<VEC, MODE_, 4>), grid, block, 0, st, p); \
    else hipLaunchKernelGGL((linear_x3_kernel<VEC, MODE_, 2>), grid, block, 0, st, p);          \
  } while (0)
  if (mode == ONSSEN_EPI_BIAS) {
    if (a_vec) ONSSEN_LINX3(true, ONSSEN_EPI_BIAS); else ONSSEN_LINX3(false, ONSSEN_EPI_BIAS);
  } else if (mode == ONSSEN_EPI_L2NORM) {
    if (a_vec) ONSSEN_LINX3(true, ONSSEN_EPI_L2NORM); else ONSSEN_LINX3(false, ONSSEN_EPI_L2NORM);
  } else if (mode == ONSSEN_EPI_SIGMOID) {
    if (a_vec) ONSSEN_LINX3(true, ONSSEN_EPI_SIGMOID); else ONSSEN_LINX3(false, ONSSEN_EPI_SIGMOID);
  } else {
    return ONSSEN_E_ARG;
  }
#undef ONSSEN_LINX3
  ONSSEN_LAUNCH_CHECK();
  return ONSSEN_OK;
}


int onssen_x3_image_f32(const float* src, int64_t s0, int64_t s1, int R, int rows, int K, uint16_t* img, void* stream) {
  if (!src || !img || R <= 0 || rows <= 0 || K <= 0) return ONSSEN_E_ARG;
  if (!aligned16(img)) return ONSSEN_E_ALIGN;
  ONSSEN_CLEAR_ERROR();
  const int KB = ceil_div(K, 32);
  const long n = (long)rows * KB * 4;
  hipLaunchKernelGGL(x3_image_kernel, dim3((unsigned)((n + 255) / 256 > 16384 ? 16384 : (n + 255) / 256)), dim3(256), 0,
                     (hipStream_t)stream, src, (long)s0, (long)s1, R, rows, K, KB, img);
  ONSSEN_LAUNCH_CHECK();
  return ONSSEN_OK;
}

int onssen_x3_image_t_f32(const float* src, int64_t ld, int M, int K, int k_shift, uint16_t* img, void* stream) {
  if (!src || !img || M <= 0 || K <= 0 || ld < M) return ONSSEN_E_ARG;
  if (!aligned16(img)) return ONSSEN_E_ALIGN;
  ONSSEN_CLEAR_ERROR();
  const int KB = ceil_div(K, 32);
  hipLaunchKernelGGL(x3_image_t_kernel, dim3((unsigned)ceil_div(M, 64), (unsigned)KB), dim3(256), 0, (hipStream_t)stream, src,
                     (long)ld, M, K, KB, k_shift, img);
  ONSSEN_LAUNCH_CHECK();
  return ONSSEN_OK;
}

static int x3_image_both_impl(const float* src, int64_t ld, int M, int K, uint16_t* img_rows, uint16_t* img_t, float* colsum,
                              void* stream) {
  if (!src || !img_rows || !img_t || M <= 0 || K <= 0 || ld < M) return ONSSEN_E_ARG;
  if (!aligned16(img_rows) || !aligned16(img_t)) return ONSSEN_E_ALIGN;
  ONSSEN_CLEAR_ERROR();
  const int KB = ceil_div(K, 32), MB = ceil_div(M, 32);
  hipLaunchKernelGGL(x3_image_both_kernel, dim3((unsigned)ceil_div(M, 64), (unsigned)KB), dim3(256), 0, (hipStream_t)stream, src,
                     (long)ld, M, K, KB, MB, img_rows, img_t, colsum);
  ONSSEN_LAUNCH_CHECK();
  return ONSSEN_OK;
}

int onssen_x3_image_both_f32(const float* src, int64_t ld, int M, int K, uint16_t* img_rows, uint16_t* img_t, void* stream) {
  return x3_image_both_impl(src, ld, M, K, img_rows, img_t, nullptr, stream);
}

int onssen_x3_image_both_colsum_f32(const float* src, int64_t ld, int M, int K, uint16_t* img_rows, uint16_t* img_t, float* colsum,
                                    void* stream) {
  if (!colsum) return ONSSEN_E_ARG;
  return x3_image_both_impl(src, ld, M, K, img_rows, img_t, colsum, stream);
}

static int linear_x3p_batched_impl(const uint16_t* a_img, int64_t a_bs, int M, int K, const uint16_t* w_img, int64_t w_bs,
                                   const float* bias, int N, int R, float* C, int64_t c_bs, int64_t c_s0, int64_t c_s1, int n_split,
                                   float* C2, int64_t c2_bs, int64_t c2_s0, int64_t c2_s1, int batch, void* stream,
                                   int n_split_odd = 0) {
  if (!a_img || !w_img || !bias || !C || M <= 0 || K <= 0 || N <= 0 || R <= 0 || batch <= 0 || batch > 65535) return ONSSEN_E_ARG;
  if (C2 && (n_split <= 0 || n_split >= N)) return ONSSEN_E_ARG;
  if (n_split_odd && (!C2 || n_split_odd < 0 || n_split_odd >= N)) return ONSSEN_E_ARG;
  if (!aligned16(a_img) || !aligned16(w_img) || (a_bs % 8) != 0 || (w_bs % 8) != 0) return ONSSEN_E_ALIGN;
  const int KB = ceil_div(K, 32);
  if ((long)lxp::BM * KB * 128 > 0x7fffffffL) return ONSSEN_E_ARG;
  ONSSEN_CLEAR_ERROR();
  LinearXpArgs p;
  p.A = a_img; p.W = w_img; p.bias = bias; p.C = C; p.c_s0 = (long)c_s0; p.c_s1 = (long)c_s1; p.R = R; p.M = M; p.N = N;
  p.KB = KB; p.group = 0; p.eps = 0.f; p.tile_group = 4;
  p.c_vec = !C2 && aligned16(C) && (N % 4) == 0 && (c_s0 % 4) == 0 && (c_s1 % 4) == 0 && (c_bs % 4) == 0;
  p.a_bs = (long)a_bs; p.w_bs = (long)w_bs; p.c_bs = (long)c_bs;
  p.C2 = C2; p.c2_s0 = (long)c2_s0; p.c2_s1 = (long)c2_s1; p.c2_bs = (long)c2_bs; p.n_split = n_split; p.n_split_odd = n_split_odd;
  p.resid = nullptr; p.r_mod = 1; p.dest = nullptr; p.dest_bs = 0; p.F = 0;
  const dim3 grid((unsigned)ceil_div(N, lxp::BN), (unsigned)ceil_div(M, lxp::BM), (unsigned)batch);
  hipLaunchKernelGGL((linear_x3p_kernel<ONSSEN_EPI_BIAS, 4, 3, true>), grid, dim3(512), 0, (hipStream_t)stream, p);
  ONSSEN_LAUNCH_CHECK();
  return ONSSEN_OK;
}

int onssen_linear_x3p_batched(const uint16_t* a_img, int64_t a_bs, int M, int K, const uint16_t* w_img, int64_t w_bs,
                              const float* bias, int N, float* C, int64_t c_bs, int64_t ldc, int batch, void* stream) {
  if (ldc < N) return ONSSEN_E_ARG;
  return linear_x3p_batched_impl(a_img, a_bs, M, K, w_img, w_bs, bias, N, 1, C, c_bs, ldc, 0, 0, nullptr, 0, 0, 0, batch, stream);
}

int onssen_linear_x3p_batched_split(const uint16_t* a_img, int64_t a_bs, int M, int K, const uint16_t* w_img, int64_t w_bs,
                                    const float* bias, int N, int R, float* C, int64_t c_bs, int64_t c_s0, int64_t c_s1,
                                    int n_split, float* C2, int64_t c2_bs, int64_t c2_s0, int64_t c2_s1, int batch, void* stream) {
  if (!C2) return ONSSEN_E_ARG;
  return linear_x3p_batched_impl(a_img, a_bs, M, K, w_img, w_bs, bias, N, R, C, c_bs, c_s0, c_s1, n_split, C2, c2_bs, c2_s0, c2_s1,
                                 batch, stream);
}

int onssen_linear_x3p_batched_split_alt(const uint16_t* a_img, int64_t a_bs, int M, int K, const uint16_t* w_img, int64_t w_bs,
                                        const float* bias, int N, int R, float* C, int64_t c_bs, int64_t c_s0, int64_t c_s1,
                                        int n_split, float* C2, int64_t c2_bs, int64_t c2_s0, int64_t c2_s1, int n_split_odd,
                                        int batch, void* stream) {
  if (!C2 || n_split_odd <= 0) return ONSSEN_E_ARG;
  return linear_x3p_batched_impl(a_img, a_bs, M, K, w_img, w_bs, bias, N, R, C, c_bs, c_s0, c_s1, n_split, C2, c2_bs, c2_s0, c2_s1,
                                 batch, stream, n_split_odd);
}

// dW of one bidirectional LSTM layer from ROW-MAJOR images (linear_x3t_kernel): dp_img [K = T*B][2*NP / 32][2][32] (dL/d pre-activation),
// y_img [K][2*Hp / 32][2][32] (the layer's output), x_img [K][ceil(Kx / 32)][2][32] (its input); per direction
//   dW_hh[d] (R-mapped rows, Hp columns) = dP_d^T h_prev_d,   dW_ih[d] (Kx columns) = dP_d^T x
// with h_prev = y rows shifted by -B (forward) / +B (reverse).
int onssen_lstm_wgrad_images_f32(const uint16_t* dp_img, const uint16_t* y_img, const uint16_t* x_img, int K, int B, int NP, int Hp,
                                 int Kx, const float* zero16, int R, float* dW_ih, int64_t ih_bs, int64_t ih_s0, int64_t ih_s1,
                                 float* dW_hh, int64_t hh_bs, int64_t hh_s0, int64_t hh_s1, void* stream) {
  if (!dp_img || !y_img || !x_img || !zero16 || !dW_ih || !dW_hh || K <= 0 || B <= 0 || R <= 0 || NP <= 0 || Hp <= 0 || Kx <= 0 ||
      (NP % 32) != 0 || (Hp % 8) != 0 || (long)K * ((2 * NP) / 32) * 128 > 0x7fffffffL)
    return ONSSEN_E_ARG;
  const int Kx8 = (Kx + 7) & ~7;          // x is read in 8-column pieces: up to 7 columns of its image's zero padding, never stored
  if (!aligned16(dp_img) || !aligned16(y_img) || !aligned16(x_img) || !aligned16(zero16)) return ONSSEN_E_ALIGN;
  ONSSEN_CLEAR_ERROR();
  LinearXtArgs p;
  p.A = dp_img; p.a_pitch = (long)(2 * NP / 32) * 128; p.a_col0[0] = 0; p.a_col0[1] = NP;
  const long y_pitch = (long)ceil_div(2 * Hp, 32) * 128, x_pitch = (long)ceil_div(Kx, 32) * 128;
  // forward direction: [h_prev (y columns [0, Hp), rows k - B) | x];  reverse: [x | h_prev (y columns [Hp, 2 Hp), rows k + B)]
  p.seg[0][0] = XtSeg{y_img, y_pitch, 0, Hp, -B, Hp};   p.seg[0][1] = XtSeg{x_img, x_pitch, 0, Kx8, 0, Kx};
  p.seg[1][0] = XtSeg{x_img, x_pitch, 0, Kx8, 0, Kx};   p.seg[1][1] = XtSeg{y_img, y_pitch, Hp, Hp, B, Hp};
  p.zero = (const unsigned short*)zero16;
  p.M = NP; p.N = Hp + Kx8; p.K = K; p.R = R; p.tile_group = 4;
  p.C[0][0] = dW_hh;           p.s0[0][0] = hh_s0; p.s1[0][0] = hh_s1;
  p.C[0][1] = dW_ih;           p.s0[0][1] = ih_s0; p.s1[0][1] = ih_s1;
  p.C[1][0] = dW_ih + ih_bs;   p.s0[1][0] = ih_s0; p.s1[1][0] = ih_s1;
  p.C[1][1] = dW_hh + hh_bs;   p.s0[1][1] = hh_s0; p.s1[1][1] = hh_s1;
  const dim3 grid((unsigned)ceil_div(p.N, 160), (unsigned)ceil_div(p.M, 256), 2);
  hipLaunchKernelGGL((linear_x3t_kernel<3>), grid, dim3(512), 0, (hipStream_t)stream, p);
  ONSSEN_LAUNCH_CHECK();
  return ONSSEN_OK;
}

// C [M][N] (row m at C + m*ldc) = A^T W for row-major x3 images A [K][ceil(M/32)][2][32] and W [K][ceil(N/32)][2][32] (columns past
// M / N: the images' zero padding): a weight gradient dW = dy^T x from the images of dy and x as they were made for the forward /
// input-gradient GEMMs (linear_x3t_kernel, one problem, one segment).
int onssen_linear_x3t(const uint16_t* a_img, const uint16_t* w_img, int K, int M, int N, const float* zero16, float* C, int64_t ldc,
                      void* stream) {
  if (!a_img || !w_img || !zero16 || !C || K <= 0 || M <= 0 || N <= 0 || ldc < N || (long)K * ceil_div(M, 32) * 128 > 0x7fffffffL)
    return ONSSEN_E_ARG;
  if (!aligned16(a_img) || !aligned16(w_img) || !aligned16(zero16)) return ONSSEN_E_ALIGN;
  ONSSEN_CLEAR_ERROR();
  LinearXtArgs p;
  p.A = a_img; p.a_pitch = (long)ceil_div(M, 32) * 128; p.a_col0[0] = p.a_col0[1] = 0;
  const long w_pitch = (long)ceil_div(N, 32) * 128;
  const int N8 = (N + 7) & ~7;
  for (int zz = 0; zz < 2; ++zz) {
    p.seg[zz][0] = XtSeg{w_img, w_pitch, 0, N8, 0, N};
    p.seg[zz][1] = XtSeg{w_img, w_pitch, 0, 0, 0, 0};
    p.C[zz][0] = p.C[zz][1] = C;
    p.s0[zz][0] = p.s0[zz][1] = ldc; p.s1[zz][0] = p.s1[zz][1] = 0;
  }
  p.zero = (const unsigned short*)zero16;
  p.M = M; p.N = N8; p.K = K; p.R = 1; p.tile_group = 4;
  const dim3 grid((unsigned)ceil_div(p.N, 160), (unsigned)ceil_div(p.M, 256), 1);
  hipLaunchKernelGGL((linear_x3t_kernel<3>), grid, dim3(512), 0, (hipStream_t)stream, p);
  ONSSEN_LAUNCH_CHECK();
  return ONSSEN_OK;
}

static int linear_x3p_impl(const uint16_t* a_img, int M, int K, const uint16_t* w_img, const float* bias, int N, int mode,
                           int group, float eps, const float* resid, int resid_mod, float* C, int R, int64_t c_s0,
                           int64_t c_s1, void* stream, float* inv_norm = nullptr) {
  if (!a_img || !w_img || !bias || !C || R <= 0 || M <= 0 || K <= 0 || N <= 0) return ONSSEN_E_ARG;
  if (!aligned16(a_img) || !aligned16(w_img)) return ONSSEN_E_ALIGN;
  const int KB = ceil_div(K, 32);
  if ((long)lxp::BM * KB * 128 > 0x7fffffffL) return ONSSEN_E_ARG;
  const bool bf16_only = (mode & ONSSEN_EPI_BF16) != 0;   // plain bf16 products: hi halves only
  mode &= ~ONSSEN_EPI_BF16;
  // pairs (group 2) and the residual exist in the 256/128 x 320 kernel only (onssen_linear_x3p_resid)
  const bool pairs = mode == ONSSEN_EPI_L2NORM && group == 2;
  if (mode == ONSSEN_EPI_L2NORM) {
    if (!pairs && (group <= 0 || (group % 4) != 0 || (80 % group) != 0 || 80 / group > 4)) return ONSSEN_E_ARG;
    if ((N % group) != 0) return ONSSEN_E_ARG;
  } else if (mode != ONSSEN_EPI_BIAS && mode != ONSSEN_EPI_SIGMOID) {
    return ONSSEN_E_ARG;
  }
  if (resid && (mode != ONSSEN_EPI_L2NORM || resid_mod <= 0)) return ONSSEN_E_ARG;
  if (inv_norm && (mode != ONSSEN_EPI_L2NORM || pairs)) return ONSSEN_E_ARG;
  ONSSEN_CLEAR_ERROR();
  LinearXpArgs p;
  p.A = a_img; p.W = w_img; p.bias = bias; p.C = C; p.c_s0 = (long)c_s0; p.c_s1 = (long)c_s1; p.R = R; p.M = M; p.N = N;
  p.KB = KB; p.group = group; p.eps = eps; p.resid = resid; p.r_mod = resid ? resid_mod : 1;
  p.dest = nullptr; p.dest_bs = 0; p.F = 0;
  p.a_bs = p.w_bs = p.c_bs = 0;
  p.C2 = inv_norm; p.c2_s0 = p.c2_s1 = p.c2_bs = 0; p.n_split = 0; p.n_split_odd = 0;
  static const int x3_gn = ONSSEN_KNOB_INT("ONSSEN_X3_GN", 4);
  p.tile_group = x3_gn < 1 ? 1 : x3_gn;
  p.c_vec = aligned16(C) && (N % 4) == 0 && (c_s0 % 4) == 0 && (c_s1 % 4) == 0;
  hipStream_t st = (hipStream_t)stream;
  // tile shape (linear_x3q_kernel, LDS-DMA staging): 256 or 128 rows x 320 or 256 columns, unless ONSSEN_X3Q=0 selects the
  // round-1 kernel (256 x 160).  One workgroup per CU: the shape with the smallest (rounds of 256 workgroups) x (tile area)
  // wins, half-height tiles charged 15 % for their lower MFMA : fragment-read ratio; the L2NORM epilogue needs whole feature
  // groups per wave (80 columns) and stays at 320 columns.  ONSSEN_X3Q=320 / 256 forces the width, ONSSEN_X3Q_BM=256 / 128 the height.
  const char* env_q = getenv("ONSSEN_X3Q");   // read per call: the tests switch it
  const char* env_bm = getenv("ONSSEN_X3Q_BM");
  const int use_q = (pairs || resid || inv_norm) ? 1 : env_q ? atoi(env_q) : 1, force_bm = env_bm ? atoi(env_bm) : 0;
  if ((pairs || resid || inv_norm) && (long)lxq::BM_MAX * KB * 128 > 0x7fffffffL) return ONSSEN_E_ARG;
  if (use_q && (long)lxq::BM_MAX * KB * 128 <= 0x7fffffffL) {
    int bm = 256, bn = 320;
    double best = 1e30;
    for (int cbm = 256; cbm >= 128; cbm -= 128)
      for (int cbn = 320; cbn >= 256; cbn -= 64) {
        if (cbn == 256 && (mode == ONSSEN_EPI_L2NORM || use_q == 320)) continue;
        if (cbn == 320 && use_q == 256 && mode != ONSSEN_EPI_L2NORM) continue;
        if (force_bm && cbm != force_bm) continue;
        const double cost = (double)ceil_div((long)ceil_div(M, cbm) * ceil_div(N, cbn), 256) * cbm * cbn * (cbm == 128 ? 1.15 : 1.0);
        if (cost < best) { best = cost; bm = cbm; bn = cbn; }
      }
    const dim3 gridq((unsigned)ceil_div(N, bn), (unsigned)ceil_div(M, bm));
#define ONSSEN_XQ2(MODE_, T_)                                                                                            \
  do {                                                                                                                   \
    if (bm == 256 && bn == 320) hipLaunchKernelGGL((linear_x3q_kernel<MODE_, T_, false, 320, 256>), gridq, dim3(512), 0, st, p);      \
    else if (bm == 256) hipLaunchKernelGGL((linear_x3q_kernel<MODE_, T_, false, 256, 256>), gridq, dim3(512), 0, st, p); \
    else if (bn == 320) hipLaunchKernelGGL((linear_x3q_kernel<MODE_, T_, false, 320, 128>), gridq, dim3(512), 0, st, p); \
    else hipLaunchKernelGGL((linear_x3q_kernel<MODE_, T_, false, 256, 128>), gridq, dim3(512), 0, st, p);                \
  } while (0)
#define ONSSEN_XQ(MODE_)                                     \
  do {                                                       \
    if (bf16_only) ONSSEN_XQ2(MODE_, 1); else ONSSEN_XQ2(MODE_, 3); \
  } while (0)
    // the bias mode (the projections): v_mfma_f32_32x32x16_bf16 tiles (linear_x3r_kernel, round 6c) unless ONSSEN_X3R=0 -- every tile
    // shape, so that a row's bits do not depend on the shape its batch selects
    const char* env_r = getenv("ONSSEN_X3R");
    const bool use_r = mode == ONSSEN_EPI_BIAS && (env_r ? atoi(env_r) != 0 : ONSSEN_X3R_DEFAULT != 0);
    // short K (the first layer's projection: 5 k-steps, then 245 MB of C): 128 x 128 tiles, 64 KB of LDS -- two workgroups per CU, one's
    // stores under the other's k-loop (round 6c experiment, ONSSEN_X3Q_SMALL = the largest KB that takes it; same bits as every x3q tile)
    const char* env_s = getenv("ONSSEN_X3Q_SMALL");
    const int small_kb = env_s ? atoi(env_s) : ONSSEN_X3Q_SMALL_DEFAULT;
    if (mode == ONSSEN_EPI_BIAS && !use_r && KB <= small_kb) {
      const dim3 grids((unsigned)ceil_div(N, 128), (unsigned)ceil_div(M, 128));
      if (bf16_only) hipLaunchKernelGGL((linear_x3q_kernel<ONSSEN_EPI_BIAS, 1, false, 128, 128>), grids, dim3(512), 0, st, p);
      else hipLaunchKernelGGL((linear_x3q_kernel<ONSSEN_EPI_BIAS, 3, false, 128, 128>), grids, dim3(512), 0, st, p);
    } else if (use_r) {
#define ONSSEN_XR(T_)                                                                                                 \
  do {                                                                                                                \
    if (bm == 256 && bn == 320) hipLaunchKernelGGL((linear_x3r_kernel<T_, 320, 256>), gridq, dim3(512), 0, st, p);    \
    else if (bm == 256) hipLaunchKernelGGL((linear_x3r_kernel<T_, 256, 256>), gridq, dim3(512), 0, st, p);            \
    else if (bn == 320) hipLaunchKernelGGL((linear_x3r_kernel<T_, 320, 128>), gridq, dim3(512), 0, st, p);            \
    else hipLaunchKernelGGL((linear_x3r_kernel<T_, 256, 128>), gridq, dim3(512), 0, st, p);                           \
  } while (0)
      if (bf16_only) ONSSEN_XR(1); else ONSSEN_XR(3);
#undef ONSSEN_XR
    } else if (mode == ONSSEN_EPI_BIAS) ONSSEN_XQ(ONSSEN_EPI_BIAS);
    else if (mode == ONSSEN_EPI_L2NORM) {
      if (bm == 256) { if (bf16_only) hipLaunchKernelGGL((linear_x3q_kernel<ONSSEN_EPI_L2NORM, 1, false, 320, 256>), gridq, dim3(512), 0, st, p);
                       else hipLaunchKernelGGL((linear_x3q_kernel<ONSSEN_EPI_L2NORM, 3, false, 320, 256>), gridq, dim3(512), 0, st, p); }
      else { if (bf16_only) hipLaunchKernelGGL((linear_x3q_kernel<ONSSEN_EPI_L2NORM, 1, false, 320, 128>), gridq, dim3(512), 0, st, p);
             else hipLaunchKernelGGL((linear_x3q_kernel<ONSSEN_EPI_L2NORM, 3, false, 320, 128>), gridq, dim3(512), 0, st, p); }
    } else ONSSEN_XQ(ONSSEN_EPI_SIGMOID);
#undef ONSSEN_XQ
#undef ONSSEN_XQ2
    ONSSEN_LAUNCH_CHECK();
    return ONSSEN_OK;
  }
  static const int xp_wms = ONSSEN_KNOB_INT("ONSSEN_X3P_WAVES", 8) == 4 ? 2 : 4;   // 8 waves unless =4
  const dim3 grid((unsigned)ceil_div(N, lxp::BN), (unsigned)ceil_div(M, lxp::BM)), block(128 * xp_wms);
#define ONSSEN_XP(MODE_)                                                                                       \
  do {                                                                                                         \
    if (xp_wms == 4 && bf16_only) hipLaunchKernelGGL((linear_x3p_kernel<MODE_, 4, 1>), grid, block, 0, st, p);    \
    else if (xp_wms == 4) hipLaunchKernelGGL((linear_x3p_kernel<MODE_, 4, 3>), grid, block, 0, st, p);           \
    else if (bf16_only) hipLaunchKernelGGL((linear_x3p_kernel<MODE_, 2, 1>), grid, block, 0, st, p);             \
    else hipLaunchKernelGGL((linear_x3p_kernel<MODE_, 2, 3>), grid, block, 0, st, p);                            \
  } while (0)
  if (mode == ONSSEN_EPI_BIAS) ONSSEN_XP(ONSSEN_EPI_BIAS);
  else if (mode == ONSSEN_EPI_L2NORM) ONSSEN_XP(ONSSEN_EPI_L2NORM);
  else ONSSEN_XP(ONSSEN_EPI_SIGMOID);
#undef ONSSEN_XP
  ONSSEN_LAUNCH_CHECK();
  return ONSSEN_OK;
}


size_t onssen_loss_dc_workspace_bytes(int B) {   // partial Grams + (gradient pass) one M matrix and sum(mag) per utterance
  return B > 0 ? (size_t)B * (lossdc::NBLK + 1) * (lossdc::ZMAX * lossdc::ZMAX + 1) * sizeof(float) : 0;
}


int onssen_linear_x3p(const uint16_t* a_img, int M, int K, const uint16_t* w_img, const float* bias, int N, int mode,
                      int group, float eps, float* C, int R, int64_t c_s0, int64_t c_s1, void* stream) {
  if ((mode & ~ONSSEN_EPI_BF16) == ONSSEN_EPI_L2NORM && group == 2) return ONSSEN_E_ARG;      // pairs: onssen_linear_x3p_resid
  return linear_x3p_impl(a_img, M, K, w_img, bias, N, mode, group, eps, nullptr, 0, C, R, c_s0, c_s1, stream);
}

int onssen_linear_x3p_norms(const uint16_t* a_img, int M, int K, const uint16_t* w_img, const float* bias, int N, int group,
                            float eps, float* C, float* inv_norm, void* stream) {
  if (!inv_norm || group <= 2) return ONSSEN_E_ARG;
  return linear_x3p_impl(a_img, M, K, w_img, bias, N, ONSSEN_EPI_L2NORM, group, eps, nullptr, 0, C, 1, N, 0, stream, inv_norm);
}

int onssen_linear_x3p_pair(const uint16_t* a_img, int M, int K, const uint16_t* w_img, const float* bias, int N, int n_split,
                           int group, float eps, float* C, int R, int64_t c_s0, int64_t c_s1, float* C2, int64_t c2_s0,
                           int64_t c2_s1, int bf16_only, void* stream) {
  if (!a_img || !w_img || !bias || !C || !C2 || R <= 0 || M <= 0 || K <= 0 || N <= 0) return ONSSEN_E_ARG;
  if (group <= 0 || (group % 4) != 0 || (80 % group) != 0 || 80 / group > 4) return ONSSEN_E_ARG;
  if (n_split <= 0 || n_split >= N || (n_split % group) != 0) return ONSSEN_E_ARG;
  if (!aligned16(a_img) || !aligned16(w_img)) return ONSSEN_E_ALIGN;
  const int KB = ceil_div(K, 32);
  if ((long)lxq::BM_MAX * KB * 128 > 0x7fffffffL) return ONSSEN_E_ARG;
  ONSSEN_CLEAR_ERROR();
  LinearXpArgs p;
  p.A = a_img; p.W = w_img; p.bias = bias; p.C = C; p.c_s0 = (long)c_s0; p.c_s1 = (long)c_s1; p.R = R; p.M = M; p.N = N;
  p.KB = KB; p.group = group; p.eps = eps; p.resid = nullptr; p.r_mod = 1;
  p.dest = nullptr; p.dest_bs = 0; p.F = 0;
  p.a_bs = p.w_bs = p.c_bs = 0;
  p.C2 = C2; p.c2_s0 = (long)c2_s0; p.c2_s1 = (long)c2_s1; p.c2_bs = 0; p.n_split = n_split; p.n_split_odd = 0;
  p.tile_group = 4;
  p.c_vec = aligned16(C) && (n_split % 4) == 0 && (c_s0 % 4) == 0 && (c_s1 % 4) == 0;
  // half-height tiles where 256-row tiles would leave CUs idle (the cost model of onssen_linear_x3p)
  const long t256 = (long)ceil_div(M, 256) * ceil_div(N, 320), t128 = (long)ceil_div(M, 128) * ceil_div(N, 320);
  const int bm = (double)ceil_div(t128, 256L) * 128 * 1.15 < (double)ceil_div(t256, 256L) * 256 ? 128 : 256;
  const dim3 gridq((unsigned)ceil_div(N, 320), (unsigned)ceil_div(M, bm));
  hipStream_t st = (hipStream_t)stream;
  if (bm == 256) { if (bf16_only) hipLaunchKernelGGL((linear_x3q_kernel<ONSSEN_EPI_L2NORM_SIGMOID, 1, false, 320, 256>), gridq, dim3(512), 0, st, p);
                   else hipLaunchKernelGGL((linear_x3q_kernel<ONSSEN_EPI_L2NORM_SIGMOID, 3, false, 320, 256>), gridq, dim3(512), 0, st, p); }
  else { if (bf16_only) hipLaunchKernelGGL((linear_x3q_kernel<ONSSEN_EPI_L2NORM_SIGMOID, 1, false, 320, 128>), gridq, dim3(512), 0, st, p);
         else hipLaunchKernelGGL((linear_x3q_kernel<ONSSEN_EPI_L2NORM_SIGMOID, 3, false, 320, 128>), gridq, dim3(512), 0, st, p); }
  ONSSEN_LAUNCH_CHECK();
  return ONSSEN_OK;
}

int onssen_linear_x3p_compact(const uint16_t* a_img, int M, int K, const uint16_t* w_img, const float* bias, int N, int group,
                              float eps, const int32_t* dest, int64_t dest_bs, int F, float* comp, int R, int64_t comp_bs,
                              int bf16_only, void* stream) {
  if (!a_img || !w_img || !bias || !dest || !comp || R <= 0 || M <= 0 || K <= 0 || N <= 0 || F <= 0) return ONSSEN_E_ARG;
  if (group <= 0 || (group % 4) != 0 || (80 % group) != 0 || 80 / group > 4 || N != F * group) return ONSSEN_E_ARG;
  if (!aligned16(a_img) || !aligned16(w_img) || !aligned16(comp) || (comp_bs % 4) != 0) return ONSSEN_E_ALIGN;
  const int KB = ceil_div(K, 32);
  if ((long)lxq::BM_MAX * KB * 128 > 0x7fffffffL) return ONSSEN_E_ARG;
  ONSSEN_CLEAR_ERROR();
  LinearXpArgs p;
  p.A = a_img; p.W = w_img; p.bias = bias; p.C = comp; p.c_s0 = 0; p.c_s1 = (long)comp_bs; p.R = R; p.M = M; p.N = N;
  p.KB = KB; p.group = group; p.eps = eps; p.resid = nullptr; p.r_mod = 1;
  p.a_bs = p.w_bs = p.c_bs = 0;
  p.C2 = nullptr; p.c2_s0 = p.c2_s1 = p.c2_bs = 0; p.n_split = 0; p.n_split_odd = 0;
  p.dest = dest; p.dest_bs = (long)dest_bs; p.F = F;
  p.tile_group = 4;
  p.c_vec = 1;
  // half-height tiles where 256-row tiles would leave CUs idle (the cost model of onssen_linear_x3p)
  const long t256 = (long)ceil_div(M, 256) * ceil_div(N, 320), t128 = (long)ceil_div(M, 128) * ceil_div(N, 320);
  const int bm = (double)ceil_div(t128, 256L) * 128 * 1.15 < (double)ceil_div(t256, 256L) * 256 ? 128 : 256;
  const dim3 gridq((unsigned)ceil_div(N, 320), (unsigned)ceil_div(M, bm));
  hipStream_t st = (hipStream_t)stream;
  if (bm == 256) { if (bf16_only) hipLaunchKernelGGL((linear_x3q_kernel<ONSSEN_EPI_L2NORM_COMPACT, 1, false, 320, 256>), gridq, dim3(512), 0, st, p);
                   else hipLaunchKernelGGL((linear_x3q_kernel<ONSSEN_EPI_L2NORM_COMPACT, 3, false, 320, 256>), gridq, dim3(512), 0, st, p); }
  else { if (bf16_only) hipLaunchKernelGGL((linear_x3q_kernel<ONSSEN_EPI_L2NORM_COMPACT, 1, false, 320, 128>), gridq, dim3(512), 0, st, p);
         else hipLaunchKernelGGL((linear_x3q_kernel<ONSSEN_EPI_L2NORM_COMPACT, 3, false, 320, 128>), gridq, dim3(512), 0, st, p); }
  ONSSEN_LAUNCH_CHECK();
  return ONSSEN_OK;
}

int onssen_linear_x3p_resid(const uint16_t* a_img, int M, int K, const uint16_t* w_img, const float* bias, int N, int group,
                            float eps, const float* resid, int resid_mod, float* C, int R, int64_t c_s0, int64_t c_s1,
                            int bf16_only, void* stream) {
  if (group != 2 && (group <= 0 || (group % 4) != 0)) return ONSSEN_E_ARG;
  return linear_x3p_impl(a_img, M, K, w_img, bias, N, ONSSEN_EPI_L2NORM | (bf16_only ? ONSSEN_EPI_BF16 : 0), group, eps, resid,
                         resid_mod, C, R, c_s0, c_s1, stream);
}

int onssen_loss_dc_f32(const float* emb, const float* one_hot, const float* mag, int B, int TF, int D, int C,
                       float* per_utt, float* total_mag, void* ws, size_t ws_bytes, void* stream) {
  if (!emb || !one_hot || !mag || !per_utt || !total_mag || !ws || B <= 0 || TF <= 0 || D <= 0 || C <= 0 ||
      D + C > lossdc::ZMAX)
    return ONSSEN_E_ARG;
  if (ws_bytes < onssen_loss_dc_workspace_bytes(B)) return ONSSEN_E_WORKSPACE;
  ONSSEN_CLEAR_ERROR();
  hipStream_t st = (hipStream_t)stream;
  if (D + C <= 32)
    hipLaunchKernelGGL(loss_dc_partial_mfma_kernel, dim3(lossdc::NBLK, (unsigned)B), dim3(256), 0, st, emb, one_hot, mag, TF, D, C,
                       (float*)ws);
  else
    hipLaunchKernelGGL(loss_dc_partial_kernel, dim3(lossdc::NBLK, (unsigned)B), dim3(256), 0, st, emb, one_hot, mag, TF, D, C,
                       (float*)ws);
  hipLaunchKernelGGL(loss_dc_final_kernel, dim3((unsigned)B), dim3(256), 0, st, (const float*)ws, D, C, per_utt, total_mag);
  ONSSEN_LAUNCH_CHECK();
  return ONSSEN_OK;
}


int onssen_dc_head_grad_images_f32(const float* emb, const float* inv_norm, const float* one_hot, const float* mag, int B, int T,
                                   int F, int D, int C, float eps, const float* g_per_utt, void* ws, size_t ws_bytes,
                                   uint16_t* img_rows, uint16_t* img_t, float* colsum, void* stream) {
  if (!emb || !inv_norm || !one_hot || !mag || !g_per_utt || !ws || !img_rows || !colsum || B <= 0 || F <= 0 || C <= 0 ||
      C > 4 || D != 20 || T < 32 || !(eps > 0.0f) || (long)B * T > 0x7fffffffL / 64)
    return ONSSEN_E_ARG;
  if (!aligned16(emb) || !aligned16(img_rows) || (img_t && !aligned16(img_t))) return ONSSEN_E_ALIGN;
  if (ws_bytes < onssen_loss_dc_workspace_bytes(B)) return ONSSEN_E_WORKSPACE;
  ONSSEN_CLEAR_ERROR();
  hipStream_t st = (hipStream_t)stream;
  float* gm = (float*)ws + (size_t)B * lossdc::NBLK * (lossdc::ZMAX * lossdc::ZMAX + 1);
  hipLaunchKernelGGL(loss_dc_gradm_kernel, dim3((unsigned)B), dim3(256), 0, st, (const float*)ws, D, C, gm);
  const dim3 grid((unsigned)ceil_div(F, 8), (unsigned)ceil_div(B * T, 32));
  hipLaunchKernelGGL(dc_head_grad_images_kernel, grid, dim3(256), 0, st, emb, inv_norm, one_hot, mag, (const float*)gm, g_per_utt, B, T,
                     F, C, eps, img_rows, img_t, colsum);
  ONSSEN_LAUNCH_CHECK();
  return ONSSEN_OK;
}

int onssen_loss_dc_grad_f32(const float* emb, const float* one_hot, const float* mag, int B, int TF, int D, int C,
                            const float* g_per_utt, float* d_emb, void* ws, size_t ws_bytes, void* stream) {
  if (!emb || !one_hot || !mag || !g_per_utt || !d_emb || !ws || B <= 0 || TF <= 0 || D <= 0 || C <= 0 || C > 4 ||
      D + C > lossdc::ZMAX)
    return ONSSEN_E_ARG;
  if (ws_bytes < onssen_loss_dc_workspace_bytes(B)) return ONSSEN_E_WORKSPACE;
  ONSSEN_CLEAR_ERROR();
  hipStream_t st = (hipStream_t)stream;
  float* gm = (float*)ws + (size_t)B * lossdc::NBLK * (lossdc::ZMAX * lossdc::ZMAX + 1);
  hipLaunchKernelGGL(loss_dc_gradm_kernel, dim3((unsigned)B), dim3(256), 0, st, (const float*)ws, D, C, gm);
  const int nb = ceil_div(TF, 256) > 64 ? 64 : ceil_div(TF, 256);
  const dim3 grid((unsigned)nb, (unsigned)B);
  if (D == 20 && aligned16(emb) && aligned16(d_emb))
    hipLaunchKernelGGL((loss_dc_grad_kernel<20>), grid, dim3(256), 0, st, emb, one_hot, mag, gm, g_per_utt, TF, D, C, d_emb);
  else
    hipLaunchKernelGGL((loss_dc_grad_kernel<0>), grid, dim3(256), 0, st, emb, one_hot, mag, gm, g_per_utt, TF, D, C, d_emb);
  ONSSEN_LAUNCH_CHECK();
  return ONSSEN_OK;
}

size_t onssen_batch_sdr_workspace_bytes(int B) {
  return B > 0 ? ((size_t)B * sdr::NBLK * sdr::PSTRIDE + (size_t)B * sdr::SMAX) * sizeof(double) : 0;
}

static int batch_sdr_impl(const float* est, const float* org, const float* mask, int B, int C, int n, float* sdr_out,
                          int* perm_out, void* ws, size_t ws_bytes, void* stream, const int32_t* lengths) {
  if (!est || !org || !sdr_out || !ws || B <= 0 || C <= 0 || C > sdr::CMAX || n <= C) return ONSSEN_E_ARG;
  if (ws_bytes < onssen_batch_sdr_workspace_bytes(B)) return ONSSEN_E_WORKSPACE;
  ONSSEN_CLEAR_ERROR();
  hipStream_t st = (hipStream_t)stream;
  if ((reinterpret_cast<uintptr_t>(ws) & 7u) != 0) return ONSSEN_E_ALIGN;
  double* partial = (double*)ws;      // fp64 sums: see loss_sdr.inc
  double* means = partial + (size_t)B * sdr::NBLK * sdr::PSTRIDE;
  const dim3 grid(sdr::NBLK, (unsigned)B);
  hipLaunchKernelGGL((sdr_partial_kernel<0>), grid, dim3(256), 0, st, est, org, mask, C, n, (const double*)nullptr, partial, lengths);
  hipLaunchKernelGGL(sdr_means_kernel, dim3((unsigned)B), dim3(64), 0, st, (const double*)partial, C, n, means, lengths);
  hipLaunchKernelGGL((sdr_partial_kernel<1>), grid, dim3(256), 0, st, est, org, mask, C, n, (const double*)means, partial, lengths);
  hipLaunchKernelGGL(sdr_final_kernel, dim3((unsigned)B), dim3(64), 0, st, (const double*)partial, C, sdr_out, perm_out);
  ONSSEN_LAUNCH_CHECK();
  return ONSSEN_OK;
}


int onssen_batch_sdr_f32(const float* est, const float* org, const float* mask, int B, int C, int n, float* sdr_out,
                         int* perm_out, void* ws, size_t ws_bytes, void* stream) {
  return batch_sdr_impl(est, org, mask, B, C, n, sdr_out, perm_out, ws, ws_bytes, stream, nullptr);
}

int onssen_batch_sdr_ragged_f32(const float* est, const float* org, const float* mask, int B, int C, int n,
                                const int32_t* lengths, float* sdr_out, int* perm_out, void* ws, size_t ws_bytes,
                                void* stream) {
  if (!lengths) return ONSSEN_E_ARG;
  return batch_sdr_impl(est, org, mask, B, C, n, sdr_out, perm_out, ws, ws_bytes, stream, lengths);
}

size_t onssen_loss_mask_workspace_bytes(int B) { return B > 0 ? (size_t)B * 32 * 4 * sizeof(float) : 0; }

int onssen_loss_mask_f32(const float* mask_a, const float* mask_b, int64_t m_sb, int64_t m_se, const float* mag_mix,
                         const float* mag_s1, const float* mag_s2, const float* cos_s1, const float* cos_s2, int B, int TF,
                         float* out, int32_t* perm_out, void* ws, size_t ws_bytes, void* stream) {
  if (!mask_a || !mask_b || !mag_mix || !mag_s1 || !mag_s2 || !out || !ws || B <= 0 || TF <= 0 ||
      ((cos_s1 == nullptr) != (cos_s2 == nullptr)))
    return ONSSEN_E_ARG;
  if (ws_bytes < onssen_loss_mask_workspace_bytes(B)) return ONSSEN_E_WORKSPACE;
  ONSSEN_CLEAR_ERROR();
  hipStream_t st = (hipStream_t)stream;
  hipLaunchKernelGGL(loss_mask_kernel, dim3(32, (unsigned)B), dim3(256), 0, st, mask_a, mask_b, (long)m_sb, (long)m_se, mag_mix,
                     mag_s1, mag_s2, cos_s1, cos_s2, TF, (float*)ws);
  hipLaunchKernelGGL(loss_mask_final_kernel, dim3((unsigned)B), dim3(64), 0, st, (const float*)ws, 32, out, (int*)perm_out);
  ONSSEN_LAUNCH_CHECK();
  return ONSSEN_OK;
}

int onssen_loss_mask_grad_f32(const float* mask_a, const float* mask_b, int64_t m_sb, int64_t m_se, const float* mag_mix,
                              const float* mag_s1, const float* mag_s2, const float* cos_s1, const float* cos_s2, int B, int TF,
                              const float* g, const int32_t* perm, float* d_mask_a, float* d_mask_b, int64_t d_sb, int64_t d_se,
                              void* stream) {
  if (!mask_a || !mask_b || !mag_mix || !mag_s1 || !mag_s2 || !g || !perm || !d_mask_a || !d_mask_b || B <= 0 || TF <= 0 ||
      ((cos_s1 == nullptr) != (cos_s2 == nullptr)))
    return ONSSEN_E_ARG;
  ONSSEN_CLEAR_ERROR();
  const int nblk = ceil_div(TF, 256) < 64 ? ceil_div(TF, 256) : 64;
  hipLaunchKernelGGL(loss_mask_grad_kernel, dim3((unsigned)nblk, (unsigned)B), dim3(256), 0, (hipStream_t)stream, mask_a, mask_b,
                     (long)m_sb, (long)m_se, mag_mix, mag_s1, mag_s2, cos_s1, cos_s2, TF, g, (const int*)perm, d_mask_a, d_mask_b,
                     (long)d_sb, (long)d_se);
  ONSSEN_LAUNCH_CHECK();
  return ONSSEN_OK;
}


// workspace layout: header | G | ybuf (L > 1) | c | h hand-off image | x3 images: layer-0 input, output A (the
// LAST layer's), output B (L > 1) | 64 KiB debug
struct BlstmWs {
  size_t g, y, c, hs, img_x, img_y, off_imgx, off_imga, off_imgb, total;
};
static bool blstm_ws_layout(int B, int T, int in_dim, int H, int L, int ug, BlstmWs* w) {
  int Hp, NP, KQ;
  if (onssen_lstm_geometry(H, ug, &Hp, &NP, &KQ, nullptr) != ONSSEN_OK || B <= 0 || T <= 0 || L <= 0 || in_dim <= 0) return false;
  w->g = align256((size_t)T * B * 2 * NP * sizeof(float));
  w->y = L > 1 ? align256((size_t)T * B * 2 * Hp * sizeof(float)) : 0;
  w->c = align256((size_t)2 * B * Hp * sizeof(float));
  w->hs = align256((size_t)2 * 2 * ceil_div(B, 4) * ceil_div(Hp, 32) * 2048);    // h hand-off image: one per (direction, group of >= 4 rows)
  w->img_x = align256((size_t)T * B * ceil_div(in_dim, 32) * 128);
  w->img_y = align256((size_t)T * B * ceil_div(2 * Hp, 32) * 128);
  w->off_imgx = ONSSEN_BLSTM_WS_HEADER_BYTES + w->g + w->y + w->c + w->hs;
  w->off_imga = w->off_imgx + w->img_x;
  w->off_imgb = w->off_imga + w->img_y;
  w->total = w->off_imgb + (L > 1 ? w->img_y : 0) + 65536;   // the last 64 KiB: debug timestamps
  return true;
}

size_t onssen_blstm_workspace_bytes(int B, int T, int in_dim, int H, int L, int ug) {
  BlstmWs w;
  return blstm_ws_layout(B, T, in_dim, H, L, ug, &w) ? w.total : 0;
}

int onssen_blstm_y_image(int B, int T, int in_dim, int H, int L, int ug, size_t* offset_bytes, int* KB) {
  BlstmWs w;
  int Hp;
  if (!blstm_ws_layout(B, T, in_dim, H, L, ug, &w) || onssen_lstm_geometry(H, ug, &Hp, nullptr, nullptr, nullptr) != ONSSEN_OK)
    return ONSSEN_E_ARG;
  if (offset_bytes) *offset_bytes = w.off_imga;
  if (KB) *KB = ceil_div(2 * Hp, 32);
  return ONSSEN_OK;
}

int onssen_blstm_x_image(int B, int T, int in_dim, int H, int L, int ug, size_t* offset_bytes, int* KB) {
  BlstmWs w;
  if (!blstm_ws_layout(B, T, in_dim, H, L, ug, &w)) return ONSSEN_E_ARG;
  if (offset_bytes) *offset_bytes = w.off_imgx;
  if (KB) *KB = ceil_div(in_dim, 32);
  return ONSSEN_OK;
}

// ONSSEN_BLSTM_WS_DIRTY: the k padding of a recurrence output image (columns 2*Hp .. 32*KB - 1 of every row) is never written by
// the recurrence; a workspace that was not zeroed for this shape gets it cleared here (stale bits there could be bf16 NaNs,
// and NaN x 0-weight = NaN in the next GEMM)
__global__ void x3_pad_zero_kernel(unsigned short* __restrict__ img, long rows, int KB, int K) {
  const int k0 = K & 31;                         // first padding column inside the last k block (0: no padding)
  if (k0 == 0) return;
  const int per = 32 - k0;
  const long total = rows * 2 * per;
  for (long e = (long)blockIdx.x * blockDim.x + threadIdx.x; e < total; e += (long)gridDim.x * blockDim.x) {
    const long row = e / (2 * per);
    const int r = (int)(e - row * 2 * per), hl = r / per, kk = k0 + r % per;
    img[(row * KB + (KB - 1)) * 64 + hl * 32 + kk] = 0;
  }
}

static int blstm_forward_impl(const float* x, int64_t xs_b, int64_t xs_t, int B, int T, int in_dim, int H, int L,
                              int ug, const float* const* wih_p_host, const float* const* whh_p_host,
                              const float* const* bias_p_host, float* y, void* ws, size_t ws_bytes, int flags,
                              void* stream, float* save_g, float* save_c, const int32_t* frames = nullptr) {
  int Hp, NP, KQ;
  int64_t we;
  if (onssen_lstm_geometry(H, ug, &Hp, &NP, &KQ, &we) != ONSSEN_OK) return ONSSEN_E_ARG;
  if (!x || !ws || !wih_p_host || !whh_p_host || !bias_p_host || B <= 0 || T <= 0 || in_dim <= 0 || L <= 0)
    return ONSSEN_E_ARG;
  // y may be NULL only in the XCD form, whose consumers can take the x3 image of the output instead
  if (!y && !((flags & ONSSEN_BLSTM_XCD) && (flags & ONSSEN_BLSTM_BF16X3))) return ONSSEN_E_ARG;
  // ragged batches: the persistent form exists for the plain split-bf16 inference recurrence (no fused first layer, no
  // bf16-only products, no saved state); the launch-per-step form takes them in both precisions
  if (frames && (flags & ONSSEN_BLSTM_XCD) &&
      (!(flags & ONSSEN_BLSTM_BF16X3) || (flags & (ONSSEN_BLSTM_FUSE_IN0 | ONSSEN_BLSTM_BF16)) || save_g || save_c))
    return ONSSEN_E_ARG;
  BlstmWs wl;
  if (!blstm_ws_layout(B, T, in_dim, H, L, ug, &wl)) return ONSSEN_E_ARG;
  if (ws_bytes < wl.total) return ONSSEN_E_WORKSPACE;
  if ((reinterpret_cast<uintptr_t>(ws) & 255u) != 0 || (y && !aligned16(y))) return ONSSEN_E_ALIGN;
  hipStream_t st = (hipStream_t)stream;
  char* wsp = (char*)ws;
  unsigned* syncw = (unsigned*)wsp;
  wsp += ONSSEN_BLSTM_WS_HEADER_BYTES;
  float* G = (float*)wsp;
  wsp += align256((size_t)T * B * 2 * NP * sizeof(float));
  float* ybuf = nullptr;
  if (L > 1) {
    ybuf = (float*)wsp;
    wsp += align256((size_t)T * B * 2 * Hp * sizeof(float));
  }
  float* cst = (float*)wsp;
  wsp += align256((size_t)2 * B * Hp * sizeof(float));
  const bool x3 = (flags & ONSSEN_BLSTM_BF16X3) != 0;
  int KQ2 = 0, Hs = 0;
  onssen_lstm_geometry_x3(H, ug, &KQ2, &Hs, nullptr);
  if (x3 && !(flags & ONSSEN_BLSTM_XCD) && KQ2 > 4 * rec::QB3) return ONSSEN_E_ARG;   // H <= 640 in the launch-per-step split-bf16 form
  uint16_t* hsb = (uint16_t*)wsp;
  const size_t hs_bytes = (size_t)2 * 2 * ceil_div(B, 4) * KQ2 * 2048;   // split-bf16 images of all groups; >= the fp32 image (2*KQ2 >= KQ)
  wsp += align256(hs_bytes);
  long long* dbg = ((flags >> 8) & 32) && T * 8 * sizeof(long long) <= 65536 ? (long long*)((char*)ws + wl.total - 65536) : nullptr;
  if (!(flags & ONSSEN_BLSTM_XCD)) {   // launch-per-step form: h_{-1} = 0 and the K padding of the hand-off images come from here
    hipError_t e = hipMemsetAsync(hsb, 0, hs_bytes, st);      // (the persistent kernels clear their own slots, K padding included,
    if (e != hipSuccess) return (int)e;                        //  before their start-up barrier: one launch less per call)
  }
  const int mt = (B > 16 && !(flags & ONSSEN_BLSTM_SPLIT_ROWS)) ? 2 : 1;
  // XCD form: activations travel between the layers (and on to the heads) as x3 images written by the recurrence
  // epilogue; wih_p_host[l] is then the x3 image of the [2*NP][K_l] input-projection matrix
  const bool images = x3 && (flags & ONSSEN_BLSTM_XCD);
  const bool bf16_only = images && (flags & ONSSEN_BLSTM_BF16);   // plain bf16 products instead of the three-term split
  uint16_t* img_x = (uint16_t*)((char*)ws + wl.off_imgx);
  uint16_t* img_ab[2] = {(uint16_t*)((char*)ws + wl.off_imga), (uint16_t*)((char*)ws + wl.off_imgb)};
  if (images && (flags & ONSSEN_BLSTM_WS_DIRTY) && ((2 * Hp) & 31)) {
    const long rows = (long)T * B;
    const long n = rows * 2 * (32 - ((2 * Hp) & 31));
    const unsigned nb = (unsigned)((n + 255) / 256 > 4096 ? 4096 : (n + 255) / 256);
    for (int i = 0; i < (L > 1 ? 2 : 1); ++i)
      hipLaunchKernelGGL(x3_pad_zero_kernel, dim3(nb), dim3(256), 0, st, img_ab[i], rows, ceil_div(2 * Hp, 32), 2 * Hp);
  }
  for (int l = 0; l < L; ++l) {
    // the last layer writes `y`; the layers before it alternate so that each reads what the previous wrote
    float* yout = ((L - 1 - l) % 2 == 0) ? y : ybuf;
    const float* yin = ((L - 1 - l) % 2 == 0) ? ybuf : y;
    int rc;
    const bool fuse0 = images && l == 0 && (flags & ONSSEN_BLSTM_FUSE_IN0);
    // the fused projection keeps <= 4 k-chunks of W_ih fragments in the LDS: in_dim <= 128, or 32k + 1 <= 129 with FUSE_TAIL
    if (fuse0 && !(in_dim <= 128 || (in_dim == 129 && (flags & ONSSEN_BLSTM_FUSE_TAIL)))) return ONSSEN_E_ARG;
    if ((flags & ONSSEN_BLSTM_G_READY) && !fuse0) {
      rc = ONSSEN_OK;              // profiling: G of this layer is what an earlier call left in the workspace
    } else if (images) {
      const uint16_t* a_img = l == 0 ? img_x : img_ab[(L - l) % 2];   // layer l-1 wrote buffer (L-1-(l-1)) % 2
      if (l == 0) {
        rc = onssen_x3_image_f32(x, xs_t, xs_b, B, T * B, in_dim, img_x, stream);
        if (rc != ONSSEN_OK) return rc;
      }
      if (fuse0) rc = ONSSEN_OK;   // x_t W_ih^T is computed inside the recurrence launch: no G, no GEMM
      else rc = onssen_linear_x3p(a_img, T * B, l == 0 ? in_dim : 2 * Hp, (const uint16_t*)wih_p_host[l], bias_p_host[l],
                             2 * NP, ONSSEN_EPI_BIAS | (bf16_only ? ONSSEN_EPI_BF16 : 0), 0, 0.f, G, B, (int64_t)B * 2 * NP, 2 * NP, stream);
    } else if (x3) {   // wih_p_host[l]: split-bf16 planes [2][2*NP][ld], ld = K rounded up to 32
      const int K = l == 0 ? in_dim : 2 * Hp, ld = ceil_div(K, 32) * 32;
      rc = onssen_linear_bf16x3(l == 0 ? x : yin, l == 0 ? xs_t : (int64_t)B * 2 * Hp, l == 0 ? xs_b : 2 * Hp, B, T * B,
                                K, (const uint16_t*)wih_p_host[l], ld, bias_p_host[l], 2 * NP, ONSSEN_EPI_BIAS, 0, 0.f,
                                nullptr, G, (int64_t)B * 2 * NP, 2 * NP, stream);
    } else if (l == 0) {
      const int Kp = ceil_div(in_dim, 4) * 4;
      rc = onssen_linear_f32(x, xs_t, xs_b, B, T * B, in_dim, wih_p_host[0], Kp, bias_p_host[0], 2 * NP,
                             ONSSEN_EPI_BIAS, 0, 0.f, nullptr, G, (int64_t)B * 2 * NP, 2 * NP, stream);
    } else {
      rc = onssen_linear_f32(yin, (int64_t)B * 2 * Hp, 2 * Hp, B, T * B, 2 * Hp, wih_p_host[l], 2 * Hp,
                             bias_p_host[l], 2 * NP, ONSSEN_EPI_BIAS, 0, 0.f, nullptr, G, (int64_t)B * 2 * NP,
                             2 * NP, stream);
    }
    if (rc != ONSSEN_OK) return rc;
    if (flags & ONSSEN_BLSTM_XCD) {
      // without ONSSEN_BLSTM_BF16X3: the exact-fp32 instantiation (whh_p_host[l] = the fp32 fragment image of
      // onssen_lstm_pack_f32, G from the exact-fp32 GEMM above, fp32 rows between the layers)
      // split-bf16: H <= 768 (ug = 24: 32 members of 24 units = every CU of an XCD; round 4), exact fp32 and the forms that fuse
      // the first layer / save state: H <= 640 (ug <= 20)
      if (ug > 24 || Hp / ug > 32 || KQ2 > 24 || (!x3 && (save_g || save_c))) return ONSSEN_E_ARG;
      if (ug > 20 && (!x3 || fuse0 || (flags & ONSSEN_BLSTM_BF16))) return ONSSEN_E_ARG;
      // bounded waits: ~0.2 s of polling on the GPU; ONSSEN_XCD_SPIN_LIMIT overrides (the host-side emulation, where a
      // 'workgroup' is a process at the mercy of the OS scheduler, raises it)
      const unsigned xcd_spin = xcd_spin_limit();
      XcdArgs xa;
      // fp32 rows only where somebody reads them (the caller's y); every layer leaves its x3 image
      xa.G = G; xa.whh = (const unsigned short*)whh_p_host[l]; xa.y = x3 ? (l == L - 1 ? y : nullptr) : yout; xa.hx = hsb; xa.sync = syncw; xa.B = B;
      xa.yimg = x3 ? img_ab[(L - 1 - l) % 2] : nullptr; xa.KBI = ceil_div(2 * Hp, 32);
      xa.wih0 = fuse0 ? (const unsigned short*)wih_p_host[0] : nullptr; xa.ximg = img_x; xa.bias0 = bias_p_host[0];
      xa.KC0 = fuse0 ? ceil_div(in_dim, 32) : 0;
      // in_dim = 32k + 1 (F = 129): the lone last column goes to the VALU; its weights follow the bias (FUSE_TAIL)
      const bool vtail = fuse0 && (flags & ONSSEN_BLSTM_FUSE_TAIL) && (in_dim % 32) == 1 && in_dim > 1;
      xa.KCM = vtail ? xa.KC0 - 1 : xa.KC0; xa.x0 = x; xa.xs_b = (long)xs_b; xa.xs_t = (long)xs_t;
      xa.wtail = vtail ? bias_p_host[0] + 2 * NP : nullptr;
      xa.T = T; xa.Hp = Hp; xa.NP = NP; xa.KQ2 = KQ2; xa.NU = Hp / ug; xa.row0 = 0; xa.nbg = 0; xa.spin_limit = xcd_spin; xa.dbg = dbg; xa.ablate = (flags >> 8) & 8;
      xa.terms = !x3 ? 0 : bf16_only ? 1 : 3;
      xa.save_g = save_g; xa.save_c = save_c;
      xa.frames = frames;
      ONSSEN_CLEAR_ERROR();
      // waves per workgroup (K is split over them): 8 = two per SIMD; ONSSEN_XCD_WAVES=4 keeps the one-per-SIMD form for comparison
      static const int xcd_nw = ONSSEN_KNOB_INT("ONSSEN_XCD_WAVES", 8) == 4 ? 4 : 8;
      switch (ug) {
        case 4: rc = launch_xcd<1>(xa, xcd_nw, st); break;
        case 8: rc = launch_xcd<2>(xa, xcd_nw, st); break;
        case 12: rc = launch_xcd<3>(xa, xcd_nw, st); break;
        case 16: rc = launch_xcd<4>(xa, xcd_nw, st); break;
        case 20: rc = launch_xcd<5>(xa, xcd_nw, st); break;
        default: rc = launch_xcd_wide(xa, st); break;      // ug = 24: 640 < H <= 768
      }
      if (rc != ONSSEN_OK) return rc;
      continue;
    }
    StepArgs sp;
    sp.G = G; sp.whh = x3 ? nullptr : whh_p_host[l]; sp.whh_x3 = x3 ? (const unsigned short*)whh_p_host[l] : nullptr;
    sp.hs = hsb; sp.KQ2 = KQ2; sp.Hs = Hs; sp.dbg = dbg; sp.y = yout; sp.c = cst; sp.B = B; sp.T = T; sp.Hp = Hp; sp.NP = NP;
    sp.KQ = KQ; sp.NU = Hp / ug; sp.step = 0; sp.ablate = (flags >> 8) & 63; sp.frames = frames;
    sp.save_g = save_g; sp.save_c = save_c;
#define ONSSEN_STEPS(MT_, NT_) rc = launch_steps<MT_, NT_>(sp, (char*)ws, T, x3, st)
    if (mt == 1) {
      switch (ug) {
        case 4: ONSSEN_STEPS(1, 1); break;
        case 8: ONSSEN_STEPS(1, 2); break;
        case 12: ONSSEN_STEPS(1, 3); break;
        case 16: ONSSEN_STEPS(1, 4); break;
        default: ONSSEN_STEPS(1, 5); break;
      }
    } else {
      switch (ug) {
        case 4: ONSSEN_STEPS(2, 1); break;
        case 8: ONSSEN_STEPS(2, 2); break;
        case 12: ONSSEN_STEPS(2, 3); break;
        case 16: ONSSEN_STEPS(2, 4); break;
        default: ONSSEN_STEPS(2, 5); break;
      }
    }
#undef ONSSEN_STEPS
    if (rc != ONSSEN_OK) return rc;
  }
  return ONSSEN_OK;
}

int onssen_blstm_forward_f32(const float* x, int64_t xs_b, int64_t xs_t, int B, int T, int in_dim, int H, int L,
                             int ug, const float* const* wih_p_host, const float* const* whh_p_host,
                             const float* const* bias_p_host, float* y, void* ws, size_t ws_bytes, int flags,
                             void* stream) {
  return blstm_forward_impl(x, xs_b, xs_t, B, T, in_dim, H, L, ug, wih_p_host, whh_p_host, bias_p_host, y, ws, ws_bytes,
                            flags, stream, nullptr, nullptr);
}

int onssen_blstm_forward_ragged_f32(const float* x, int64_t xs_b, int64_t xs_t, int B, int T, const int32_t* frames,
                                    int in_dim, int H, int L, int ug, const float* const* wih_p_host,
                                    const float* const* whh_p_host, const float* const* bias_p_host, float* y, void* ws,
                                    size_t ws_bytes, int flags, void* stream) {
  if (!frames) return ONSSEN_E_ARG;
  return blstm_forward_impl(x, xs_b, xs_t, B, T, in_dim, H, L, ug, wih_p_host, whh_p_host, bias_p_host, y, ws, ws_bytes,
                            flags, stream, nullptr, nullptr, frames);
}

// ---- two-layer stack, software-pipelined over consecutive calls (round 6) -------------------------------------------
// workspace: header | G0 | G1 | h hand-off images of 8 groups | x3 images: input, layer-0 output, layer-1 output | 64 KiB debug
struct Pipe2Ws {
  size_t g, hs, img_x, img_y, off_g0, off_g1, off_hs, off_imgx, off_img0, off_img1, total;
};
static bool pipe2_ws_layout(int B, int T, int in_dim, int H, int ug, Pipe2Ws* w) {
  int Hp, NP, KQ2 = 0, Hs = 0;
  if (onssen_lstm_geometry(H, ug, &Hp, &NP, nullptr, nullptr) != ONSSEN_OK || B <= 0 || B > 32 || T <= 0 || in_dim <= 0) return false;
  if (onssen_lstm_geometry_x3(H, ug, &KQ2, &Hs, nullptr) != ONSSEN_OK) return false;
  w->g = align256((size_t)T * B * 2 * NP * sizeof(float));
  w->hs = align256((size_t)8 * 2 * KQ2 * 2048);                 // 8 groups x 2 slots x KQ2 chunks of 2 KiB
  w->img_x = align256((size_t)T * B * ceil_div(in_dim, 32) * 128);
  w->img_y = align256((size_t)T * B * ceil_div(2 * Hp, 32) * 128);
  w->off_g0 = ONSSEN_BLSTM_WS_HEADER_BYTES;
  w->off_g1 = w->off_g0 + w->g;
  w->off_hs = w->off_g1 + w->g;
  w->off_imgx = w->off_hs + w->hs;
  w->off_img0 = w->off_imgx + w->img_x;
  w->off_img1 = w->off_img0 + w->img_y;
  w->total = w->off_img1 + w->img_y + 65536;
  return true;
}

size_t onssen_blstm_pipe2_workspace_bytes(int B, int T, int in_dim, int H, int ug) {
  Pipe2Ws w;
  return pipe2_ws_layout(B, T, in_dim, H, ug, &w) ? w.total : 0;
}

int onssen_blstm_pipe2_y_image(int B, int T, int in_dim, int H, int ug, size_t* offset_bytes, int* KB) {
  Pipe2Ws w;
  int Hp;
  if (!pipe2_ws_layout(B, T, in_dim, H, ug, &w) || onssen_lstm_geometry(H, ug, &Hp, nullptr, nullptr, nullptr) != ONSSEN_OK)
    return ONSSEN_E_ARG;
  if (offset_bytes) *offset_bytes = w.off_img1;
  if (KB) *KB = ceil_div(2 * Hp, 32);
  return ONSSEN_OK;
}

// T_cap lays the workspace out (>= every T that passes through it); the uniform form has T_cap = T = T_prev and no frames
static int blstm_pipe2_impl(const float* x, int64_t xs_b, int64_t xs_t, int B, int T_cap, int T, const int32_t* frames, int T_prev,
                            const int32_t* frames_prev, int in_dim, int H, int ug, const float* const* wih_p_host,
                            const float* const* whh_p_host, const float* const* bias_p_host, void* ws, size_t ws_bytes, int flags,
                            void* stream) {
  int Hp, NP, KQ2 = 0, Hs = 0;
  if (onssen_lstm_geometry(H, ug, &Hp, &NP, nullptr, nullptr) != ONSSEN_OK) return ONSSEN_E_ARG;
  if (!x || !ws || !wih_p_host || !whh_p_host || !bias_p_host || B <= 0 || B > 32 || T <= 0 || in_dim <= 0) return ONSSEN_E_ARG;
  if (T > T_cap || T_prev <= 0 || T_prev > T_cap) return ONSSEN_E_ARG;
  // ragged rows keep the bits of their own batch-1 run: stacked tiles only, i.e. <= 16 rows; both batches bring their frames
  if ((frames != nullptr) != (frames_prev != nullptr) || (frames && B > 16)) return ONSSEN_E_ARG;
  // the plain split-bf16 persistent recurrence only (no fused first layer, no bf16-only products)
  if ((flags & 0xff & ~ONSSEN_BLSTM_G_READY) != (ONSSEN_BLSTM_BF16X3 | ONSSEN_BLSTM_XCD)) return ONSSEN_E_ARG;
  const bool g_ready = (flags & ONSSEN_BLSTM_G_READY) != 0;     // measurement aid: the pair launch by itself, on the projections an earlier call left
  Pipe2Ws wl;
  if (!pipe2_ws_layout(B, T_cap, in_dim, H, ug, &wl)) return ONSSEN_E_ARG;
  if (ws_bytes < wl.total) return ONSSEN_E_WORKSPACE;
  if ((reinterpret_cast<uintptr_t>(ws) & 255u) != 0) return ONSSEN_E_ALIGN;
  onssen_lstm_geometry_x3(H, ug, &KQ2, &Hs, nullptr);
  if (ug > 20 || Hp / ug > 32 || KQ2 > 24) return ONSSEN_E_ARG;
  hipStream_t st = (hipStream_t)stream;
  char* base = (char*)ws;
  float* G0 = (float*)(base + wl.off_g0);
  float* G1 = (float*)(base + wl.off_g1);
  uint16_t* img_x = (uint16_t*)(base + wl.off_imgx);
  uint16_t* img0 = (uint16_t*)(base + wl.off_img0);
  uint16_t* img1 = (uint16_t*)(base + wl.off_img1);
  // layer 0 of THIS batch: input image, input projection
  int rc = ONSSEN_OK;
  if (!g_ready) {
    rc = onssen_x3_image_f32(x, xs_t, xs_b, B, T * B, in_dim, img_x, stream);
    if (rc != ONSSEN_OK) return rc;
    rc = onssen_linear_x3p(img_x, T * B, in_dim, (const uint16_t*)wih_p_host[0], bias_p_host[0], 2 * NP, ONSSEN_EPI_BIAS, 0, 0.f, G0, B,
                           (int64_t)B * 2 * NP, 2 * NP, stream);
    if (rc != ONSSEN_OK) return rc;
  }
  // ONE launch: layer 1 of the batch before (its G1 was left by the call before) beside layer 0 of this one
  XcdArgs xa;
  xa.G = G1; xa.whh = (const unsigned short*)whh_p_host[1]; xa.y = nullptr; xa.yimg = img1;
  xa.G_b = G0; xa.whh_b = (const unsigned short*)whh_p_host[0]; xa.yimg_b = img0;
  xa.hx = (unsigned short*)(base + wl.off_hs); xa.sync = (unsigned*)base; xa.B = B; xa.KBI = ceil_div(2 * Hp, 32);
  xa.wih0 = nullptr; xa.ximg = img_x; xa.bias0 = bias_p_host[0]; xa.KC0 = 0; xa.KCM = 0; xa.x0 = x; xa.xs_b = (long)xs_b; xa.xs_t = (long)xs_t;
  xa.wtail = nullptr;
  xa.T = T_prev; xa.T_b = T; xa.frames = frames_prev; xa.frames_b = frames;
  xa.Hp = Hp; xa.NP = NP; xa.KQ2 = KQ2; xa.NU = Hp / ug; xa.row0 = 0; xa.nbg = 0; xa.spin_limit = xcd_spin_limit();
  xa.dbg = nullptr; xa.ablate = (flags >> 8) & 8; xa.terms = 3; xa.save_g = nullptr; xa.save_c = nullptr;
  ONSSEN_CLEAR_ERROR();
  switch (ug) {
    case 4: rc = launch_xcd_pair<1>(xa, st); break;
    case 8: rc = launch_xcd_pair<2>(xa, st); break;
    case 12: rc = launch_xcd_pair<3>(xa, st); break;
    case 16: rc = launch_xcd_pair<4>(xa, st); break;
    case 20: rc = launch_xcd_pair<5>(xa, st); break;
    default: return ONSSEN_E_ARG;
  }
  if (rc != ONSSEN_OK || g_ready) return rc;
  // layer 1's input projection of THIS batch, for the next call
  return onssen_linear_x3p(img0, T * B, 2 * Hp, (const uint16_t*)wih_p_host[1], bias_p_host[1], 2 * NP, ONSSEN_EPI_BIAS, 0, 0.f, G1, B,
                           (int64_t)B * 2 * NP, 2 * NP, stream);
}

int onssen_blstm_pipe2_forward_f32(const float* x, int64_t xs_b, int64_t xs_t, int B, int T, int in_dim, int H, int ug,
                                   const float* const* wih_p_host, const float* const* whh_p_host,
                                   const float* const* bias_p_host, void* ws, size_t ws_bytes, int flags, void* stream) {
  return blstm_pipe2_impl(x, xs_b, xs_t, B, T, T, nullptr, T, nullptr, in_dim, H, ug, wih_p_host, whh_p_host, bias_p_host, ws, ws_bytes,
                          flags, stream);
}

int onssen_blstm_pipe2_forward_ragged_f32(const float* x, int64_t xs_b, int64_t xs_t, int B, int T_cap, int T, const int32_t* frames,
                                          int T_prev, const int32_t* frames_prev, int in_dim, int H, int ug,
                                          const float* const* wih_p_host, const float* const* whh_p_host,
                                          const float* const* bias_p_host, void* ws, size_t ws_bytes, int flags, void* stream) {
  if (!frames || !frames_prev) return ONSSEN_E_ARG;
  return blstm_pipe2_impl(x, xs_b, xs_t, B, T_cap, T, frames, T_prev, frames_prev, in_dim, H, ug, wih_p_host, whh_p_host, bias_p_host, ws,
                          ws_bytes, flags, stream);
}

// ---- training: one layer forward with saved state, and its backward recurrence ----------------
int onssen_lstm_train_forward_f32(const float* x, int64_t xs_b, int64_t xs_t, int B, int T, int in_dim, int H, int ug,
                                  const uint16_t* wih_img, const uint16_t* whh_x3, const float* bias_p, float* y,
                                  float* gates, float* cs, void* ws, size_t ws_bytes, void* stream) {
  if (!y || !gates || !cs || !aligned16(gates)) return ONSSEN_E_ARG;
  const float* wih[1] = {(const float*)wih_img};
  const float* whh[1] = {(const float*)whh_x3};
  const float* bias[1] = {bias_p};
  return blstm_forward_impl(x, xs_b, xs_t, B, T, in_dim, H, 1, ug, wih, whh, bias, y, ws, ws_bytes,
                            ONSSEN_BLSTM_BF16X3 | ONSSEN_BLSTM_XCD, stream, gates, cs);
}

static bool lstm_bwd_geometry(int H, int ug, int* Hp, int* NP, int* KQB, int* NUB) {
  if (onssen_lstm_geometry(H, ug, Hp, NP, nullptr, nullptr) != ONSSEN_OK) return false;
  *KQB = ceil_div(*NP, 32);
  *NUB = ceil_div(*Hp, 16);
  return true;
}

int onssen_lstm_train_forward_form_f32(const float* x, int64_t xs_b, int64_t xs_t, int B, int T, int in_dim, int H, int ug,
                                       const void* wih, const void* whh, const float* bias_p, float* y, float* gates,
                                       float* cs, void* ws, size_t ws_bytes, int flags, void* stream) {
  if (!y || !gates || !cs || !aligned16(gates)) return ONSSEN_E_ARG;
  // the forms that can save state: the persistent split-bf16 launch, and the launch-per-step recurrence in either precision
  const int form = flags & (ONSSEN_BLSTM_XCD | ONSSEN_BLSTM_BF16X3);
  if ((flags & ~(ONSSEN_BLSTM_XCD | ONSSEN_BLSTM_BF16X3)) != 0 || form == ONSSEN_BLSTM_XCD) return ONSSEN_E_ARG;
  const float* wih_a[1] = {(const float*)wih};
  const float* whh_a[1] = {(const float*)whh};
  const float* bias[1] = {bias_p};
  return blstm_forward_impl(x, xs_b, xs_t, B, T, in_dim, H, 1, ug, wih_a, whh_a, bias, y, ws, ws_bytes, flags, stream, gates, cs);
}

int64_t onssen_lstm_whhT_elems(int H, int ug) {
  int Hp, NP, KQB, NUB;
  if (!lstm_bwd_geometry(H, ug, &Hp, &NP, &KQB, &NUB)) return 0;
  return (int64_t)NUB * KQB * 1024;
}

int onssen_lstm_pack_whhT_bf16x3(const float* w_hh, int H, int ug, uint16_t* out, void* stream) {
  int Hp, NP, KQB, NUB;
  if (!w_hh || !out || !lstm_bwd_geometry(H, ug, &Hp, &NP, &KQB, &NUB)) return ONSSEN_E_ARG;
  ONSSEN_CLEAR_ERROR();
  const long n = (long)NUB * KQB * 512;
  hipLaunchKernelGGL(pack_whhT_bf16x3_kernel, dim3((unsigned)((n + 255) / 256 > 4096 ? 4096 : (n + 255) / 256)), dim3(256), 0,
                     (hipStream_t)stream, w_hh, H, Hp, ug, KQB, NUB, out);
  ONSSEN_LAUNCH_CHECK();
  return ONSSEN_OK;
}

int64_t onssen_lstm_whhR_elems(int H, int ug) {
  int Hp, NP, KQB, NUB;
  if (!lstm_bwd_geometry(H, ug, &Hp, &NP, &KQB, &NUB)) return 0;
  return (int64_t)(Hp / ug) * ceil_div(4 * ug, 32) * NUB * 1024;
}

int onssen_lstm_pack_whhR_bf16x3(const float* w_hh, int H, int ug, uint16_t* out, void* stream) {
  int Hp, NP, KQB, NUB;
  if (!w_hh || !out || !lstm_bwd_geometry(H, ug, &Hp, &NP, &KQB, &NUB)) return ONSSEN_E_ARG;
  if (!aligned16(out)) return ONSSEN_E_ALIGN;
  ONSSEN_CLEAR_ERROR();
  const int KC = ceil_div(4 * ug, 32);
  const long n = (long)(Hp / ug) * KC * NUB * 64;
  hipLaunchKernelGGL(pack_whhR_bf16x3_kernel, dim3((unsigned)((n + 255) / 256 > 4096 ? 4096 : (n + 255) / 256)), dim3(256), 0,
                     (hipStream_t)stream, w_hh, H, Hp, ug, KC, NUB, out);
  ONSSEN_LAUNCH_CHECK();
  return ONSSEN_OK;
}

int onssen_lstm_pack_train_f32(int L, int in_dim, int H, int ug, const float* const* w_ih_host, const float* const* w_hh_host,
                               const float* const* b_ih_host, const float* const* b_hh_host, float* const* wih_p_host,
                               float* const* bias_p_host, uint16_t* const* wih_img_host, uint16_t* const* whh_x3_host,
                               uint16_t* const* whhR_host, void* stream) {
  int Hp, NP, KQB, NUB, KQ2;
  if (L <= 0 || in_dim <= 0 || !w_ih_host || !w_hh_host || !b_ih_host || !b_hh_host || !wih_p_host || !bias_p_host || !wih_img_host ||
      !whh_x3_host || !lstm_bwd_geometry(H, ug, &Hp, &NP, &KQB, &NUB) || onssen_lstm_geometry_x3(H, ug, &KQ2, nullptr, nullptr) != ONSSEN_OK)
    return ONSSEN_E_ARG;
  for (int i = 0; i < 2 * L; ++i)
    if (!w_ih_host[i] || !w_hh_host[i] || !b_ih_host[i] || !b_hh_host[i] || !wih_p_host[i] || !bias_p_host[i] || !wih_img_host[i] ||
        !whh_x3_host[i] || !aligned16(wih_img_host[i]) || !aligned16(whh_x3_host[i]) ||
        (whhR_host && (!whhR_host[i] || !aligned16(whhR_host[i]))))
      return ONSSEN_E_ARG;
  ONSSEN_CLEAR_ERROR();
  const int per = whhR_host ? 3 : 2, group = packt::MAXJ / per;       // (layer, direction) pairs per launch
  auto blocks = [](long n) { const long b = (n + 255) / 256; return (int)(b > 2048 ? 2048 : b); };
  for (int i0 = 0; i0 < 2 * L; i0 += group) {
    PackTrainArgs a;
    a.H = H; a.Hp = Hp; a.UG = ug; a.KQ2 = KQ2; a.KC = ceil_div(4 * ug, 32); a.NTB = NUB; a.njobs = 0;
    int fb = 0;
    for (int i = i0; i < 2 * L && i < i0 + group; ++i) {
      const int l = i / 2, bidir = l > 0;
      const int ind = bidir ? 2 * H : in_dim, Kp = bidir ? 2 * Hp : ceil_div(in_dim, 4) * 4, K = bidir ? 2 * Hp : in_dim, KB = ceil_div(K, 32);
      PackJob j{};
      j.w = w_ih_host[i]; j.b_ih = b_ih_host[i]; j.b_hh = b_hh_host[i]; j.wih_p = wih_p_host[i]; j.bias_p = bias_p_host[i];
      j.out = wih_img_host[i]; j.type = 0; j.in_dim = ind; j.bidir = bidir; j.Kp = Kp; j.K = K; j.KB = KB;
      j.first_block = fb; j.nblocks = blocks((long)NP * KB * 4); fb += j.nblocks;
      a.job[a.njobs++] = j;
      PackJob h{};
      h.w = w_hh_host[i]; h.out = whh_x3_host[i]; h.type = 1;
      h.first_block = fb; h.nblocks = blocks((long)(Hp / ug) * KQ2 * (ug / 4) * 64); fb += h.nblocks;
      a.job[a.njobs++] = h;
      if (whhR_host) {
        PackJob r{};
        r.w = w_hh_host[i]; r.out = whhR_host[i]; r.type = 2;
        r.first_block = fb; r.nblocks = blocks((long)(Hp / ug) * a.KC * NUB * 64); fb += r.nblocks;
        a.job[a.njobs++] = r;
      }
    }
    hipLaunchKernelGGL(lstm_pack_train_kernel, dim3((unsigned)fb), dim3(256), 0, (hipStream_t)stream, a);
  }
  ONSSEN_LAUNCH_CHECK();
  return ONSSEN_OK;
}

static int bwd_rows_per_group(int B) {
  int rg = B <= 16 ? 4 : B <= 32 ? 8 : 16;
  static const int rg_env = ONSSEN_KNOB_INT("ONSSEN_XCD_RG", 0);
  if (rg_env == 4 || rg_env == 8 || rg_env == 16) rg = rg_env;
  return rg;
}

size_t onssen_lstm_train_backward_workspace_bytes(int B, int H, int ug, int form) {
  int Hp, NP, KQB, NUB;
  if (B <= 0 || !lstm_bwd_geometry(H, ug, &Hp, &NP, &KQB, &NUB)) return 0;
  if (form == ONSSEN_LSTM_BWD_XCD) {
    const int NU = Hp / ug, RG = bwd_rows_per_group(B);
    return ONSSEN_BLSTM_WS_HEADER_BYTES + align256((size_t)8 * 2 * NU * NU * RG * ug * sizeof(float));
  }
  return align256((size_t)2 * 2 * ceil_div(B, 16) * KQB * 2048) + align256((size_t)2 * B * Hp * sizeof(float));
}

static int lstm_train_backward_impl(int B, int T, int H, int ug, const uint16_t* whh_img, const float* dy, float* gates_dp,
                                    const float* cs, void* ws, size_t ws_bytes, int form, float* db_rows, uint16_t* dp_img, void* stream);

int onssen_lstm_train_backward_f32(int B, int T, int H, int ug, const uint16_t* whh_img, const float* dy, float* gates_dp,
                                   const float* cs, void* ws, size_t ws_bytes, int form, float* db_rows, void* stream) {
  return lstm_train_backward_impl(B, T, H, ug, whh_img, dy, gates_dp, cs, ws, ws_bytes, form, db_rows, nullptr, stream);
}

int onssen_lstm_train_backward_img_f32(int B, int T, int H, int ug, const uint16_t* whh_img, const float* dy, const float* gates,
                                       const float* cs, void* ws, size_t ws_bytes, float* db_rows, uint16_t* dp_img, void* stream) {
  int Hp, NP;
  if (!dp_img || !aligned16(dp_img) || onssen_lstm_geometry(H, ug, &Hp, &NP, nullptr, nullptr) != ONSSEN_OK || (2 * NP) % 32 != 0)
    return ONSSEN_E_ARG;
  return lstm_train_backward_impl(B, T, H, ug, whh_img, dy, const_cast<float*>(gates), cs, ws, ws_bytes, ONSSEN_LSTM_BWD_XCD, db_rows,
                                  dp_img, stream);
}

static int lstm_train_backward_impl(int B, int T, int H, int ug, const uint16_t* whh_img, const float* dy, float* gates_dp,
                                    const float* cs, void* ws, size_t ws_bytes, int form, float* db_rows, uint16_t* dp_img, void* stream) {
  int Hp, NP, KQB, NUB;
  if (!whh_img || !dy || !gates_dp || !cs || !ws || B <= 0 || T <= 0 || !lstm_bwd_geometry(H, ug, &Hp, &NP, &KQB, &NUB) ||
      (form != ONSSEN_LSTM_BWD_STEPS && form != ONSSEN_LSTM_BWD_XCD) || (db_rows && form != ONSSEN_LSTM_BWD_XCD) ||
      (db_rows && !aligned16(db_rows)))
    return ONSSEN_E_ARG;
  if (ws_bytes < onssen_lstm_train_backward_workspace_bytes(B, H, ug, form)) return ONSSEN_E_WORKSPACE;
  if ((reinterpret_cast<uintptr_t>(ws) & 255u) != 0 || !aligned16(gates_dp)) return ONSSEN_E_ALIGN;
  hipStream_t st = (hipStream_t)stream;
  if (form == ONSSEN_LSTM_BWD_XCD) {
    if (Hp / ug > 32 || NUB > 40) return ONSSEN_E_ARG;
    const unsigned xcd_spin = xcd_spin_limit();
    static const int ablate_env = ONSSEN_KNOB_INT("ONSSEN_BWD_ABLATE", 0);
    static const int delay_env = ONSSEN_KNOB_INT("ONSSEN_BWD_DELAY", 0);
    static const bool bwd_unstacked = ONSSEN_KNOB_INT("ONSSEN_BWD_UNSTACKED", 0) != 0;      // debug builds: the three-term form of rounds 2-4, for A/B
    static const bool bwd_wide = ONSSEN_KNOB_INT("ONSSEN_BWD_WIDE", ONSSEN_BWD_WIDE) != 0;      // the wide poll of round 6 (lstm_bwd.inc: RGW)
    XcdBwdArgs xa;
    xa.gd = gates_dp; xa.cs = cs; xa.dy = dy; xa.wR = whh_img; xa.sync = (unsigned*)ws;
    xa.xch = (float*)((char*)ws + ONSSEN_BLSTM_WS_HEADER_BYTES);
    xa.B = B; xa.T = T; xa.Hp = Hp; xa.NP = NP; xa.NU = Hp / ug; xa.NTB = NUB; xa.RG = bwd_rows_per_group(B);
    xa.spin_limit = xcd_spin; xa.ablate = ablate_env; xa.delay = delay_env; xa.db_rows = db_rows; xa.dp_img = dp_img;
    // ONSSEN_XCD_PROFILE builds only (ONSSEN_BWD_DBG=1, tools/bwd_timeline.py): 8 timestamps per step of workgroup 0 in the tail of ws
    static const bool dbg_env = ONSSEN_KNOB_INT("ONSSEN_BWD_DBG", 0) != 0;
    xa.dbg = dbg_env && ws_bytes >= onssen_lstm_train_backward_workspace_bytes(B, H, ug, form) + (size_t)T * 64
                 ? (long long*)((char*)ws + onssen_lstm_train_backward_workspace_bytes(B, H, ug, form)) : nullptr;
    ONSSEN_CLEAR_ERROR();
    const dim3 grid((unsigned)(8 * xa.NU));
    const int E = xa.RG * ug, parts = 4 * E <= 320 ? 4 : 2 * E <= 320 ? 2 : 1;   // polling lanes per element (320 polling threads)
    for (int r0 = 0; r0 < B; r0 += 4 * xa.RG) {
      const int rows = B - r0 < 4 * xa.RG ? B - r0 : 4 * xa.RG;
      xa.row0 = r0;
      xa.nbg = ceil_div(rows, xa.RG);
#define ONSSEN_BWD_UG(UG_)                                                                                   \
  do {                                                                                                       \
    if (bwd_wide && !bwd_unstacked) {   /* round 6: the wide poll (16-byte loads, 16 lanes per unit) */       \
      if (xa.RG == 4) hipLaunchKernelGGL((lstm_xcd_bwd_kernel<UG_, 1, true, 4>), grid, dim3(512), 0, st, xa);  \
      else if (xa.RG == 8) hipLaunchKernelGGL((lstm_xcd_bwd_kernel<UG_, 1, true, 8>), grid, dim3(512), 0, st, xa); \
      else hipLaunchKernelGGL((lstm_xcd_bwd_kernel<UG_, 1, false, 16>), grid, dim3(512), 0, st, xa);          \
    } else if (xa.RG <= 8 && !bwd_unstacked) {                                                               \
      if (parts == 4) hipLaunchKernelGGL((lstm_xcd_bwd_kernel<UG_, 4, true>), grid, dim3(512), 0, st, xa);     \
      else if (parts == 2) hipLaunchKernelGGL((lstm_xcd_bwd_kernel<UG_, 2, true>), grid, dim3(512), 0, st, xa);\
      else hipLaunchKernelGGL((lstm_xcd_bwd_kernel<UG_, 1, true>), grid, dim3(512), 0, st, xa);                \
    } else if (parts == 4) hipLaunchKernelGGL((lstm_xcd_bwd_kernel<UG_, 4, false>), grid, dim3(512), 0, st, xa); \
    else if (parts == 2) hipLaunchKernelGGL((lstm_xcd_bwd_kernel<UG_, 2, false>), grid, dim3(512), 0, st, xa); \
    else hipLaunchKernelGGL((lstm_xcd_bwd_kernel<UG_, 1, false>), grid, dim3(512), 0, st, xa);                 \
  } while (0)
      switch (ug) {
        case 4: ONSSEN_BWD_UG(4); break;
        case 8: ONSSEN_BWD_UG(8); break;
        case 12: ONSSEN_BWD_UG(12); break;
        case 16: ONSSEN_BWD_UG(16); break;
        default: ONSSEN_BWD_UG(20); break;
      }
#undef ONSSEN_BWD_UG
    }
    ONSSEN_LAUNCH_CHECK();
    return ONSSEN_OK;
  }
  const size_t img_bytes = align256((size_t)2 * 2 * ceil_div(B, 16) * KQB * 2048);
  hipError_t e = hipMemsetAsync(ws, 0, img_bytes, st);   // rows past B and the K tail of the images stay zero
  if (e != hipSuccess) return (int)e;
  BwdArgs p;
  p.gd = gates_dp; p.cs = cs; p.dy = dy; p.wT = whh_img; p.ds = (unsigned short*)ws; p.dc = (float*)((char*)ws + img_bytes);
  p.B = B; p.T = T; p.Hp = Hp; p.NP = NP; p.UG = ug; p.KQB = KQB; p.NUB = NUB;
  ONSSEN_CLEAR_ERROR();
  const dim3 grid((unsigned)NUB, 2, (unsigned)ceil_div(B, 16));
  for (int s = 0; s < T; ++s) {
    p.step = s;
    hipLaunchKernelGGL(lstm_bwd_step_kernel, grid, dim3(64 * recb::NW), 0, st, p);
  }
  ONSSEN_LAUNCH_CHECK();
  return ONSSEN_OK;
}

size_t onssen_bn_rows_workspace_bytes(int64_t M, int C) {
  return M > 0 && C > 0 ? (size_t)((M + bnr::STRIP - 1) / bnr::STRIP) * 3 * (size_t)C * sizeof(float) : 0;   // (count-free strip sums + shift)
}

int onssen_bn_rows_train_f32(const float* x, int64_t M, int C, const float* gamma, const float* beta, float eps, float* y,
                             float* mean, float* invstd, void* ws, size_t ws_bytes, void* stream) {
  if (!x || !gamma || !beta || !y || !mean || !invstd || !ws || M <= 1 || C <= 0 || M > 0x7fffffffL * bnr::STRIP) return ONSSEN_E_ARG;
  if (ws_bytes < onssen_bn_rows_workspace_bytes(M, C)) return ONSSEN_E_WORKSPACE;
  ONSSEN_CLEAR_ERROR();
  hipStream_t st = (hipStream_t)stream;
  const int nblk = (int)((M + bnr::STRIP - 1) / bnr::STRIP);
  const dim3 gp((unsigned)nblk, (unsigned)ceil_div(C, 256));
  hipLaunchKernelGGL((bn_rows_partial_kernel<0>), gp, dim3(256), 0, st, x, (const float*)nullptr, (const float*)nullptr,
                     (const float*)nullptr, (long)M, C, (float*)ws);
  hipLaunchKernelGGL((bn_rows_final_kernel<0>), dim3((unsigned)ceil_div(C, 16)), dim3(256), 0, st, (const float*)ws, nblk, (long)M, C,
                     eps, mean, invstd);
  const long nb = ((long)M * C + 255) / 256;
  hipLaunchKernelGGL((bn_rows_apply_kernel<0>), dim3((unsigned)(nb > 16384 ? 16384 : nb)), dim3(256), 0, st, x, (const float*)nullptr,
                     (const float*)mean, (const float*)invstd, gamma, beta, (const float*)nullptr, (long)M, C, y);
  ONSSEN_LAUNCH_CHECK();
  return ONSSEN_OK;
}

int onssen_bn_rows_grad_f32(const float* x, const float* dy, int64_t M, int C, const float* gamma, const float* mean,
                            const float* invstd, float* dx, float* dgamma, float* dbeta, void* ws, size_t ws_bytes, void* stream) {
  if (!x || !dy || !gamma || !mean || !invstd || !dx || !dgamma || !dbeta || !ws || M <= 1 || C <= 0) return ONSSEN_E_ARG;
  if (ws_bytes < onssen_bn_rows_workspace_bytes(M, C)) return ONSSEN_E_WORKSPACE;
  ONSSEN_CLEAR_ERROR();
  hipStream_t st = (hipStream_t)stream;
  const int nblk = (int)((M + bnr::STRIP - 1) / bnr::STRIP);
  const dim3 gp((unsigned)nblk, (unsigned)ceil_div(C, 256));
  hipLaunchKernelGGL((bn_rows_partial_kernel<1>), gp, dim3(256), 0, st, x, dy, mean, invstd, (long)M, C, (float*)ws);
  hipLaunchKernelGGL((bn_rows_final_kernel<1>), dim3((unsigned)ceil_div(C, 16)), dim3(256), 0, st, (const float*)ws, nblk, (long)M, C,
                     0.0f, dbeta, dgamma);
  const long nb = ((long)M * C + 255) / 256;
  hipLaunchKernelGGL((bn_rows_apply_kernel<1>), dim3((unsigned)(nb > 16384 ? 16384 : nb)), dim3(256), 0, st, x, dy, mean, invstd, gamma,
                     (const float*)dbeta, (const float*)dgamma, (long)M, C, dx);
  ONSSEN_LAUNCH_CHECK();
  return ONSSEN_OK;
}

int onssen_l2norm_rows_f32(const float* x, int64_t rows, int D, float eps, float* y, void* stream) {
  if (!x || !y || rows <= 0 || D <= 0 || D > 64 || (D % 4) != 0 || !(eps > 0.0f)) return ONSSEN_E_ARG;
  if (!aligned16(x) || !aligned16(y)) return ONSSEN_E_ALIGN;
  ONSSEN_CLEAR_ERROR();
  const int dq = D <= 20 ? 5 : D <= 32 ? 8 : 16;
  const long nb = (rows + 4 * (64 / dq) - 1) / (4 * (64 / dq));      // 4 waves per workgroup, 64 / dq rows per wave and pass
#define ONSSEN_L2N(DQ_) hipLaunchKernelGGL((l2norm_rows_kernel<false, DQ_>), dim3((unsigned)(nb > 8192 ? 8192 : nb)), dim3(256), 0, (hipStream_t)stream, x, \
                                           (const float*)nullptr, (long)rows, D, eps, y)
  if (dq == 5) ONSSEN_L2N(5); else if (dq == 8) ONSSEN_L2N(8); else ONSSEN_L2N(16);
#undef ONSSEN_L2N
  ONSSEN_LAUNCH_CHECK();
  return ONSSEN_OK;
}

int onssen_l2norm_rows_grad_f32(const float* x, const float* g, int64_t rows, int D, float eps, float* dx, void* stream) {
  if (!x || !g || !dx || rows <= 0 || D <= 0 || D > 64 || (D % 4) != 0 || !(eps > 0.0f)) return ONSSEN_E_ARG;
  if (!aligned16(x) || !aligned16(g) || !aligned16(dx)) return ONSSEN_E_ALIGN;
  ONSSEN_CLEAR_ERROR();
  const int dq = D <= 20 ? 5 : D <= 32 ? 8 : 16;
  const long nb = (rows + 4 * (64 / dq) - 1) / (4 * (64 / dq));      // 4 waves per workgroup, 64 / dq rows per wave and pass
#define ONSSEN_L2N(DQ_) hipLaunchKernelGGL((l2norm_rows_kernel<true, DQ_>), dim3((unsigned)(nb > 8192 ? 8192 : nb)), dim3(256), 0, (hipStream_t)stream, x, g, \
                                           (long)rows, D, eps, dx)
  if (dq == 5) ONSSEN_L2N(5); else if (dq == 8) ONSSEN_L2N(8); else ONSSEN_L2N(16);
#undef ONSSEN_L2N
  ONSSEN_LAUNCH_CHECK();
  return ONSSEN_OK;
}

int onssen_l2norm_rows_grad_y_f32(const float* y, const float* inv_norm, const float* g, int64_t rows, int D, float eps, float* dx,
                                  void* stream) {
  if (!y || !inv_norm || !g || !dx || rows <= 0 || D <= 0 || D > 64 || (D % 4) != 0 || !(eps > 0.0f)) return ONSSEN_E_ARG;
  if (!aligned16(y) || !aligned16(g) || !aligned16(dx)) return ONSSEN_E_ALIGN;
  ONSSEN_CLEAR_ERROR();
  const int dq = D <= 20 ? 5 : D <= 32 ? 8 : 16;
  const long nb = (rows + 4 * (64 / dq) - 1) / (4 * (64 / dq));      // 4 waves per workgroup, 64 / dq rows per wave and pass
#define ONSSEN_L2N(DQ_) hipLaunchKernelGGL((l2norm_rows_grad_y_kernel<DQ_>), dim3((unsigned)(nb > 8192 ? 8192 : nb)), dim3(256), 0, (hipStream_t)stream, y, \
                                           inv_norm, g, (long)rows, D, eps, dx)
  if (dq == 5) ONSSEN_L2N(5); else if (dq == 8) ONSSEN_L2N(8); else ONSSEN_L2N(16);
#undef ONSSEN_L2N
  ONSSEN_LAUNCH_CHECK();
  return ONSSEN_OK;
}

int onssen_dropout_f32(const float* x, int64_t n, float p, uint64_t seed, float* out, void* stream) {
  if (!x || !out || n <= 0 || !(p >= 0.0f) || !(p < 1.0f)) return ONSSEN_E_ARG;
  const int vec = (n % 4) == 0 && aligned16(x) && aligned16(out);
  const double t = (double)p * 4294967296.0;
  const unsigned thr = t >= 4294967295.0 ? 4294967295u : (unsigned)t;     // hash < thr: dropped
  const long work = vec ? n / 4 : n;
  const long nb = (work + 255) / 256;
  ONSSEN_CLEAR_ERROR();
  hipLaunchKernelGGL(dropout_kernel, dim3((unsigned)(nb > 16384 ? 16384 : nb)), dim3(256), 0, (hipStream_t)stream, x, (long)n, thr,
                     1.0f / (1.0f - p), (unsigned long long)seed, out, vec);
  ONSSEN_LAUNCH_CHECK();
  return ONSSEN_OK;
}

static long adam_blocks(const int64_t* numel, int n) {
  long b = 0;
  for (int i = 0; i < n; ++i) b += (numel[i] + opt::CHUNK - 1) / opt::CHUNK;
  return b;
}

size_t onssen_clip_adam_workspace_bytes(const int64_t* numel_host, int n) {
  if (!numel_host || n <= 0) return 0;
  for (int i = 0; i < n; ++i)
    if (numel_host[i] <= 0) return 0;
  return align256((size_t)(1 + adam_blocks(numel_host, n)) * sizeof(float));
}

int onssen_clip_adam_f32(int n, float* const* p_host, float* const* g_host, float* const* m_host, float* const* v_host,
                         const int64_t* numel_host, float max_norm, double lr, double beta1, double beta2, double eps, int step,
                         int write_grads, void* ws, size_t ws_bytes, void* stream) {
  if (n <= 0 || !p_host || !g_host || !m_host || !v_host || !numel_host || step < 1 || !(lr >= 0.) || !(beta1 >= 0. && beta1 < 1.) ||
      !(beta2 >= 0. && beta2 < 1.) || !(eps >= 0.))
    return ONSSEN_E_ARG;
  for (int i = 0; i < n; ++i)
    if (!p_host[i] || !g_host[i] || !m_host[i] || !v_host[i] || numel_host[i] <= 0) return ONSSEN_E_ARG;
  const bool clip = max_norm > 0.f && max_norm < INFINITY;
  const long nblk = adam_blocks(numel_host, n);
  if (nblk > 0x7fffffffL) return ONSSEN_E_ARG;
  if (clip && (!ws || ws_bytes < onssen_clip_adam_workspace_bytes(numel_host, n))) return ONSSEN_E_WORKSPACE;
  if (clip && (reinterpret_cast<uintptr_t>(ws) & 15u)) return ONSSEN_E_ALIGN;
  hipStream_t st = (hipStream_t)stream;
  float* wsf = (float*)ws;
  // hyper-parameters arrive as doubles (Python floats) and are rounded ONCE, after 1 - beta has been formed: 1 - float(0.999) is
  // off by 1.3e-5 relative, which is what torch avoids by the same route
  const double bc1 = 1.0 - pow(beta1, (double)step), bc2 = 1.0 - pow(beta2, (double)step);
  ONSSEN_CLEAR_ERROR();
  for (int pass = clip ? 0 : 1; pass < 2; ++pass) {
    long done_blocks = 0;
    for (int i0 = 0; i0 < n; i0 += opt::MAXT) {
      AdamArgs a;
      a.n = n - i0 < opt::MAXT ? n - i0 : opt::MAXT;
      int b = 0;
      for (int i = 0; i < a.n; ++i) {
        a.p[i] = p_host[i0 + i]; a.g[i] = g_host[i0 + i]; a.m[i] = m_host[i0 + i]; a.v[i] = v_host[i0 + i];
        a.numel[i] = (long)numel_host[i0 + i];
        a.first_block[i] = b;
        b += (int)((numel_host[i0 + i] + opt::CHUNK - 1) / opt::CHUNK);
      }
      a.first_block[a.n] = b;
      a.lr_bc1 = (float)(lr / bc1); a.omb1 = (float)(1.0 - beta1); a.beta2 = (float)beta2; a.omb2 = (float)(1.0 - beta2);
      a.eps = (float)eps; a.inv_sqrt_bc2 = (float)(1.0 / sqrt(bc2));
      a.max_norm = max_norm; a.write_grads = write_grads;
      a.partials = clip ? wsf + 1 + done_blocks : nullptr;
      a.norm = clip ? wsf : nullptr;
      if (pass == 0) hipLaunchKernelGGL(grad_sqnorm_partial_kernel, dim3((unsigned)b), dim3(256), 0, st, a);
      else hipLaunchKernelGGL(clip_adam_kernel, dim3((unsigned)b), dim3(256), 0, st, a);
      done_blocks += b;
    }
    if (pass == 0) hipLaunchKernelGGL(grad_norm_final_kernel, dim3(1), dim3(256), 0, st, (const float*)(wsf + 1), (int)nblk, wsf);
  }
  ONSSEN_LAUNCH_CHECK();
  return ONSSEN_OK;
}

int onssen_phase_input_f32(const float* x_mag, const float* mask, int64_t m_sb, int64_t m_sc, int64_t m_st,
                           int64_t m_sf, const float* x_phase, int B, int C, int T, int F, float* out, void* stream) {
  if (!x_mag || !mask || !x_phase || !out || B <= 0 || C <= 0 || T <= 0 || F <= 0) return ONSSEN_E_ARG;
  const long total = (long)C * B * T * 3 * F;
  const long nb = (total + 255) / 256;
  ONSSEN_CLEAR_ERROR();
  hipLaunchKernelGGL(phase_input_kernel, dim3((unsigned)(nb > 8192 ? 8192 : nb)), dim3(256), 0, (hipStream_t)stream,
                     x_mag, mask, (long)m_sb, (long)m_sc, (long)m_st, (long)m_sf, x_phase, B, C, T, F, out);
  ONSSEN_LAUNCH_CHECK();
  return ONSSEN_OK;
}

int onssen_labels_f32(const float* stft_mix, const float* stft_s1, const float* stft_s2, const float* feature_mix, int B,
                      int T, int F, float db_threshold, float* utt_max, float* one_hot, float* mag_mix, float* mag_s1,
                      float* mag_s2, float* cos_s1, float* cos_s2, void* stream) {
  if (!stft_mix || !stft_s1 || !stft_s2 || !feature_mix || !utt_max || !one_hot || !mag_mix || !mag_s1 || !mag_s2 ||
      B <= 0 || T <= 0 || F <= 0 || ((cos_s1 == nullptr) != (cos_s2 == nullptr)))
    return ONSSEN_E_ARG;
  ONSSEN_CLEAR_ERROR();
  hipStream_t st = (hipStream_t)stream;
  const long per_utt = (long)T * F, total = per_utt * B;
  hipLaunchKernelGGL(utt_max_kernel, dim3((unsigned)B), dim3(256), 0, st, feature_mix, per_utt, utt_max);
  const long nb = (total + 255) / 256;
  hipLaunchKernelGGL(labels_kernel, dim3((unsigned)(nb > 8192 ? 8192 : nb)), dim3(256), 0, st, stft_mix, stft_s1, stft_s2,
                     feature_mix, utt_max, per_utt, total, db_threshold, one_hot, mag_mix, mag_s1, mag_s2, cos_s1, cos_s2);
  ONSSEN_LAUNCH_CHECK();
  return ONSSEN_OK;
}

int onssen_param_guard_u32(const void* const* ptrs, const int64_t* numel, int n, int samples, int mode, uint32_t* ref, uint32_t* flag,
                           void* stream) {
  if (!ptrs || !numel || !ref || !flag || n <= 0 || samples <= 0 || (mode != 0 && mode != 1)) return ONSSEN_E_ARG;
  ONSSEN_CLEAR_ERROR();
  hipLaunchKernelGGL(param_guard_kernel, dim3((unsigned)n), dim3(256), 0, (hipStream_t)stream, reinterpret_cast<const float* const*>(ptrs),
                     reinterpret_cast<const long long*>(numel), samples, mode, ref, flag);
  ONSSEN_LAUNCH_CHECK();
  return ONSSEN_OK;
}

static unsigned ew_blocks(long total) { const long nb = (total + 255) / 256; return (unsigned)(nb > 8192 ? 8192 : nb); }

int onssen_log_magnitude_f32(const float* stft_ri, int64_t n, float epsilon, float* out, void* stream) {
  if (!stft_ri || !out || n <= 0) return ONSSEN_E_ARG;
  ONSSEN_CLEAR_ERROR();
  hipLaunchKernelGGL(log_magnitude_kernel, dim3(ew_blocks(n)), dim3(256), 0, (hipStream_t)stream, stft_ri, (long)n, epsilon, out);
  ONSSEN_LAUNCH_CHECK();
  return ONSSEN_OK;
}

int onssen_cos_difference_f32(const float* stft_1, const float* stft_2, int64_t n, float* out, void* stream) {
  if (!stft_1 || !stft_2 || !out || n <= 0) return ONSSEN_E_ARG;
  ONSSEN_CLEAR_ERROR();
  hipLaunchKernelGGL(cos_difference_kernel, dim3(ew_blocks(n)), dim3(256), 0, (hipStream_t)stream, stft_1, stft_2, (long)n, out);
  ONSSEN_LAUNCH_CHECK();
  return ONSSEN_OK;
}

int onssen_one_hot_f32(const float* feature_mix, const float* mag_s1, const float* mag_s2, int B, int64_t per_utt,
                       float db_threshold, float* utt_max, float* one_hot, void* stream) {
  if (!feature_mix || !mag_s1 || !mag_s2 || !utt_max || !one_hot || B <= 0 || per_utt <= 0) return ONSSEN_E_ARG;
  ONSSEN_CLEAR_ERROR();
  hipStream_t st = (hipStream_t)stream;
  const long total = (long)per_utt * B;
  hipLaunchKernelGGL(utt_max_kernel, dim3((unsigned)B), dim3(256), 0, st, feature_mix, (long)per_utt, utt_max);
  hipLaunchKernelGGL(one_hot_kernel, dim3(ew_blocks(total)), dim3(256), 0, st, feature_mix, mag_s1, mag_s2, utt_max, (long)per_utt,
                     total, db_threshold, one_hot);
  ONSSEN_LAUNCH_CHECK();
  return ONSSEN_OK;
}

// workspace: [B][stride] float header (feature max, centroids, partial sums, done flag) | [B][km::IW] ints + status word |
// compacted active rows [B][T*F][D] (persistent form)
static size_t dc_cluster_header_floats(int B, int D) { return (size_t)B * (1 + 2 * D + km::NBLK * 2 * (D + 1) + 1); }
size_t onssen_dc_cluster_status_offset(int B, int D) {
  if (B <= 0 || D <= 0 || D > km::DMAX) return 0;
  return align256(dc_cluster_header_floats(B, D) * sizeof(float)) + (size_t)B * km::IW * sizeof(int);
}
size_t onssen_dc_cluster_workspace_bytes(int B, int T, int F, int D) {
  if (B <= 0 || T <= 0 || F <= 0 || D <= 0 || D > km::DMAX) return 0;
  return align256(onssen_dc_cluster_status_offset(B, D) + 256) + (size_t)B * T * F * D * sizeof(float);
}

static int dc_cluster_impl(const float* emb, const float* feature, int B, int T, int F, int D, float db_threshold,
                           int iters, float tol, float* masks, void* ws, size_t ws_bytes, int flags, void* stream, const int32_t* frames) {
  if (!emb || !feature || !masks || !ws || B <= 0 || T <= 0 || F <= 0 || D <= 0 || D > km::DMAX || iters < 0 || !(tol >= 0.f))
    return ONSSEN_E_ARG;
  if (ws_bytes < onssen_dc_cluster_workspace_bytes(B, T, F, D)) return ONSSEN_E_WORKSPACE;
  if (!aligned16(emb) || (reinterpret_cast<uintptr_t>(ws) & 255u)) return ONSSEN_E_ALIGN;
  ONSSEN_CLEAR_ERROR();
  hipStream_t st = (hipStream_t)stream;
  const long per_utt = (long)T * F, stride = 1 + 2 * D + km::NBLK * 2 * (D + 1) + 1;
  float* w = (float*)ws;
  int* iw = (int*)((char*)ws + align256(dc_cluster_header_floats(B, D) * sizeof(float)));
  unsigned* status = (unsigned*)((char*)ws + onssen_dc_cluster_status_offset(B, D));
  float* comp = (float*)((char*)ws + align256(onssen_dc_cluster_status_offset(B, D) + 256));
  const bool persistent = !(flags & ONSSEN_DC_CLUSTER_LAUNCH_PER_ITERATION);
  const dim3 sgrid(km::NBLK, (unsigned)B);
  hipLaunchKernelGGL((kmeans2_search_kernel<0>), sgrid, dim3(256), 0, st, emb, feature, per_utt, D, db_threshold, w, stride, frames, F);
  hipLaunchKernelGGL((kmeans2_pick_kernel<0>), dim3((unsigned)B), dim3(64), 0, st, emb, per_utt, D, w, stride, iw);
#define ONSSEN_KM_ASSIGN(MODE_, OUT_)                                                                                      \
  do {                                                                                                                   \
    if (D == 20) hipLaunchKernelGGL((kmeans2_assign_kernel<MODE_, 20>), dim3(km::NBLK, (unsigned)B), dim3(256), 0, st, emb, \
                                    feature, per_utt, D, db_threshold, w, stride, OUT_, frames, F);                         \
    else hipLaunchKernelGGL((kmeans2_assign_kernel<MODE_, 0>), dim3(km::NBLK, (unsigned)B), dim3(256), 0, st, emb, feature,  \
                            per_utt, D, db_threshold, w, stride, OUT_, frames, F);                                           \
  } while (0)
  if (persistent) {
    // active bins compacted once (the same pass finds the second centroid), then ALL Lloyd iterations in one launch per <= 32
    // utterances (km::NBP workgroups of km::LT threads each, one per CU: 256 workgroups fill the chip exactly)
    hipLaunchKernelGGL((kmeans2_count_kernel<false>), sgrid, dim3(256), 0, st, feature, per_utt, db_threshold, w, stride, iw, frames, F, D);
    if (D == 20) hipLaunchKernelGGL((kmeans2_compact_kernel<20>), sgrid, dim3(256), 0, st, emb, feature, per_utt, D, db_threshold, w, stride, iw, comp, frames, F);
    else hipLaunchKernelGGL((kmeans2_compact_kernel<0>), sgrid, dim3(256), 0, st, emb, feature, per_utt, D, db_threshold, w, stride, iw, comp, frames, F);
    hipLaunchKernelGGL((kmeans2_pick_kernel<1>), dim3((unsigned)B), dim3(64), 0, st, emb, per_utt, D, w, stride, (int*)nullptr);
    // (a wait that gives up leaves status = 1: the host sees it and runs the launch-per-iteration form)
    const unsigned spin = xcd_spin_limit();
    // one workgroup per CU: as many utterances per launch as the device has CUs / NBP (32 on a whole MI355X; fewer in a
    // partitioned mode -- a launch that cannot be co-resident would only be caught by its bounded waits)
    static const int lloyd_utts = [] {
      int dev = 0, cus = 0;
      if (hipGetDevice(&dev) != hipSuccess || hipDeviceGetAttribute(&cus, hipDeviceAttributeMultiprocessorCount, dev) != hipSuccess) cus = 256;
      const int n = cus / km::NBP;
      return n < 1 ? 1 : n > 32 ? 32 : n;
    }();
    for (int u0 = 0; u0 < B && iters > 0; u0 += lloyd_utts) {
      const int nutt = B - u0 < lloyd_utts ? B - u0 : lloyd_utts;
      const dim3 lgrid((unsigned)(ceil_div(nutt, 8) * 8 * km::NBP));
      if (D == 20) hipLaunchKernelGGL((kmeans2_lloyd_kernel<20>), lgrid, dim3(km::LT), 0, st, (const float*)comp, per_utt, D, iters, w, stride, iw, u0, nutt, spin, status, tol);
      else hipLaunchKernelGGL((kmeans2_lloyd_kernel<0>), lgrid, dim3(km::LT), 0, st, (const float*)comp, per_utt, D, iters, w, stride, iw, u0, nutt, spin, status, tol);
    }
  } else {
    hipLaunchKernelGGL((kmeans2_search_kernel<1>), sgrid, dim3(256), 0, st, emb, feature, per_utt, D, db_threshold, w, stride, frames, F);
    hipLaunchKernelGGL((kmeans2_pick_kernel<1>), dim3((unsigned)B), dim3(64), 0, st, emb, per_utt, D, w, stride, (int*)nullptr);
    for (int it = 0; it < iters; ++it) {
      ONSSEN_KM_ASSIGN(0, (float*)nullptr);
      hipLaunchKernelGGL(kmeans2_update_kernel, dim3((unsigned)B), dim3(256), 0, st, D, km::NBLK, w, stride, tol);
    }
  }
  ONSSEN_KM_ASSIGN(1, masks);
#undef ONSSEN_KM_ASSIGN
  ONSSEN_LAUNCH_CHECK();
  return ONSSEN_OK;
}

// ---- compacted form (round 4): index -> (the fc_dc GEMM scatters the active rows) -> cluster ---------------------------------
size_t onssen_dc_compact_workspace_bytes(int B, int T, int F, int D) {
  const size_t base = onssen_dc_cluster_workspace_bytes(B, T, F, D);
  return base ? align256(base) + align256((size_t)B * T * F * sizeof(int32_t) + 16) : 0;   // (+16: onssen_linear_x3p_compact reads the map in 16-byte words)
}

int onssen_dc_compact_layout(int B, int T, int F, int D, size_t* comp_offset, size_t* dest_offset) {
  if (onssen_dc_cluster_workspace_bytes(B, T, F, D) == 0) return ONSSEN_E_ARG;
  if (comp_offset) *comp_offset = align256(onssen_dc_cluster_status_offset(B, D) + 256);
  if (dest_offset) *dest_offset = align256(onssen_dc_cluster_workspace_bytes(B, T, F, D));
  return ONSSEN_OK;
}

int onssen_dc_index_f32(const float* feature, int B, int T, const int32_t* frames, int F, int D, float db_threshold, void* ws,
                        size_t ws_bytes, void* stream) {
  if (!feature || !ws || B <= 0 || T <= 0 || F <= 0 || D <= 0 || D > km::DMAX) return ONSSEN_E_ARG;
  if (ws_bytes < onssen_dc_compact_workspace_bytes(B, T, F, D)) return ONSSEN_E_WORKSPACE;
  if (reinterpret_cast<uintptr_t>(ws) & 255u) return ONSSEN_E_ALIGN;
  ONSSEN_CLEAR_ERROR();
  hipStream_t st = (hipStream_t)stream;
  const long per_utt = (long)T * F, stride = 1 + 2 * D + km::NBLK * 2 * (D + 1) + 1;
  float* w = (float*)ws;
  int* iw = (int*)((char*)ws + align256(dc_cluster_header_floats(B, D) * sizeof(float)));
  size_t dest_off = 0;
  onssen_dc_compact_layout(B, T, F, D, nullptr, &dest_off);
  int* dest = (int*)((char*)ws + dest_off);
  const dim3 sgrid(km::NBLK, (unsigned)B);
  hipLaunchKernelGGL((kmeans2_search_kernel<0>), sgrid, dim3(256), 0, st, (const float*)nullptr, feature, per_utt, D, db_threshold, w, stride, frames, F);
  hipLaunchKernelGGL((kmeans2_count_kernel<true>), sgrid, dim3(256), 0, st, feature, per_utt, db_threshold, w, stride, iw, frames, F, D);
  hipLaunchKernelGGL(kmeans2_index_kernel, sgrid, dim3(256), 0, st, feature, per_utt, db_threshold, (const float*)w, stride, iw, dest, frames, F, D);
  ONSSEN_LAUNCH_CHECK();
  return ONSSEN_OK;
}

int onssen_dc_cluster_compact_f32(int B, int T, int F, int D, int iters, float tol, float* masks, void* ws, size_t ws_bytes,
                                  int flags, void* stream) {
  if (!masks || !ws || B <= 0 || T <= 0 || F <= 0 || D <= 0 || D > km::DMAX || iters < 0 || !(tol >= 0.f)) return ONSSEN_E_ARG;
  if (flags & ONSSEN_DC_CLUSTER_LAUNCH_PER_ITERATION) return ONSSEN_E_ARG;      // the compacted form IS the persistent form
  if (ws_bytes < onssen_dc_compact_workspace_bytes(B, T, F, D)) return ONSSEN_E_WORKSPACE;
  if (reinterpret_cast<uintptr_t>(ws) & 255u) return ONSSEN_E_ALIGN;
  ONSSEN_CLEAR_ERROR();
  hipStream_t st = (hipStream_t)stream;
  const long per_utt = (long)T * F, stride = 1 + 2 * D + km::NBLK * 2 * (D + 1) + 1;
  float* w = (float*)ws;
  int* iw = (int*)((char*)ws + align256(dc_cluster_header_floats(B, D) * sizeof(float)));
  unsigned* status = (unsigned*)((char*)ws + onssen_dc_cluster_status_offset(B, D));
  size_t comp_off = 0, dest_off = 0;
  onssen_dc_compact_layout(B, T, F, D, &comp_off, &dest_off);
  const float* comp = (const float*)((char*)ws + comp_off);
  const int* dest = (const int*)((char*)ws + dest_off);
  const dim3 sgrid(km::NBLK, (unsigned)B);
  // (the farthest-point initialisation rides in the Lloyd launch's first pass: INIT)
  const unsigned spin = xcd_spin_limit();
  static const int lloyd_utts = [] {
    int dev = 0, cus = 0;
    if (hipGetDevice(&dev) != hipSuccess || hipDeviceGetAttribute(&cus, hipDeviceAttributeMultiprocessorCount, dev) != hipSuccess) cus = 256;
    const int n = cus / km::NBP;
    return n < 1 ? 1 : n > 32 ? 32 : n;
  }();
  for (int u0 = 0; u0 < B; u0 += lloyd_utts) {     // (also with iters = 0: pass 0 initialises the centroids)
    const int nutt = B - u0 < lloyd_utts ? B - u0 : lloyd_utts;
    const dim3 lgrid((unsigned)(ceil_div(nutt, 8) * 8 * km::NBP));
    if (D == 20) hipLaunchKernelGGL((kmeans2_lloyd_kernel<20, true>), lgrid, dim3(km::LT), 0, st, comp, per_utt, D, iters, w, stride, iw, u0, nutt, spin, status, tol, dest);
    else hipLaunchKernelGGL((kmeans2_lloyd_kernel<0, true>), lgrid, dim3(km::LT), 0, st, comp, per_utt, D, iters, w, stride, iw, u0, nutt, spin, status, tol, dest);
  }
  if (D == 20) hipLaunchKernelGGL((kmeans2_mask_compact_kernel<20>), dim3(km::NBLK * 4, (unsigned)B), dim3(256), 0, st, comp, dest, per_utt, D, (const float*)w, stride, masks);
  else hipLaunchKernelGGL((kmeans2_mask_compact_kernel<0>), dim3(km::NBLK * 4, (unsigned)B), dim3(256), 0, st, comp, dest, per_utt, D, (const float*)w, stride, masks);
  ONSSEN_LAUNCH_CHECK();
  return ONSSEN_OK;
}

int onssen_dc_cluster_f32(const float* emb, const float* feature, int B, int T, int F, int D, float db_threshold,
                          int iters, float tol, float* masks, void* ws, size_t ws_bytes, int flags, void* stream) {
  return dc_cluster_impl(emb, feature, B, T, F, D, db_threshold, iters, tol, masks, ws, ws_bytes, flags, stream, nullptr);
}

int onssen_dc_cluster_ragged_f32(const float* emb, const float* feature, int B, int T, const int32_t* frames, int F, int D,
                                 float db_threshold, int iters, float tol, float* masks, void* ws, size_t ws_bytes, int flags,
                                 void* stream) {
  if (!frames) return ONSSEN_E_ARG;
  return dc_cluster_impl(emb, feature, B, T, F, D, db_threshold, iters, tol, masks, ws, ws_bytes, flags, stream, frames);
}

#ifdef ONSSEN_FFT_PROFILE
int onssen_debug_fft_stamps(long long* host_out, int n) {      // profile builds only; not part of the ABI
  return (int)hipMemcpyFromSymbol(host_out, HIP_SYMBOL(g_fft_stamps), (size_t)n * sizeof(long long), 0, hipMemcpyDeviceToHost);
}
#endif

static int mask_istft_impl(const float* stft_ri, const float* mask, int64_t m_sb, int64_t m_sc, int64_t m_st,
                           int64_t m_sf, int B, int C, int T, int n_fft, int hop, int length, float* out,
                           void* stream, const int32_t* frames, const int32_t* lengths) {
  if (!stft_ri || !out || B <= 0 || C <= 0 || T <= 0 || hop <= 0 || length <= 0 || hop > n_fft) return ONSSEN_E_ARG;
  if (reinterpret_cast<uintptr_t>(stft_ri) & 7u) return ONSSEN_E_ALIGN;               // (re, im) pairs are read as one 8-byte word
  // a chunk of FR hops of output needs FR + ceil(n_fft/hop) - 1 frames (one more when the chunk
  // origin n_fft/2 is not hop-aligned); FB frames fit in LDS
  // speakers go through the inverse FFT in pairs (one complex transform for two real frames) when there are at least
  // two; FB frames of both speakers then share the LDS, so the longer transforms keep fewer frames per workgroup
  const bool hop_aligned = (n_fft % hop) == 0 && ((n_fft / 2) % hop) == 0;
  const int halo = ceil_div(n_fft, hop) - 1 + (hop_aligned ? 0 : 1);
  bool pair = C >= 2 && n_fft <= 512;
  int FB = pair ? (n_fft <= 256 ? 16 : 8) : (n_fft <= 512 ? 16 : 8);
  if (pair && FB - halo <= 0) {      // very small hops: the unpaired form keeps more frames per workgroup
    pair = false;
    FB = 16;
  }
  // the wsj0-2mix transform (256, two speakers) exists with 8, 12 and 16 frames per workgroup.  A grid that fits the chip in one
  // go (3 workgroups per CU) is latency-bound -- its time is the rounds of ONE workgroup, so the fewest frames per workgroup
  // win (one utterance of 1 000 frames: 7.7 / 9.5 / 11.4 us with 8 / 12 / 16); a larger grid is throughput-bound and the
  // halo frames that every chunk transforms again cost more than the shorter workgroups save (32 x 400: 35.1 / 31.6 / 28.4 us)
  static const int fb_knob = ONSSEN_KNOB_INT("ONSSEN_ISTFT_FB", 0);      // debug builds: tools/ab_variants.py
  if (pair && n_fft == 256 && FB - halo > 0) {
    if (fb_knob == 8 || fb_knob == 12 || fb_knob == 16) {
      if (fb_knob - halo > 0) FB = fb_knob;
    } else {
      for (int fb = 8; fb < 16; fb += 4) {
        if (fb - halo <= 0) continue;
        if ((long)ceil_div(length, (fb - halo) * hop) * ceil_div(C, 2) * B <= 256L * 3) { FB = fb; break; }
      }
    }
  }
  const int FR = FB - halo;
  if (FR <= 0) return ONSSEN_E_ARG;
  ONSSEN_CLEAR_ERROR();
  const dim3 grid((unsigned)ceil_div(length, FR * hop), (unsigned)(pair ? ceil_div(C, 2) : C), (unsigned)B), block(256);
  hipStream_t st = (hipStream_t)stream;
#define ONSSEN_ISTFT(N_, FB_, PAIR_)                                                                                    \
  hipLaunchKernelGGL((mask_istft_kernel<N_, FB_, PAIR_>), grid, block, 0, st, stft_ri, mask, (long)m_sb, (long)m_sc, \
                     (long)m_st, (long)m_sf, C, T, hop, length, FR, out, frames, lengths)
  if (n_fft == 256) {
    if (!pair) ONSSEN_ISTFT(256, 16, false);
    else if (FB == 8) ONSSEN_ISTFT(256, 8, true);
    else if (FB == 12) ONSSEN_ISTFT(256, 12, true);
    else ONSSEN_ISTFT(256, 16, true);
  }
  else if (n_fft == 512) { if (pair) ONSSEN_ISTFT(512, 8, true); else ONSSEN_ISTFT(512, 16, false); }
  else if (n_fft == 1024) ONSSEN_ISTFT(1024, 8, false);
  else
    return ONSSEN_E_ARG;
  ONSSEN_LAUNCH_CHECK();
  return ONSSEN_OK;
}

int onssen_mask_istft_f32(const float* stft_ri, const float* mask, int64_t m_sb, int64_t m_sc, int64_t m_st,
                          int64_t m_sf, int B, int C, int T, int n_fft, int hop, int length, float* out,
                          void* stream) {
  return mask_istft_impl(stft_ri, mask, m_sb, m_sc, m_st, m_sf, B, C, T, n_fft, hop, length, out, stream, nullptr, nullptr);
}

int onssen_mask_istft_ragged_f32(const float* stft_ri, const float* mask, int64_t m_sb, int64_t m_sc, int64_t m_st,
                                 int64_t m_sf, int B, int C, int T, const int32_t* frames, int n_fft, int hop, int length,
                                 const int32_t* lengths, float* out, void* stream) {
  if (!frames || !lengths) return ONSSEN_E_ARG;
  return mask_istft_impl(stft_ri, mask, m_sb, m_sc, m_st, m_sf, B, C, T, n_fft, hop, length, out, stream, frames, lengths);
}

}  // extern "C"
