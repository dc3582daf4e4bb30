/* Deep-clustering separation through the C ABI alone (include/onssen_hip.h): no Python, no torch.
 *
 * What egs/wsj0-2mix/deep_clustering/evaluate.py:31-45 of the reference does for a batch of mixtures -- STFT, log-magnitude,
 * the network (onssen/nn/deep_clustering.py:6-43), the 40 dB threshold + 2-means, binary masks, mask x mixture STFT, iSTFT --
 * as the sequence of library calls a host program makes: pack the nn.LSTM / BatchNorm1d / nn.Linear parameters once, then per
 * batch  onssen_stft_logmag_f32 -> onssen_dc_index_f32 -> onssen_blstm_forward_f32 -> onssen_linear_x3p_compact ->
 * onssen_dc_cluster_compact_f32 -> onssen_mask_istft_f32.  Plain pointers and sizes; device memory from the HIP runtime.
 *
 *   separate_dc <in.bin> <out.bin>
 * in.bin  (little endian): int32 magic 0x44435345, B, n_samples, n_fft, hop, H, L, D; then float32 arrays in state_dict order:
 *         per layer, per direction (forward, reverse): weight_ih (4H x in), weight_hh (4H x H), bias_ih (4H), bias_hh (4H);
 *         bn.weight, bn.bias, bn.running_mean, bn.running_var (2H each), bn.eps (1 float); fc_dc.weight (F*D x 2H),
 *         fc_dc.bias (F*D); the mixtures (B x n_samples).
 * out.bin: float32 (B x 2 x n_samples) separated signals.
 * tests/test_gpu_c_abi_example.py writes in.bin from an onssen_amd.nn.deep_clustering module and checks out.bin against
 * onssen_amd.separation.separate_dc bit for bit.
 *
 * Build (also done by __graft_entry__.build()):
 *   gcc -std=c99 -O2 -Iinclude -I/opt/rocm/include examples/separate_dc.c -Lonssen_amd -lonssen_hip -L/opt/rocm/lib -lamdhip64 \
 *       -Wl,-rpath,'$ORIGIN/../onssen_amd' -o examples/separate_dc
 */
#define __HIP_PLATFORM_AMD__ 1
#include <hip/hip_runtime_api.h>
#include <stdint.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>

#include "onssen_hip.h"

#define CHECK_HIP(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { fprintf(stderr, "%s:%d: HIP error %d (%s)\n", __FILE__, __LINE__, (int)e_, hipGetErrorString(e_)); exit(2); } } while (0)
#define CHECK(x) do { int r_ = (x); if (r_ != ONSSEN_OK) { fprintf(stderr, "%s:%d: %s -> %d\n", __FILE__, __LINE__, #x, r_); exit(3); } } while (0)

static void* dmalloc(size_t bytes) { void* p = NULL; CHECK_HIP(hipMalloc(&p, bytes ? bytes : 16)); return p; }
static float* read_to_device(FILE* f, size_t n) {       /* n floats from the file into a fresh device array */
  float* h = (float*)malloc(n * sizeof(float));
  if (!h || fread(h, sizeof(float), n, f) != n) { fprintf(stderr, "short read (%zu floats)\n", n); exit(4); }
  float* d = (float*)dmalloc(n * sizeof(float));
  CHECK_HIP(hipMemcpy(d, h, n * sizeof(float), hipMemcpyHostToDevice));
  free(h);
  return d;
}

int main(int argc, char** argv) {
  if (argc != 3) { fprintf(stderr, "usage: %s in.bin out.bin\n", argv[0]); return 1; }
  if (onssen_abi_version() != ONSSEN_ABI_VERSION) { fprintf(stderr, "library ABI %d, header %d\n", onssen_abi_version(), ONSSEN_ABI_VERSION); return 1; }
  FILE* f = fopen(argv[1], "rb");
  if (!f) { perror(argv[1]); return 1; }
  int32_t hdr[8];
  if (fread(hdr, 4, 8, f) != 8 || hdr[0] != 0x44435345) { fprintf(stderr, "bad header\n"); return 1; }
  const int B = hdr[1], n = hdr[2], n_fft = hdr[3], hop = hdr[4], H = hdr[5], L = hdr[6], D = hdr[7];
  const int F = n_fft / 2 + 1, T = 1 + n / hop, N = F * D;
  void* stream = NULL;                                  /* the null stream; any hipStream_t works */

  /* ---- once per model: pack the parameters into the images the kernels read ------------------------------------ */
  const int ug = 4 * ((H + 127) / 128);                 /* unit group of the XCD-local persistent recurrence: <= 32 members */
  int Hp, NP, KQ, KQ2, Hs;
  int64_t whh_elems, whh_x3_elems;
  CHECK(onssen_lstm_geometry(H, ug, &Hp, &NP, &KQ, &whh_elems));
  CHECK(onssen_lstm_geometry_x3(H, ug, &KQ2, &Hs, &whh_x3_elems));
  const void** wih_img = (const void**)calloc((size_t)L, sizeof(void*));
  const void** whh_img = (const void**)calloc((size_t)L, sizeof(void*));
  const void** bias_p = (const void**)calloc((size_t)L, sizeof(void*));
  for (int l = 0; l < L; ++l) {
    const int in_l = l == 0 ? F : 2 * H;
    const int K_l = l == 0 ? F : 2 * Hp, Kp = l == 0 ? (F + 3) / 4 * 4 : 2 * Hp;   /* columns of the packed projection matrix */
    float* wih_p = (float*)dmalloc((size_t)2 * NP * Kp * 4);           /* [2 directions][NP][Kp] */
    float* whh_f = (float*)dmalloc((size_t)whh_elems * 4);             /* fp32 fragment image: not used by this path, but packed together */
    float* b_p = (float*)dmalloc((size_t)2 * NP * 4);
    uint16_t* whh3 = (uint16_t*)dmalloc((size_t)2 * whh_x3_elems * 2);
    for (int d = 0; d < 2; ++d) {
      float* w_ih = read_to_device(f, (size_t)4 * H * in_l);
      float* w_hh = read_to_device(f, (size_t)4 * H * H);
      float* b_ih = read_to_device(f, (size_t)4 * H);
      float* b_hh = read_to_device(f, (size_t)4 * H);
      CHECK(onssen_lstm_pack_f32(w_ih, w_hh, b_ih, b_hh, in_l, l > 0, H, ug, wih_p + (size_t)d * NP * Kp, whh_f, b_p + (size_t)d * NP, stream));
      CHECK(onssen_lstm_pack_whh_bf16x3(w_hh, H, ug, whh3 + (size_t)d * whh_x3_elems, stream));
      CHECK_HIP(hipDeviceSynchronize());
      CHECK_HIP(hipFree(w_ih)); CHECK_HIP(hipFree(w_hh)); CHECK_HIP(hipFree(b_ih)); CHECK_HIP(hipFree(b_hh));
    }
    /* the persistent form reads the projection matrix of both directions as ONE x3 image [2*NP][ceil(K_l/32)][2][32] */
    uint16_t* img = (uint16_t*)dmalloc((size_t)2 * NP * ((K_l + 31) / 32) * 64 * 2);
    CHECK(onssen_x3_image_f32(wih_p, Kp, 0, 1, 2 * NP, K_l, img, stream));
    CHECK_HIP(hipDeviceSynchronize());
    CHECK_HIP(hipFree(wih_p)); CHECK_HIP(hipFree(whh_f));
    wih_img[l] = img; whh_img[l] = whh3; bias_p[l] = b_p;
  }
  /* the embedding head: eval-mode BatchNorm1d(2H) folded into fc_dc, re-laid for the [fwd(Hp) | rev(Hp)] activations */
  float* gamma = read_to_device(f, (size_t)2 * H);
  float* beta = read_to_device(f, (size_t)2 * H);
  float* mean = read_to_device(f, (size_t)2 * H);
  float* var = read_to_device(f, (size_t)2 * H);
  float bn_eps;
  if (fread(&bn_eps, 4, 1, f) != 1) return 4;
  float* fc_w = read_to_device(f, (size_t)N * 2 * H);
  float* fc_b = read_to_device(f, (size_t)N);
  float* head_w = (float*)dmalloc((size_t)N * 2 * Hp * 4);
  float* head_b = (float*)dmalloc((size_t)N * 4);
  CHECK(onssen_head_pack_f32(fc_w, fc_b, N, H, Hp, gamma, beta, mean, var, bn_eps, head_w, head_b, stream));
  uint16_t* head_img = (uint16_t*)dmalloc((size_t)N * ((2 * Hp + 31) / 32) * 64 * 2);
  CHECK(onssen_x3_image_f32(head_w, 2 * Hp, 0, 1, N, 2 * Hp, head_img, stream));

  /* ---- per batch ----------------------------------------------------------------------------------------------- */
  float* wav = read_to_device(f, (size_t)B * n);
  fclose(f);
  float* logmag = (float*)dmalloc((size_t)B * T * F * 4);
  float* stft_ri = (float*)dmalloc((size_t)B * T * F * 2 * 4);
  CHECK(onssen_stft_logmag_f32(wav, B, n, n, n_fft, hop, 1e-7f, logmag, stft_ri, stream));

  /* clustering workspace: everything in front of the compacted array (centroids, counters, status word) starts out zero */
  const size_t cws_bytes = onssen_dc_compact_workspace_bytes(B, T, F, D);
  size_t comp_off, dest_off;
  CHECK(onssen_dc_compact_layout(B, T, F, D, &comp_off, &dest_off));
  char* cws = (char*)dmalloc(cws_bytes);
  CHECK_HIP(hipMemset(cws, 0, comp_off));
  /* 1. which bins are clustered, and where each lands in the compacted array: known before the network runs */
  CHECK(onssen_dc_index_f32(logmag, B, T, NULL, F, D, 40.0f, cws, cws_bytes, stream));

  /* 2. the BLSTM stack: one projection GEMM + ONE persistent launch per layer; the header of its workspace starts out zero */
  const size_t ws_bytes = onssen_blstm_workspace_bytes(B, T, F, H, L, ug);
  char* ws = (char*)dmalloc(ws_bytes);
  CHECK_HIP(hipMemset(ws, 0, ws_bytes));
  CHECK(onssen_blstm_forward_f32(logmag, (int64_t)T * F, F, B, T, F, H, L, ug, (const float* const*)wih_img,
                                 (const float* const*)whh_img, (const float* const*)bias_p, NULL, ws, ws_bytes,
                                 ONSSEN_BLSTM_BF16X3 | ONSSEN_BLSTM_XCD, stream));
  size_t y_off;
  int KB;
  CHECK(onssen_blstm_y_image(B, T, F, H, L, ug, &y_off, &KB));

  /* 3. fc_dc + L2 norm per bin, storing only the active bins' rows, each at its place in the compacted array */
  CHECK(onssen_linear_x3p_compact((const uint16_t*)(ws + y_off), T * B, 2 * Hp, head_img, head_b, N, D, 1e-12f,
                                  (const int32_t*)(cws + dest_off), (int64_t)T * F, F, (float*)(cws + comp_off), B,
                                  (int64_t)T * F * D, 0, stream));
  /* 4. initialisation + Lloyd iterations + binary masks (B, T, F, 2) */
  float* masks = (float*)dmalloc((size_t)B * T * F * 2 * 4);
  CHECK(onssen_dc_cluster_compact_f32(B, T, F, D, 20, 1e-4f, masks, cws, cws_bytes, 0, stream));
  /* 5. mask x mixture STFT, inverse STFT, overlap-add: (B, 2, n) */
  float* out = (float*)dmalloc((size_t)B * 2 * n * 4);
  CHECK(onssen_mask_istft_f32(stft_ri, masks, (int64_t)T * F * 2, 1, (int64_t)F * 2, 2, B, 2, T, n_fft, hop, n, out, stream));
  CHECK_HIP(hipDeviceSynchronize());

  /* the bounded waits report through status words instead of hanging: examine them before trusting the outputs */
  uint32_t st[3], cst;
  CHECK_HIP(hipMemcpy(st, ws + 280 * 4, sizeof st, hipMemcpyDeviceToHost));
  CHECK_HIP(hipMemcpy(&cst, cws + onssen_dc_cluster_status_offset(B, D), 4, hipMemcpyDeviceToHost));
  if (st[0] != 0 || st[2] != 0 || cst != 0) {
    fprintf(stderr, "aborted launch: recurrence %u, non-finite %u, clustering %u (re-run without ONSSEN_BLSTM_XCD / with "
            "onssen_dc_cluster_f32's launch-per-iteration form)\n", st[0], st[2], cst);
    return 5;
  }
  float* h_out = (float*)malloc((size_t)B * 2 * n * 4);
  CHECK_HIP(hipMemcpy(h_out, out, (size_t)B * 2 * n * 4, hipMemcpyDeviceToHost));
  FILE* g = fopen(argv[2], "wb");
  if (!g || fwrite(h_out, 4, (size_t)B * 2 * n, g) != (size_t)B * 2 * n) { perror(argv[2]); return 1; }
  fclose(g);
  printf("separated %d mixtures of %d samples (T = %d frames, H = %d, L = %d, ug = %d, placement-independent protocol: %u)\n",
         B, n, T, H, L, ug, st[1]);
  return 0;
}
