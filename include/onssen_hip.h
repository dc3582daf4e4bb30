/*
 * onssen_hip.h -- C ABI of libonssen_hip.so, the MI355X (gfx950) implementation of
 * onssen's STFT-domain separation forward pass.
 *
 * The upstream project is pure Python on PyTorch ATen ops; it has no FFI of its
 * own ("C ABI ... new; nothing in the reference to mirror").
 * Each entry point below therefore cites the reference *call site* it replaces
 * (paths relative to the upstream tree).  The binding a maintainer adds on the
 * reference side is a ctypes stub; see INTEGRATION.md.
 *
 * Conventions
 *  - All pointers are DEVICE pointers (HBM) unless the parameter name ends in
 *    `_host`.  The caller owns every buffer including workspaces; the library
 *    never allocates, frees or synchronises (safe under hipGraph capture).
 *  - `stream` is a hipStream_t passed as void*; work is stream-ordered.
 *  - Return value: 0 = ok; >0 = hipError_t of a failed launch; <0 = ONSSEN_E_*.
 *    No exceptions cross the ABI.  Functions are re-entrant; one in-flight call
 *    per workspace.
 *  - dtype: fp32 at every boundary.  Contractions run either on the exact-fp32 MFMA
 *    (v_mfma_f32_16x16x4_f32; the *_f32 GEMM and the default BLSTM flags) or in split-bf16
 *    (three v_mfma_f32_16x16x32_bf16 per fp32 product, fp32 accumulate, ~1e-5 relative;
 *    ONSSEN_BLSTM_BF16X3, *_bf16x3, *_x3p) -- the caller chooses.  STFT / iSTFT butterflies
 *    run in fp64 like the reference's NumPy FFT.
 */
#ifndef ONSSEN_HIP_H
#define ONSSEN_HIP_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define ONSSEN_ABI_VERSION 14   /* 14: onssen_blstm_pipe2_forward_ragged_f32 (the pipelined pair launch over a stream of RAGGED batches of whole utterances: each half of the launch runs its own batch's time steps and row lengths).  13: onssen_blstm_pipe2_* (a two-layer stack software-pipelined over consecutive calls: layer 1 of batch n-1 and layer 0 of batch n in ONE persistent launch, each on half of the XCDs).  12: onssen_log_magnitude_f32, onssen_cos_difference_f32, onssen_one_hot_f32 (the reference's stand-alone feature helpers).  11: onssen_wav_info, onssen_wav_read_batch_f32 (host-side batch RIFF reader of the file loader), onssen_lstm_pack_wih_image_f32, onssen_clip_adam_f32, onssen_lstm_train_backward_img_f32, onssen_lstm_pack_train_f32.  10: onssen_linear_x3p_norms, onssen_l2norm_rows_grad_y_f32, onssen_linear_x3p_batched_split_alt, onssen_x3_image_both_colsum_f32, onssen_dc_head_grad_images_f32, onssen_lstm_wgrad_images_f32, onssen_linear_x3t, onssen_blstm_x_image; ug = 24 (640 < H <= 768) in the persistent split-bf16 recurrence.  9: ragged batches of whole utterances (onssen_*_ragged_f32), the compacted deep-clustering back end, `tol` of onssen_dc_cluster_*, onssen_lstm_train_forward_form_f32.  8: onssen_linear_x3p_resid, onssen_linear_x3p_pair, onssen_x3_image_both_f32.  7: onssen_xcd_spin_limit, onssen_debug_cotenant_spin, chimera mask-loss gradient, compacted clustering.  6: onssen_dropout_f32, onssen_loss_dc_grad_f32, onssen_linear_x3p_batched_split, db_rows of onssen_lstm_train_backward_f32, l2norm_rows and bn_rows kernels; backward recurrence exchanges tagged partial sums.  5: status word [282] (non-finite h), W_hh fragment images unit-major, fp64 SDR workspace */

#define ONSSEN_OK 0
#define ONSSEN_E_ARG (-1)         /* invalid argument / unsupported shape */
#define ONSSEN_E_WORKSPACE (-2)   /* workspace too small */
#define ONSSEN_E_ALIGN (-3)       /* pointer / stride alignment requirement violated */
/* onssen_wav_*: per-file status / return codes (host-side reader, no device work) */
#define ONSSEN_WAV_TRUNCATED 1          /* the file holds more frames than the row: the first row_stride frames were read */
#define ONSSEN_WAV_E_OPEN (-16)         /* cannot open the file */
#define ONSSEN_WAV_E_FORMAT (-17)       /* not RIFF/WAVE, no fmt / data chunk, malformed */
#define ONSSEN_WAV_E_UNSUPPORTED (-18)  /* sample format other than PCM 8/16/24/32-bit or IEEE float 32/64 */
#define ONSSEN_WAV_E_SOME_FAILED (-19)  /* onssen_wav_read_batch_f32: at least one status_host[i] < 0 */

/* flags of onssen_blstm_forward_f32 */
#define ONSSEN_BLSTM_SPLIT_ROWS 1 /* 16 batch rows per recurrence workgroup instead of 32: more workgroups */
#define ONSSEN_BLSTM_BF16X3 2     /* recurrent product h W_hh^T in split-bf16 (3 bf16 MFMAs per fp32 product,
                                     ~1e-5 relative); whh_p_host[l] must then point to the images made by
                                     onssen_lstm_pack_whh_bf16x3 ([2 directions][whh_x3_elems] uint16), and
                                     wih_p_host[l] to onssen_linear_pack_bf16x3 planes of the [2*NP][K_l] matrix
                                     (ld = K_l rounded up to 32): the input projections run split-bf16 too.
                                     H <= 640. */
#define ONSSEN_BLSTM_XCD 4        /* (with BF16X3) one persistent launch per layer: every (direction, 4 / 8 / 16-row group)
                                     recurrence runs inside one XCD, W_hh register-resident, h_t exchanged through
                                     that XCD's L2 as data-tagged bf16 words (no flags).  Needs ceil(H/ug) <= 32,
                                     H <= 640, ug <= 20 -- or (round 4, split-bf16 only, without FUSE_IN0 / BF16 and
                                     not in the training forward) ug = 24: 640 < H <= 768.  The kernel verifies the placement itself and otherwise
                                     uses placement-independent (write-through / system-scope) accesses; bounded
                                     waits: ws word [281] = 1 reports that, word [280] != 0 an aborted launch
                                     (outputs invalid), word [282] = 1 a non-finite activation (a NaN cannot carry
                                     the exchange's tag: it was replaced by 0 -- outputs are NOT nn.LSTM's NaNs;
                                     the launch-per-step form propagates them).
                                     In this form the activations travel between the layers as x3 images (see
                                     onssen_linear_x3p) written by the recurrence epilogue: wih_p_host[l] must be the
                                     x3 image (onssen_x3_image_f32) of the packed [2*NP][K_l] input-projection
                                     matrix, and the last layer's output image stays in the workspace for the
                                     heads (onssen_blstm_y_image).
                                     WITHOUT ONSSEN_BLSTM_BF16X3 (round 3): the same persistent launch in EXACT fp32
                                     (v_mfma_f32_16x16x4_f32; h exchanged as tagged fp32 words): wih_p_host / whh_p_host /
                                     bias_p_host are the fp32 arrays of onssen_lstm_pack_f32 (as for the launch-per-step
                                     form), G comes from the exact-fp32 GEMM, the layers hand fp32 rows to each other,
                                     y must not be NULL; same limits (H <= 640, ug <= 20). */
#define ONSSEN_BLSTM_FUSE_IN0 16   /* (with XCD, in_dim <= 128, or = 129 with FUSE_TAIL) the first layer's input projection is computed inside its
                                     recurrence launch: wih_p_host[0] must be the B-fragment image made by
                                     onssen_lstm_pack_wih_bf16x3 of the layer's two W_ih, back to back; no G is
                                     written or read for that layer. */
#define ONSSEN_BLSTM_FUSE_TAIL 32   /* (with FUSE_IN0, in_dim = 32k + 1, e.g. F = 129) the lone last input column is a rank-1
                                     update on the VALU instead of a whole MFMA k-chunk: bias_p_host[0] then holds 4*NP
                                     floats -- the bias, then column in_dim-1 of the packed W_ih ([2*NP]). */
#define ONSSEN_BLSTM_BF16 64        /* (with BF16X3 | XCD) OPT-IN reduced precision: every product of the stack (input
                                     projections, h W_hh^T) uses the bf16 hi halves only -- one MFMA instead of
                                     three, bf16-grade results (~1e-2 relative), outside the 1e-4 parity contract.
                                     Same images, same workspace; accumulation, gates and cell state stay fp32. */
#define ONSSEN_BLSTM_G_READY 128     /* measurement aid (bench.py times the recurrence kernel by itself): skip the input
                                       projection of every layer -- G is what an earlier call with the same arguments left
                                       in the workspace (L = 1 calls only make sense) */
#define ONSSEN_BLSTM_WS_DIRTY 65536  /* (with XCD | BF16X3) the workspace behind its header was NOT zeroed for this shape (a grow-only
                                       scratch buffer reused across ragged batches of changing length): the call clears the k
                                       padding of the recurrence output images itself (two tiny launches) before it runs.  The
                                       header must still have been zeroed once by the owner. */
/* Debug flags (0 in production).  Launch-per-step form only: bits 8..11 switch off parts of the kernel for profiling
 * ablations (results are then meaningless): 0x100 h loads, 0x200 W_hh loads, 0x400 MFMA, 0x800 G/c loads; 0x1000 selects
 * libm-grade gate non-linearities.  ONSSEN_BLSTM_XCD form: 0x800 = TEST bit, rotates the exchange groups across the XCDs
 * so that the placement-independent accesses run; its profiling switches are build variants (lstm.inc). */

/* epilogue modes of onssen_linear_f32 */
#define ONSSEN_EPI_BIAS 0     /* C = A W^T + b                                   (nn.Linear)            */
#define ONSSEN_EPI_L2NORM 1   /* ... then x / max(||x||_2, eps) over `group` consecutive outputs        */
#define ONSSEN_EPI_SIGMOID 2  /* ... then logistic                                                      */
#define ONSSEN_EPI_BF16 0x100 /* OR-ed into `mode` of onssen_linear_x3p: plain bf16 products (the hi halves of the x3 images only,
                                 fp32 accumulate; ~2^-9 relative per product) instead of split-bf16                       */
#define ONSSEN_EPI_RELU 3     /* ... then max(x, 0), times `resid` (laid out like C) if given -- onssen_linear_f32 only
                                 (enhance: fc_pre / fc_post, onssen/nn/enhancement.py:49-51)           */

int onssen_abi_version(void);
const char* onssen_error_string(int code);

/* ---------------------------------------------------------------------------------------------
 * K1+K2  framed STFT + log-magnitude.
 * Replaces onssen/data/feature_utils.py:20 (librosa.core.stft(sig, n_fft, hop_length), transposed)
 * fused with :49-51 get_log_magnitude and :54-64 get_phase.
 *   wav      (B, n_samples) float32, row stride `wav_stride` elements
 *   logmag   (B, T, F) float32 = log10(|X| + eps),  T = 1 + n_samples/hop, F = n_fft/2 + 1
 *   stft_ri  (B, T, F, 2) float32 (Re, Im) of the complex64 STFT, or NULL
 * n_fft in {256, 512, 1024}; reflect padding needs n_samples > n_fft/2.
 */
int onssen_stft_logmag_f32(const float* wav, int B, int n_samples, int64_t wav_stride, int n_fft, int hop,
                           float eps, float* logmag, float* stft_ri, void* stream);

/* ---------------------------------------------------------------------------------------------
 * Packed-weight geometry of one LSTM direction for unit-group size `ug` (a multiple of 4, <= 20).
 *   Hp = H rounded up to ug; NP = 4*Hp packed gate columns; KQ = ceil(Hp/16) k-chunks;
 *   whh_elems = floats of the MFMA-fragment-ordered recurrent weight image.
 */
int onssen_lstm_geometry(int H, int ug, int* Hp, int* NP, int* KQ, int64_t* whh_elems);

/* Pack one (layer, direction) of nn.LSTM parameters (state_dict layout, gate rows i,f,g,o;
 * onssen/nn/deep_clustering.py:15-22) into the images the kernels consume.
 *   w_ih (4H, in_dim), w_hh (4H, H), b_ih, b_hh (4H)
 *   bidir_in = 0: the layer input has in_dim plain columns; wih_p is [NP][Kp], Kp = in_dim rounded up to 4
 *   bidir_in = 1: the layer input is the previous layer's [fwd(H) | rev(H)]; it is stored by this
 *                 library as [fwd(Hp) | rev(Hp)], so wih_p is [NP][2*Hp] (in_dim must equal 2H)
 *   whh_p   [NU][KQ][ug/4][64][4]   bias_p [NP] = b_ih + b_hh (gate-permuted)
 */
int onssen_lstm_pack_f32(const float* w_ih, const float* w_hh, const float* b_ih, const float* b_hh, int in_dim,
                         int bidir_in, int H, int ug, float* wih_p, float* whh_p, float* bias_p, void* stream);

/* (round 5) One direction's packed W_ih, its bias and the x3 image of the packed matrix in ONE pass: what onssen_lstm_pack_f32's
 * wih_p / bias_p and onssen_x3_image_f32(wih_p, ...) give, bit for bit, without the fp32 whh image nobody reads on the persistent
 * training path.  The training forward re-packs every step (the optimizer moves the weights between two forwards), so
 * the pack is part of the training step (onssen/utils/train.py:75-86).  wih_img: this direction's NP rows of the
 * [2*NP][ceil(K/32)][2][32] image (K = in_dim, or 2*Hp with bidir_in), 16-byte aligned. */
int onssen_lstm_pack_wih_image_f32(const float* w_ih, const float* b_ih, const float* b_hh, int in_dim, int bidir_in, int H, int ug,
                                   float* wih_p, float* bias_p, uint16_t* wih_img, void* stream);

/* (round 5) ... and the whole stack's training images in ONE launch: per (layer l, direction d), index i = 2 l + d of every HOST
 * pointer array: onssen_lstm_pack_wih_image_f32 (layer 0: in_dim plain columns; deeper layers: bidir_in), onssen_lstm_pack_whh_bf16x3
 * and -- when whhR_host is not NULL -- onssen_lstm_pack_whhR_bf16x3, bit for bit what the single calls give.  wih_img_host[i]:
 * direction d's NP rows of layer l's image. */
int onssen_lstm_pack_train_f32(int L, int in_dim, int H, int ug, const float* const* w_ih_host, const float* const* w_hh_host,
                               const float* const* b_ih_host, const float* const* b_hh_host, float* const* wih_p_host,
                               float* const* bias_p_host, uint16_t* const* wih_img_host, uint16_t* const* whh_x3_host,
                               uint16_t* const* whhR_host, void* stream);

/* Split-bf16 image of one direction's W_hh (see ONSSEN_BLSTM_BF16X3): hi = bf16(w), lo = bf16(w - hi), in
 * v_mfma_f32_16x16x32_bf16 B-fragment order [NU][KQ2][ug/4][hi|lo][64][8], KQ2 = ceil(Hp/32). */
int onssen_lstm_geometry_x3(int H, int ug, int* KQ2, int* Hs, int64_t* whh_x3_elems);
int onssen_lstm_pack_whh_bf16x3(const float* w_hh, int H, int ug, uint16_t* whh_x3, void* stream);
/* The same B-fragment image for one direction's W_ih [4H][in_dim] (ONSSEN_BLSTM_FUSE_IN0):
 * [NU][ceil(in_dim/32)][ug/4][hi|lo][64][8] uint16, zero beyond in_dim. */
int onssen_lstm_pack_wih_bf16x3(const float* w_ih, int in_dim, int H, int ug, uint16_t* wih_x3, void* stream);

/* Pack a head nn.Linear(2H -> N) for the [fwd(Hp) | rev(Hp)] activation layout, optionally folding an
 * eval-mode nn.BatchNorm1d(2H) that precedes it (onssen/nn/deep_clustering.py:36-39):
 *   w_p[n][d*Hp+j] = w[n][d*H+j] * s[d*H+j],  b_p[n] = b[n] + sum_k w[n][k] * (beta[k] - mean[k]*s[k]),
 *   s = gamma / sqrt(var + bn_eps).  Pass bn_gamma = NULL for no BatchNorm (chimera heads).
 */
int onssen_head_pack_f32(const float* w, const float* b, int N, int H, int Hp, const float* bn_gamma,
                         const float* bn_beta, const float* bn_mean, const float* bn_var, float bn_eps,
                         float* w_p, float* b_p, void* stream);

/* ---------------------------------------------------------------------------------------------
 * K3/K7/K8/K9  C = epilogue(A W^T + bias)   (exact-fp32 MFMA GEMM, fused epilogue).
 * Replaces nn.Linear + F.normalize / torch.sigmoid at onssen/nn/deep_clustering.py:39-42,
 * onssen/nn/chimera.py:37-42, onssen/nn/phase_network.py:53-66, and nn.LSTM's input projection.
 *   logical row m in [0, M):  i0 = m / R, i1 = m % R
 *     A row  at  A + i0*a_s0 + i1*a_s1  (K contiguous floats)
 *     C row  at  C + i0*c_s0 + i1*c_s1  (N contiguous floats); `resid` (nullable) is laid out like C
 *   W[n][k] row-major with leading dimension ldw (multiple of 4, zero-filled beyond K), bias[N]
 *   mode L2NORM: `group` must divide 80 and N; the residual (phase head) is added before the norm.
 */
int onssen_linear_f32(const float* A, int64_t a_s0, int64_t a_s1, int R, int M, int K, const float* W, int ldw,
                      const float* bias, int N, int mode, int group, float eps, const float* resid, float* C,
                      int64_t c_s0, int64_t c_s1, void* stream);

/* Split-bf16 form of onssen_linear_f32: every fp32 product is a_hi*b_hi + a_hi*b_lo + a_lo*b_hi on the bf16
 * MFMA pipe with fp32 accumulation (~1e-5 relative on a dot product; measured 1e-6 abs on the network
 * outputs).  A is plain fp32 and split on the fly; the weights are pre-split by onssen_linear_pack_bf16x3 into
 * `w_planes` = [N][ldw] bf16 hi plane followed by the [N][ldw] lo plane, ldw a multiple of 32 (zero beyond K).
 * Same row maps, epilogues and argument meaning as onssen_linear_f32; L2NORM `group` must divide 160.
 */
int onssen_linear_pack_bf16x3(const float* w, int N, int K, int ld_in, int ld_out, uint16_t* w_planes, void* stream);
int onssen_linear_bf16x3(const float* A, int64_t a_s0, int64_t a_s1, int R, int M, int K, const uint16_t* w_planes,
                         int ldw, const float* bias, int N, int mode, int group, float eps, const float* resid,
                         float* C, int64_t c_s0, int64_t c_s1, void* stream);

/* `batch` independent problems C_z = A_z W_z^T + bias of one shape in ONE launch (gridDim.z): A_z = a_img + z*a_bs,
 * W_z = w_img + z*w_bs (uint16 elements, multiples of 8), C_z = C + z*c_bs with row stride ldc; the bias is shared.
 * Used by the training path: the two directions' weight-gradient GEMMs fill the chip only together. */
int onssen_linear_x3p_batched(const uint16_t* a_img, int64_t a_bs, int M, int K, const uint16_t* w_img, int64_t w_bs,
                              const float* bias, int N, float* C, int64_t c_bs, int64_t ldc, int batch, void* stream);
/* The same with row map and TWO dense outputs: row m of problem z goes to C + z*c_bs + (m / R)*c_s0 + (m % R)*c_s1 for the
 * columns n < n_split and to C2 + z*c2_bs + (m / R)*c2_s0 + (m % R)*c2_s1 (column n - n_split) for the rest.  The training
 * path uses it to write [dW_ih | dW_hh] of both directions straight into nn.LSTM's row order (packed row 4u + gate ->
 * row gate*H + u: R = 4, c_s0 = ld, c_s1 = H*ld) and into two contiguous matrices: no gather, no strided copies. */
int onssen_linear_x3p_batched_split(const uint16_t* a_img, int64_t a_bs, int M, int K, const uint16_t* w_img, int64_t w_bs,
                                    const float* bias, int N, int R, float* C, int64_t c_bs, int64_t c_s0, int64_t c_s1,
                                    int n_split, float* C2, int64_t c2_bs, int64_t c2_s0, int64_t c2_s1, int batch, void* stream);
/* ... with the roles of the two outputs alternating: even problems as above, ODD problems write their columns n < n_split_odd
 * to C2 (+ z*c2_bs, column n) and the rest to C (+ z*c_bs, column n - n_split_odd).  With w_bs smaller than a problem's N rows the
 * problems' W windows overlap: the training path lays out [h_prev forward | x | h_prev reverse] (Hp | Kx | Hp image rows),
 * w_bs = Hp rows -- problem 0 reads [h_f | x], problem 1 [x | h_r], and the layer input's transposed image exists once. */
int onssen_linear_x3p_batched_split_alt(const uint16_t* a_img, int64_t a_bs, int M, int K, const uint16_t* w_img, int64_t w_bs,
                                        const float* bias, int N, int R, float* C, int64_t c_bs, int64_t c_s0, int64_t c_s1,
                                        int n_split, float* C2, int64_t c2_bs, int64_t c2_s0, int64_t c2_s1, int n_split_odd,
                                        int batch, void* stream);

/* The weight gradients of one bidirectional LSTM layer from ROW-MAJOR x3 images, as the kernels that produced them left them --
 * no transposed image is made (round 4; gfx950's transposing LDS read delivers the MFMA fragments):
 *   dp_img [K = T*B][2*NP / 32][2][32]  dL/d(pre-activation), columns direction*NP + packed gate column
 *   y_img  [K][ceil(2*Hp / 32)][2][32]   the layer's output (onssen_blstm_y_image),   x_img [K][ceil(Kx / 32)][2][32]  its input
 *   dW_hh[d] = dP_d^T h_prev_d (h_prev = y rows k - B for the forward, k + B for the reverse direction),  dW_ih[d] = dP_d^T x,
 *   direction d at dW_* + d*bs, packed row m at (m / R)*s0 + (m % R)*s1 (R = 4, s0 = ld, s1 = H*ld writes nn.LSTM's row order).
 * NP % 32 == 0, Hp % 8 == 0 (x_img's zero padding past Kx is read up to the next multiple of 8, never stored).  zero16: 16 readable zero bytes.  Bit-identical to
 * onssen_linear_x3p_batched_split_alt on the transposed images. */
int onssen_lstm_wgrad_images_f32(const uint16_t* dp_img, const uint16_t* y_img, const uint16_t* x_img, int K, int B, int NP, int Hp,
                                 int Kx, const float* zero16, int R, float* dW_ih, int64_t ih_bs, int64_t ih_s0, int64_t ih_s1,
                                 float* dW_hh, int64_t hh_bs, int64_t hh_s0, int64_t hh_s1, void* stream);
/* The same kernel for one plain weight gradient: C [M][N] (row stride ldc) = A^T W, contraction over the K rows of the row-major
 * x3 images A [K][ceil(M/32)][2][32] and W [K][ceil(N/32)][2][32] (their zero padding past M / N is read, never stored). */
int onssen_linear_x3t(const uint16_t* a_img, const uint16_t* w_img, int K, int M, int N, const float* zero16, float* C, int64_t ldc,
                      void* stream);

/* x3 image of the TRANSPOSE of a row-major fp32 matrix src [K][ld >= M], optionally shifted along k: image row m
 * (0 <= m < M), element k (0 <= k < K) = src[(k + k_shift) * ld + m], 0 where k + k_shift is outside [0, K).  Operands of
 * the training path's weight-gradient GEMMs (contraction over the T*B rows of row-major activations; k_shift = -+B turns
 * the layer output into "h of the step before" for the forward / reverse direction). */
int onssen_x3_image_t_f32(const float* src, int64_t ld, int M, int K, int k_shift, uint16_t* img, void* stream);
/* Both images of a row-major [K][M] matrix (leading dimension ld) in one pass: img_rows = its x3 image [K][ceil(M/32)][2][32]
 * (what onssen_x3_image_f32 gives), img_t = the image of its transpose [M][ceil(K/32)][2][32] (onssen_x3_image_t_f32 with
 * k_shift 0).  The training path's dP feeds the input-gradient GEMM as the first and the weight-gradient GEMM as the second. */
int onssen_x3_image_both_f32(const float* src, int64_t ld, int M, int K, uint16_t* img_rows, uint16_t* img_t, void* stream);
/* ... also leaving colsum [ceil(K/32)][M]: the column sums of every 32-row block of src -- their sum over the blocks is the
 * column sum of src (the bias gradient of the layer whose output gradient src is) without another pass over it. */
int onssen_x3_image_both_colsum_f32(const float* src, int64_t ld, int M, int K, uint16_t* img_rows, uint16_t* img_t, float* colsum,
                                    void* stream);

/* Split-bf16 GEMM over PRE-SPLIT operands.  An "x3 image" of a row-major [rows][K] fp32 matrix is
 * [rows][KB][2][32] bf16, KB = ceil(K/32): per row and 32-wide k block, 32 x hi = bf16(x) then 32 x lo =
 * bf16(x - hi), zeros beyond K (16-byte aligned).  onssen_x3_image_f32 builds one from fp32 rows (row m at
 * src + (m / R)*s0 + (m % R)*s1); the BLSTM's split-bf16 recurrence writes its output in this form as well.
 *   C row m at C + (m / R)*c_s0 + (m % R)*c_s1 = epilogue(a_img[m] . w_img[n] + bias[n]), n < N
 *   mode BIAS | SIGMOID | L2NORM (group in {20, 40, 80}: multiple of 4 dividing 80, N % group == 0; no residual).
 * Same arithmetic as onssen_linear_bf16x3 (three bf16 MFMAs per product, fp32 accumulate); ~2x its speed because
 * staging is a plain copy and the epilogue stays in registers.
 */
int onssen_x3_image_f32(const float* src, int64_t s0, int64_t s1, int R, int rows, int K, uint16_t* img, void* stream);
int onssen_linear_x3p(const uint16_t* a_img, int M, int K, const uint16_t* w_img, const float* bias, int N, int mode,
                      int group, float eps, float* C, int R, int64_t c_s0, int64_t c_s1, void* stream);
/* TWO heads over the same activations in one launch (onssen/nn/chimera.py:37-45: fc_dc + F.normalize and fc_mi + sigmoid
 * both read the BLSTM output): w_img / bias hold the N rows of both layers, rows [0, n_split) are normalised over `group`
 * consecutive outputs (a multiple of 4 dividing 80; n_split % group == 0) and go to C, rows [n_split, N) pass through the
 * logistic and go to columns n - n_split of C2 (row m at C2 + (m / R)*c2_s0 + (m % R)*c2_s1).  The embedding's last
 * 320-column tile is mostly padding (2 580 = 8 x 320 + 20): the second head's columns fill it, its own launch disappears. */
int onssen_linear_x3p_pair(const uint16_t* a_img, int M, int K, const uint16_t* w_img, const float* bias, int N, int n_split,
                           int group, float eps, float* C, int R, int64_t c_s0, int64_t c_s1, float* C2, int64_t c2_s0,
                           int64_t c2_s1, int bf16_only, void* stream);
/* The embedding head of a TRAINING forward (onssen/nn/deep_clustering.py:39-41: fc_dc, then F.normalize over each bin's D
 * features): C [M][N] = the normalised rows as with ONSSEN_EPI_L2NORM, and inv_norm [M][N / group] = 1 / max(||x||, eps) of
 * every group (group a multiple of 4 dividing 80).  With the two, the backward of the normalisation needs no raw product
 * (onssen_l2norm_rows_grad_y_f32): the 4 M N bytes of x are neither written nor read back, and the separate normalisation
 * pass of the forward disappears. */
int onssen_linear_x3p_norms(const uint16_t* a_img, int M, int K, const uint16_t* w_img, const float* bias, int N, int group,
                            float eps, float* C, float* inv_norm, void* stream);

/* The same GEMM with phase_net's head epilogue (onssen/nn/phase_network.py:58-66: fc_phase(bn(rnn)) + mix phase, then
 * F.normalize over (re, im)): C = normalise_group(a_img . w_img^T + bias + resid), group = 2 (pairs of consecutive
 * outputs, N even) or a multiple of 4 dividing 80.  resid has C's row addressing with the row-in-block index taken
 * modulo resid_mod: row m reads resid + (m / R)*c_s0 + ((m % R) % resid_mod)*c_s1 -- both speakers of the phase network
 * (batch rows b and b + B of the shared phase BLSTM) add the SAME mixture phase, so one launch serves both.
 * bf16_only != 0: plain bf16 products (ONSSEN_EPI_BF16). */
int onssen_linear_x3p_resid(const uint16_t* a_img, int M, int K, const uint16_t* w_img, const float* bias, int N, int group,
                            float eps, const float* resid, int resid_mod, float* C, int R, int64_t c_s0, int64_t c_s1,
                            int bf16_only, void* stream);

/* ---------------------------------------------------------------------------------------------
 * K3+K4  stacked bidirectional LSTM, eval semantics (zero initial state, no dropout).
 * Replaces `rnn_output, hidden = self.rnn(x)` (onssen/nn/deep_clustering.py:35, chimera.py:35,
 * phase_network.py:48,55).
 *   x         element (b, t, k) at x + b*xs_b + t*xs_t + k, k < in_dim
 *   wih_p_host / whh_p_host / bias_p_host: HOST arrays of L device pointers; entry l holds both
 *             directions back to back: wih [2*NP][Kp_l], whh [2][whh_elems], bias [2*NP]
 *   y         (T, B, 2*Hp) time-major output of the last layer: [fwd(Hp) | rev(Hp)], padded units are 0.  May be
 *             NULL in the ONSSEN_BLSTM_XCD form when the caller only consumes the x3 image (onssen_blstm_y_image):
 *             the recurrence then skips its fp32 stores.
 *   ws        workspace of onssen_blstm_workspace_bytes() bytes, 256-byte aligned, ZEROED ONCE by its owner when
 *             it is allocated and never again.  Its first ONSSEN_BLSTM_WS_HEADER_BYTES hold the exchange state
 *             of the ONSSEN_BLSTM_XCD form (flags, generations, status words): everything in there is monotonic,
 *             and no call memsets it (so a hipGraph replay does not depend on a memset node reaching the kernel's
 *             L2); the k padding of the x3 images further back is never written and must read as zero (the h hand-off
 *             area needs nothing: the ONSSEN_BLSTM_XCD kernels clear their slots at start-up, the launch-per-step form
 *             memsets it).  u32 word
 *             [280] != 0: a launch gave up waiting (outputs invalid); word [281] = 1: some launch used the
 *             placement-independent accesses; word [282] = 1: a non-finite activation was replaced by 0 (see
 *             ONSSEN_BLSTM_XCD).  The host resets [280] and [282] after reporting them.
 * Default form: one input-projection GEMM + T recurrence launches per layer; capture the call in a hipGraph
 * to amortise launch cost.  ONSSEN_BLSTM_XCD: one GEMM + ONE persistent launch per layer (DESIGN.md,
 * "XCD-local persistent recurrence").
 */
#define ONSSEN_BLSTM_WS_HEADER_BYTES 32768
size_t onssen_blstm_workspace_bytes(int B, int T, int in_dim, int H, int L, int ug);
/* Where the ONSSEN_BLSTM_XCD form leaves the x3 image [T*B][KB][2][32] (row m = t*B + b, KB = ceil(2*Hp/32),
 * k = d*Hp + j) of the LAST layer's output: byte offset inside the workspace.  Feed it to onssen_linear_x3p. */
int onssen_blstm_y_image(int B, int T, int in_dim, int H, int L, int ug, size_t* offset_bytes, int* KB);
/* ... and the x3 image [T*B][ceil(in_dim/32)][2][32] of the stack's INPUT (row m = t*B + b), made by the XCD form for its first
 * projection GEMM (not with ONSSEN_BLSTM_FUSE_IN0): the training path's weight-gradient GEMM over row-major images reads it back. */
int onssen_blstm_x_image(int B, int T, int in_dim, int H, int L, int ug, size_t* offset_bytes, int* KB);
int onssen_blstm_forward_f32(const float* x, int64_t xs_b, int64_t xs_t, int B, int T, int in_dim, int H, int L,
                             int ug, const float* const* wih_p_host, const float* const* whh_p_host,
                             const float* const* bias_p_host, float* y, void* ws, size_t ws_bytes, int flags,
                             void* stream);

/* ---- Two-layer stack, SOFTWARE-PIPELINED over consecutive calls (round 6) -----------------------------------------------
 * The same `self.rnn(x)` (onssen/nn/deep_clustering.py:35) for num_layers = 2 and B <= 32, arranged for THROUGHPUT over a stream
 * of batches (the evaluation loop of onssen/utils/test.py:29-41 hands over one batch after the other): a B <= 32 layer fills the
 * chip's 8 XCDs only with 8-row groups, whose time step costs what a 16-row group's does bar the MFMAs (1.42 vs 1.88 us at
 * H = 600) -- so call n runs layer 1 of batch n-1 and layer 0 of batch n in ONE persistent launch, 16-row groups (stacked 8-row
 * groups up to B = 16), each layer on its own half of the XCDs:
 *     x3 image + input projection of x (batch n)  ->  [ layer 1 (batch n-1)  ||  layer 0 (batch n) ]  ->  layer 1's input
 *     projection of batch n (kept in ws for call n+1).
 * After call n the x3 image [T*B][KB][2][32] of layer 1's output FOR BATCH n-1 is at onssen_blstm_pipe2_y_image (feed it to
 * onssen_linear_x3p*); call 0's image is the stack's answer to an all-zero layer-1 input projection (finite, meaningless), and one
 * extra call (any x) drains the last batch.  Per batch the results are bit for bit those of onssen_blstm_forward_f32 on 16-row
 * groups without ONSSEN_BLSTM_FUSE_IN0 (e.g. the same rows inside a B = 64 call); against the default B = 32 call (stacked 8-row
 * groups, which add the lo x lo products) they differ in the last bits, inside the same tolerance.
 *   flags: ONSSEN_BLSTM_BF16X3 | ONSSEN_BLSTM_XCD exactly (the plain split-bf16 persistent recurrence), H <= 640; plus, as a
 *   measurement aid, ONSSEN_BLSTM_G_READY: only the persistent launch, on the projections an earlier call left in ws.
 *   wih_p_host / whh_p_host / bias_p_host: HOST arrays of 2 device pointers as for onssen_blstm_forward_f32 in that form (x3 image
 *   of the packed projection, the two directions' onssen_lstm_pack_whh_bf16x3 images, packed bias).
 *   ws: onssen_blstm_pipe2_workspace_bytes() bytes, 256-byte aligned, ZEROED ONCE by its owner; header words as above. */
size_t onssen_blstm_pipe2_workspace_bytes(int B, int T, int in_dim, int H, int ug);
int onssen_blstm_pipe2_y_image(int B, int T, int in_dim, int H, int ug, size_t* offset_bytes, int* KB);
int onssen_blstm_pipe2_forward_f32(const float* x, int64_t xs_b, int64_t xs_t, int B, int T, int in_dim, int H, int ug,
                                   const float* const* wih_p_host, const float* const* whh_p_host,
                                   const float* const* bias_p_host, void* ws, size_t ws_bytes, int flags, void* stream);
/* ... over a stream of RAGGED batches of whole utterances (round 6c; the reference evaluates them one by one, onssen/utils/test.py:29-41;
 * onssen_blstm_forward_ragged_f32 runs K of different lengths per call): batch n is B <= 16 rows padded to ITS longest utterance,
 * T frames, row b live for frames[b] <= T of them; the launch's other half still works on batch n-1 (T_prev, frames_prev -- the
 * values that call n-1 was given; call 0: any valid pair, e.g. this call's own).  The workspace is laid out for T_cap >= every T
 * of the stream (onssen_blstm_pipe2_workspace_bytes / _y_image with T = T_cap); the image of batch n-1 holds T_prev * B rows, zeros
 * at t >= frames_prev[b].  Every row's outputs at its own frames are bit for bit those of onssen_blstm_forward_ragged_f32 (i.e. of
 * its own batch-1 run): stacked tiles in both. */
int onssen_blstm_pipe2_forward_ragged_f32(const float* x, int64_t xs_b, int64_t xs_t, int B, int T_cap, int T, const int32_t* frames,
                                          int T_prev, const int32_t* frames_prev, int in_dim, int H, int ug,
                                          const float* const* wih_p_host, const float* const* whh_p_host,
                                          const float* const* bias_p_host, void* ws, size_t ws_bytes, int flags, void* stream);

/* ---- Training: nn.LSTM forward with saved state and its backward recurrence ------------------
 * What `loss.backward()` does for `self.rnn` (onssen/utils/train.py:80-84; nn.LSTM autograd), one layer at a time so that
 * the caller can apply the inter-layer dropout (nn.LSTM(dropout=0.3), onssen/nn/deep_clustering.py:15-22) in between.
 *
 * onssen_lstm_train_forward_f32: one bidirectional layer in the ONSSEN_BLSTM_XCD | ONSSEN_BLSTM_BF16X3 form (same
 *   operands as onssen_blstm_forward_f32 with L = 1: wih_img = x3 image of the packed [2*NP][in_dim] projection,
 *   whh_x3 = the two directions' onssen_lstm_pack_whh_bf16x3 images, bias_p [2*NP]; ws as for L = 1).  Besides
 *   y [T][B][2][Hp] it leaves gates [T][B][2][NP] = the activations (i, f, g, o) of every unit in the packed column
 *   layout of G (column = ugi*4*ug + ju*4 + gate) and cs [T][B][2][Hp] = the cell states c_t.  H <= 768 (ug = 24 above 640).
 * onssen_lstm_train_backward_f32: given dy [T][B][2][Hp] = dL/dy (padded units 0), overwrites `gates_dp` in place with
 *   dL/d(pre-activation) (same layout) -- the operand of the weight / input gradient GEMMs:
 *     dW_ih(packed) = dP^T x,  dW_hh(packed, per direction) = dP_d^T h_prev,  db = sum_rows dP,  dx = dP W_ih(packed).
 *   form ONSSEN_LSTM_BWD_XCD (default of the Python layer; H <= 640): ONE persistent launch per layer, every (direction, row
 *     group) inside one XCD like the forward; each member keeps the rows of W_hh of its own gate columns in registers,
 *     multiplies them with the dP it has just produced and the members reduce-scatter fp32 partial sums of dh through
 *     the XCD's L2.  whh_img = the two directions' onssen_lstm_pack_whhR_bf16x3 images (onssen_lstm_whhR_elems uint16
 *     each).  ws: zeroed ONCE by its owner (header words [280] / [281] report aborts / the placement-independent
 *     protocol exactly as in onssen_blstm_forward_f32).
 *   form ONSSEN_LSTM_BWD_STEPS: one launch per time step (T launches); whh_img = onssen_lstm_pack_whhT_bf16x3 images
 *     (onssen_lstm_whhT_elems).  No requirements on ws contents.
 *   Split-bf16 products, fp32 accumulation and state in both forms.
 *   db_rows (ONSSEN_LSTM_BWD_XCD only, may be NULL): [B][2][NP], receives sum_t dP[t][b] per batch row -- the kernel has every
 *   dP in registers as it goes; the bias gradient is the sum of its B rows instead of a pass over all T*B rows of dP. */
/* onssen_lstm_train_backward_img_f32 (round 5): the ONSSEN_LSTM_BWD_XCD form that leaves dP as the x3 image
 *   dp_img [T*B][2*NP/32][2][32] (row m = t*B + b, column k = d*NP + packed gate column; onssen_x3_image_f32's layout, bit for bit
 *   what it makes of the fp32 dP) INSTEAD of overwriting the gates: the gradient GEMMs read that image, the fp32 -> image pass over
 *   all of dP disappears.  `gates` is only read (padded rows / columns of dp_img that no unit owns are NOT written: 2*NP % 32 == 0
 *   and Hp == H give an image without such holes; otherwise zero it first). */
#define ONSSEN_LSTM_BWD_STEPS 0
#define ONSSEN_LSTM_BWD_XCD 1
int onssen_lstm_train_forward_f32(const float* x, int64_t x_stride_b, int64_t x_stride_t, int B, int T, int in_dim, int H,
                                  int ug, const uint16_t* wih_img, const uint16_t* whh_x3, const float* bias_p, float* y,
                                  float* gates, float* cs, void* ws, size_t ws_bytes, void* stream);
int onssen_lstm_train_backward_img_f32(int B, int T, int H, int ug, const uint16_t* whh_img, const float* dy, const float* gates,
                                       const float* cs, void* ws, size_t ws_bytes, float* db_rows, uint16_t* dp_img, void* stream);
/* The same training forward in a chosen FORM of onssen_blstm_forward_f32 (round 4: the training path no longer leaves the
 * library when the persistent launch cannot be used -- the re-run of a step whose persistent launch aborted, H > 640):
 *   flags = ONSSEN_BLSTM_XCD | ONSSEN_BLSTM_BF16X3   the persistent launch (= onssen_lstm_train_forward_f32; wih = x3 image,
 *                                                    whh = onssen_lstm_pack_whh_bf16x3 images)
 *   flags = ONSSEN_BLSTM_BF16X3                      one launch per time step, split-bf16 (wih = onssen_linear_pack_bf16x3
 *                                                    planes, whh as above; H <= 640)
 *   flags = 0                                        one launch per time step, exact fp32 (wih / whh = onssen_lstm_pack_f32
 *                                                    arrays; any H)
 * gates / cs as above; pair it with onssen_lstm_train_backward_f32 (either form: both read the same saved state). */
int onssen_lstm_train_forward_form_f32(const float* x, int64_t x_stride_b, int64_t x_stride_t, int B, int T, int in_dim, int H,
                                       int ug, const void* wih, const void* whh, const float* bias_p, float* y, float* gates,
                                       float* cs, void* ws, size_t ws_bytes, int flags, void* stream);
int64_t onssen_lstm_whhT_elems(int H, int ug);
int onssen_lstm_pack_whhT_bf16x3(const float* w_hh, int H, int ug, uint16_t* out, void* stream);
int64_t onssen_lstm_whhR_elems(int H, int ug);
int onssen_lstm_pack_whhR_bf16x3(const float* w_hh, int H, int ug, uint16_t* out, void* stream);
size_t onssen_lstm_train_backward_workspace_bytes(int B, int H, int ug, int form);
int onssen_lstm_train_backward_f32(int B, int T, int H, int ug, const uint16_t* whh_img, const float* dy, float* gates_dp,
                                   const float* cs, void* ws, size_t ws_bytes, int form, float* db_rows, void* stream);
/* The inter-layer dropout of nn.LSTM(dropout=p) (onssen/nn/deep_clustering.py:15-22) as ONE pass:
 *   out[i] = keep(seed, i) ? x[i] / (1 - p) : 0,   keep = counter-based hash of (seed, i) compared with p
 * -- no mask tensor: the backward pass calls it again on dL/d(out) with the same seed.  Like the reference's (the RNN
 * library's own generator) the random stream is not torch's; the caller draws `seed` from torch's generator, so runs
 * repeat under torch.manual_seed.  0 <= p < 1; out may alias x; n elements, 16-byte aligned when n % 4 == 0. */
int onssen_dropout_f32(const float* x, int64_t n, float p, uint64_t seed, float* out, void* stream);
/* (round 5) Gradient-norm clipping + Adam over n parameter tensors in two passes (onssen/utils/train.py:83-84:
 * `clip_grad_norm_(model.parameters(), 5)` then `optimizer.step()`; Adam without weight decay / amsgrad, torch.optim.Adam's
 * formulas):  coef = min(1, max_norm / (||g|| + 1e-6)), g' = coef g, m += (g' - m)(1 - beta1), v = beta2 v + (1 - beta2) g'^2,
 * p -= lr / (1 - beta1^step) * m / (sqrt(v) / sqrt(1 - beta2^step) + eps).  The scaled gradient is not written back unless
 * write_grads != 0.  max_norm <= 0 or +inf: no clipping (one pass, ws may be NULL).  *_host: HOST arrays of n DEVICE pointers /
 * element counts; step >= 1 is the step being taken; lr / betas / eps are doubles (rounded once, after 1 - beta is formed).  ws: onssen_clip_adam_workspace_bytes() bytes, 16-byte aligned; after the
 * call ws[0] (float) holds ||g|| before clipping (clip_grad_norm_'s return value). */
size_t onssen_clip_adam_workspace_bytes(const int64_t* numel_host, int n);
int onssen_clip_adam_f32(int n, float* const* p_host, float* const* g_host, float* const* m_host, float* const* v_host,
                         const int64_t* numel_host, float max_norm, double lr, double beta1, double beta2, double eps, int step,
                         int write_grads, void* ws, size_t ws_bytes, void* stream);
/* F.normalize(x, p=2, dim=-1, eps) over `rows` rows of D floats and its backward -- the embedding's unit norm per TF bin in a
 * TRAINING forward (onssen/nn/deep_clustering.py:40-41; inference normalises in the GEMM epilogue):
 *   y = x / max(||x||, eps);   dx = (g - y (y . g)) / ||x||  where ||x|| > eps,  g / eps  elsewhere.   D % 4 == 0, D <= 64. */
int onssen_l2norm_rows_f32(const float* x, int64_t rows, int D, float eps, float* y, void* stream);
/* Train-mode BatchNorm1d over the rows of a row-major (M, C) matrix (onssen/nn/deep_clustering.py:12,36-38: per channel, over all
 * B*T frames of the batch): y = (x - mean) * invstd * gamma + beta with the batch's biased variance, mean / invstd returned
 * for the backward pass (the caller updates running_mean / running_var from mean and 1 / invstd^2 - eps, unbiased by
 * M / (M - 1), as nn.BatchNorm1d does); and its backward: dbeta = sum dy, dgamma = sum dy * xhat,
 * dx = gamma * invstd * (dy - dbeta / M - xhat * dgamma / M).  Column sums are partial sums of 32-row strips added in a
 * fixed order (deterministic).  ws: onssen_bn_rows_workspace_bytes(M, C). */
size_t onssen_bn_rows_workspace_bytes(int64_t M, int C);
int onssen_bn_rows_train_f32(const float* x, int64_t M, int C, const float* gamma, const float* beta, float eps, float* y,
                             float* mean, float* invstd, void* ws, size_t ws_bytes, void* stream);
int onssen_bn_rows_grad_f32(const float* x, const float* dy, int64_t M, int C, const float* gamma, const float* mean,
                            const float* invstd, float* dx, float* dgamma, float* dbeta, void* ws, size_t ws_bytes, void* stream);
int onssen_l2norm_rows_grad_f32(const float* x, const float* g, int64_t rows, int D, float eps, float* dx, void* stream);
/* ... from the normalisation's output y and inv_norm = 1 / max(||x||, eps) per row (what onssen_linear_x3p_norms leaves):
 * dx = (g - y (y . g)) * inv_norm where the norm was not clamped, g * inv_norm otherwise. */
int onssen_l2norm_rows_grad_y_f32(const float* y, const float* inv_norm, const float* g, int64_t rows, int D, float eps, float* dx,
                                  void* stream);

/* ---------------------------------------------------------------------------------------------
 * K11 glue  recurrent input of the phase network for all C speakers at once:
 *   out[(s*B + b), t, :] = cat(x_mag[b,t,:] * mask[b,s,t,:], x_phase[b,t,:,:].view(2F))      (C*B, T, 3F)
 * Replaces `mag_A = x_mag * mask_A; input_A = torch.cat((mag_A, x_phase.view(B,T,-1)), 2)` (and _B) at
 * onssen/nn/phase_network.py:45-47,54.  mask element (b,s,t,f) at mask + b*m_sb + s*m_sc + t*m_st + f*m_sf.
 */
int onssen_phase_input_f32(const float* x_mag, const float* mask, int64_t m_sb, int64_t m_sc, int64_t m_st,
                           int64_t m_sf, const float* x_phase, int B, int C, int T, int F, float* out, void* stream);

/* ---------------------------------------------------------------------------------------------
 * N3  training-label features of a chunk from the complex STFTs of mixture and sources ((B,T,F,2) float32 each,
 * as written by onssen_stft_logmag_f32) and the mixture's log-magnitude (B,T,F).
 * Replaces get_one_hot / get_cos_difference / np.abs in onssen/data/feature_utils.py:77-95 and
 * onssen/data/wsj0_2mix.py:130-152:
 *   one_hot (B,T,F,2) float32: e_argmax(|s1|,|s2|) (speaker 0 on ties), all-zero where
 *           feature < max_over_the_chunk(feature) - db_threshold/20;   utt_max (B) scratch
 *   mag_mix, mag_s1, mag_s2 (B,T,F);  cos_s1, cos_s2 (B,T,F) = cos(angle(mix) - angle(s)), both NULL to skip
 */
int onssen_labels_f32(const float* stft_mix, const float* stft_s1, const float* stft_s2, const float* feature_mix, int B,
                      int T, int F, float db_threshold, float* utt_max, float* one_hot, float* mag_mix, float* mag_s1,
                      float* mag_s2, float* cos_s1, float* cos_s2, void* stream);

/* ---------------------------------------------------------------------------------------------
 * Weight guard: do packed weight images still belong to the live parameters?  `ptrs` / `numel` are DEVICE arrays of n parameter
 * tensors (float32, any 4-byte type) and their element counts.  mode 0: ref[t] = a 32-bit fold of a strided sample (<= `samples`
 * elements, first and last included) of tensor t -- call when the images are built; mode 1: the same fold compared with ref[t],
 * `*flag |= 1` on a difference (sticky; the caller reads and clears it asynchronously).  One launch of n small workgroups,
 * stream-ordered, capturable.  Nothing in the reference corresponds to it: there a module's weights ARE its parameters
 * (onssen/utils/test.py:21-35 loads a checkpoint and runs); here it keeps a cached re-layout honest against updates that bypass
 * autograd's version counter (fused optimizers, p.data arithmetic).
 */
int onssen_param_guard_u32(const void* const* ptrs, const int64_t* numel, int n, int samples, int mode, uint32_t* ref, uint32_t* flag,
                           void* stream);

/* ---------------------------------------------------------------------------------------------
 * The reference's stand-alone feature helpers on ONE utterance's arrays (onssen_amd/data/feature_utils.py binds them under
 * the reference's names).  `stft*` are complex arrays as interleaved (re, im) float32 pairs, `n` complex elements.
 *   onssen_log_magnitude_f32   out[e] = log10f(|stft[e]| + epsilon)           get_log_magnitude, onssen/data/feature_utils.py:49-51
 *   onssen_cos_difference_f32  out[e] = cos(angle(stft_1[e]) - angle(stft_2[e]))   get_cos_difference, feature_utils.py:77-80
 *   onssen_one_hot_f32         one_hot (B, per_utt, 2) float32 from B utterances of per_utt bins each: e_argmax(mag_s1, mag_s2)
 *                              (speaker 0 on ties), all-zero where feature < max_over_the_utterance(feature) - db_threshold/20;
 *                              utt_max (B) scratch                             get_one_hot, feature_utils.py:83-95
 */
int onssen_log_magnitude_f32(const float* stft_ri, int64_t n, float epsilon, float* out, void* stream);
int onssen_cos_difference_f32(const float* stft_1, const float* stft_2, int64_t n, float* out, void* stream);
int onssen_one_hot_f32(const float* feature_mix, const float* mag_s1, const float* mag_s2, int B, int64_t per_utt,
                       float db_threshold, float* utt_max, float* one_hot, void* stream);

/* ---------------------------------------------------------------------------------------------
 * N2  deep-clustering back end: per utterance, 2-means over the D-dimensional embeddings of the bins with
 * feature >= max(feature) - db_threshold/20, then binary masks (B,T,F,2): mask[...,0] = label, mask[...,1] =
 * 1 - label on active bins, 0 in both on silent bins.
 * Replaces `KMeans(n_clusters=2, random_state=0).fit_predict(emb)` + the mask fill at
 * egs/wsj0-2mix/deep_clustering/evaluate.py:36-41 (sklearn on the host upstream).  Deterministic farthest-point
 * initialisation and at most `iters` Lloyd iterations; an utterance stops earlier when its assignment reaches its exact
 * fixed point or -- sklearn's own rule, KMeans(tol=1e-4) being the reference's default -- when the summed squared shift of
 * the two centroids is <= tol x the mean per-feature variance of the clustered rows (tol = 0: the fixed point only).  The
 * rows are unit vectors (F.normalize, onssen/nn/deep_clustering.py:41): that variance is taken as (1 - |mean|^2) / D.
 * Cluster numbering is arbitrary, as it is upstream.  D <= 32.
 * Default form: the active bins are compacted once into the workspace and ALL iterations run in one persistent launch
 * (8 workgroups per utterance that meet at a counter; every wait is bounded by onssen_xcd_spin_limit: a wait that gives up
 * sets the u32 at ws + onssen_dc_cluster_status_offset(B, D) -- the masks of that call are then not the converged ones and
 * the caller repeats it with ONSSEN_DC_CLUSTER_LAUNCH_PER_ITERATION).  flags = ONSSEN_DC_CLUSTER_LAUNCH_PER_ITERATION: two
 * launches per iteration over the uncompacted embeddings, no inter-workgroup waits.  The owner zeroes the status word once.
 * emb 16-byte aligned, ws 256-byte aligned.
 */
#define ONSSEN_DC_CLUSTER_LAUNCH_PER_ITERATION 1
size_t onssen_dc_cluster_workspace_bytes(int B, int T, int F, int D);
size_t onssen_dc_cluster_status_offset(int B, int D);
int onssen_dc_cluster_f32(const float* emb, const float* feature, int B, int T, int F, int D, float db_threshold,
                          int iters, float tol, float* masks, void* ws, size_t ws_bytes, int flags, void* stream);

/* ---------------------------------------------------------------------------------------------
 * N1  deep-clustering loss: VALUE (onssen_loss_dc_f32) and its GRADIENT w.r.t. the embedding (onssen_loss_dc_grad_f32):
 *   per_utt[b] = ||V^T V||_F - 2 ||V^T Y||_F + ||Y^T Y||_F,  V = w * (sum_c Y) * emb, Y = w * one_hot,
 *   w_r = sqrt(mag_r / total_mag[b]),  total_mag[b] = sum_r mag_r            (Frobenius NORMS, as upstream)
 * Replaces onssen/loss/loss_dc.py:24-43 + loss_util.py:4-11; the caller forms upstream's (B,B) product
 * per_utt[None,:] * total_mag[:,None] (loss_dc.py:44) and its mean (utils/train.py:78-79).
 *   emb (B, TF, D), one_hot (B, TF, C) as float, mag (B, TF); D + C <= 34.
 */
/* Mask-inference term of the chimera losses: out[b] = min over the two speaker assignments of
 *   sum |mask_x * mag_mix - t_1| + sum |mask_y * mag_mix - t_2|,  t_s = mag_s (MSA: cos_* = NULL) or
 *   min(mag_mix, relu(mag_s * cos_s)) (PSA).  Replaces onssen/loss/loss_chimera.py:25-29 and :53-57.
 *   mask element (b, e) at mask_* + b*m_sb + e*m_se (the strided views of the (B,T,F,2) mask buffer); maps (B, TF).
 *   perm_out (B) int32 or NULL: the assignment that won (0: A->1, B->2; 1: A->2, B->1; ties -> 0) -- what the gradient follows.
 * onssen_loss_mask_grad_f32: d out[b] / d mask_{A,B} times the incoming gradient g[b], one pass over the maps:
 *   d_mask_A[b,e] = g[b] * mag_mix * sign(mask_A * mag_mix - t_A)  (sign(0) = 0, like torch.abs' backward), t_A the target
 *   `perm` gives speaker A; written at d_mask_* + b*d_sb + e*d_se (e.g. the two planes of one interleaved (B,TF,2) buffer). */
size_t onssen_loss_mask_workspace_bytes(int B);
int onssen_loss_mask_f32(const float* mask_a, const float* mask_b, int64_t m_sb, int64_t m_se, const float* mag_mix,
                         const float* mag_s1, const float* mag_s2, const float* cos_s1, const float* cos_s2, int B, int TF,
                         float* out, int32_t* perm_out, void* ws, size_t ws_bytes, void* stream);
int onssen_loss_mask_grad_f32(const float* mask_a, const float* mask_b, int64_t m_sb, int64_t m_se, const float* mag_mix,
                              const float* mag_s1, const float* mag_s2, const float* cos_s1, const float* cos_s2, int B, int TF,
                              const float* g, const int32_t* perm, float* d_mask_a, float* d_mask_b, int64_t d_sb, int64_t d_se,
                              void* stream);
size_t onssen_loss_dc_workspace_bytes(int B);
int onssen_loss_dc_f32(const float* emb, const float* one_hot, const float* mag, int B, int TF, int D, int C,
                       float* per_utt, float* total_mag, void* ws, size_t ws_bytes, void* stream);
/* d_emb (B, TF, D) = sum_b g_per_utt[b] * d per_utt[b] / d emb -- what autograd derives from loss_dc.py:36-44 (the norms of the
 * affinity blocks): with Z = [V | Y] (weighted as above) and G = Z^T Z, dV = Z [2 Gvv/||Gvv|| ; -2 Gvy^T/||Gvy||], scaled back
 * through the weights.  `ws` must be the workspace onssen_loss_dc_f32 has just filled for the SAME emb / one_hot / mag (its
 * partial Grams are reused); the embedding is read once and d_emb written once.  C <= 4. */
int onssen_loss_dc_grad_f32(const float* emb, const float* one_hot, const float* mag, int B, int TF, int D, int C,
                            const float* g_per_utt, float* d_emb, void* ws, size_t ws_bytes, void* stream);
/* Fusion at the train-step level (the labels are in hand there): from the normalised embedding `emb` [B*T][F*D] and the
 * reciprocal norms `inv_norm` [B*T][F] of onssen_linear_x3p_norms, with the partial Grams onssen_loss_dc_f32 left in ws, straight
 * to the operands of fc_dc's gradient GEMMs -- the gradient of loss_dc w.r.t. the embedding, taken through F.normalize, written as
 * the row-major x3 image [B*T][ceil(F*D/32)][2][32] (dx = draw W), the transposed x3 image [F*D][ceil(B*T/32)][2][32]
 * (dW = draw^T x; img_t may be NULL: onssen_linear_x3t takes the row-major one) and colsum [ceil(B*T/32)][F*D] (db = its sum over the blocks).  Neither d(loss)/d(embedding) nor the gradient of
 * the raw product exists in memory.  D = 20, C <= 4, T >= 32.  g_per_utt [B]: dL/d(per-utterance loss). */
int onssen_dc_head_grad_images_f32(const float* emb, const float* inv_norm, const float* one_hot, const float* mag, int B, int T,
                                   int F, int D, int C, float eps, const float* g_per_utt, void* ws, size_t ws_bytes,
                                   uint16_t* img_rows, uint16_t* img_t, float* colsum, void* stream);

/* ---------------------------------------------------------------------------------------------
 * N4  batch SI-SDR with the best source permutation (evaluation metric of tester.eval):
 *   sdr_out[b] = max over permutations P of (1/C) sum_i SDR(est[b,i], org[b,P(i)]), signals zero-mean, optional
 *   (B, n) mask applied after centring; perm_out[b] (nullable) = index of P in lexicographic order.
 * Replaces calc_sdr_torch + batch_SDR_torch (onssen/evaluate/sdr.py:11-87).  est, org (B, C, n); C <= 4.
 * Means, Gram matrix and the SDR table are accumulated / evaluated in fp64 (the residual power from a Gram matrix cancels:
 * fp32 sums were wrong by 0.1 dB at 50 dB and NaN from ~70 dB); ws 8-byte aligned.
 */
size_t onssen_batch_sdr_workspace_bytes(int B);
int onssen_batch_sdr_f32(const float* est, const float* org, const float* mask, int B, int C, int n, float* sdr_out,
                         int* perm_out, void* ws, size_t ws_bytes, void* stream);

/* ---------------------------------------------------------------------------------------------
 * K10  mask-apply + inverse STFT overlap-add.
 * Replaces `stft_est = stft_mix * mask; librosa.core.istft(stft_est[i].T, hop_length, length)` at
 * egs/wsj0-2mix/deep_clustering/evaluate.py:42-45 and egs/wsj0-2mix/chimera/evaluate.py:40-43.
 *   stft_ri  (B, T, F, 2) float32;  mask element (b,c,t,f) at mask + b*m_sb + c*m_sc + t*m_st + f*m_sf
 *            (mask = NULL means all-ones);  out (B, C, length) float32
 * n_fft = 2(F-1) in {256, 512, 1024}; periodic Hann; divide by the window sum-of-squares where > tiny.
 */
int onssen_mask_istft_f32(const float* stft_ri, const float* mask, int64_t m_sb, int64_t m_sc, int64_t m_st,
                          int64_t m_sf, int B, int C, int T, int n_fft, int hop, int length, float* out,
                          void* stream);

/* ---------------------------------------------------------------------------------------------
 * Ragged batches (round 4): the reference evaluates WHOLE utterances one at a time (onssen/utils/test.py:29-41 with the
 * batch-1 loader of onssen/data/wsj0_2mix.py:231-245), a shape that leaves most of the chip idle.  These entry points
 * run K utterances of different lengths in one launch sequence: every buffer keeps the padded batch shape (T = the longest
 * utterance's frames, n = the longest sample count) and a device array gives each row's own extent.  Row b's results
 * inside its own extent are bit-identical to what the uniform entry point returns for that utterance alone (B = 1,
 * T = frames[b]); outside it they are defined as stated below.
 *
 * onssen_stft_logmag_ragged_f32: n_per_utt[b] samples of row b are valid (n_fft/2 < n_per_utt[b] <= n_max; the reflect
 *   padding mirrors about the row's own last sample); T = 1 + n_max/hop frames are written per row, the frames past
 *   1 + n_per_utt[b]/hop as the transform of silence (log-magnitude log10(eps), spectrum 0 -- to rounding, ~1e-16).
 * onssen_blstm_forward_ragged_f32: frames[b] <= T live frames of row b; at t >= frames[b] the row holds h = c = 0 in both
 *   directions (the reverse direction therefore starts from zero state at the row's own last frame) and its output rows are
 *   0 -- whatever the input holds there (a NaN of the padding does not spread).  ONSSEN_BLSTM_XCD | ONSSEN_BLSTM_BF16X3
 *   without ONSSEN_BLSTM_FUSE_IN0 / ONSSEN_BLSTM_BF16 runs the persistent form; the launch-per-step form takes ragged
 *   batches in both precisions; other flag combinations return ONSSEN_E_ARG.
 * onssen_dc_cluster_ragged_f32: utterance b owns the first frames[b]*F bins of its slab; its padding is never active
 *   (masks 0) and takes no part in the maximum, the initialisation or the sums.
 * onssen_mask_istft_ragged_f32: frames[b] frames and lengths[b] <= length output samples per row (librosa.istft(...,
 *   length = lengths[b]) semantics); out (B, C, length), zeros past lengths[b].
 * onssen_batch_sdr_ragged_f32: row b holds lengths[b] <= n samples (est, org, mask keep the row stride n).
 */
int onssen_stft_logmag_ragged_f32(const float* wav, int B, int n_max, int64_t wav_stride, const int32_t* n_per_utt, int n_fft,
                                  int hop, float eps, float* logmag, float* stft_ri, void* stream);
int onssen_blstm_forward_ragged_f32(const float* x, int64_t xs_b, int64_t xs_t, int B, int T, const int32_t* frames,
                                    int in_dim, int H, int L, int ug, const float* const* wih_p_host,
                                    const float* const* whh_p_host, const float* const* bias_p_host, float* y, void* ws,
                                    size_t ws_bytes, int flags, void* stream);
int onssen_dc_cluster_ragged_f32(const float* emb, const float* feature, int B, int T, const int32_t* frames, int F, int D,
                                 float db_threshold, int iters, float tol, float* masks, void* ws, size_t ws_bytes, int flags,
                                 void* stream);
int onssen_mask_istft_ragged_f32(const float* stft_ri, const float* mask, int64_t m_sb, int64_t m_sc, int64_t m_st,
                                 int64_t m_sf, int B, int C, int T, const int32_t* frames, int n_fft, int hop, int length,
                                 const int32_t* lengths, float* out, void* stream);
int onssen_batch_sdr_ragged_f32(const float* est, const float* org, const float* mask, int B, int C, int n,
                                const int32_t* lengths, float* sdr_out, int* perm_out, void* ws, size_t ws_bytes,
                                void* stream);

/* ---------------------------------------------------------------------------------------------
 * Deep-clustering separation without the embedding round trip (round 4).  Which bins are clustered is decided by the
 * mixture's log-magnitude alone (`m = np.max(feature_mix) - 40/20; emb = embedding[feature_mix >= m, :]`,
 * egs/wsj0-2mix/deep_clustering/evaluate.py:36-37) and is known before the network has run; the silent bins' embeddings
 * are never read (their masks are 0).  So:
 *   1. onssen_dc_index_f32 (after the STFT): per utterance the threshold, the number of active bins and a target map
 *      dest[b][t*F + f] = row of bin (t, f) in the utterance's compacted array, -1 for a silent bin (int32, inside ws);
 *   2. onssen_linear_x3p_compact: the fc_dc GEMM of onssen_linear_x3p with ONSSEN_EPI_L2NORM whose epilogue stores only the
 *      normalised rows of the active bins, each straight into its row of the compacted array (comp, inside ws);
 *   3. onssen_dc_cluster_compact_f32: farthest-point initialisation, all Lloyd iterations in one persistent launch and the
 *      mask pass on the compacted array -- the same arithmetic as onssen_dc_cluster_f32 in its default form, bit-identical
 *      masks, without the 10 320 B/frame embedding write, its re-read and the compaction pass.
 * ws: onssen_dc_compact_workspace_bytes() bytes, 256-byte aligned, the part in front of the compacted array zeroed by its
 * owner like onssen_dc_cluster_f32's (same status word: onssen_dc_cluster_status_offset); onssen_dc_compact_layout gives the
 * byte offsets of `comp` ([B][T*F][D] floats, rows past an utterance's active count unused) and `dest` ([B][T*F] int32).
 * frames: NULL, or per-utterance frame counts of a ragged batch (see the ragged entry points).
 * onssen_linear_x3p_compact: a_img / w_img / bias / N / group / eps as onssen_linear_x3p (N = F * group); row m of the GEMM is
 * (utterance m % R, frame m / R); dest_bs = ints per utterance in dest, comp_bs = floats per utterance in comp; dest must be
 * readable for 12 bytes past its last entry (the map is fetched in 16-byte words; the workspace layout above pads it). */
size_t onssen_dc_compact_workspace_bytes(int B, int T, int F, int D);
int onssen_dc_compact_layout(int B, int T, int F, int D, size_t* comp_offset, size_t* dest_offset);
int onssen_dc_index_f32(const float* feature, int B, int T, const int32_t* frames, int F, int D, float db_threshold, void* ws,
                        size_t ws_bytes, void* stream);
int onssen_linear_x3p_compact(const uint16_t* a_img, int M, int K, const uint16_t* w_img, const float* bias, int N, int group,
                              float eps, const int32_t* dest, int64_t dest_bs, int F, float* comp, int R, int64_t comp_bs,
                              int bf16_only, void* stream);
int onssen_dc_cluster_compact_f32(int B, int T, int F, int D, int iters, float tol, float* masks, void* ws, size_t ws_bytes,
                                  int flags, void* stream);

/* ---------------------------------------------------------------------------------------------
 * H1  host side of the data front end (round 5): a batch of RIFF/WAVE files -> float32 mono rows of one host buffer.
 * Replaces `librosa.load(fn, sr=None)` (onssen/data/feature_utils.py:15), called three times per training sample from
 * Dataset.__getitem__ on the training thread (onssen/data/wsj0_2mix.py:114-116, num_workers 0: :28-32).  HOST pointers
 * only; no device work, no stream.  onssen_wav_read_batch_f32 reads `count` files on up to `threads` host threads into
 * out_host + i * row_stride (floats; a pinned buffer makes the following H2D copy one asynchronous transfer):
 *   frames_host[i] = frames written (<= row_stride), rates_host[i] = the file's sample rate (resampling, if the rate is not
 *   the recipe's, stays with the caller as in feature_utils.py:17-20), status_host[i] = 0 | ONSSEN_WAV_TRUNCATED | ONSSEN_WAV_E_*.
 * Samples: PCM 8-bit (x - 128) / 128, 16-bit x / 2^15, 24-bit (left-justified in 32) and 32-bit x / 2^31, IEEE float 32 / 64
 * -> float32; several channels -> their float32 mean (what librosa.load(mono=True) returns for these files).
 * Returns ONSSEN_OK, ONSSEN_E_ARG, or ONSSEN_WAV_E_SOME_FAILED when a file failed (the others were still read).
 * onssen_wav_info: the header only; bits_host < 0 marks IEEE float. */
int onssen_wav_info(const char* path_host, int64_t* frames_host, int32_t* rate_host, int32_t* channels_host, int32_t* bits_host);
int onssen_wav_read_batch_f32(const char* const* paths_host, int count, float* out_host, int64_t row_stride, int32_t* frames_host,
                              int32_t* rates_host, int32_t* status_host, int threads);

/* Calibration probe (not part of the separation path): n dependent launches of a near-empty kernel with
 * `workgroups` x 256 threads on `stream`; bracket it with events to measure this box's launch-boundary floor. */
int onssen_debug_launch_chain(float* scratch, int n, int workgroups, void* stream);

/* Co-tenant probe (not part of the separation path; tools/cotenant_probe.py): `workgroups` x `threads` threads that only
 * hold their compute units for `ticks` ticks of the 100 MHz wall clock on `stream` -- a stand-in for RCCL's channel
 * kernels running beside the persistent recurrences (a recurrence group needs 30 of its XCD's 32 CUs at the same time).
 * heavy != 0: >= 112 live VGPRs per lane and 32 KB of LDS (cannot share a CU with a recurrence workgroup). */
int onssen_debug_cotenant_spin(int workgroups, int threads, long long ticks, int heavy, void* stream);

/* Bound of every wait inside the persistent kernels (ONSSEN_BLSTM_XCD forward, training forward / backward), in polling
 * passes: default 400000 (~0.2 s), initial value from ONSSEN_XCD_SPIN_LIMIT.  new_limit >= 0 sets it (0: every wait gives
 * up at once -- abort-path tests; data-parallel training raises it, see onssen_amd/dist.py), new_limit < 0 only queries.
 * Returns the previous value.  Process-wide; takes effect at the next launch. */
long long onssen_xcd_spin_limit(long long new_limit);

#ifdef __cplusplus
}
#endif
#endif /* ONSSEN_HIP_H */
