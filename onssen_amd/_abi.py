"""ctypes binding of the C ABI declared in include/onssen_hip.h.

Pointer arguments are plain integers (``tensor.data_ptr()``); nothing here
imports torch, so the same binding drives the product library
(libonssen_hip.so, device pointers) and the host-side kernel-test build
(tests/emu, host pointers).
"""
import ctypes as C
import os

EPI_BIAS, EPI_L2NORM, EPI_SIGMOID, EPI_RELU = 0, 1, 2, 3
EPI_BF16 = 0x100   # OR-ed into the mode of linear_x3p: plain bf16 products
ABI_VERSION = 14
WAV_TRUNCATED = 1
BLSTM_SPLIT_ROWS = 1
BLSTM_BF16X3 = 2
BLSTM_XCD = 4
BLSTM_FUSE_IN0 = 16
BLSTM_FUSE_TAIL = 32
BLSTM_BF16 = 64
BLSTM_G_READY = 128
BLSTM_WS_DIRTY = 65536
LSTM_BWD_STEPS, LSTM_BWD_XCD = 0, 1
DC_CLUSTER_LAUNCH_PER_ITERATION = 1
BLSTM_WS_HEADER = 32768   # ONSSEN_BLSTM_WS_HEADER_BYTES: zeroed once by the workspace owner

_vp, _i, _i64, _f, _sz = C.c_void_p, C.c_int, C.c_int64, C.c_float, C.c_size_t
_pp = C.POINTER(C.c_void_p)

# name -> (restype, argtypes); mirrors include/onssen_hip.h one to one
SIGNATURES = {
    "onssen_abi_version": (_i, []),
    "onssen_error_string": (C.c_char_p, [_i]),
    "onssen_stft_logmag_f32": (_i, [_vp, _i, _i, _i64, _i, _i, _f, _vp, _vp, _vp]),
    "onssen_lstm_geometry": (_i, [_i, _i, C.POINTER(_i), C.POINTER(_i), C.POINTER(_i), C.POINTER(_i64)]),
    "onssen_lstm_geometry_x3": (_i, [_i, _i, C.POINTER(_i), C.POINTER(_i), C.POINTER(_i64)]),
    "onssen_lstm_pack_whh_bf16x3": (_i, [_vp, _i, _i, _vp, _vp]),
    "onssen_lstm_pack_wih_bf16x3": (_i, [_vp, _i, _i, _i, _vp, _vp]),
    "onssen_lstm_pack_f32": (_i, [_vp, _vp, _vp, _vp, _i, _i, _i, _i, _vp, _vp, _vp, _vp]),
    "onssen_lstm_pack_train_f32": (_i, [_i, _i, _i, _i, _vp, _vp, _vp, _vp, _vp, _vp, _vp, _vp, _vp, _vp]),
    "onssen_lstm_pack_wih_image_f32": (_i, [_vp, _vp, _vp, _i, _i, _i, _i, _vp, _vp, _vp, _vp]),
    "onssen_clip_adam_workspace_bytes": (_sz, [_vp, _i]),
    "onssen_clip_adam_f32": (_i, [_i, _vp, _vp, _vp, _vp, _vp, _f, C.c_double, C.c_double, C.c_double, C.c_double, _i, _i, _vp, _sz, _vp]),
    "onssen_head_pack_f32": (_i, [_vp, _vp, _i, _i, _i, _vp, _vp, _vp, _vp, _f, _vp, _vp, _vp]),
    "onssen_linear_f32": (_i, [_vp, _i64, _i64, _i, _i, _i, _vp, _i, _vp, _i, _i, _i, _f, _vp, _vp, _i64, _i64, _vp]),
    "onssen_linear_pack_bf16x3": (_i, [_vp, _i, _i, _i, _i, _vp, _vp]),
    "onssen_linear_bf16x3": (_i, [_vp, _i64, _i64, _i, _i, _i, _vp, _i, _vp, _i, _i, _i, _f, _vp, _vp, _i64, _i64, _vp]),
    "onssen_x3_image_f32": (_i, [_vp, _i64, _i64, _i, _i, _i, _vp, _vp]),
    "onssen_linear_x3p": (_i, [_vp, _i, _i, _vp, _vp, _i, _i, _i, _f, _vp, _i, _i64, _i64, _vp]),
    "onssen_linear_x3p_resid": (_i, [_vp, _i, _i, _vp, _vp, _i, _i, _f, _vp, _i, _vp, _i, _i64, _i64, _i, _vp]),
    "onssen_linear_x3p_pair": (_i, [_vp, _i, _i, _vp, _vp, _i, _i, _i, _f, _vp, _i, _i64, _i64, _vp, _i64, _i64, _i, _vp]),
    "onssen_blstm_workspace_bytes": (_sz, [_i, _i, _i, _i, _i, _i]),
    "onssen_blstm_y_image": (_i, [_i, _i, _i, _i, _i, _i, _vp, _vp]),
    "onssen_blstm_x_image": (_i, [_i, _i, _i, _i, _i, _i, _vp, _vp]),
    "onssen_blstm_forward_f32": (_i, [_vp, _i64, _i64, _i, _i, _i, _i, _i, _i, _pp, _pp, _pp, _vp, _vp, _sz, _i, _vp]),
    "onssen_blstm_pipe2_workspace_bytes": (_sz, [_i, _i, _i, _i, _i]),
    "onssen_blstm_pipe2_y_image": (_i, [_i, _i, _i, _i, _i, _vp, _vp]),
    "onssen_blstm_pipe2_forward_f32": (_i, [_vp, _i64, _i64, _i, _i, _i, _i, _i, _pp, _pp, _pp, _vp, _sz, _i, _vp]),
    "onssen_blstm_pipe2_forward_ragged_f32": (_i, [_vp, _i64, _i64, _i, _i, _i, _vp, _i, _vp, _i, _i, _i, _pp, _pp, _pp, _vp, _sz, _i, _vp]),
    "onssen_linear_x3p_batched": (_i, [_vp, _i64, _i, _i, _vp, _i64, _vp, _i, _vp, _i64, _i64, _i, _vp]),
    "onssen_linear_x3p_batched_split": (_i, [_vp, _i64, _i, _i, _vp, _i64, _vp, _i, _i, _vp, _i64, _i64, _i64, _i, _vp, _i64, _i64, _i64,
                                             _i, _vp]),
    "onssen_linear_x3p_batched_split_alt": (_i, [_vp, _i64, _i, _i, _vp, _i64, _vp, _i, _i, _vp, _i64, _i64, _i64, _i, _vp, _i64, _i64,
                                                 _i64, _i, _i, _vp]),
    "onssen_linear_x3t": (_i, [_vp, _vp, _i, _i, _i, _vp, _vp, _i64, _vp]),
    "onssen_lstm_wgrad_images_f32": (_i, [_vp, _vp, _vp, _i, _i, _i, _i, _i, _vp, _i, _vp, _i64, _i64, _i64, _vp, _i64, _i64, _i64, _vp]),
    "onssen_x3_image_t_f32": (_i, [_vp, _i64, _i, _i, _i, _vp, _vp]),
    "onssen_x3_image_both_f32": (_i, [_vp, _i64, _i, _i, _vp, _vp, _vp]),
    "onssen_x3_image_both_colsum_f32": (_i, [_vp, _i64, _i, _i, _vp, _vp, _vp, _vp]),
    "onssen_lstm_train_forward_f32": (_i, [_vp, _i64, _i64, _i, _i, _i, _i, _i, _vp, _vp, _vp, _vp, _vp, _vp, _vp, _sz, _vp]),
    "onssen_lstm_train_forward_form_f32": (_i, [_vp, _i64, _i64, _i, _i, _i, _i, _i, _vp, _vp, _vp, _vp, _vp, _vp, _vp, _sz, _i, _vp]),
    "onssen_lstm_whhT_elems": (_i64, [_i, _i]),
    "onssen_lstm_pack_whhT_bf16x3": (_i, [_vp, _i, _i, _vp, _vp]),
    "onssen_lstm_whhR_elems": (_i64, [_i, _i]),
    "onssen_lstm_pack_whhR_bf16x3": (_i, [_vp, _i, _i, _vp, _vp]),
    "onssen_lstm_train_backward_workspace_bytes": (_sz, [_i, _i, _i, _i]),
    "onssen_lstm_train_backward_f32": (_i, [_i, _i, _i, _i, _vp, _vp, _vp, _vp, _vp, _sz, _i, _vp, _vp]),
    "onssen_lstm_train_backward_img_f32": (_i, [_i, _i, _i, _i, _vp, _vp, _vp, _vp, _vp, _sz, _vp, _vp, _vp]),
    "onssen_dropout_f32": (_i, [_vp, _i64, _f, C.c_uint64, _vp, _vp]),
    "onssen_l2norm_rows_f32": (_i, [_vp, _i64, _i, _f, _vp, _vp]),
    "onssen_bn_rows_workspace_bytes": (_sz, [_i64, _i]),
    "onssen_bn_rows_train_f32": (_i, [_vp, _i64, _i, _vp, _vp, _f, _vp, _vp, _vp, _vp, _sz, _vp]),
    "onssen_bn_rows_grad_f32": (_i, [_vp, _vp, _i64, _i, _vp, _vp, _vp, _vp, _vp, _vp, _vp, _sz, _vp]),
    "onssen_l2norm_rows_grad_f32": (_i, [_vp, _vp, _i64, _i, _f, _vp, _vp]),
    "onssen_l2norm_rows_grad_y_f32": (_i, [_vp, _vp, _vp, _i64, _i, _f, _vp, _vp]),
    "onssen_linear_x3p_norms": (_i, [_vp, _i, _i, _vp, _vp, _i, _i, _f, _vp, _vp, _vp]),
    "onssen_phase_input_f32": (_i, [_vp, _vp, _i64, _i64, _i64, _i64, _vp, _i, _i, _i, _i, _vp, _vp]),
    "onssen_debug_launch_chain": (_i, [_vp, _i, _i, _vp]),
    "onssen_debug_cotenant_spin": (_i, [_i, _i, C.c_longlong, _i, _vp]),
    "onssen_xcd_spin_limit": (C.c_longlong, [C.c_longlong]),
    "onssen_wav_info": (_i, [C.c_char_p, C.POINTER(_i64), C.POINTER(C.c_int32), C.POINTER(C.c_int32), C.POINTER(C.c_int32)]),
    "onssen_wav_read_batch_f32": (_i, [C.POINTER(C.c_char_p), _i, _vp, _i64, _vp, _vp, _vp, _i]),
    "onssen_labels_f32": (_i, [_vp, _vp, _vp, _vp, _i, _i, _i, _f, _vp, _vp, _vp, _vp, _vp, _vp, _vp, _vp]),
    "onssen_param_guard_u32": (_i, [_vp, _vp, _i, _i, _i, _vp, _vp, _vp]),
    "onssen_log_magnitude_f32": (_i, [_vp, _i64, _f, _vp, _vp]),
    "onssen_cos_difference_f32": (_i, [_vp, _vp, _i64, _vp, _vp]),
    "onssen_one_hot_f32": (_i, [_vp, _vp, _vp, _i, _i64, _f, _vp, _vp, _vp]),
    "onssen_dc_cluster_workspace_bytes": (_sz, [_i, _i, _i, _i]),
    "onssen_dc_cluster_status_offset": (_sz, [_i, _i]),
    "onssen_loss_mask_workspace_bytes": (_sz, [_i]),
    "onssen_loss_mask_f32": (_i, [_vp, _vp, _i64, _i64, _vp, _vp, _vp, _vp, _vp, _i, _i, _vp, _vp, _vp, _sz, _vp]),
    "onssen_loss_mask_grad_f32": (_i, [_vp, _vp, _i64, _i64, _vp, _vp, _vp, _vp, _vp, _i, _i, _vp, _vp, _vp, _vp, _i64, _i64, _vp]),
    "onssen_loss_dc_workspace_bytes": (_sz, [_i]),
    "onssen_batch_sdr_workspace_bytes": (_sz, [_i]),
    "onssen_batch_sdr_f32": (_i, [_vp, _vp, _vp, _i, _i, _i, _vp, _vp, _vp, _sz, _vp]),
    "onssen_loss_dc_f32": (_i, [_vp, _vp, _vp, _i, _i, _i, _i, _vp, _vp, _vp, _sz, _vp]),
    "onssen_loss_dc_grad_f32": (_i, [_vp, _vp, _vp, _i, _i, _i, _i, _vp, _vp, _vp, _sz, _vp]),
    "onssen_dc_head_grad_images_f32": (_i, [_vp, _vp, _vp, _vp, _i, _i, _i, _i, _i, _f, _vp, _vp, _sz, _vp, _vp, _vp, _vp]),
    "onssen_dc_cluster_f32": (_i, [_vp, _vp, _i, _i, _i, _i, _f, _i, _f, _vp, _vp, _sz, _i, _vp]),
    "onssen_mask_istft_f32": (_i, [_vp, _vp, _i64, _i64, _i64, _i64, _i, _i, _i, _i, _i, _i, _vp, _vp]),
    # deep-clustering separation without the embedding round trip (round 4)
    "onssen_dc_compact_workspace_bytes": (_sz, [_i, _i, _i, _i]),
    "onssen_dc_compact_layout": (_i, [_i, _i, _i, _i, _vp, _vp]),
    "onssen_dc_index_f32": (_i, [_vp, _i, _i, _vp, _i, _i, _f, _vp, _sz, _vp]),
    "onssen_linear_x3p_compact": (_i, [_vp, _i, _i, _vp, _vp, _i, _i, _f, _vp, _i64, _i, _vp, _i, _i64, _i, _vp]),
    "onssen_dc_cluster_compact_f32": (_i, [_i, _i, _i, _i, _i, _f, _vp, _vp, _sz, _i, _vp]),
    # ragged batches of whole utterances (round 4)
    "onssen_stft_logmag_ragged_f32": (_i, [_vp, _i, _i, _i64, _vp, _i, _i, _f, _vp, _vp, _vp]),
    "onssen_blstm_forward_ragged_f32": (_i, [_vp, _i64, _i64, _i, _i, _vp, _i, _i, _i, _i, _pp, _pp, _pp, _vp, _vp, _sz, _i, _vp]),
    "onssen_dc_cluster_ragged_f32": (_i, [_vp, _vp, _i, _i, _vp, _i, _i, _f, _i, _f, _vp, _vp, _sz, _i, _vp]),
    "onssen_mask_istft_ragged_f32": (_i, [_vp, _vp, _i64, _i64, _i64, _i64, _i, _i, _i, _vp, _i, _i, _i, _vp, _vp, _vp]),
    "onssen_batch_sdr_ragged_f32": (_i, [_vp, _vp, _vp, _i, _i, _i, _vp, _vp, _vp, _vp, _sz, _vp]),
}


class OnssenError(RuntimeError):
    pass


class Lib:
    """Thin, checked wrapper around a loaded libonssen_*.so."""

    def __init__(self, path):
        self.path = path
        self.dll = C.CDLL(path)
        for name, (res, args) in SIGNATURES.items():
            fn = getattr(self.dll, name)   # AttributeError if the symbol is not exported
            fn.restype, fn.argtypes = res, args
        if self.dll.onssen_abi_version() != ABI_VERSION:
            raise OnssenError(f"{path}: ABI version {self.dll.onssen_abi_version()} != {ABI_VERSION}")

    def check(self, rc, what):
        if rc != 0:
            msg = self.dll.onssen_error_string(rc)
            raise OnssenError(f"{what} failed with code {rc}: {msg.decode() if msg else '?'}")

    # ---- geometry -------------------------------------------------------
    def lstm_geometry(self, H, ug):
        hp, np_, kq, we = _i(), _i(), _i(), _i64()
        self.check(self.dll.onssen_lstm_geometry(H, ug, hp, np_, kq, we), "onssen_lstm_geometry")
        return hp.value, np_.value, kq.value, we.value

    def lstm_geometry_x3(self, H, ug):
        kq2, hs, we = _i(), _i(), _i64()
        self.check(self.dll.onssen_lstm_geometry_x3(H, ug, kq2, hs, we), "onssen_lstm_geometry_x3")
        return kq2.value, hs.value, we.value

    def lstm_pack_whh_bf16x3(self, w_hh, H, ug, out, stream):
        self.check(self.dll.onssen_lstm_pack_whh_bf16x3(w_hh, H, ug, out, stream), "onssen_lstm_pack_whh_bf16x3")

    def lstm_pack_wih_bf16x3(self, w_ih, in_dim, H, ug, out, stream):
        self.check(self.dll.onssen_lstm_pack_wih_bf16x3(w_ih, in_dim, H, ug, out, stream), "onssen_lstm_pack_wih_bf16x3")

    # ---- host-side wav reader (no device work) ---------------------------
    def wav_info(self, path):
        """(frames, rate, channels, bits) of a RIFF/WAVE file's header; bits < 0 marks IEEE float."""
        fr, rate, ch, bits = _i64(), C.c_int32(), C.c_int32(), C.c_int32()
        self.check(self.dll.onssen_wav_info(os.fsencode(path), fr, rate, ch, bits), f"onssen_wav_info({path!r})")
        return fr.value, rate.value, ch.value, bits.value

    def wav_read_batch(self, paths, out_ptr, row_stride, frames_ptr, rates_ptr, status_ptr, threads):
        """Read ``paths`` into rows of a host float32 buffer (see include/onssen_hip.h).  Returns the library's code: 0, or
        ONSSEN_WAV_E_SOME_FAILED (-19) with the per-file status telling which; other codes raise."""
        arr = (C.c_char_p * len(paths))(*[os.fsencode(p) for p in paths])
        rc = self.dll.onssen_wav_read_batch_f32(arr, len(paths), out_ptr, row_stride, frames_ptr, rates_ptr, status_ptr, threads)
        if rc not in (0, -19):
            self.check(rc, "onssen_wav_read_batch_f32")
        return rc

    def batch_sdr_workspace_bytes(self, B):
        return int(self.dll.onssen_batch_sdr_workspace_bytes(B))

    def batch_sdr(self, est, org, mask, B, Cn, n, sdr_out, perm_out, ws, ws_bytes, stream, lengths=None):
        if lengths is not None:      # ragged batch: row b holds lengths[b] <= n samples
            self.check(self.dll.onssen_batch_sdr_ragged_f32(est, org, mask, B, Cn, n, lengths, sdr_out, perm_out, ws, ws_bytes,
                                                            stream), "onssen_batch_sdr_ragged_f32")
            return
        self.check(self.dll.onssen_batch_sdr_f32(est, org, mask, B, Cn, n, sdr_out, perm_out, ws, ws_bytes, stream),
                   "onssen_batch_sdr_f32")

    def loss_mask_workspace_bytes(self, B):
        return int(self.dll.onssen_loss_mask_workspace_bytes(B))

    def loss_mask(self, ma, mb, m_sb, m_se, mag, s1, s2, c1, c2, B, TF, out, ws, ws_bytes, stream, perm=None):
        self.check(self.dll.onssen_loss_mask_f32(ma, mb, m_sb, m_se, mag, s1, s2, c1, c2, B, TF, out, perm, ws, ws_bytes, stream),
                   "onssen_loss_mask_f32")

    def loss_mask_grad(self, ma, mb, m_sb, m_se, mag, s1, s2, c1, c2, B, TF, g, perm, da, db, d_sb, d_se, stream):
        self.check(self.dll.onssen_loss_mask_grad_f32(ma, mb, m_sb, m_se, mag, s1, s2, c1, c2, B, TF, g, perm, da, db, d_sb, d_se,
                                                      stream), "onssen_loss_mask_grad_f32")

    def loss_dc_workspace_bytes(self, B):
        return int(self.dll.onssen_loss_dc_workspace_bytes(B))

    def loss_dc(self, emb, one_hot, mag, B, TF, D, Cc, per_utt, total_mag, ws, ws_bytes, stream):
        self.check(self.dll.onssen_loss_dc_f32(emb, one_hot, mag, B, TF, D, Cc, per_utt, total_mag, ws, ws_bytes, stream),
                   "onssen_loss_dc_f32")

    def loss_dc_grad(self, emb, one_hot, mag, B, TF, D, Cc, g_per_utt, d_emb, ws, ws_bytes, stream):
        self.check(self.dll.onssen_loss_dc_grad_f32(emb, one_hot, mag, B, TF, D, Cc, g_per_utt, d_emb, ws, ws_bytes, stream),
                   "onssen_loss_dc_grad_f32")

    def blstm_workspace_bytes(self, B, T, in_dim, H, L, ug):
        return int(self.dll.onssen_blstm_workspace_bytes(B, T, in_dim, H, L, ug))

    def blstm_y_image(self, B, T, in_dim, H, L, ug):
        """(byte offset inside the workspace, KB) of the last layer's x3 output image (ONSSEN_BLSTM_XCD form)."""
        off, kb = C.c_size_t(0), C.c_int(0)
        self.check(self.dll.onssen_blstm_y_image(B, T, in_dim, H, L, ug, C.byref(off), C.byref(kb)), "onssen_blstm_y_image")
        return int(off.value), int(kb.value)

    def blstm_x_image(self, B, T, in_dim, H, L, ug):
        """(byte offset inside the workspace, KB) of the last layer's x3 output image (ONSSEN_BLSTM_XCD form)."""
        off, kb = C.c_size_t(0), C.c_int(0)
        self.check(self.dll.onssen_blstm_x_image(B, T, in_dim, H, L, ug, C.byref(off), C.byref(kb)), "onssen_blstm_x_image")
        return int(off.value), int(kb.value)

    # ---- kernels (pointers are ints) --------------------------------------
    def stft_logmag(self, wav, B, n, stride, n_fft, hop, eps, logmag, stft_ri, stream, n_per_utt=None):
        if n_per_utt is not None:    # ragged batch: n = the longest row, n_per_utt[b] the valid samples of row b
            self.check(self.dll.onssen_stft_logmag_ragged_f32(wav, B, n, stride, n_per_utt, n_fft, hop, eps, logmag, stft_ri, stream),
                       "onssen_stft_logmag_ragged_f32")
            return
        self.check(self.dll.onssen_stft_logmag_f32(wav, B, n, stride, n_fft, hop, eps, logmag, stft_ri, stream),
                   "onssen_stft_logmag_f32")

    def lstm_pack(self, w_ih, w_hh, b_ih, b_hh, in_dim, bidir_in, H, ug, wih_p, whh_p, bias_p, stream):
        self.check(self.dll.onssen_lstm_pack_f32(w_ih, w_hh, b_ih, b_hh, in_dim, bidir_in, H, ug, wih_p, whh_p,
                                                 bias_p, stream), "onssen_lstm_pack_f32")

    def lstm_pack_train(self, L, in_dim, H, ug, w_ih, w_hh, b_ih, b_hh, wih_p, bias_p, wih_img, whh_x3, whhR, stream):
        """Lists of 2 L device pointers, index 2 l + d (whhR may be None): include/onssen_hip.h: onssen_lstm_pack_train_f32."""
        arr = lambda xs: (C.c_void_p * (2 * L))(*xs) if xs is not None else None
        self.check(self.dll.onssen_lstm_pack_train_f32(L, in_dim, H, ug, arr(w_ih), arr(w_hh), arr(b_ih), arr(b_hh), arr(wih_p), arr(bias_p),
                                                       arr(wih_img), arr(whh_x3), arr(whhR), stream), "onssen_lstm_pack_train_f32")

    def lstm_pack_wih_image(self, w_ih, b_ih, b_hh, in_dim, bidir_in, H, ug, wih_p, bias_p, wih_img, stream):
        self.check(self.dll.onssen_lstm_pack_wih_image_f32(w_ih, b_ih, b_hh, in_dim, bidir_in, H, ug, wih_p, bias_p, wih_img, stream),
                   "onssen_lstm_pack_wih_image_f32")

    def clip_adam(self, p, g, m, v, numel, max_norm, lr, beta1, beta2, eps, step, ws, ws_bytes, stream, write_grads=False):
        """p / g / m / v: lists of device pointers; numel: list of element counts (include/onssen_hip.h: onssen_clip_adam_f32)."""
        n = len(p)
        arr = lambda xs: (C.c_void_p * n)(*xs)
        ne = (C.c_int64 * n)(*numel)
        self.check(self.dll.onssen_clip_adam_f32(n, arr(p), arr(g), arr(m), arr(v), ne, max_norm, lr, beta1, beta2, eps, step,
                                                 1 if write_grads else 0, ws, ws_bytes, stream), "onssen_clip_adam_f32")

    def clip_adam_workspace_bytes(self, numel):
        ne = (C.c_int64 * len(numel))(*numel)
        return int(self.dll.onssen_clip_adam_workspace_bytes(ne, len(numel)))

    def head_pack(self, w, b, N, H, Hp, g, beta, mean, var, bn_eps, w_p, b_p, stream):
        self.check(self.dll.onssen_head_pack_f32(w, b, N, H, Hp, g, beta, mean, var, bn_eps, w_p, b_p, stream),
                   "onssen_head_pack_f32")

    def linear(self, A, a_s0, a_s1, R, M, K, W, ldw, bias, N, mode, group, eps, resid, Cp, c_s0, c_s1, stream):
        self.check(self.dll.onssen_linear_f32(A, a_s0, a_s1, R, M, K, W, ldw, bias, N, mode, group, eps, resid, Cp,
                                              c_s0, c_s1, stream), "onssen_linear_f32")

    def linear_pack_bf16x3(self, w, N, K, ld_in, ld_out, planes, stream):
        self.check(self.dll.onssen_linear_pack_bf16x3(w, N, K, ld_in, ld_out, planes, stream),
                   "onssen_linear_pack_bf16x3")

    def linear_bf16x3(self, A, a_s0, a_s1, R, M, K, planes, ldw, bias, N, mode, group, eps, resid, Cp, c_s0, c_s1,
                      stream):
        self.check(self.dll.onssen_linear_bf16x3(A, a_s0, a_s1, R, M, K, planes, ldw, bias, N, mode, group, eps, resid,
                                                 Cp, c_s0, c_s1, stream), "onssen_linear_bf16x3")

    def x3_image(self, src, s0, s1, R, rows, K, img, stream):
        self.check(self.dll.onssen_x3_image_f32(src, s0, s1, R, rows, K, img, stream), "onssen_x3_image_f32")

    def linear_x3p(self, a_img, M, K, w_img, bias, N, mode, group, eps, Cp, R, c_s0, c_s1, stream):
        self.check(self.dll.onssen_linear_x3p(a_img, M, K, w_img, bias, N, mode, group, eps, Cp, R, c_s0, c_s1, stream),
                   "onssen_linear_x3p")

    def linear_x3p_pair(self, a_img, M, K, w_img, bias, N, n_split, group, eps, Cp, R, c_s0, c_s1, C2p, c2_s0, c2_s1, bf16_only, stream):
        self.check(self.dll.onssen_linear_x3p_pair(a_img, M, K, w_img, bias, N, n_split, group, eps, Cp, R, c_s0, c_s1, C2p, c2_s0,
                                                   c2_s1, int(bool(bf16_only)), stream), "onssen_linear_x3p_pair")

    def linear_x3p_resid(self, a_img, M, K, w_img, bias, N, group, eps, resid, resid_mod, Cp, R, c_s0, c_s1, bf16_only, stream):
        self.check(self.dll.onssen_linear_x3p_resid(a_img, M, K, w_img, bias, N, group, eps, resid, resid_mod, Cp, R, c_s0, c_s1,
                                                    int(bool(bf16_only)), stream), "onssen_linear_x3p_resid")

    def blstm_forward(self, x, xs_b, xs_t, B, T, in_dim, H, L, ug, wih_ptrs, whh_ptrs, bias_ptrs, y, ws, ws_bytes,
                      flags, stream, frames=None):
        arr = C.c_void_p * L
        if frames is not None:       # ragged batch: frames[b] <= T live frames of row b
            self.check(self.dll.onssen_blstm_forward_ragged_f32(x, xs_b, xs_t, B, T, frames, in_dim, H, L, ug, arr(*wih_ptrs),
                                                                arr(*whh_ptrs), arr(*bias_ptrs), y, ws, ws_bytes, flags, stream),
                       "onssen_blstm_forward_ragged_f32")
            return
        self.check(self.dll.onssen_blstm_forward_f32(x, xs_b, xs_t, B, T, in_dim, H, L, ug, arr(*wih_ptrs),
                                                     arr(*whh_ptrs), arr(*bias_ptrs), y, ws, ws_bytes, flags, stream),
                   "onssen_blstm_forward_f32")

    # ---- two-layer stack software-pipelined over consecutive calls (round 6)
    def blstm_pipe2_workspace_bytes(self, B, T, in_dim, H, ug):
        return int(self.dll.onssen_blstm_pipe2_workspace_bytes(B, T, in_dim, H, ug))

    def blstm_pipe2_y_image(self, B, T, in_dim, H, ug):
        """(byte offset inside the workspace, KB) of layer 1's x3 output image of the batch handed over one call earlier."""
        off, kb = C.c_size_t(0), C.c_int(0)
        self.check(self.dll.onssen_blstm_pipe2_y_image(B, T, in_dim, H, ug, C.byref(off), C.byref(kb)), "onssen_blstm_pipe2_y_image")
        return int(off.value), int(kb.value)

    def blstm_pipe2_forward(self, x, xs_b, xs_t, B, T, in_dim, H, ug, wih_ptrs, whh_ptrs, bias_ptrs, ws, ws_bytes, flags, stream):
        arr = C.c_void_p * 2
        self.check(self.dll.onssen_blstm_pipe2_forward_f32(x, xs_b, xs_t, B, T, in_dim, H, ug, arr(*wih_ptrs), arr(*whh_ptrs),
                                                           arr(*bias_ptrs), ws, ws_bytes, flags, stream),
                   "onssen_blstm_pipe2_forward_f32")

    def blstm_pipe2_forward_ragged(self, x, xs_b, xs_t, B, T_cap, T, frames, T_prev, frames_prev, in_dim, H, ug, wih_ptrs, whh_ptrs,
                                   bias_ptrs, ws, ws_bytes, flags, stream):
        """The pair launch over a stream of ragged batches: this call's batch (T, frames) beside the one before (T_prev, frames_prev)."""
        arr = C.c_void_p * 2
        self.check(self.dll.onssen_blstm_pipe2_forward_ragged_f32(x, xs_b, xs_t, B, T_cap, T, frames, T_prev, frames_prev, in_dim, H, ug,
                                                                  arr(*wih_ptrs), arr(*whh_ptrs), arr(*bias_ptrs), ws, ws_bytes, flags,
                                                                  stream),
                   "onssen_blstm_pipe2_forward_ragged_f32")

    # ---- training (row N1)
    def linear_x3p_batched(self, a_img, a_bs, M, K, w_img, w_bs, bias, N, out, c_bs, ldc, batch, stream):
        self.check(self.dll.onssen_linear_x3p_batched(a_img, a_bs, M, K, w_img, w_bs, bias, N, out, c_bs, ldc, batch, stream),
                   "onssen_linear_x3p_batched")

    def linear_x3p_batched_split(self, a_img, a_bs, M, K, w_img, w_bs, bias, N, R, out, c_bs, c_s0, c_s1, n_split, out2, c2_bs,
                                 c2_s0, c2_s1, batch, stream):
        self.check(self.dll.onssen_linear_x3p_batched_split(a_img, a_bs, M, K, w_img, w_bs, bias, N, R, out, c_bs, c_s0, c_s1, n_split,
                                                            out2, c2_bs, c2_s0, c2_s1, batch, stream), "onssen_linear_x3p_batched_split")

    def linear_x3p_batched_split_alt(self, a_img, a_bs, M, K, w_img, w_bs, bias, N, R, out, c_bs, c_s0, c_s1, n_split, out2, c2_bs,
                                     c2_s0, c2_s1, n_split_odd, batch, stream):
        self.check(self.dll.onssen_linear_x3p_batched_split_alt(a_img, a_bs, M, K, w_img, w_bs, bias, N, R, out, c_bs, c_s0, c_s1,
                                                                n_split, out2, c2_bs, c2_s0, c2_s1, n_split_odd, batch, stream),
                   "onssen_linear_x3p_batched_split_alt")

    def linear_x3t(self, a_img, w_img, K, M, N, zero16, Cp, ldc, stream):
        self.check(self.dll.onssen_linear_x3t(a_img, w_img, K, M, N, zero16, Cp, ldc, stream), "onssen_linear_x3t")

    def lstm_wgrad_images(self, dp_img, y_img, x_img, K, B, NP, Hp, Kx, zero16, R, dW_ih, ih_bs, ih_s0, ih_s1, dW_hh, hh_bs, hh_s0,
                          hh_s1, stream):
        self.check(self.dll.onssen_lstm_wgrad_images_f32(dp_img, y_img, x_img, K, B, NP, Hp, Kx, zero16, R, dW_ih, ih_bs, ih_s0, ih_s1,
                                                         dW_hh, hh_bs, hh_s0, hh_s1, stream), "onssen_lstm_wgrad_images_f32")

    def x3_image_t(self, src, ld, M, K, k_shift, img, stream):
        self.check(self.dll.onssen_x3_image_t_f32(src, ld, M, K, k_shift, img, stream), "onssen_x3_image_t_f32")

    def x3_image_both(self, src, ld, M, K, img_rows, img_t, stream):
        self.check(self.dll.onssen_x3_image_both_f32(src, ld, M, K, img_rows, img_t, stream), "onssen_x3_image_both_f32")

    def x3_image_both_colsum(self, src, ld, M, K, img_rows, img_t, colsum, stream):
        self.check(self.dll.onssen_x3_image_both_colsum_f32(src, ld, M, K, img_rows, img_t, colsum, stream),
                   "onssen_x3_image_both_colsum_f32")

    def lstm_train_forward(self, x, xs_b, xs_t, B, T, in_dim, H, ug, wih_img, whh_x3, bias, y, gates, cs, ws, ws_bytes, stream):
        self.check(self.dll.onssen_lstm_train_forward_f32(x, xs_b, xs_t, B, T, in_dim, H, ug, wih_img, whh_x3, bias, y, gates,
                                                          cs, ws, ws_bytes, stream), "onssen_lstm_train_forward_f32")

    def lstm_train_forward_form(self, x, xs_b, xs_t, B, T, in_dim, H, ug, wih, whh, bias, y, gates, cs, ws, ws_bytes, flags, stream):
        self.check(self.dll.onssen_lstm_train_forward_form_f32(x, xs_b, xs_t, B, T, in_dim, H, ug, wih, whh, bias, y, gates,
                                                               cs, ws, ws_bytes, flags, stream), "onssen_lstm_train_forward_form_f32")

    def lstm_whhT_elems(self, H, ug):
        return int(self.dll.onssen_lstm_whhT_elems(H, ug))

    def lstm_pack_whhT_bf16x3(self, w_hh, H, ug, out, stream):
        self.check(self.dll.onssen_lstm_pack_whhT_bf16x3(w_hh, H, ug, out, stream), "onssen_lstm_pack_whhT_bf16x3")

    def lstm_whhR_elems(self, H, ug):
        return int(self.dll.onssen_lstm_whhR_elems(H, ug))

    def lstm_pack_whhR_bf16x3(self, w_hh, H, ug, out, stream):
        self.check(self.dll.onssen_lstm_pack_whhR_bf16x3(w_hh, H, ug, out, stream), "onssen_lstm_pack_whhR_bf16x3")

    def lstm_train_backward_workspace_bytes(self, B, H, ug, form):
        return int(self.dll.onssen_lstm_train_backward_workspace_bytes(B, H, ug, form))

    def lstm_train_backward(self, B, T, H, ug, whh_img, dy, gates_dp, cs, ws, ws_bytes, form, stream, db_rows=None):
        self.check(self.dll.onssen_lstm_train_backward_f32(B, T, H, ug, whh_img, dy, gates_dp, cs, ws, ws_bytes, form, db_rows, stream),
                   "onssen_lstm_train_backward_f32")

    def lstm_train_backward_img(self, B, T, H, ug, whh_img, dy, gates, cs, ws, ws_bytes, stream, db_rows, dp_img):
        self.check(self.dll.onssen_lstm_train_backward_img_f32(B, T, H, ug, whh_img, dy, gates, cs, ws, ws_bytes, db_rows, dp_img, stream),
                   "onssen_lstm_train_backward_img_f32")

    def bn_rows_workspace_bytes(self, M, Cc):
        return int(self.dll.onssen_bn_rows_workspace_bytes(M, Cc))

    def bn_rows_train(self, x, M, Cc, gamma, beta, eps, y, mean, invstd, ws, ws_bytes, stream):
        self.check(self.dll.onssen_bn_rows_train_f32(x, M, Cc, gamma, beta, eps, y, mean, invstd, ws, ws_bytes, stream),
                   "onssen_bn_rows_train_f32")

    def bn_rows_grad(self, x, dy, M, Cc, gamma, mean, invstd, dx, dgamma, dbeta, ws, ws_bytes, stream):
        self.check(self.dll.onssen_bn_rows_grad_f32(x, dy, M, Cc, gamma, mean, invstd, dx, dgamma, dbeta, ws, ws_bytes, stream),
                   "onssen_bn_rows_grad_f32")

    def l2norm_rows(self, x, rows, D, eps, y, stream):
        self.check(self.dll.onssen_l2norm_rows_f32(x, rows, D, eps, y, stream), "onssen_l2norm_rows_f32")

    def l2norm_rows_grad(self, x, g, rows, D, eps, dx, stream):
        self.check(self.dll.onssen_l2norm_rows_grad_f32(x, g, rows, D, eps, dx, stream), "onssen_l2norm_rows_grad_f32")

    def dc_head_grad_images(self, emb, inv_norm, one_hot, mag, B, T, F, D, Cc, eps, g_per_utt, ws, ws_bytes, img_rows, img_t, colsum, stream):
        self.check(self.dll.onssen_dc_head_grad_images_f32(emb, inv_norm, one_hot, mag, B, T, F, D, Cc, eps, g_per_utt, ws, ws_bytes,
                                                           img_rows, img_t, colsum, stream), "onssen_dc_head_grad_images_f32")

    def l2norm_rows_grad_y(self, y, inv_norm, g, rows, D, eps, dx, stream):
        self.check(self.dll.onssen_l2norm_rows_grad_y_f32(y, inv_norm, g, rows, D, eps, dx, stream), "onssen_l2norm_rows_grad_y_f32")

    def linear_x3p_norms(self, a_img, M, K, w_img, bias, N, group, eps, Cp, inv_norm, stream):
        self.check(self.dll.onssen_linear_x3p_norms(a_img, M, K, w_img, bias, N, group, eps, Cp, inv_norm, stream),
                   "onssen_linear_x3p_norms")

    def dropout(self, x, n, p, seed, out, stream):
        self.check(self.dll.onssen_dropout_f32(x, n, p, seed, out, stream), "onssen_dropout_f32")

    def mask_istft(self, stft_ri, mask, m_sb, m_sc, m_st, m_sf, B, Cn, T, n_fft, hop, length, out, stream, frames=None,
                   lengths=None):
        if frames is not None:       # ragged batch: frames[b] frames in, lengths[b] <= length samples out per row
            self.check(self.dll.onssen_mask_istft_ragged_f32(stft_ri, mask, m_sb, m_sc, m_st, m_sf, B, Cn, T, frames, n_fft, hop,
                                                             length, lengths, out, stream), "onssen_mask_istft_ragged_f32")
            return
        self.check(self.dll.onssen_mask_istft_f32(stft_ri, mask, m_sb, m_sc, m_st, m_sf, B, Cn, T, n_fft, hop,
                                                  length, out, stream), "onssen_mask_istft_f32")

    def phase_input(self, x_mag, mask, m_sb, m_sc, m_st, m_sf, x_phase, B, Cn, T, F, out, stream):
        self.check(self.dll.onssen_phase_input_f32(x_mag, mask, m_sb, m_sc, m_st, m_sf, x_phase, B, Cn, T, F, out,
                                                   stream), "onssen_phase_input_f32")

    def labels(self, mix, s1, s2, feat, B, T, F, db, utt_max, one_hot, mag_mix, mag_s1, mag_s2, cos_s1, cos_s2, stream):
        self.check(self.dll.onssen_labels_f32(mix, s1, s2, feat, B, T, F, db, utt_max, one_hot, mag_mix, mag_s1, mag_s2,
                                              cos_s1, cos_s2, stream), "onssen_labels_f32")

    def param_guard(self, ptrs, numel, n, samples, mode, ref, flag, stream):
        self.check(self.dll.onssen_param_guard_u32(ptrs, numel, n, samples, mode, ref, flag, stream), "onssen_param_guard_u32")

    def log_magnitude(self, stft_ri, n, eps, out, stream):
        self.check(self.dll.onssen_log_magnitude_f32(stft_ri, n, eps, out, stream), "onssen_log_magnitude_f32")

    def cos_difference(self, s1, s2, n, out, stream):
        self.check(self.dll.onssen_cos_difference_f32(s1, s2, n, out, stream), "onssen_cos_difference_f32")

    def one_hot(self, feat, m1, m2, B, per_utt, db, utt_max, out, stream):
        self.check(self.dll.onssen_one_hot_f32(feat, m1, m2, B, per_utt, db, utt_max, out, stream), "onssen_one_hot_f32")

    def dc_compact_layout(self, B, T, F, D):
        """(workspace bytes, byte offset of the compacted array, byte offset of the target map)."""
        co, do = _sz(), _sz()
        self.check(self.dll.onssen_dc_compact_layout(B, T, F, D, C.byref(co), C.byref(do)), "onssen_dc_compact_layout")
        return int(self.dll.onssen_dc_compact_workspace_bytes(B, T, F, D)), co.value, do.value

    def dc_index(self, feat, B, T, F, D, db, ws, ws_bytes, stream, frames=None):
        self.check(self.dll.onssen_dc_index_f32(feat, B, T, frames, F, D, db, ws, ws_bytes, stream), "onssen_dc_index_f32")

    def linear_x3p_compact(self, a_img, M, K, w_img, bias, N, group, eps, dest, dest_bs, F, comp, R, comp_bs, bf16_only, stream):
        self.check(self.dll.onssen_linear_x3p_compact(a_img, M, K, w_img, bias, N, group, eps, dest, dest_bs, F, comp, R, comp_bs,
                                                      int(bool(bf16_only)), stream), "onssen_linear_x3p_compact")

    def dc_cluster_compact(self, B, T, F, D, iters, masks, ws, ws_bytes, stream, flags=0, tol=1e-4):
        self.check(self.dll.onssen_dc_cluster_compact_f32(B, T, F, D, iters, tol, masks, ws, ws_bytes, flags, stream),
                   "onssen_dc_cluster_compact_f32")

    def dc_cluster(self, emb, feat, B, T, F, D, db, iters, masks, ws, ws_bytes, stream, flags=0, frames=None, tol=1e-4):
        if frames is not None:       # ragged batch: utterance b owns frames[b] * F bins
            self.check(self.dll.onssen_dc_cluster_ragged_f32(emb, feat, B, T, frames, F, D, db, iters, tol, masks, ws, ws_bytes, flags,
                                                             stream), "onssen_dc_cluster_ragged_f32")
            return
        self.check(self.dll.onssen_dc_cluster_f32(emb, feat, B, T, F, D, db, iters, tol, masks, ws, ws_bytes, flags, stream),
                   "onssen_dc_cluster_f32")
