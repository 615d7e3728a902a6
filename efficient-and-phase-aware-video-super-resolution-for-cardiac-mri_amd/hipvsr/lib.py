"""ctypes binding of librefinenet_hip.so (the C ABI declared in include/refinenet_hip.h).

The library is the product: there is no CPU or PyTorch fallback.  ``load()`` raises if the shared object is
missing, and every wrapper raises ``HipKernelError`` on a non-zero return code.
"""
import ctypes as C
import os

import torch  # noqa: F401  (first: the process must bind ONE HIP runtime - torch's - before the library below is loaded;
#                            loaded the other way round, the library's own libamdhip64 comes in first and torch's copy
#                            then reports "no ROCm-capable device")

HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.path.join(HERE, 'librefinenet_hip.so')
ABI_VERSION = 7          # RNH_ABI_VERSION of include/refinenet_hip.h this binding was written against

MAX_SRC, MAX_DST = 16, 4
EPI_STORE, EPI_PS, EPI_LSTM, EPI_LSTM_BWD = 0, 1, 2, 3
TILE_128x128, TILE_128x128_G, TILE_256x64, TILE_128x160, TILE_256x128 = 0, 1, 2, 3, 6
TILE_COLS = {TILE_128x128: 128, TILE_128x128_G: 128, TILE_256x64: 64, TILE_128x160: 160, TILE_256x128: 128}
TILE_ROWS = {TILE_128x128: 128, TILE_128x128_G: 128, TILE_256x64: 256, TILE_128x160: 128, TILE_256x128: 256}
WTILE_128x64, WTILE_64x128, WTILE_64x64 = 0x42, 0x24, 0x22      # rnh_conv_wgrad: MI << 4 | NI
TILE_DIRECT = 16
LOSS_L1, LOSS_CHARBONNIER = 0, 1
LOSS_BLOCKS = 64

EXPORTS = ['rnh_conv_igemm', 'rnh_pack_weights', 'rnh_conv_wgrad', 'rnh_wgrad_reduce', 'rnh_inconv_prelu_fwd',
           'rnh_inconv_prelu_bwd', 'rnh_inconv_bwd_ws_floats', 'rnh_outconv_fwd', 'rnh_outconv_fwd_ld', 'rnh_outconv_dgrad',
           'rnh_outconv_wgrad', 'rnh_outconv_wgrad_ws_floats', 'rnh_lstm_gates_bwd', 'rnh_loss_fwd_bwd', 'rnh_loss_total', 'rnh_ew_add',
           'rnh_phase_plane', 'rnh_last_error', 'rnh_abi_version', 'rnh_struct_sizes', 'rnh_uptail_compose',
           'rnh_uptail_dgrad', 'rnh_uptail_expand', 'rnh_uptail_wcontract', 'rnh_uptail_fwd', 'rnh_uptail_fwd_ws_floats',
           'rnh_uptail_g_floats', 'rnh_uptail_xcorr_supported', 'rnh_uptail_xcorr_ws_floats', 'rnh_uptail_xcorr',
           'rnh_uptail_bf16_supported', 'rnh_uptail_fwd_bf16_ws_floats', 'rnh_uptail_fwd_bf16', 'rnh_uptail_dgrad_bf16_ws_floats',
           'rnh_uptail_dgrad_bf16', 'rnh_uptail_xcorr_bf16', 'rnh_xcol_combine_m', 'rnh_xcol_gather_m', 'rnh_phase_wgrad', 'rnh_phase_wgrad_ws_floats', 'rnh_xcol_dgrad',
           'rnh_xcol_pack', 'rnh_xcol_unpack', 'rnh_xcol_combine', 'rnh_xcol_gather', 'rnh_conv_wino', 'rnh_wino_pack_weights', 'rnh_phase_bias_add',
           'rnh_wino_wgrad_supported', 'rnh_wino_wgrad_ws_floats', 'rnh_wino_wgrad', 'rnh_cine_gather', 'rnh_adam_step',
           'rnh_metrics_ws_floats', 'rnh_metrics_psnr_ssim',
           # bf16-storage path
           'rnh_conv_bf16', 'rnh_conv_bf16_pair', 'rnh_conv_wino_pair', 'rnh_pack_weights_bf16', 'rnh_wgrad_bf16', 'rnh_ew_add_m', 'rnh_lstm_gates_bwd_m', 'rnh_cast',
           'rnh_phase_plane_m', 'rnh_struct_sizes_bf16',
           # F(4x4, 3x3) ConvLSTM cell (ABI 5)
           'rnh_wino44_v_floats', 'rnh_wino44_transform', 'rnh_wino44_pack_weights', 'rnh_wino44_cell', 'rnh_wino44_cell_pair', 'rnh_wino44_conv',
           'rnh_wino44_tmajor_floats', 'rnh_wino44_tmajor', 'rnh_wino44_wgrad_gemm', 'rnh_wino44_wgrad_finish',
           # f16 weights for the upsampler's forward in the bf16-storage path (ABI 6)
           'rnh_pack_weights_f16',
           # gate backward + transform of the gate gradients in one launch (ABI 6)
           'rnh_wino44_gates_bwd_supported', 'rnh_wino44_gates_bwd',
           # weight gradient in F(4x4)-tile Winograd form, both transforms fused (ABI 6)
           'rnh_wino44f_wgrad_supported', 'rnh_wino44f_wgrad_ws_floats', 'rnh_wino44f_wgrad',
           'rnh_wino44f_wgrad_v_supported', 'rnh_wino44f_wgrad_v']
DT_F32, DT_BF16 = 0, 1


def dt_of(t):
    """RNH_DT_* tag of a tensor (fp32 or bf16; anything else is refused)."""
    if t.dtype == torch.float32:
        return DT_F32
    if t.dtype == torch.bfloat16:
        return DT_BF16
    raise HipKernelError(f'expected an fp32 or bf16 tensor, got {t.dtype}')


class HipKernelError(RuntimeError):
    pass


class Src(C.Structure):
    _fields_ = [('ptr', C.c_void_p), ('ptr2', C.c_void_p), ('C', C.c_int32), ('c0', C.c_int32), ('nch', C.c_int32),
                ('img_off', C.c_int32), ('scale', C.c_int32), ('sub_y', C.c_int32), ('sub_x', C.c_int32),
                ('_pad', C.c_int32)]


class Dst(C.Structure):
    _fields_ = [('ptr', C.c_void_p), ('C', C.c_int32), ('c0', C.c_int32), ('ncols', C.c_int32),
                ('accumulate', C.c_int32), ('img_off', C.c_int32), ('_pad', C.c_int32)]


class Wino44CellArgs(C.Structure):
    """rnh_wino44_cell_args_t"""
    _fields_ = [('v', C.c_void_p * 2), ('vchunks', C.c_int32 * 2), ('nsrc', C.c_int32), ('B', C.c_int32), ('H', C.c_int32), ('W', C.c_int32),
                ('Npad', C.c_int32), ('hd', C.c_int32), ('_pad', C.c_int32), ('wp', C.c_void_p), ('bias', C.c_void_p), ('c_prev', C.c_void_p),
                ('h_out', C.c_void_p), ('c_out', C.c_void_p), ('gates_out', C.c_void_p)]


class Wino44WgradArgs(C.Structure):
    """rnh_wino44_wgrad_args_t"""
    _fields_ = [('a', (C.c_void_p * 2) * 8), ('a_k8', (C.c_int64 * 2) * 8), ('a_k80', (C.c_int64 * 2) * 8), ('z', C.c_void_p), ('z_k8', C.c_int64),
                ('part', C.c_void_p), ('nprob', C.c_int32), ('CO', C.c_int32), ('T8', C.c_int32), ('S', C.c_int32)]


class Wino44ConvArgs(C.Structure):
    """rnh_wino44_conv_args_t"""
    _fields_ = [('v', C.c_void_p * 16), ('vchunks', C.c_int32 * 16), ('vblock_off', C.c_int32 * 16), ('nsrc', C.c_int32), ('B', C.c_int32), ('H', C.c_int32),
                ('W', C.c_int32), ('Npad', C.c_int32), ('_pad', C.c_int32 * 3), ('wp', C.c_void_p), ('bias', C.c_void_p), ('ndst', C.c_int32), ('ps_r', C.c_int32), ('ps_cq', C.c_int32), ('_pad2', C.c_int32), ('dst', Dst * MAX_DST)]


class ConvArgs(C.Structure):
    _fields_ = [('src', Src * MAX_SRC), ('nsrc', C.c_int32), ('B', C.c_int32), ('H', C.c_int32), ('W', C.c_int32),
                ('ntaps', C.c_int32), ('nk', C.c_int32), ('wp', C.c_void_p), ('bias', C.c_void_p),
                ('Npad', C.c_int32), ('epilogue', C.c_int32), ('tile', C.c_int32), ('ndst', C.c_int32),
                ('dst', Dst * MAX_DST), ('ps_r', C.c_int32), ('ps_cq', C.c_int32), ('hd', C.c_int32),
                ('_pad', C.c_int32), ('c_prev', C.c_void_p), ('h_out', C.c_void_p), ('c_out', C.c_void_p),
                ('gates_out', C.c_void_p)]


class Wino44VSrc(C.Structure):
    """rnh_wino44_vsrc_t: the transformed image that stands for one x source of rnh_wino44f_wgrad_v."""
    _fields_ = [('v', C.c_void_p), ('frame_stride', C.c_int64), ('nchunks', C.c_int32), ('c_first', C.c_int32)]


class WgradArgs(C.Structure):
    _fields_ = [('xs', Src * MAX_SRC), ('nxs', C.c_int32), ('xcols_pad', C.c_int32), ('ys', Src * MAX_SRC),
                ('nys', C.c_int32), ('ycols_pad', C.c_int32), ('xgrp', C.c_void_p), ('ygrp', C.c_void_p),
                ('B', C.c_int32), ('H', C.c_int32), ('W', C.c_int32), ('ntaps', C.c_int32), ('tile', C.c_int32),
                ('nsplit', C.c_int32), ('slab', C.c_void_p), ('bslab', C.c_void_p), ('zero_page', C.c_void_p)]


class MSrc(C.Structure):
    _fields_ = [('ptr', C.c_void_p), ('dtype', C.c_int32), ('C', C.c_int32), ('c0', C.c_int32), ('nch', C.c_int32),
                ('img_off', C.c_int32), ('scale', C.c_int32), ('sub_y', C.c_int32), ('sub_x', C.c_int32)]


class MDst(C.Structure):
    _fields_ = [('ptr', C.c_void_p), ('dtype', C.c_int32), ('C', C.c_int32), ('c0', C.c_int32), ('ncols', C.c_int32),
                ('accumulate', C.c_int32), ('img_off', C.c_int32), ('_pad', C.c_int32)]


class ConvBf16Args(C.Structure):
    _fields_ = [('src', MSrc * MAX_SRC), ('nsrc', C.c_int32), ('B', C.c_int32), ('H', C.c_int32), ('W', C.c_int32),
                ('ntaps', C.c_int32), ('nchunks', C.c_int32), ('wp', C.c_void_p), ('bias', C.c_void_p),
                ('Npad', C.c_int32), ('epilogue', C.c_int32), ('ndst', C.c_int32), ('ps_r', C.c_int32), ('ps_cq', C.c_int32),
                ('hd', C.c_int32), ('dst', MDst * MAX_DST), ('c_prev', C.c_void_p), ('c_out', C.c_void_p),
                ('h_out', C.c_void_p), ('gates_out', C.c_void_p), ('h_dtype', C.c_int32), ('gates_dtype', C.c_int32),
                ('bw_dh', C.c_void_p), ('bw_dc_next', C.c_void_p), ('bw_gates', C.c_void_p), ('bw_c_prev', C.c_void_p),
                ('bw_c_next', C.c_void_p), ('bw_dgates', C.c_void_p), ('bw_dc_prev', C.c_void_p), ('bw_dh_dtype', C.c_int32),
                ('bw_dgates_dtype', C.c_int32), ('bw_rec_dtype', C.c_int32), ('wp_f16', C.c_int32)]


class WgradBf16Args(C.Structure):
    _fields_ = [('xs', MSrc * MAX_SRC), ('nxs', C.c_int32), ('xrows_pad', C.c_int32), ('ys', MSrc * MAX_SRC),
                ('nys', C.c_int32), ('ycols_pad', C.c_int32), ('B', C.c_int32), ('H', C.c_int32), ('W', C.c_int32),
                ('ntaps', C.c_int32), ('nsplit', C.c_int32), ('slab', C.c_void_p), ('bslab', C.c_void_p)]


class CineSample(C.Structure):
    _fields_ = [('lr_off', C.c_int64), ('hr_off', C.c_int64), ('code_off', C.c_int64), ('Tc', C.c_int32), ('Hl', C.c_int32),
                ('Wl', C.c_int32), ('Hh', C.c_int32), ('Wh', C.c_int32), ('lr_start', C.c_int32), ('hr_start', C.c_int32),
                ('y0', C.c_int32), ('x0', C.c_int32), ('hflip', C.c_int32), ('vflip', C.c_int32), ('reserved', C.c_int32 * 3)]


_lib = None


def load():
    """Load the shared library (once).  Raises if it has not been built: build with csrc/build.sh or
    ``__graft_entry__.build()``."""
    global _lib
    if _lib is not None:
        return _lib
    if not os.path.exists(LIB_PATH):
        raise HipKernelError(f'{LIB_PATH} is missing: build it with csrc/build.sh (hipcc --offload-arch=gfx950); '
                             'there is no fallback path')
    lib = C.CDLL(LIB_PATH)
    for name in EXPORTS:
        if not hasattr(lib, name):
            raise HipKernelError(f'{LIB_PATH} does not export {name}')
    vp, i32, i64, f32 = C.c_void_p, C.c_int, C.c_int64, C.c_float
    lib.rnh_last_error.restype = C.c_char_p
    lib.rnh_conv_igemm.argtypes = [C.POINTER(ConvArgs), vp]
    lib.rnh_pack_weights.argtypes = [vp, vp, vp, vp, vp, vp, vp, vp, vp, i32, i32, i32, i32, i32, i32, i32, vp]
    lib.rnh_conv_wgrad.argtypes = [C.POINTER(WgradArgs), vp]
    lib.rnh_wgrad_reduce.argtypes = [vp, vp, i32, i32, i32, i32, vp, vp, i32, vp, vp, i32, vp]
    lib.rnh_inconv_prelu_fwd.argtypes = [vp, vp, vp, vp, vp, i32, i32, i32, i32, i32, vp]
    lib.rnh_inconv_prelu_bwd.argtypes = [vp, vp, vp, vp, vp, vp, vp, vp, vp, i32, i32, i32, i32, i32, i32, vp]
    lib.rnh_inconv_bwd_ws_floats.argtypes = [i32, i32]
    lib.rnh_inconv_bwd_ws_floats.restype = i64
    lib.rnh_outconv_fwd.argtypes = [vp, vp, vp, vp, i32, i32, i32, i32, i32, vp]
    lib.rnh_outconv_fwd_ld.argtypes = [vp, vp, i64, i64, i32, vp, vp, i32, i32, i32, i32, i32, i32, i32, vp]
    lib.rnh_outconv_dgrad.argtypes = [vp, vp, vp, i32, i32, i32, i32, i32, vp]
    lib.rnh_outconv_wgrad.argtypes = [vp, vp, vp, vp, vp, i32, i32, i32, i32, i32, i32, vp]
    lib.rnh_outconv_wgrad_ws_floats.argtypes = [i32, i32]
    lib.rnh_outconv_wgrad_ws_floats.restype = i64
    lib.rnh_lstm_gates_bwd.argtypes = [vp, vp, vp, vp, vp, vp, vp, vp, i64, i32, vp]
    lib.rnh_loss_fwd_bwd.argtypes = [vp, vp, vp, vp, vp, vp, i32, i32, i64, i32, f32, vp]
    lib.rnh_loss_total.argtypes = [vp, vp, vp, i32, i32, i32, vp]
    lib.rnh_ew_add.argtypes = [vp, vp, vp, vp, i64, i32, vp]
    lib.rnh_phase_plane.argtypes = [vp, vp, i32, i32, i32, i32, vp]
    lib.rnh_uptail_fwd.argtypes = [vp, vp, vp, vp, vp, vp, vp, i32, i32, i32, i32, i32, i32, i32, vp]
    lib.rnh_uptail_fwd_ws_floats.argtypes = [i32, i32, i32, i32]
    lib.rnh_uptail_fwd_ws_floats.restype = i64
    lib.rnh_uptail_compose.argtypes = [vp, vp, vp, i32, i32, i32, i32, vp]
    lib.rnh_uptail_g_floats.argtypes = [i32, i32, i32]
    lib.rnh_uptail_g_floats.restype = i64
    lib.rnh_uptail_xcorr_supported.argtypes = [i32, i32, i32]
    lib.rnh_uptail_xcorr_ws_floats.argtypes = [i32, i32, i32, i32, i32]
    lib.rnh_uptail_xcorr_ws_floats.restype = i64
    lib.rnh_uptail_xcorr.argtypes = [vp, vp, vp, vp, vp, i32, i32, i32, i32, i32, vp]
    lib.rnh_uptail_bf16_supported.argtypes = [i32, i32, i32]
    lib.rnh_uptail_fwd_bf16_ws_floats.argtypes = [i32, i32, i32, i32]
    lib.rnh_uptail_fwd_bf16_ws_floats.restype = i64
    lib.rnh_uptail_fwd_bf16.argtypes = [vp, vp, vp, vp, vp, vp, vp, i32, i32, i32, i32, i32, i32, i32, vp]
    lib.rnh_uptail_dgrad_bf16_ws_floats.argtypes = []
    lib.rnh_uptail_dgrad_bf16_ws_floats.restype = i64
    lib.rnh_uptail_dgrad_bf16.argtypes = [vp, vp, vp, vp, i32, i32, i32, i32, i32, i32, vp]
    lib.rnh_uptail_xcorr_bf16.argtypes = [vp, vp, vp, vp, vp, i32, i32, i32, i32, i32, vp]
    lib.rnh_phase_wgrad.argtypes = [vp, vp, vp, vp, i32, i32, i32, i32, i32, i32, i32, i32, i32, i32, i32, vp]
    lib.rnh_phase_wgrad_ws_floats.argtypes = [i32, i32, i32, i32]
    lib.rnh_phase_wgrad_ws_floats.restype = i64
    lib.rnh_xcol_dgrad.argtypes = [vp, vp, vp, vp, vp, i32, i32, i32, i32, i32, i32, i32, i32, i32, i32, vp]
    lib.rnh_xcol_combine_m.argtypes = [vp, vp, vp, i32, i64, i32, i32, i32, i32, i32, vp]
    lib.rnh_xcol_gather_m.argtypes = [vp, i32, vp, i32, i64, i32, i32, i32, i32, i32, vp]
    lib.rnh_xcol_pack.argtypes = [vp, vp, i32, i32, i32, i32, i32, i32, i32, vp]
    lib.rnh_xcol_unpack.argtypes = [vp, vp, vp, vp, i32, i32, i32, i32, i32, i32, i32, i32, vp]
    lib.rnh_xcol_combine.argtypes = [vp, vp, vp, vp, vp, i64, i32, i32, i32, i32, i32, vp]
    lib.rnh_xcol_gather.argtypes = [vp, vp, i64, i32, i32, i32, i32, i32, vp]
    lib.rnh_conv_wino.argtypes = [C.POINTER(ConvArgs), vp]
    lib.rnh_wino_pack_weights.argtypes = [vp, vp, vp, vp, vp, vp, vp, vp, i32, i32, i32, i32, i32, i32, vp]
    lib.rnh_phase_bias_add.argtypes = [vp, vp, vp, vp, i32, i32, i32, i32, i32, i32, i32, i32, i32, i32, vp]
    lib.rnh_wino_wgrad_supported.argtypes = [C.POINTER(WgradArgs)]
    lib.rnh_wino_wgrad_ws_floats.argtypes = [C.POINTER(WgradArgs), C.POINTER(C.c_int64)]
    lib.rnh_wino_wgrad.argtypes = [C.POINTER(WgradArgs), vp, vp, vp, i32, vp, vp, i32, vp]
    lib.rnh_uptail_dgrad.argtypes = [vp, vp, vp, i32, i32, i32, i32, i32, i32, vp]
    lib.rnh_uptail_expand.argtypes = [vp, vp, i32, i32, i32, i32, i32, i32, vp]
    lib.rnh_uptail_wcontract.argtypes = [vp, vp, vp, vp, vp, vp, vp, vp, vp, i32, i32, i32, i32, i32, i32, vp]
    lib.rnh_cine_gather.argtypes = [vp, i64, C.POINTER(CineSample), vp, i32, i32, i32, i32, i32, i32, i32, f32, f32, vp, vp, vp, vp]
    lib.rnh_adam_step.argtypes = [vp, vp, vp, vp, i64, i32, f32, f32, f32, f32, f32, vp]
    lib.rnh_metrics_ws_floats.argtypes = [i32, i32, i32]
    lib.rnh_metrics_ws_floats.restype = i64
    lib.rnh_metrics_psnr_ssim.argtypes = [vp, vp, i32, i32, i32, i32, i32, i32, f32, f32, f32, f32, C.POINTER(C.c_float), vp, vp, vp]
    lib.rnh_conv_bf16.argtypes = [C.POINTER(ConvBf16Args), vp]
    lib.rnh_conv_bf16_pair.argtypes = [C.POINTER(ConvBf16Args), C.POINTER(ConvBf16Args), vp]
    lib.rnh_conv_wino_pair.argtypes = [C.POINTER(ConvArgs), C.POINTER(ConvArgs), vp]
    lib.rnh_wino44_v_floats.argtypes = [i32, i32, i32, i32]
    lib.rnh_wino44_v_floats.restype = i64
    lib.rnh_wino44_transform.argtypes = [vp, i32, i32, i32, i32, i32, i32, vp, vp]
    lib.rnh_wino44_gates_bwd_supported.argtypes = [i32, i32, i32]
    lib.rnh_wino44_gates_bwd.argtypes = [vp] * 9 + [i32, i32, i32, i32, vp]
    lib.rnh_wino44f_wgrad_supported.argtypes = [vp]
    lib.rnh_wino44f_wgrad_ws_floats.argtypes = [vp, vp]
    lib.rnh_wino44f_wgrad.argtypes = [vp, vp, vp, vp, vp, vp, i32, vp, vp, i32, vp]
    lib.rnh_wino44f_wgrad_v_supported.argtypes = [vp, vp, i32]
    lib.rnh_wino44f_wgrad_v.argtypes = [vp, vp, i32, vp, vp, vp, vp, i32, vp, vp, i32, vp]
    lib.rnh_wino44_pack_weights.argtypes = [vp, vp, vp, vp, vp, vp, vp, i32, i32, i32, i32, i32, vp]
    lib.rnh_wino44_cell.argtypes = [C.POINTER(Wino44CellArgs), vp]
    lib.rnh_wino44_cell_pair.argtypes = [C.POINTER(Wino44CellArgs), C.POINTER(Wino44CellArgs), vp]
    lib.rnh_wino44_conv.argtypes = [C.POINTER(Wino44ConvArgs), vp]
    lib.rnh_wino44_tmajor_floats.argtypes = [i32, i32, i32, i32]
    lib.rnh_wino44_tmajor_floats.restype = i64
    lib.rnh_wino44_tmajor.argtypes = [vp, i32, i32, i32, i32, i32, i32, i32, vp, vp]
    lib.rnh_wino44_wgrad_gemm.argtypes = [C.POINTER(Wino44WgradArgs), vp]
    lib.rnh_wino44_wgrad_finish.argtypes = [C.POINTER(Wino44WgradArgs), vp, i32, i32, vp, vp, vp, i32, i32, vp]
    lib.rnh_pack_weights_bf16.argtypes = [vp, vp, vp, vp, vp, vp, vp, vp, vp, i32, i32, i32, i32, i32, i32, i32, vp]
    lib.rnh_pack_weights_f16.argtypes = lib.rnh_pack_weights_bf16.argtypes
    lib.rnh_wgrad_bf16.argtypes = [C.POINTER(WgradBf16Args), vp]
    lib.rnh_ew_add_m.argtypes = [vp, i32, vp, i32, vp, i32, vp, i32, i64, i32, vp]
    lib.rnh_lstm_gates_bwd_m.argtypes = [vp, i32, vp, i32, vp, vp, i32, vp, vp, vp, i32, vp, i64, i32, vp]
    lib.rnh_cast.argtypes = [vp, i32, vp, i32, i64, vp]
    lib.rnh_phase_plane_m.argtypes = [vp, vp, i32, i32, i32, i32, i32, vp]
    lib.rnh_struct_sizes_bf16.argtypes = [C.POINTER(C.c_int32 * 4)]
    lib.rnh_struct_sizes_bf16.restype = None
    msz = (C.c_int32 * 4)()
    lib.rnh_struct_sizes_bf16(C.byref(msz))
    mine_m = [C.sizeof(MSrc), C.sizeof(MDst), C.sizeof(ConvBf16Args), C.sizeof(WgradBf16Args)]
    if list(msz) != mine_m:
        raise HipKernelError(f'bf16 struct layout mismatch between the binding {mine_m} and the library {list(msz)}')
    lib.rnh_struct_sizes.argtypes = [C.POINTER(C.c_int32 * 4)]
    lib.rnh_struct_sizes.restype = None
    sizes = (C.c_int32 * 4)()
    lib.rnh_struct_sizes(C.byref(sizes))
    mine = [C.sizeof(Src), C.sizeof(Dst), C.sizeof(ConvArgs), C.sizeof(WgradArgs)]
    if list(sizes) != mine:
        raise HipKernelError(f'struct layout mismatch between the binding {mine} and the library {list(sizes)}')
    if lib.rnh_abi_version() != ABI_VERSION:
        raise HipKernelError('ABI version mismatch')
    _lib = lib
    return lib


def check(rc, what):
    if rc != 0:
        msg = load().rnh_last_error().decode('utf-8', 'replace')
        raise HipKernelError(f'{what} failed with code {rc}: {msg}')
