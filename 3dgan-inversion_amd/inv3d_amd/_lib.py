"""ctypes binding of libeg3d_hip.so (the C-ABI declared in include/eg3d_hip.h).

The library is built in-tree by `make -C 3dgan-inversion_amd` (or `__graft_entry__.build()`).  There is NO fallback:
if the shared object is missing, or a kernel returns a non-zero status, this module raises -- the product path never
silently drops to PyTorch / CPU code.
"""
import ctypes as C
import os

import torch

_HERE = os.path.dirname(os.path.abspath(__file__))
# EG3D_DETERMINISTIC=1: the deterministic build of the same sources (csrc/det.h: every floating-point atomic an exact fixed-point
# accumulation -- bit-identical results from run to run); its accumulator workspace is lent on first use (det_enable below)
DETERMINISTIC = os.environ.get('EG3D_DETERMINISTIC', '0') != '0'
LIB_PATH = os.path.join(_HERE, os.environ.get('EG3D_LIBNAME', 'libeg3d_hip_det.so' if DETERMINISTIC else 'libeg3d_hip.so'))     # EG3D_LIBNAME: A/B builds in one GPU session

F32, F16, F64 = 0, 1, 2
EPI_STORE, EPI_ATOMIC, EPI_FWD, EPI_BWD, EPI_BWD_ACT = 0, 1, 2, 3, 4
ACT_IDS = dict(linear=1, relu=2, lrelu=3, tanh=4, sigmoid=5, elu=6, selu=7, softplus=8, swish=9)

c_float_p = C.POINTER(C.c_float)


class ConvClass(C.Structure):
    _fields_ = [('Ha', C.c_int32), ('Wa', C.c_int32), ('out_py', C.c_int32), ('out_px', C.c_int32), ('ntaps', C.c_int32),
                ('dy', C.c_int32 * 9), ('dx', C.c_int32 * 9), ('wtap', C.c_int32 * 9)]


class ActBwd(C.Structure):
    """eg3d_act_bwd: the producing layer's activation backward fused into a data-gradient epilogue (EPI_BWD_ACT)."""
    _fields_ = [('d', C.c_void_p), ('bias', C.c_void_p), ('noise', C.c_void_p), ('noise_nstride', C.c_int64), ('noise_strength', C.c_void_p),
                ('act', C.c_int32), ('alpha', C.c_float), ('gain', C.c_float), ('clamp', C.c_float),
                ('dbias', C.c_void_p), ('dd', C.c_void_p), ('dnoise', C.c_void_p), ('dnoise_nstride', C.c_int64), ('dstrength', C.c_void_p)]


class WgfItem(C.Structure):
    """eg3d_wgf_item: one layer of eg3d_weight_grad_finish_batched."""
    _fields_ = [('g', C.c_void_p), ('w', C.c_void_p), ('s', C.c_void_p), ('d', C.c_void_p), ('dd', C.c_void_p), ('dw', C.c_void_p),
                ('slab_stride', C.c_int64), ('N', C.c_int32), ('O', C.c_int32), ('I', C.c_int32), ('T', C.c_int32), ('nslab', C.c_int32)]


WGF_BATCH_MAX = 32


class PackItem(C.Structure):
    """eg3d_pack_item: one layer of eg3d_pack_conv_weights_batched."""
    _fields_ = [('w', C.c_void_p), ('wf', C.c_void_p), ('wa', C.c_void_p), ('wsq', C.c_void_p),
                ('O', C.c_int32), ('I', C.c_int32), ('T', C.c_int32), ('O_pad', C.c_int32), ('oscale', C.c_void_p)]


PACK_BATCH_MAX = 40


class ConvParams(C.Structure):
    _fields_ = [('x', C.c_void_p), ('w', C.c_void_p), ('out', C.c_void_p),
                ('N', C.c_int32), ('Hi', C.c_int32), ('Wi', C.c_int32), ('Ck', C.c_int32), ('ldx', C.c_int32),
                ('Nc', C.c_int32), ('w_row', C.c_int32),
                ('Ho', C.c_int32), ('Wo', C.c_int32), ('ldo', C.c_int32),
                ('in_stride', C.c_int32), ('out_stride', C.c_int32),
                ('ncls', C.c_int32), ('cls', ConvClass * 4),
                ('in_scale', C.c_void_p), ('epi', C.c_int32), ('ksplit', C.c_int32),
                ('out_scale', C.c_void_p), ('bias', C.c_void_p), ('noise', C.c_void_p), ('noise_nstride', C.c_int64),
                ('noise_strength', C.c_void_p), ('act', C.c_int32), ('alpha', C.c_float), ('gain', C.c_float),
                ('clamp', C.c_float), ('addend', C.c_void_p), ('xin', C.c_void_p), ('ds', C.c_void_p), ('precision', C.c_int32), ('a_amax', C.c_void_p), ('a_amax_mul', C.c_float), ('ds_replicas', C.c_int32), ('out_amax', C.c_void_p), ('act_bwd', ActBwd), ('w_presplit', C.c_int32),
                ('addend_up2', C.c_int32), ('addend_taps', C.c_float * 4)]


class ConvWsParams(C.Structure):
    _fields_ = [('x', C.c_void_p), ('in_scale', C.c_void_p), ('x_amax', C.c_void_p), ('x_amax_mul', C.c_float),
                ('w', C.c_void_p), ('w_scale', C.c_void_p), ('out', C.c_void_p),
                ('N', C.c_int32), ('H', C.c_int32), ('W', C.c_int32), ('Ck', C.c_int32), ('ldx', C.c_int32),
                ('Nc', C.c_int32), ('ldo', C.c_int32), ('wtaps', C.c_int32),
                ('dy', C.c_int32 * 9), ('dx', C.c_int32 * 9), ('wtap', C.c_int32 * 9), ('products', C.c_int32),
                ('in_stride', C.c_int32), ('Hx', C.c_int32), ('Wx', C.c_int32), ('out_stride', C.c_int32)]


class ConvV2Params(C.Structure):
    _fields_ = [('a', C.c_void_p), ('w', C.c_void_p), ('a_scale', C.c_void_p), ('w_scale', C.c_void_p), ('out', C.c_void_p),
                ('N', C.c_int32), ('Hi', C.c_int32), ('Wi', C.c_int32), ('Ck', C.c_int32), ('Nc', C.c_int32), ('wtaps', C.c_int32),
                ('Ho', C.c_int32), ('Wo', C.c_int32), ('ldo', C.c_int32), ('in_stride', C.c_int32), ('out_stride', C.c_int32),
                ('ncls', C.c_int32), ('cls', ConvClass * 4), ('epi', C.c_int32),
                ('out_scale', C.c_void_p), ('bias', C.c_void_p), ('noise', C.c_void_p), ('noise_nstride', C.c_int64),
                ('noise_strength', C.c_void_p), ('act', C.c_int32), ('alpha', C.c_float), ('gain', C.c_float), ('clamp', C.c_float),
                ('addend', C.c_void_p), ('xin', C.c_void_p), ('ds', C.c_void_p), ('out_amax', C.c_void_p), ('act_bwd', ActBwd), ('products', C.c_int32), ('ksplit', C.c_int32), ('patch_rows', C.c_int32),
                ('rgb_w', C.c_void_p), ('rgb_s', C.c_void_p), ('rgb_bias', C.c_void_p), ('rgb_out', C.c_void_p), ('rgb_clamp', C.c_float), ('rgb_ldw', C.c_int32), ('rgb_nout', C.c_int32)]


class ConvUp2Params(C.Structure):
    _fields_ = [('a', C.c_void_p), ('w', C.c_void_p), ('a_scale', C.c_void_p), ('w_scale', C.c_void_p), ('out', C.c_void_p),
                ('N', C.c_int32), ('Hi', C.c_int32), ('Wi', C.c_int32), ('Ck', C.c_int32), ('Nc', C.c_int32),
                ('Hc', C.c_int32), ('Wc', C.c_int32), ('Ho', C.c_int32), ('Wo', C.c_int32), ('ldo', C.c_int32),
                ('wtap', C.c_int32 * 9), ('epi', C.c_int32), ('products', C.c_int32), ('ksplit', C.c_int32), ('patch_rows', C.c_int32)]


class WgradParams(C.Structure):
    _fields_ = [('x', C.c_void_p), ('g', C.c_void_p), ('dw', C.c_void_p),
                ('N', C.c_int32), ('Hi', C.c_int32), ('Wi', C.c_int32), ('Ck', C.c_int32), ('ldx', C.c_int32),
                ('Nc', C.c_int32), ('w_row', C.c_int32),
                ('Ho', C.c_int32), ('Wo', C.c_int32), ('ldg', C.c_int32),
                ('in_stride', C.c_int32), ('out_stride', C.c_int32),
                ('ncls', C.c_int32), ('cls', ConvClass * 4),
                ('in_scale', C.c_void_p), ('psplit', C.c_int32), ('precision', C.c_int32), ('g_amax', C.c_void_p), ('g_amax_mul', C.c_float)]


class WgradV2Params(C.Structure):
    _fields_ = [('g', C.c_void_p), ('x', C.c_void_p), ('g_scale', C.c_void_p), ('x_scale', C.c_void_p), ('dw', C.c_void_p),
                ('N', C.c_int32), ('H', C.c_int32), ('W', C.c_int32), ('Co', C.c_int32), ('Ci', C.c_int32), ('w_row', C.c_int32),
                ('ntaps', C.c_int32), ('dy', C.c_int32 * 9), ('dx', C.c_int32 * 9), ('wtap', C.c_int32 * 9),
                ('products', C.c_int32), ('row_groups', C.c_int32), ('slabs', C.c_int32)]


class RenderParams(C.Structure):
    _fields_ = [('planes', C.c_void_p), ('N', C.c_int32), ('Hp', C.c_int32), ('Wp', C.c_int32), ('ldp', C.c_int32),
                ('C', C.c_int32), ('origins', C.c_void_p), ('dirs', C.c_void_p), ('R', C.c_int32), ('u1', C.c_void_p),
                ('u2', C.c_void_p), ('Dc', C.c_int32), ('Df', C.c_int32), ('ray_start', C.c_float), ('ray_end', C.c_float),
                ('ray_limits', C.c_void_p), ('disparity', C.c_int32), ('box_warp', C.c_float), ('white_back', C.c_int32),
                ('w0', C.c_void_p), ('b0', C.c_void_p), ('w1', C.c_void_p), ('b1', C.c_void_p), ('Hdim', C.c_int32),
                ('Cout', C.c_int32), ('rgb', C.c_void_p), ('depth', C.c_void_p), ('wsum', C.c_void_p),
                ('depth_minmax', C.c_void_p), ('fine_depths', C.c_void_p), ('save_sigma', C.c_void_p), ('save_rgb', C.c_void_p),
                ('ray_tile_width', C.c_int32), ('pos_rows', C.c_void_p), ('feat_rows', C.c_void_p), ('dbg_inds', C.c_void_p), ('dbg_ranks', C.c_void_p), ('dbg_cdf', C.c_void_p)]


class RenderBwdParams(C.Structure):
    _fields_ = [('fwd', RenderParams), ('depth_out', C.c_void_p), ('d_rgb', C.c_void_p), ('d_depth', C.c_void_p),
                ('d_wsum', C.c_void_p), ('df_rows', C.c_void_p), ('df_pos', C.c_void_p), ('ag_rows', C.c_void_p), ('gc_rows', C.c_void_p), ('d_origins', C.c_void_p),
                ('d_dirs', C.c_void_p),
                ('dump_dpre', C.c_void_p), ('dump_h', C.c_void_p), ('dump_dout', C.c_void_p), ('dump_feat', C.c_void_p), ('df_amax', C.c_void_p),
                ('gram_w0', C.c_void_p), ('gram_b0', C.c_void_p), ('gram_w1', C.c_void_p), ('gram_b1', C.c_void_p),
                ('gram_scale0', C.c_float), ('gram_scale1', C.c_float), ('gram_bias_scale', C.c_float)]


class RenderSizes(C.Structure):
    _fields_ = [(k, C.c_int64) for k in ('S', 'rgb', 'depth', 'wsum', 'depth_minmax', 'fine_depths', 'save_sigma', 'save_rgb', 'pos_rows',
                                         'df_rows', 'df_pos', 'ag_rows', 'gc_rows', 'dump_dpre', 'dump_h', 'dump_dout', 'dump_feat', 'feat_rows')]


SPLIT_W_BATCH_MAX = 24


class SplitWItem(C.Structure):
    _fields_ = [('w', C.c_void_p), ('image', C.c_void_p), ('scale_out', C.c_void_p), ('amax', C.c_void_p), ('O', C.c_int32), ('I', C.c_int32), ('T', C.c_int32),
                ('w_row', C.c_int32)]


class SplitWBatch(C.Structure):
    _fields_ = [('n', C.c_int32), ('pad_', C.c_int32), ('items', SplitWItem * SPLIT_W_BATCH_MAX)]


class TorgbSmallParams(C.Structure):
    _fields_ = [('x', C.c_void_p), ('w', C.c_void_p), ('s', C.c_void_p), ('bias', C.c_void_p), ('addend', C.c_void_p), ('out', C.c_void_p),
                ('N', C.c_int32), ('H', C.c_int32), ('W', C.c_int32), ('C', C.c_int32), ('Cp', C.c_int32), ('ldx', C.c_int32), ('ldo', C.c_int32),
                ('w_row', C.c_int32), ('addend_up2', C.c_int32), ('clamp', C.c_float), ('addend_taps', C.c_float * 4),
                ('pre_z', C.c_void_p), ('pre_d', C.c_void_p), ('pre_bias', C.c_void_p), ('pre_noise', C.c_void_p), ('pre_strength', C.c_void_p), ('x_amax', C.c_void_p),
                ('pre_noise_nstride', C.c_int64), ('pre_slope', C.c_float), ('pre_gain', C.c_float), ('pre_clamp', C.c_float), ('pad_', C.c_int32)]


class TorgbSmallBwdParams(C.Structure):
    _fields_ = [('dy', C.c_void_p), ('wa', C.c_void_p), ('s', C.c_void_p), ('xin', C.c_void_p), ('addend', C.c_void_p), ('dx', C.c_void_p), ('ds', C.c_void_p),
                ('out_amax', C.c_void_p), ('N', C.c_int32), ('H', C.c_int32), ('W', C.c_int32), ('C', C.c_int32), ('Cp', C.c_int32), ('ldg', C.c_int32),
                ('ldx', C.c_int32), ('wa_row', C.c_int32), ('act_on', C.c_int32), ('no_mid', C.c_int32), ('act_bwd', ActBwd),
                ('add_scale', C.c_void_p), ('add_ds', C.c_void_p)]


class Conv3x3DirectParams(C.Structure):
    _fields_ = [('x', C.c_void_p), ('w', C.c_void_p), ('y', C.c_void_p), ('pooled', C.c_void_p), ('N', C.c_int32), ('H', C.c_int32), ('W', C.c_int32),
                ('Ci', C.c_int32), ('Co', C.c_int32), ('G', C.c_int32), ('act', C.c_int32), ('alpha', C.c_float), ('gain', C.c_float), ('pad_', C.c_int32),
                ('ga', C.c_void_p), ('gb', C.c_void_p)]


ADAM_ITEMS_MAX = 32


class AdamItem(C.Structure):
    _fields_ = [('p', C.c_void_p), ('g', C.c_void_p), ('g2', C.c_void_p), ('m', C.c_void_p), ('v', C.c_void_p), ('n', C.c_int64), ('normalize', C.c_int32),
                ('pad_', C.c_int32)]


class AdamList(C.Structure):
    _fields_ = [('n', C.c_int32), ('bump_step', C.c_int32), ('beta1', C.c_float), ('beta2', C.c_float), ('eps', C.c_float), ('pad_', C.c_float),
                ('lr', C.c_void_p), ('step', C.c_void_p), ('items', AdamItem * ADAM_ITEMS_MAX), ('skip', C.c_void_p)]


UNIT_LEVELS_MAX = 8


class UnitLevel(C.Structure):
    _fields_ = [('x', C.c_void_p), ('scale', C.c_void_p), ('feat', C.c_void_p), ('dx', C.c_void_p), ('HW', C.c_int32), ('C', C.c_int32), ('ldx', C.c_int32),
                ('mul', C.c_float)]


class UnitLevels(C.Structure):
    _fields_ = [('n', C.c_int32), ('N', C.c_int32), ('eps_inside', C.c_int32), ('eps', C.c_float), ('feat_nstride', C.c_int64),
                ('levels', UnitLevel * UNIT_LEVELS_MAX)]


class FlreluParams(C.Structure):
    _fields_ = [('x', C.c_void_p), ('b', C.c_void_p), ('y', C.c_void_p), ('fu', C.c_void_p), ('fd', C.c_void_p), ('mask', C.c_void_p),
                ('dtype', C.c_int32), ('N', C.c_int32), ('C', C.c_int32), ('H', C.c_int32), ('W', C.c_int32),
                ('fuh', C.c_int32), ('fuw', C.c_int32), ('fdh', C.c_int32), ('fdw', C.c_int32), ('up', C.c_int32), ('down', C.c_int32),
                ('px0', C.c_int32), ('px1', C.c_int32), ('py0', C.c_int32), ('py1', C.c_int32),
                ('qx0', C.c_int32), ('qx1', C.c_int32), ('qy0', C.c_int32), ('qy1', C.c_int32), ('Ho', C.c_int32), ('Wo', C.c_int32),
                ('flip_fu', C.c_int32), ('flip_fd', C.c_int32), ('mode', C.c_int32), ('gain1', C.c_float), ('gain2', C.c_float),
                ('gain', C.c_float), ('slope', C.c_float), ('clamp', C.c_float)]


class StyleLayer(C.Structure):
    _fields_ = [('weight', C.c_void_p), ('bias', C.c_void_p), ('out', C.c_void_p), ('dout', C.c_void_p), ('C', C.c_int32), ('wrow', C.c_int32),
                ('wgain', C.c_float), ('bgain', C.c_float), ('post', C.c_float), ('Co', C.c_int32),
                ('wsq', C.c_void_p), ('d', C.c_void_p), ('dd', C.c_void_p), ('dout_extra', C.c_void_p), ('dweight', C.c_void_p), ('dbias', C.c_void_p)]


STYLE_BANK_MAX = 32


class StyleBank(C.Structure):
    _fields_ = [('ws', C.c_void_p), ('dws', C.c_void_p), ('N', C.c_int32), ('L', C.c_int32), ('D', C.c_int32), ('nlayers', C.c_int32),
                ('layers', StyleLayer * STYLE_BANK_MAX)]


_SIGS = {
    'eg3d_abi_version': (C.c_int, []),
    'eg3d_status_string': (C.c_char_p, [C.c_int]),
    'eg3d_bias_act': (C.c_int, [C.c_void_p] * 6 + [C.c_int, C.c_int64, C.c_int, C.c_int, C.c_int, C.c_int, C.c_float, C.c_float,
                                C.c_float, C.c_void_p]),
    'eg3d_upfirdn2d': (C.c_int, [C.c_void_p, C.c_void_p, C.c_void_p, C.c_int, C.c_int, C.c_int, C.c_int, C.c_int,
                                 C.POINTER(C.c_int64), C.c_int, C.c_int, C.c_int, C.c_int, C.c_int, C.c_int, C.c_int, C.c_int,
                                 C.c_int, C.c_int, C.c_int, C.c_float, C.c_int, C.c_int, C.POINTER(C.c_int64), C.c_void_p]),
    'eg3d_conv2d_igemm_f32': (C.c_int, [C.POINTER(ConvParams), C.c_void_p]),
    'eg3d_split_weight_pieces': (C.c_int, [C.c_void_p, C.c_void_p, C.c_int64, C.c_void_p]),
    'eg3d_conv2d_igemm_config': (C.c_int, [C.POINTER(ConvParams)]),
    'eg3d_conv2d_igemm_act_bwd_ok': (C.c_int, [C.POINTER(ConvParams)]),
    'eg3d_conv2d_wgrad_f32': (C.c_int, [C.POINTER(WgradParams), C.c_void_p]),
    'eg3d_conv2d_wgrad_batched': (C.c_int, [C.POINTER(WgradParams), C.c_int, C.c_void_p]),
    'eg3d_conv2d_v2_supported': (C.c_int, [C.POINTER(ConvV2Params)]),
    'eg3d_conv2d_v2': (C.c_int, [C.POINTER(ConvV2Params), C.c_void_p]),
    'eg3d_conv2d_v3_supported': (C.c_int, [C.POINTER(ConvV2Params)]),
    'eg3d_conv2d_v3': (C.c_int, [C.POINTER(ConvV2Params), C.c_void_p]),
    'eg3d_conv2d_ws_supported': (C.c_int, [C.POINTER(ConvWsParams)]),
    'eg3d_conv2d_ws': (C.c_int, [C.POINTER(ConvWsParams), C.c_void_p]),
    'eg3d_conv2d_wgrad_v2_supported': (C.c_int, [C.POINTER(WgradV2Params)]),
    'eg3d_conv2d_wgrad_v2': (C.c_int, [C.POINTER(WgradV2Params), C.c_void_p]),
    'eg3d_conv2d_wgrad_v2_up_supported': (C.c_int, [C.POINTER(WgradV2Params)]),
    'eg3d_conv2d_wgrad_v2_up': (C.c_int, [C.POINTER(WgradV2Params), C.c_void_p]),
    'eg3d_conv2d_wgrad_v2_slabs': (C.c_int, [C.POINTER(WgradV2Params)]),
    'eg3d_weight_grad_finish_slabs': (C.c_int, [C.c_void_p, C.c_int, C.c_int64, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_int, C.c_int, C.c_int, C.c_int, C.c_void_p]),
    'eg3d_split_activation_bytes': (C.c_int64, [C.c_int, C.c_int, C.c_int, C.c_int]),
    'eg3d_split_activation': (C.c_int, [C.c_void_p] * 6 + [C.c_int] * 5 + [C.c_void_p]),
    'eg3d_split_weight': (C.c_int, [C.c_void_p] * 4 + [C.c_int] * 4 + [C.c_void_p]),
    'eg3d_split_weights_batched': (C.c_int, [C.POINTER(SplitWBatch), C.c_void_p]),
    'eg3d_absmax': (C.c_int, [C.c_void_p, C.c_int64, C.c_void_p, C.c_void_p]),
    'eg3d_upconv_epilogue_fwd': (C.c_int, [C.c_void_p, C.c_void_p] + [C.c_int] * 6 + [C.c_void_p, C.c_int, C.c_float, C.c_void_p, C.c_void_p, C.c_int64,
                                           C.c_void_p, C.c_void_p, C.c_int, C.c_float, C.c_float, C.c_float, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p]),
    'eg3d_filtered_lrelu_act': (C.c_int, [C.c_void_p, C.c_void_p] + [C.c_int] * 8 + [C.c_float] * 3 + [C.c_int, C.c_void_p]),
    'eg3d_conv2d_v2_s2adj_supported': (C.c_int, [C.POINTER(ConvV2Params)]),
    'eg3d_conv2d_v2_s2adj': (C.c_int, [C.POINTER(ConvV2Params), C.c_void_p]),
    'eg3d_conv2d_v3_s2adj_supported': (C.c_int, [C.POINTER(ConvV2Params)]),
    'eg3d_conv2d_v3_s2adj': (C.c_int, [C.POINTER(ConvV2Params), C.c_void_p]),
    'eg3d_fir44_adjoint_split_bytes': (C.c_int64, [C.c_int] * 4),
    'eg3d_fir44_adjoint_split': (C.c_int, [C.c_void_p] * 4 + [C.c_int] * 5 + [C.c_float, C.c_void_p]),
    'eg3d_conv2d_up2_supported': (C.c_int, [C.POINTER(ConvUp2Params)]),
    'eg3d_conv2d_up2': (C.c_int, [C.POINTER(ConvUp2Params), C.c_void_p]),
    'eg3d_probe_mfma_f16': (C.c_int, [C.c_void_p, C.c_void_p, C.c_int, C.c_int, C.c_void_p]),
    'eg3d_grid_sample_nhwc_fwd': (C.c_int, [C.c_void_p, C.c_void_p, C.c_void_p] + [C.c_int] * 6 + [C.c_void_p]),
    'eg3d_grid_sample_nhwc_bwd': (C.c_int, [C.c_void_p] * 5 + [C.c_int] * 6 + [C.c_void_p]),
    'eg3d_det_enabled': (C.c_int, []),
    'eg3d_det_workspace_bytes': (C.c_int64, [C.c_int64]),
    'eg3d_det_set_workspace': (C.c_int, [C.c_void_p, C.c_int64, C.c_void_p]),
    'eg3d_det_misses': (C.c_int, [C.POINTER(C.c_uint32), C.c_void_p]),
    'eg3d_det_accumulate': (C.c_int, [C.c_void_p, C.c_int64, C.c_void_p, C.c_void_p]),
    'eg3d_modconv_epilogue_fwd': (C.c_int, [C.c_void_p, C.c_void_p, C.c_int, C.c_int, C.c_int, C.c_int, C.c_int, C.c_int,
                                            C.c_void_p, C.c_int, C.c_int, C.c_int, C.c_float, C.c_void_p, C.c_void_p, C.c_int64,
                                            C.c_void_p, C.c_void_p, C.c_int, C.c_float, C.c_float, C.c_float, C.c_void_p, C.c_void_p]),
    'eg3d_modconv_epilogue_bwd': (C.c_int, [C.c_void_p, C.c_void_p, C.c_void_p, C.c_int, C.c_int, C.c_int, C.c_int, C.c_void_p,
                                            C.c_void_p, C.c_int64, C.c_void_p, C.c_void_p, C.c_int, C.c_float, C.c_float,
                                            C.c_float, C.c_void_p, C.c_void_p, C.c_void_p, C.c_int64, C.c_void_p, C.c_void_p, C.c_void_p]),
    'eg3d_dgrad_finish': (C.c_int, [C.c_void_p] * 6 + [C.c_int, C.c_int, C.c_int, C.c_int, C.c_void_p]),
    'eg3d_dgrad_finish_act': (C.c_int, [C.c_void_p] * 6 + [C.c_int] * 4 + [C.POINTER(ActBwd), C.c_void_p, C.c_void_p]),
    'eg3d_torgb_dgrad_act': (C.c_int, [C.c_void_p] * 7 + [C.c_int] * 4 + [C.POINTER(ActBwd), C.c_void_p, C.c_void_p]),
    'eg3d_torgb_dgrad_act_split': (C.c_int, [C.c_void_p] * 7 + [C.c_int] * 4 + [C.POINTER(ActBwd)] + [C.c_void_p] * 5),
    'eg3d_upfirdn2d_nhwc': (C.c_int, [C.c_void_p, C.c_void_p, C.c_void_p] + [C.c_int] * 13 + [C.c_float, C.c_int, C.c_int,
                                                                                             C.c_int, C.c_void_p]),
    'eg3d_upfirdn2d_nhwc_add': (C.c_int, [C.c_void_p] * 4 + [C.c_int] * 13 + [C.c_float, C.c_int, C.c_int, C.c_void_p]),
    'eg3d_weight_sqsum': (C.c_int, [C.c_void_p, C.c_void_p, C.c_int, C.c_int, C.c_int, C.c_void_p]),
    'eg3d_demod_fwd': (C.c_int, [C.c_void_p, C.c_void_p, C.c_void_p, C.c_int, C.c_int, C.c_int, C.c_void_p]),
    'eg3d_demod_bwd': (C.c_int, [C.c_void_p] * 6 + [C.c_int, C.c_int, C.c_int, C.c_void_p]),
    'eg3d_noise_reg_workspace_floats': (C.c_int64, [C.POINTER(C.c_int32), C.c_int]),
    'eg3d_noise_regularizer': (C.c_int, [C.POINTER(C.c_void_p), C.POINTER(C.c_void_p), C.POINTER(C.c_int32), C.c_int, C.c_void_p, C.c_void_p,
                                         C.c_float, C.c_void_p]),
    'eg3d_noise_normalize': (C.c_int, [C.POINTER(C.c_void_p), C.POINTER(C.c_int32), C.c_int, C.c_void_p, C.c_void_p]),
    'eg3d_maxpool2d_fwd': (C.c_int, [C.c_void_p] * 3 + [C.c_int] * 7 + [C.c_void_p]),
    'eg3d_maxpool2d_bwd': (C.c_int, [C.c_void_p] * 3 + [C.c_int] * 7 + [C.c_void_p]),
    'eg3d_unit_normalize_fwd': (C.c_int, [C.c_void_p] * 3 + [C.c_int] * 4 + [C.c_float, C.c_float, C.c_int64, C.c_int, C.c_void_p]),
    'eg3d_image_prepare_fwd': (C.c_int, [C.c_void_p, C.c_void_p, C.c_int, C.c_int, C.c_int, C.c_int, C.c_float, C.c_float, C.c_void_p]),
    'eg3d_image_prepare_bwd': (C.c_int, [C.c_void_p, C.c_void_p, C.c_int, C.c_int, C.c_int, C.c_int, C.c_float, C.c_void_p]),
    'eg3d_conv3x3_direct': (C.c_int, [C.POINTER(Conv3x3DirectParams), C.c_void_p]),
    'eg3d_pool2_act_bwd': (C.c_int, [C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_int, C.c_int, C.c_int, C.c_int, C.c_float, C.c_float, C.c_void_p]),
    'eg3d_sqdist_fwd': (C.c_int, [C.c_void_p, C.c_void_p, C.c_void_p, C.c_int, C.c_int64, C.c_void_p]),
    'eg3d_sqdist_bwd': (C.c_int, [C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_int, C.c_int64, C.c_void_p]),
    'eg3d_sqdist_sum_fwd': (C.c_int, [C.c_void_p, C.c_void_p, C.c_int64, C.c_void_p, C.c_float, C.c_void_p, C.c_float, C.c_void_p]),
    'eg3d_sqdist_sum_bwd': (C.c_int, [C.c_void_p, C.c_void_p, C.c_void_p, C.c_float, C.c_void_p, C.c_int64, C.c_void_p]),
    'eg3d_tv_norm_fwd': (C.c_int, [C.c_void_p, C.c_int, C.c_int, C.c_int, C.c_void_p, C.c_float, C.c_void_p, C.c_float, C.c_void_p]),
    'eg3d_tv_norm_bwd': (C.c_int, [C.c_void_p, C.c_void_p, C.c_float, C.c_void_p, C.c_int, C.c_int, C.c_int, C.c_void_p]),
    'eg3d_slice_rgb4_fwd': (C.c_int, [C.c_void_p, C.c_void_p, C.c_int64, C.c_int, C.c_void_p]),
    'eg3d_slice_rgb4_bwd': (C.c_int, [C.c_void_p, C.c_void_p, C.c_int64, C.c_int, C.c_void_p]),
    'eg3d_slice_rgb4_bwd_add': (C.c_int, [C.c_void_p, C.c_void_p, C.c_void_p, C.c_int64, C.c_int, C.c_void_p]),
    'eg3d_warp_project_fwd': (C.c_int, [C.c_void_p] * 5 + [C.c_int64, C.c_void_p]),
    'eg3d_warp_project_bwd': (C.c_int, [C.c_void_p] * 8 + [C.c_int64, C.c_void_p]),
    'eg3d_torgb_small_supported': (C.c_int, [C.POINTER(TorgbSmallParams)]),
    'eg3d_torgb_small_fwd': (C.c_int, [C.POINTER(TorgbSmallParams), C.c_void_p]),
    'eg3d_torgb_small_bwd_supported': (C.c_int, [C.POINTER(TorgbSmallBwdParams)]),
    'eg3d_torgb_small_bwd': (C.c_int, [C.POINTER(TorgbSmallBwdParams), C.c_void_p]),
    'eg3d_torgb_mid_supported': (C.c_int, [C.POINTER(TorgbSmallParams)]),
    'eg3d_torgb_mid_bwd_supported': (C.c_int, [C.POINTER(TorgbSmallBwdParams)]),
    'eg3d_adam_step': (C.c_int, [C.POINTER(AdamList), C.c_void_p, C.c_void_p]),
    'eg3d_pose_chain_fwd': (C.c_int, [C.c_void_p, C.c_void_p, C.c_void_p, C.c_int, C.c_int, C.c_float, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p]),
    'eg3d_pose_chain_bwd': (C.c_int, [C.c_void_p, C.c_void_p, C.c_void_p, C.c_int, C.c_int, C.c_void_p, C.c_void_p, C.c_void_p]),
    'eg3d_early_stop_flag': (C.c_int, [C.c_void_p, C.c_float, C.c_void_p, C.c_void_p]),
    'eg3d_unit_normalize_levels': (C.c_int, [C.POINTER(UnitLevels), C.c_int, C.c_void_p]),
    'eg3d_unit_normalize_bwd': (C.c_int, [C.c_void_p] * 4 + [C.c_int] * 4 + [C.c_float, C.c_float, C.c_int64, C.c_int, C.c_void_p]),
    'eg3d_pack_conv_weight': (C.c_int, [C.c_void_p] * 4 + [C.c_int] * 3 + [C.c_void_p]),
    'eg3d_weight_grad_finish': (C.c_int, [C.c_void_p] * 6 + [C.c_int] * 4 + [C.c_void_p]),
    'eg3d_weight_grad_finish_batched': (C.c_int, [C.POINTER(WgfItem), C.c_int, C.c_void_p]),
    'eg3d_pack_conv_weights_batched': (C.c_int, [C.POINTER(PackItem), C.c_int, C.c_void_p]),
    'eg3d_pack_conv_weight_padded': (C.c_int, [C.c_void_p] * 4 + [C.c_int] * 4 + [C.c_void_p]),
    'eg3d_pack_conv_weight_scaled': (C.c_int, [C.c_void_p] * 4 + [C.c_int] * 3 + [C.c_void_p]),
    'eg3d_unpack_weight_grad': (C.c_int, [C.c_void_p] * 5 + [C.c_int] * 4 + [C.c_void_p]),
    'eg3d_rows_gram': (C.c_int, [C.c_void_p, C.c_void_p, C.c_int64, C.c_int, C.c_int, C.c_void_p, C.c_void_p, C.c_void_p]),
    'eg3d_rows_gram_scaled': (C.c_int, [C.c_void_p, C.c_void_p, C.c_int64, C.c_int, C.c_int, C.c_void_p, C.c_void_p, C.c_float, C.c_float, C.c_void_p]),
    'eg3d_filtered_lrelu': (C.c_int, [C.POINTER(FlreluParams), C.c_void_p]),
    'eg3d_style_affine_fwd': (C.c_int, [C.POINTER(StyleBank), C.c_void_p]),
    'eg3d_style_affine_bwd': (C.c_int, [C.POINTER(StyleBank), C.c_void_p]),
    'eg3d_ray_gen_fwd': (C.c_int, [C.c_void_p] * 4 + [C.c_int, C.c_int, C.c_void_p]),
    'eg3d_ray_gen_bwd': (C.c_int, [C.c_void_p] * 6 + [C.c_int, C.c_int, C.c_void_p]),
    'eg3d_render_fwd': (C.c_int, [C.POINTER(RenderParams), C.c_void_p]),
    'eg3d_render_finalize': (C.c_int, [C.c_void_p, C.c_void_p, C.c_int64, C.c_void_p]),
    'eg3d_render_bwd': (C.c_int, [C.POINTER(RenderBwdParams), C.c_void_p]),
    'eg3d_render_query_sizes': (C.c_int, [C.c_void_p, C.c_void_p]),
    'eg3d_triplane_scatter_workspace_ints': (C.c_int64, [C.c_int64, C.c_int, C.c_int, C.c_int]),
    'eg3d_triplane_scatter': (C.c_int, [C.c_void_p, C.c_void_p, C.c_int64, C.c_int64, C.c_void_p, C.c_int, C.c_int, C.c_int, C.c_int, C.c_float,
                                        C.c_void_p, C.c_int, C.c_int, C.c_void_p, C.c_void_p]),
    'eg3d_sample_decode': (C.c_int, [C.POINTER(RenderParams), C.c_void_p, C.c_int64, C.c_void_p, C.c_void_p, C.c_void_p]),
}

EXPORTED_SYMBOLS = tuple(_SIGS)
_lib = None


class Eg3dHipError(RuntimeError):
    pass


def lib():
    """Load (once) and return the ctypes handle.  Raises if the extension has not been built."""
    global _lib
    if _lib is None:
        if not os.path.exists(LIB_PATH):
            raise Eg3dHipError(f'{LIB_PATH} not found: build it with `make -C 3dgan-inversion_amd` '
                               f'(or python -c "import __graft_entry__ as g; g.build()"). There is no fallback path.')
        h = C.CDLL(LIB_PATH)
        for name, (res, args) in _SIGS.items():
            fn = getattr(h, name)          # AttributeError if a declared symbol is not exported
            fn.restype = res
            fn.argtypes = args
        _lib = h
        if h.eg3d_det_enabled() and torch.cuda.is_available():
            det_enable()
    return _lib


_det_ws = None


def det_enable(max_elements: int = None, device=None):
    """Lend the deterministic build its accumulator workspace (32 bytes per float a single library call accumulates into; the default
    covers a 512^2 x 128 split-K output and the plane gradients of a batch of 8; EG3D_DET_ELEMENTS overrides it).  Called by hipops on the first GPU use when EG3D_DETERMINISTIC=1; no-op in the normal build."""
    global _det_ws
    h = lib()
    if not h.eg3d_det_enabled():
        return False
    if _det_ws is None or (max_elements is not None and _det_ws[1] < max_elements):
        n = int(max_elements or int(os.environ.get('EG3D_DET_ELEMENTS', str(96 * 1024 * 1024))))
        nbytes = h.eg3d_det_workspace_bytes(n)
        ws = torch.zeros(nbytes // 8 + 1, dtype=torch.int64, device=device or 'cuda')
        check(h.eg3d_det_set_workspace(ws.data_ptr(), nbytes, stream_ptr()), 'det_set_workspace')
        _det_ws = (ws, n)
    return True


def det_misses() -> int:
    """Additions of the deterministic build that fell back to a float atomic (a target the call did not bind): must stay 0."""
    out = C.c_uint32(0)
    check(lib().eg3d_det_misses(C.byref(out), stream_ptr()), 'det_misses')
    return int(out.value)


def check(status, what=''):
    if status != 0:
        msg = lib().eg3d_status_string(int(status)).decode()
        raise Eg3dHipError(f'{what or "eg3d kernel"} failed: status {status} ({msg})')


def stream_ptr():
    return C.c_void_p(torch.cuda.current_stream().cuda_stream)


def ptr(t):
    """Device pointer of a tensor (None -> NULL)."""
    if t is None:
        return None
    return C.c_void_p(t.data_ptr())


def require_cuda(*tensors):
    for t in tensors:
        if t is not None and not t.is_cuda:
            raise Eg3dHipError('eg3d HIP path needs tensors on an AMD GPU (cuda device); there is no CPU fallback. '
                               'Use oracle/eg3d_oracle.py only as a test checker.')
