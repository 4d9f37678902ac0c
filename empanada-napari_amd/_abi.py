"""ctypes binding of libempanada_hip.so (include/empanada_hip.h).

There is deliberately no fallback: if the library is missing or a symbol
cannot be resolved the import of the GPU engine fails loudly.
"""
import ctypes as C
import os

from . import build as _build

_lib = None

c_int, c_i32, c_i64, c_f32, c_f64 = C.c_int, C.c_int32, C.c_int64, C.c_float, C.c_double
vp, cp, sz = C.c_void_p, C.c_char_p, C.c_size_t


class PdlConfig(C.Structure):
    _fields_ = [
        ('num_classes', c_i32), ('stage4_stride', c_i32), ('decoder_channels', c_i32),
        ('aspp_channels', c_i32), ('n_stages', c_i32), ('low_level_stages', c_i32 * 3),
        ('low_level_proj_sem', c_i32 * 3), ('low_level_proj_ins', c_i32 * 3),
        ('atrous_rates', c_i32 * 3), ('ins_decoder', c_i32), ('num_fc', c_i32),
        ('subdivision_num_points', c_i32), ('arch', c_i32), ('fpn_dim', c_i32), ('fpn_layers', c_i32),
        ('encoder', c_i32), ('rn_stem', c_i32), ('rn_widths', c_i32 * 4), ('rn_depths', c_i32 * 4),
        ('rn_groups', c_i32 * 4), ('rn_strides', c_i32 * 4), ('rn_se', c_i32),
    ]


# name -> (restype, argtypes); must cover every EMP_API prototype of the header
PROTOTYPES = {
    'emp_last_error': (cp, []),
    'emp_abi_version': (c_int, []),
    'emp_device_count': (c_int, []),
    'emp_pdl_create': (c_int, [C.POINTER(PdlConfig), C.POINTER(vp)]),
    'emp_pdl_destroy': (None, [vp]),
    'emp_pdl_set_param': (c_int, [vp, cp, vp, C.POINTER(c_i64), c_int, vp]),
    'emp_pdl_finalize': (c_int, [vp]),
    'emp_pdl_set_precision': (c_int, [vp, c_int]),
    'emp_pdl_precision': (c_int, [vp]),
    'emp_pdl_num_params': (c_int, [vp]),
    'emp_pdl_param_name': (cp, [vp, c_int]),
    'emp_pdl_reserve': (c_int, [vp, c_int, c_int, c_int]),
    'emp_pdl_arena_bytes': (sz, [vp]),
    'emp_pdl_forward': (c_int, [vp, vp, c_int, c_f32, c_f32, c_int, c_int, c_int, c_int, c_int, vp, vp, vp, vp]),
    'emp_pdl_forward_padded': (c_int, [vp, vp, c_int, c_f32, c_f32, c_int, c_int, c_int, c_int, c_int, c_int, c_int, vp,
                                       vp, vp, vp]),
    'emp_pdl_flops': (c_f64, [vp, c_int, c_int, c_int, c_int]),
    'emp_pdl_profile': (c_int, [vp, c_int]),
    'emp_pdl_profile_read': (c_int, [vp, C.POINTER(c_f64), C.POINTER(c_f64), C.POINTER(c_int)]),
    'emp_pdl_tap': (c_int, [vp, cp, C.POINTER(vp), C.POINTER(c_i64)]),
    'emp_pdl_num_taps': (c_int, [vp]),
    'emp_pdl_tap_raw': (c_int, [vp, cp, C.POINTER(vp), C.POINTER(c_i64)]),
    'emp_pdl_tap_name': (cp, [vp, c_int]),
    'emp_copy_d2d': (c_int, [vp, vp, sz, vp]),
    'emp_conv2d_nhwc_f16': (c_int, [vp, c_int, c_int, c_int, c_int, c_int, vp, vp, vp, vp, c_int, vp, c_int, c_int,
                                    c_int, c_int, c_int, c_int, c_int, c_int, c_int, vp]),
    'emp_conv256_pack_weights': (c_int, [vp, vp, c_int, c_int, c_int, c_int, vp]),
    'emp_conv2d_grouped_nhwc_f32': (c_int, [vp, c_int, c_int, c_int, c_int, c_int, c_int, c_int, vp, vp, vp, c_int, c_int, c_int,
                                            c_int, c_int, c_int, c_int, c_int, vp]),
    'emp_conv2d_nhwc_f32': (c_int, [vp, c_int, c_int, c_int, c_int, c_int, vp, vp, vp, vp, c_int, vp, c_int, c_int,
                                    c_int, c_int, c_int, c_int, c_int, c_int, vp]),
    'emp_conv2d_nhwc_f16x3': (c_int, [vp, c_int, c_int, c_int, c_int, c_int, vp, vp, vp, vp, c_int, vp, c_int, c_int,
                                      c_int, c_int, c_int, c_int, c_int, c_int, c_int, c_int, vp]),
    'emp_conv2d_nhwc_f16x3_ex': (c_int, [vp, c_int, c_int, c_int, c_int, c_int, vp, vp, vp, vp, c_int, vp, c_int, c_int, c_int,
                                         c_int, c_int, c_int, c_int, c_int, c_int, c_int, vp, c_int, c_int, c_int, c_int, c_int,
                                         vp, vp, c_int, vp, vp]),
    'emp_sepconv_x3_pack': (c_int, [vp, vp, c_int, c_int, c_int, vp, vp, vp]),
    'emp_sepconv_x3_nhwc_f32': (c_int, [vp, c_int, c_int, c_int, c_int, c_int, vp, vp, vp, c_int, c_int, vp, c_int, vp, vp, c_int, vp,
                                        c_int, vp]),
    'emp_hl32_from_f32': (c_int, [vp, vp, c_i64, c_int, c_int, c_int, vp]),
    'emp_hl32_to_f32': (c_int, [vp, vp, c_i64, c_int, c_int, c_int, vp]),
    'emp_x3p_pack_weights': (c_int, [vp, vp, c_int, c_int, vp]),
    'emp_conv2d_hl32_f16x3': (c_int, [vp, c_int, c_int, c_int, c_int, c_int, vp, vp, vp, vp, c_int, c_int, vp, c_int, c_int, c_int,
                                      c_int, c_int, c_int, c_int, c_int, c_int, vp]),
    'emp_conv2d_hl32_f16x3_ksplit': (c_int, [vp, c_int, c_int, c_int, c_int, c_int, vp, vp, vp, vp, c_int, c_int, vp, c_int, c_int, vp, c_int, c_int,
                                             c_int, c_int, c_int, c_int, c_int, c_int, c_int, vp, c_i64, vp]),
    'emp_conv1x1_dual_nhwc_f16': (c_int, [vp, c_int, c_int, c_int, c_int, c_int, vp, c_int, c_int, c_int, c_int, c_int, vp, vp, vp,
                                  c_int, c_int, c_int, c_int, vp]),
    'emp_dwconv_nhwc_f16': (c_int, [vp, c_int, c_int, c_int, c_int, c_int, vp, c_int, vp, c_int, vp]),
    'emp_sepconv5x5_pack_pw': (c_int, [vp, c_int, c_int, c_int, vp, vp]),
    'emp_sepconv5x5_nhwc_f16': (c_int, [vp, c_int, c_int, c_int, c_int, c_int, vp, vp, vp, c_int, c_int, vp,
                                        c_int, vp, vp, c_int, vp, vp]),
    'emp_sepconvp_pack_dw': (c_int, [vp, c_int, c_int, vp, vp]),
    'emp_sepconvp_pack_pw': (c_int, [vp, c_int, c_int, c_int, vp, vp]),
    'emp_sepconvp_nhwc_f16': (c_int, [vp, c_int, c_int, c_int, c_int, c_int, c_int, vp, vp, vp, c_int, c_int, vp,
                                      c_int, vp, vp, c_int, vp, vp]),
    'emp_sepconvp_ws_pack_pw': (c_int, [vp, c_int, c_int, c_int, vp, vp]),
    'emp_sepconvp_ws_nhwc_f16': (c_int, [vp, c_int, c_int, c_int, c_int, c_int, c_int, vp, vp, vp, c_int, c_int, vp,
                                         c_int, vp, vp, c_int, vp, vp]),
    'emp_sepconv3x3_nhwc_f16': (c_int, [vp, c_int, c_int, c_int, c_int, c_int, vp, vp, vp, c_int, c_int, vp, c_int, vp]),
    'emp_sm_create': (vp, [c_i64, c_i64, c_f64, c_f64, c_int]),
    'emp_sm_destroy': (None, [vp]),
    'emp_sm_push_slice_runs': (c_int, [vp, vp, c_i64, c_i64, c_i64]),
    'emp_sm_push_slices_runs': (c_int, [vp, c_i64, vp, vp, c_i64, c_i64]),
    'emp_sm_push_slice_objects': (c_int, [vp, c_i64, vp, vp, vp, vp, vp]),
    'emp_sm_num_slices': (c_i64, [vp]),
    'emp_lsa_maximize': (c_int, [vp, c_i64, c_i64, vp, vp]),
    'emp_lsa_maximize_sparse': (c_int, [c_i64, c_i64, c_i64, vp, vp, vp, vp, vp]),
    'emp_sm_prepare': (c_int, [vp, c_i64, c_i64]),
    'emp_sm_state_size': (c_int, [vp, c_i64, vp, vp]),
    'emp_sm_export_state': (c_int, [vp, c_i64, vp, vp, vp, vp]),
    'emp_sm_import_state': (c_int, [vp, c_i64, c_i64, vp, vp, vp, c_i64, c_int]),
    'emp_sm_begin_backward': (c_int, [vp]),
    'emp_sm_step_begin': (c_int, [vp, c_i64, C.POINTER(c_int), C.POINTER(c_int)]),
    'emp_sm_iou': (vp, [vp]),
    'emp_sm_step_apply': (c_int, [vp, vp, vp, c_i64]),
    'emp_sm_run': (c_int, [vp, c_i64, c_int, c_i64, c_int, vp]),
    'emp_sm_solver_stats': (c_int, [vp, C.POINTER(c_i64), C.POINTER(c_i64)]),
    'emp_sm_pending_shape': (c_int, [vp, vp, vp]),
    'emp_sm_tracker_init': (c_int, [vp, c_int, c_i64, c_i64, c_i64]),
    'emp_sm_track': (c_int, [vp, c_i64, c_i64]),
    'emp_sm_track_range': (c_int, [vp, c_i64, c_i64, c_i64]),
    'emp_sm_tracker_finish': (c_int, [vp]),
    'emp_sm_num_tracks': (c_i64, [vp]),
    'emp_sm_track_info': (c_int, [vp, c_i64, C.POINTER(c_i64), C.POINTER(c_i64), C.POINTER(c_i64)]),
    'emp_sm_track_runs': (c_int, [vp, c_i64, vp, vp]),
    'emp_sm_tracks_info': (c_int, [vp, vp, vp, vp, C.POINTER(c_i64)]),
    'emp_sm_tracks_runs': (c_int, [vp, vp, vp]),
    'emp_sm_slice_num_objects': (c_i64, [vp, c_i64]),
    'emp_sm_slice_object_info': (c_int, [vp, c_i64, c_i64, C.POINTER(c_i64), C.POINTER(c_i64), C.POINTER(c_i64)]),
    'emp_sm_slice_object_runs': (c_int, [vp, c_i64, c_i64, vp, vp]),
    'emp_logits_to_prob': (c_int, [vp, vp, c_int, c_int, c_int, c_int, vp]),
    'emp_median_slices': (c_int, [C.POINTER(vp), c_int, vp, sz, vp]),
    'emp_median_recursive': (c_int, [vp, vp, c_int, c_int, c_int, vp, sz, vp]),
    'emp_instance_cells_work_bytes': (sz, [c_int, c_int, c_int]),
    'emp_instance_cells': (c_int, [vp, vp, c_int, c_int, c_int, c_f32, c_int, c_int, c_int, vp, vp, vp, c_int, vp, vp]),
    'emp_panoptic_merge_work_bytes': (sz, [c_int, c_int, c_int]),
    'emp_ccl8_work_bytes': (sz, [c_int, c_int, c_int]),
    'emp_ccl8': (c_int, [vp, c_int, c_int, c_int, vp, vp, vp, vp]),
    'emp_force_connected_work_bytes': (sz, [c_int, c_int, c_int]),
    'emp_force_connected': (c_int, [vp, c_int, c_int, c_int, C.POINTER(c_i32), c_int, c_i64, vp, vp, vp]),
    'emp_ccl26': (c_int, [vp, c_int, c_int, c_int, vp, vp, vp, vp]),
    'emp_morph_cross3d': (c_int, [vp, vp, c_int, c_int, c_int, c_int, vp]),
    'emp_fill_holes_slices': (c_int, [vp, c_i64, c_i64, c_i64]),
    'emp_rle_extract_work_bytes': (sz, [c_int, c_int, c_int]),
    'emp_rle_extract': (c_int, [vp, c_int, c_int, c_int, vp, vp, c_int, vp, vp]),
    'emp_ccl_range': (c_int, [vp, c_int, c_int, c_int, c_int, c_int, c_i64, c_i64, vp, vp, vp, vp]),
    'emp_rle_extract_range': (c_int, [vp, c_int, c_int, c_int, c_int, c_i64, c_i64, vp, vp, c_int, vp, vp]),
    'emp_rle_fill': (c_int, [vp, vp, vp, c_i64, vp, c_i64, c_int, vp]),
    'emp_rle_fill_ordered': (c_int, [vp, vp, vp, vp, c_i64, vp, c_i64, c_int, vp, vp]),
    'emp_rle_pair_intersections': (c_int, [vp, vp, vp, vp, c_i64, vp]),
    'emp_ranges_vote': (c_int, [vp, c_i64, c_int, vp, C.POINTER(c_i64)]),
    'emp_gather_segments_i64': (c_int, [vp, vp, vp, vp, vp, vp, c_i64, vp, vp]),
    'emp_panoptic_merge': (c_int, [vp, vp, c_int, c_int, c_int, c_int, c_f32, C.POINTER(c_i32), c_int, c_i64, c_i64,
                                   c_i64, c_int, vp, vp, vp]),
}


class EmpError(RuntimeError):
    pass


def lib_path():
    return _build.LIB


def load(build_if_missing=False):
    """Load the shared library and bind every prototype; raises if anything is missing."""
    global _lib
    if _lib is not None:
        return _lib
    # torch bundles its own HIP runtime (SONAME libamdhip64.so.7); it must be loaded first so
    # that this library binds to the SAME runtime instance (streams and pointers are shared)
    import torch  # noqa: F401
    path = lib_path()
    if not os.path.exists(path):
        if build_if_missing:
            _build.build_all()
        else:
            raise EmpError(f'{path} not found: run `python __graft_entry__.py` (build()) first; '
                           'the HIP engine has no CPU fallback')
    lib = C.CDLL(path)
    for name, (res, args) in PROTOTYPES.items():
        try:
            fn = getattr(lib, name)
        except AttributeError as e:
            raise EmpError(f'{path} does not export {name}') from e
        fn.restype = res
        fn.argtypes = args
    _lib = lib
    return lib


def check(rc, what=''):
    if rc != 0:
        msg = load().emp_last_error()
        raise EmpError(f'{what} failed ({rc}): {msg.decode() if msg else "?"}')


def ptr(t):
    """device/host pointer of a torch tensor (or None)."""
    return None if t is None else C.c_void_p(t.data_ptr())


def stream_ptr(device=None):
    import torch
    return C.c_void_p(torch.cuda.current_stream(device).cuda_stream)
