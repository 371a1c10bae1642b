"""ctypes binding of libfcn8s_hip.so (C ABI declared in include/fcn8s_hip.h).

There is no CPU fallback: if the shared library is missing the import of this
module raises, and every compute entry point needs an AMD GPU (gfx950).
"""
from __future__ import annotations

import ctypes as C
import os

_HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.path.join(_HERE, "libfcn8s_hip.so")

OK, ERR_BAD_ARG, ERR_SHAPE, ERR_OOM, ERR_HIP, ERR_STATE, ERR_NOT_FOUND, ERR_RCCL = range(8)
HOST, DEVICE = 0, 1
IMG_U8, IMG_F32 = 0, 1
OPT_TF_ADAM, OPT_SGD_MOMENTUM, OPT_NONE = 0, 1, 2
MAX_BUCKETS = 8          # the number of gradient buckets is a run-time value: lib.fcn8s_num_buckets(handle) / fcn8s_layout_num_buckets(cfg)
COMM_ID_BYTES = 128
NUM_STAGE_SLOTS = 3
PREC_F32, PREC_BF16_FC, PREC_F32X3, PREC_BF16_FWD, PREC_F32X2, PREC_BF16_FWD_X2, PREC_BF16_TRAIN = 0, 1, 2, 3, 4, 5, 6


class Config(C.Structure):
    _fields_ = [("num_classes", C.c_int32), ("fc6_ksize", C.c_int32), ("widths", C.c_int32 * 7),
                ("device_id", C.c_int32), ("seed", C.c_uint64),
                ("ext_params", C.c_void_p), ("ext_grads", C.c_void_p)]


# name -> (restype, argtypes); every symbol include/fcn8s_hip.h declares
_p, _i, _f, _sz = C.c_void_p, C.c_int, C.c_float, C.c_size_t
_i64, _dp, _i64p, _fp = C.c_int64, C.POINTER(C.c_double), C.POINTER(C.c_int64), C.POINTER(C.c_float)
SIGNATURES = {
    "fcn8s_param_floats": (_sz, [C.POINTER(Config)]),
    "fcn8s_layout_num_params": (_i, [C.POINTER(Config)]),
    "fcn8s_layout_param": (_i, [C.POINTER(Config), _i, C.c_char_p, C.POINTER(C.c_int32), C.POINTER(_i64 * 4), _i64p]),
    "fcn8s_layout_num_buckets": (_i, [C.POINTER(Config)]),
    "fcn8s_layout_bucket": (_i, [C.POINTER(Config), _i, C.POINTER(_sz), C.POINTER(_sz)]),
    "fcn8s_create": (_i, [C.POINTER(Config), C.POINTER(_p)]),
    "fcn8s_destroy": (_i, [_p]),
    "fcn8s_last_error": (C.c_char_p, [_p]),
    "fcn8s_set_stream": (_i, [_p, _p]),
    "fcn8s_synchronize": (_i, [_p]),
    "fcn8s_num_params": (_i, [_p]),
    "fcn8s_param_info": (_i, [_p, _i, C.POINTER(C.c_char_p), C.POINTER(C.c_int32), C.POINTER(_i64 * 4), _i64p]),
    "fcn8s_param_index": (_i, [_p, C.c_char_p]),
    "fcn8s_set_param": (_i, [_p, C.c_char_p, _p, _sz]),
    "fcn8s_get_param": (_i, [_p, C.c_char_p, _p, _sz]),
    "fcn8s_get_grad": (_i, [_p, C.c_char_p, _p, _sz]),
    "fcn8s_param_buffer": (_p, [_p, C.POINTER(_sz)]),
    "fcn8s_grad_buffer": (_p, [_p, C.POINTER(_sz)]),
    "fcn8s_num_buckets": (_i, [_p]),
    "fcn8s_bucket_range": (_i, [_p, _i, C.POINTER(_sz), C.POINTER(_sz)]),
    "fcn8s_init_params": (_i, [_p, C.c_uint64]),
    "fcn8s_onehot_to_ids": (_i, [_p, _p, _i, _i64, _i, _p, _p]),
    "fcn8s_train_step": (_i, [_p, _p, _i, _p, _i, _i, _i, _f, _f, _f, _i, _fp, _i64p]),
    "fcn8s_forward_loss": (_i, [_p, _p, _i, _p, _i, _i, _i, _f, _f, _i]),
    "fcn8s_backward_bucket": (_i, [_p, _i]),
    "fcn8s_bucket_complete_after": (_i, [_p, _i]),
    "fcn8s_bucket_wait": (_i, [_p, _i, _p]),
    "fcn8s_device_pci_bus_id": (_i, [_i, C.c_char_p, _sz]),
    "fcn8s_comm_unique_id": (_i, [_p, _sz]),
    "fcn8s_comm_init": (_i, [_p, _p, _sz, _i, _i]),
    "fcn8s_comm_destroy": (_i, [_p]),
    "fcn8s_comm_info": (_i, [_p, C.POINTER(_i), C.POINTER(_i), C.POINTER(_i)]),
    "fcn8s_allreduce_bucket": (_i, [_p, _i]),
    "fcn8s_comm_wait": (_i, [_p]),
    "fcn8s_comm_broadcast_params": (_i, [_p, _i]),
    "fcn8s_comm_allreduce_metrics": (_i, [_p]),
    "fcn8s_apply_update": (_i, [_p, _i, _f, _f]),
    "fcn8s_read_loss": (_i, [_p, _fp]),
    "fcn8s_eval_step": (_i, [_p, _p, _i, _p, _i, _i, _i, _f, _i]),
    "fcn8s_metrics_reset": (_i, [_p]),
    "fcn8s_metrics_get": (_i, [_p, _dp, _dp, _dp]),
    "fcn8s_metrics_get_ex": (_i, [_p, _dp, _dp, _dp, _i]),
    "fcn8s_stage_inputs": (_i, [_p, _i, _p, _i, _p, _i, _i, _i, C.POINTER(_p), C.POINTER(_p)]),
    "fcn8s_stage_wait": (_i, [_p, _i]),
    "fcn8s_stage_release": (_i, [_p, _i]),
    "fcn8s_metrics_raw": (_i, [_p, _p, _dp, _i64p]),
    "fcn8s_metrics_set_raw": (_i, [_p, _p, C.c_double, _i64]),
    "fcn8s_predict": (_i, [_p, _p, _i, _i, _i, _i, _i, _p, _i]),
    "fcn8s_global_step": (_i64, [_p]),
    "fcn8s_set_global_step": (_i, [_p, _i64]),
    "fcn8s_get_opt_state": (_i, [_p, _p, _p, _sz]),
    "fcn8s_set_opt_state": (_i, [_p, _p, _p, _sz]),
    "fcn8s_freeze_params": (_i, [_p, _i]),
    "fcn8s_set_precision": (_i, [_p, _i]),
    "fcn8s_get_precision": (_i, [_p]),
    "fcn8s_set_option": (_i, [_p, C.c_char_p, _i64]),
    "fcn8s_get_option": (_i, [_p, C.c_char_p, _i64p]),
    "fcn8s_get_activation": (_i, [_p, C.c_char_p, _p, _sz]),
    "fcn8s_get_dropout_masks": (_i, [_p, _p, _sz, _p, _sz]),
    "fcn8s_get_pool_routing": (_i, [_p, _i, _p, _sz]),
    "fcn8s_get_relu_record": (_i, [_p, C.c_char_p, _p, _sz]),
    "fcn8s_crc32c": (C.c_uint32, [_p, _sz, C.c_uint32]),
    "fcn8s_profile_enable": (_i, [_p, _i]),
    "fcn8s_profile_reset": (_i, [_p]),
    "fcn8s_profile_num_groups": (_i, [_p]),
    "fcn8s_profile_get": (_i, [_p, _i, C.POINTER(C.c_char_p), _dp, _i64p, _dp, _dp]),
    "fcn8s_op_preprocess": (_i, [_p, _p, _i, _p, _i64]),
    "fcn8s_op_augment_u8": (_i, [_p, _p, _p, _p, _p, _p, _p, _i, _i, _i, _i, _i, _i]),
    "fcn8s_op_resample_u8": (_i, [_p, _p, _p, _p, _p, _p, _i, _i, _i, _i, _i, _i]),
    "fcn8s_op_conv2d": (_i, [_p, _p, _p, _p, _p, _i, _i, _i, _i, _i, _i, _i]),
    "fcn8s_op_conv2d_winograd": (_i, [_p, _p, _p, _p, _p, _i, _i, _i, _i, _i, _i, _i, _i]),
    "fcn8s_op_conv3x3_winograd_fwd_bwd": (_i, [_p] * 11 + [_i] * 8),
    "fcn8s_op_conv2d_bf16": (_i, [_p, _p, _p, _p, _p, _i, _i, _i, _i, _i, _i, _i]),
    "fcn8s_op_conv2d_bf16_train": (_i, [_p, _p, _p, _p, _p, _i, _p, _p, _p, _p, _p, _i, _i, _i, _i, _i, _i]),
    "fcn8s_op_conv2d_bwd": (_i, [_p, _p, _p, _p, _p, _p, _p, _i, _i, _i, _i, _i, _i]),
    "fcn8s_op_maxpool2x2": (_i, [_p, _p, _p, _i, _i, _i, _i]),
    "fcn8s_op_maxpool2x2_bwd": (_i, [_p, _p, _p, _p, _i, _i, _i, _i, _i]),
    "fcn8s_op_conv2d_transpose": (_i, [_p, _p, _p, _p, _p, _p, _i, _i, _i, _i, _i, _i]),
    "fcn8s_op_conv2d_transpose_bwd": (_i, [_p, _p, _p, _p, _p, _p, _p, _i, _i, _i, _i, _i, _i]),
    "fcn8s_op_softmax_xent": (_i, [_p, _p, _p, _p, _p, _i64, _i]),
    "fcn8s_op_softmax_argmax": (_i, [_p, _p, _p, _p, _i64, _i]),
    "fcn8s_op_confusion": (_i, [_p, _p, _p, _i64, _p, _i]),
    "fcn8s_op_tf_adam": (_i, [_p, _p, _p, _p, _p, _i64, _i, _f, _f, _f, _f, _f]),
    "fcn8s_op_sgd_momentum": (_i, [_p, _p, _p, _p, _i64, _f, _f, _f]),
}


def _load():
    if not os.path.isfile(LIB_PATH):
        raise ImportError(
            "libfcn8s_hip.so not found at %s. Build it with `python -c 'import __graft_entry__ as g; g.build()'` "
            "or `make -C fcn8s_tensorflow_amd/csrc`. There is no CPU fallback for the FCN-8s hot path." % LIB_PATH)
    # PyTorch-ROCm ships its own HIP runtime (torch/lib/libamdhip64.so); this library is linked against the one under /opt/rocm.  Whichever
    # is loaded first serves both, and a process that ends up with the two of them sees "no ROCm-capable device" from the second
    # (observed: `from fcn8s_tensorflow_amd.fcn8s import FCN8s` before any `import torch`).  The engine keeps its buffers in torch tensors
    # anyway, so torch goes first -- always the same single runtime, whatever the caller's import order.
    try:
        import torch  # noqa: F401
    except ImportError:
        pass
    lib = C.CDLL(LIB_PATH)
    for name, (res, args) in SIGNATURES.items():
        fn = getattr(lib, name)          # AttributeError if the library does not export a declared symbol
        fn.restype = res
        fn.argtypes = args
    return lib


lib = _load()


class Fcn8sError(RuntimeError):
    pass


def check(rc, handle=None):
    """Map a C status to the Python exception the reference would raise."""
    if rc == OK:
        return
    msg = lib.fcn8s_last_error(handle)
    msg = msg.decode() if msg else "error %d" % rc
    if rc in (ERR_BAD_ARG, ERR_SHAPE, ERR_NOT_FOUND):
        raise ValueError(msg)
    if rc == ERR_OOM:
        raise MemoryError(msg)
    raise Fcn8sError(msg)
