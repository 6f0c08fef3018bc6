"""ctypes binding of libgm_hip.so (C-ABI declared in include/gm_hip.h).

There is NO fallback: if the HIP library is missing or a call fails, this raises.  The product
path never routes through torch eager kernels or the CPU oracle for the ops bound here."""
import ctypes
import os
from ctypes import POINTER, c_char_p, c_float, c_int, c_int32, c_int64, c_void_p

HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.environ.get("GM_LIB_PATH") or os.path.join(HERE, "libgm_hip.so")

GM_EINVAL = -10001
ACT_ID, ACT_RELU, ACT_SIGMOID = 0, 1, 2
ACT = {"id": ACT_ID, None: ACT_ID, "relu": ACT_RELU, "sigmoid": ACT_SIGMOID}
LOSS = {"ns": 0, "mm": 1, "w": 2, "ls": 3, "ra": 4, "fisher": 5, "f_total_variation": 6,
        "f_forward_kl": 7, "f_reverse_kl": 8, "f_pearson": 9, "f_hellinger": 10,
        "f_jensen_shannon": 11}


class Slot(ctypes.Structure):
    """gm_slot: ((ctr ? *ctr : 0) * mul + add) % ring * stride."""
    _fields_ = [("ctr", c_void_p), ("mul", c_int32), ("add", c_int32), ("ring", c_int32),
                ("stride", c_int64)]


def slot(ctr=0, mul=0, add=0, ring=0, stride=0):
    return Slot(ctr or None, mul, add, ring, stride)


NO_SLOT = Slot(None, 0, 0, 0, 0)


class HeadBwdArgs(ctypes.Structure):
    """gm_head_bwd_args (include/gm_hip.h): gm_head_bwd_fused's arguments as one block."""
    _fields_ = [("H", c_void_p), ("ldh", c_int64), ("dS", c_void_p), ("w2", c_void_p),
                ("b2", c_void_p), ("rowloss", c_void_p), ("dH", c_void_p), ("lddh", c_int64),
                ("gw2", c_void_p), ("gb2", c_void_p), ("loss_out", c_void_p), ("loss_slot", Slot),
                ("inv_b", c_float), ("gen_mode", c_int), ("B", c_int), ("Hd", c_int),
                ("with_adam", c_int), ("mW", c_void_p), ("vW", c_void_p), ("mb", c_void_p),
                ("vb", c_void_p), ("sched", c_void_p), ("sched_slot", Slot),
                ("beta1", ctypes.c_double), ("beta2", ctypes.c_double), ("eps", ctypes.c_double),
                ("weight_decay", ctypes.c_double), ("clamp", c_float), ("tick", c_void_p),
                ("gw2_add", c_void_p), ("pen_s", c_void_p), ("pen_h", c_void_p), ("pen_ldh", c_int64),
                ("pen_t", c_void_p), ("pen_ldt", c_int64), ("pen_rows", c_int), ("gb2_add", c_void_p)]



class HeadFoldArgs(ctypes.Structure):
    """gm_head_fold_args (include/gm_hip.h): the folded critic head's partial dots + loss settings."""
    _fields_ = [("part", c_void_p), ("ldp", c_int64), ("nparts", c_int), ("snap", c_void_p),
                ("variant", c_int), ("out_act", c_int), ("hyper", c_float * 8), ("n_hyper", c_int),
                ("pen", c_void_p), ("S", c_void_p), ("dS", c_void_p), ("rowloss", c_void_p)]


class DwAdamArgs(ctypes.Structure):
    """gm_dw_adam_args (include/gm_hip.h): gm_linear_bwd_dw_adam's arguments as one block."""
    _fields_ = [("dA", c_void_p), ("lda", c_int64), ("X", c_void_p), ("ldx", c_int64),
                ("x_slot", Slot), ("dW", c_void_p), ("db", c_void_p), ("M", c_int), ("K", c_int),
                ("N", c_int), ("pW", c_void_p), ("mW", c_void_p), ("vW", c_void_p),
                ("pb", c_void_p), ("mb", c_void_p), ("vb", c_void_p), ("sched", c_void_p),
                ("sched_slot", Slot), ("beta1", ctypes.c_double), ("beta2", ctypes.c_double),
                ("eps", ctypes.c_double), ("weight_decay", ctypes.c_double), ("clamp", c_float)]


class Finalize2Args(ctypes.Structure):
    """gm_finalize2_args (include/gm_hip.h): the two loss sums + counter tick that ride in a VAE batch's last launch."""
    _fields_ = [("pa", c_void_p), ("na", c_int), ("scale_a", c_float), ("out_a", c_void_p), ("slot_a", Slot),
                ("pb", c_void_p), ("nb", c_int), ("scale_b", c_float), ("out_b", c_void_p), ("slot_b", Slot),
                ("tick", c_void_p), ("done", c_void_p)]


class DrawOp(ctypes.Structure):
    """gm_draw_op (include/gm_hip.h): one draw of the per-iteration host RNG program."""
    _fields_ = [("kind", c_int32), ("n", c_int32), ("a", c_int64), ("b", c_int32), ("c", c_int32),
                ("dst", c_void_p), ("iter_stride", c_int64), ("e0", c_int64), ("e1", c_int64)]


class StageSeg(ctypes.Structure):
    """gm_stage_seg (include/gm_hip.h)."""
    _fields_ = [("src", c_void_p), ("dst", c_void_p), ("bytes_per_iter", c_int64),
                ("blocks", ctypes.c_int32), ("reserved", ctypes.c_int32),
                ("src_block_stride", c_int64), ("dst_block_stride", c_int64)]


DRAW_SAMPLER, DRAW_NORMAL, DRAW_UNIFORM, DRAW_INFO = 0, 1, 2, 3
GM_EUNSUPPORTED = -10002

_P = c_void_p      # device pointers travel as integers (tensor.data_ptr())

_SIGNATURES = {
    "gm_version": (c_int, []),
    "gm_arch": (c_char_p, []),
    "gm_last_error": (c_char_p, []),
    "gm_tick": (c_int, [_P, _P, c_int64]),
    "gm_copy_slot_f32": (c_int, [_P, _P, Slot, _P, Slot, c_int64]),
    "gm_gather_rows": (c_int, [_P, _P, c_int64, _P, Slot, _P, c_int64, c_int, c_int]),
    "gm_linear_fwd": (c_int, [_P, _P, c_int64, Slot, _P, _P, _P, c_int64, c_int, c_int, c_int,
                              c_int]),
    "gm_linear_bwd_dx": (c_int, [_P, _P, c_int64, _P, _P, c_int64, _P, c_int64, c_int, c_int,
                                 c_int, c_int]),
    "gm_linear_bwd_dw": (c_int, [_P, _P, c_int64, _P, c_int64, Slot, _P, _P, c_int, c_int, c_int,
                                 c_int]),
    "gm_gan_loss": (c_int, [_P, c_int, c_int, _P, _P, c_int, c_int, POINTER(c_float), c_int,
                            c_float, _P, Slot, _P, _P, _P, _P]),
    "gm_gan_loss_phase": (c_int, [_P, c_int, c_int, _P, _P, c_int, c_int, POINTER(c_float), c_int,
                                  c_float, _P, Slot, _P, _P, _P, _P, c_int, _P, c_float]),
    "gm_l1_rows_dp": (c_int, [_P, _P, c_int64, _P, c_int64, c_int, c_int, c_int, c_int, _P, _P, c_int64, _P]),
    "gm_began_dloss_dp": (c_int, [_P, _P, c_int, c_int, _P, _P, Slot]),
    "gm_std_sums": (c_int, [_P, _P, c_int64, c_int, c_int, _P, _P]),
    "gm_std_from_sums": (c_int, [_P, _P, c_int64, _P]),
    "gm_adam": (c_int, [_P, _P, _P, _P, _P, c_int64, _P, Slot, ctypes.c_double, ctypes.c_double,
                        ctypes.c_double, ctypes.c_double, c_float]),
    "gm_interp": (c_int, [_P, _P, Slot, _P, c_int64, _P, c_int64, _P, c_int64, c_int, c_int]),
    "gm_gp_u": (c_int, [_P, _P, _P, c_int64, _P, _P, c_int64, c_int, c_int]),
    "gm_gp_norm": (c_int, [_P, _P, c_int64, _P, c_int64, _P, c_float, c_float, c_float, c_int,
                           c_int]),
    "gm_gp_dw2": (c_int, [_P, _P, _P, c_int64, _P, c_int64, _P, c_int, c_int]),
    "gm_bir_reparam": (c_int, [_P, _P, c_int64, _P, Slot, _P, c_int64, c_int, c_int]),
    "gm_bir_mmd": (c_int, [_P, _P, c_int64, _P, Slot, _P, _P, c_int64, c_int, c_int, c_float]),
    "gm_vae_reparam": (c_int, [_P, _P, c_int64, _P, Slot, _P, c_int64, _P, Slot, c_int, c_int]),
    "gm_vae_reparam_bwd": (c_int, [_P, _P, c_int64, _P, Slot, _P, c_int64, _P, c_int64, c_int,
                                   c_int]),
    "gm_sqerr_sigmoid_bwd": (c_int, [_P, _P, c_int64, _P, c_int64, _P, c_int64, _P, c_int, c_int]),
    "gm_sum_finalize": (c_int, [_P, _P, c_int, c_float, _P, Slot]),
    "gm_sum_finalize_tick": (c_int, [_P, _P, c_int, c_float, _P, Slot, _P]),
    "gm_sum_finalize2_tick": (c_int, [_P, _P, c_int, c_float, _P, Slot, _P, c_int, c_float, _P, Slot, _P]),
    "gm_vae_reparam_wide": (c_int, [_P, _P, c_int64, _P, Slot, _P, c_int64, _P, c_int, c_int, c_int]),
    "gm_vae_bwd_mid": (c_int, [_P, _P, c_int64, _P, _P, c_int64, _P, Slot, _P, c_int64, _P, _P, c_int64, _P, c_int64,
                               c_int, c_int, c_int]),
    "gm_vae_reparam_fwd": (c_int, [_P, _P, c_int64, _P, Slot, _P, c_int64, _P, c_int, c_int, c_int, _P, _P, _P,
                                   c_int64, c_int, c_int]),
    "gm_linear_bwd_dw_adam": (c_int, [_P, _P, c_int64, _P, c_int64, Slot, _P, _P, c_int, c_int, c_int,
                                      _P, _P, _P, _P, _P, _P, _P, Slot, ctypes.c_double,
                                      ctypes.c_double, ctypes.c_double, ctypes.c_double, c_float]),
    "gm_linear_fwd_interp": (c_int, [_P, _P, c_int64, Slot, _P, _P, _P, c_int64, c_int, c_int, c_int, c_int,
                                     _P, Slot, _P, c_int64, _P, c_int64, c_int]),
    "gm_linear_fwd_gather": (c_int, [_P, _P, c_int64, Slot, _P, _P, _P, c_int64, c_int, c_int, c_int,
                                     c_int, _P, c_int64, _P, Slot, _P, c_int64, c_int, c_int]),
    "gm_linear_fwd_gather_bits": (c_int, [_P, _P, c_int64, Slot, _P, _P, _P, c_int64, c_int, c_int, c_int,
                                          c_int, _P, c_int, c_int64, _P, Slot, _P, c_int64, c_int, c_int]),
    "gm_gather_rows_bits": (c_int, [_P, _P, c_int, c_int64, _P, Slot, _P, c_int64, c_int, c_int]),
    "gm_linear_bwd_dx_head": (c_int, [_P, _P, c_int64, _P, _P, c_int64, _P, c_int64, c_int, c_int,
                                      c_int, c_int, POINTER(HeadBwdArgs)]),
    "gm_linear_bwd_dx_head_fold": (c_int, [_P, _P, c_int64, _P, _P, c_int64, _P, c_int64, c_int, c_int,
                                           c_int, c_int, POINTER(HeadBwdArgs), POINTER(HeadFoldArgs)]),
    "gm_linear_fwd_sqerr": (c_int, [_P, _P, c_int64, _P, _P, _P, c_int64, c_int, c_int, c_int, _P, c_int64,
                                    _P, c_int64, _P, c_int64]),
    "gm_linear_bwd_dx_reparam": (c_int, [_P, _P, c_int64, _P, _P, c_int64, c_int, c_int, c_int, _P, c_int64,
                                         _P, Slot, _P, c_int64]),
    "gm_linear_fwd_headpart": (c_int, [_P, _P, c_int64, Slot, _P, _P, _P, c_int64, c_int, c_int, c_int, c_int,
                                       _P, _P, _P, c_int64, _P]),
    "gm_linear_bwd_dw_adam_head_fold": (c_int, [_P, _P, c_int64, _P, c_int64, Slot, _P, _P, c_int, c_int,
                                                c_int, _P, _P, _P, _P, _P, _P, _P, Slot, ctypes.c_double,
                                                ctypes.c_double, ctypes.c_double, ctypes.c_double, c_float,
                                                POINTER(HeadBwdArgs), POINTER(HeadFoldArgs)]),
    "gm_gather_rows_bits_packed": (c_int, [_P, _P, c_int, c_int64, _P, Slot, _P, c_int]),
    "gm_linear_fwd_gather_bits_packed": (c_int, [_P, _P, c_int64, Slot, _P, _P, _P, c_int64, c_int, c_int, c_int,
                                                 c_int, _P, c_int, c_int64, _P, Slot, _P, c_int]),
    "gm_linear_fwd_headpart_bits": (c_int, [_P, _P, c_int64, Slot, _P, _P, _P, c_int64, c_int, c_int, c_int, c_int,
                                            _P, _P, _P, c_int64, _P, _P, c_int, c_int]),
    "gm_linear_bwd_dw_adam_head_fold_bits": (c_int, [_P, _P, c_int64, _P, c_int64, Slot, _P, _P, c_int, c_int,
                                                     c_int, _P, _P, _P, _P, _P, _P, _P, Slot, ctypes.c_double,
                                                     ctypes.c_double, ctypes.c_double, ctypes.c_double, c_float,
                                                     POINTER(HeadBwdArgs), POINTER(HeadFoldArgs), _P, c_int, c_int]),
    "gm_linear_bwd_dw_adam_pair": (c_int, [_P, POINTER(DwAdamArgs), POINTER(DwAdamArgs)]),
    "gm_linear_bwd_dw_adam_pair_finalize": (c_int, [_P, POINTER(DwAdamArgs), POINTER(DwAdamArgs),
                                                    POINTER(Finalize2Args)]),
    "gm_linear_bwd_dw_adam_head": (c_int, [_P, _P, c_int64, _P, c_int64, Slot, _P, _P, c_int, c_int,
                                           c_int, _P, _P, _P, _P, _P, _P, _P, Slot, ctypes.c_double,
                                           ctypes.c_double, ctypes.c_double, ctypes.c_double, c_float,
                                           POINTER(HeadBwdArgs)]),
    "gm_linear_bwd_dw_adam_head_ex": (c_int, [_P, _P, c_int64, _P, c_int64, Slot, _P, _P, c_int, c_int,
                                              c_int, _P, _P, _P, _P, _P, _P, _P, Slot, ctypes.c_double,
                                              ctypes.c_double, ctypes.c_double, ctypes.c_double, c_float,
                                              POINTER(HeadBwdArgs), c_int]),
    "gm_gp_dw2_store": (c_int, [_P, _P, _P, c_int64, _P, c_int64, _P, c_int, c_int]),
    "gm_head_gp": (c_int, [_P, _P, c_int64, _P, _P, _P, _P, c_int64, c_int, c_int]),
    "gm_head_bwd_fused": (c_int, [_P, _P, c_int64, _P, _P, _P, _P, _P, c_int64, _P, _P, _P, Slot,
                                  c_float, c_int, c_int, c_int, c_int, _P, _P, _P, _P, _P, Slot,
                                  ctypes.c_double, ctypes.c_double, ctypes.c_double,
                                  ctypes.c_double, c_float, _P]),
    "gm_head_fwd_loss": (c_int, [_P, c_int, c_int, _P, c_int64, _P, _P, c_int, c_int, c_int,
                                 POINTER(c_float), c_int, c_float, _P, _P, _P, _P, _P, c_int64]),
    "gm_head_fwd_loss_final": (c_int, [_P, c_int, c_int, _P, c_int64, _P, _P, c_int, c_int, c_int,
                                       POINTER(c_float), c_int, c_float, _P, _P, _P, _P, _P, c_int64,
                                       _P, Slot, _P, _P]),
    "gm_head_bwd": (c_int, [_P, _P, c_int64, _P, _P, _P, _P, c_int64, _P, _P, _P, Slot, c_float,
                            c_int, c_int, c_int]),
    "gm_clock_probe": (c_int, [_P, c_int, _P, _P]),
    "gm_stream_create": (c_int, [POINTER(c_void_p)]),
    "gm_stream_destroy": (c_int, [_P]),
    "gm_stream_wait_event": (c_int, [_P, _P]),
    "gm_l1_rows": (c_int, [_P, _P, c_int64, _P, c_int64, c_int, c_int, c_int, _P, _P, c_int64, _P]),
    "gm_began_dloss": (c_int, [_P, _P, c_int, _P, _P, Slot]),
    "gm_began_update": (c_int, [_P, _P, _P, _P, c_float, c_float, c_int64, _P]),
    "gm_adam_scaled": (c_int, [_P, _P, _P, _P, _P, c_int64, _P, Slot, ctypes.c_double, ctypes.c_double,
                               ctypes.c_double, ctypes.c_double, c_float, _P]),
    "gm_linear_bwd_dx_add": (c_int, [_P, _P, c_int64, _P, _P, c_int64, _P, c_int64, c_int, c_int,
                                     c_int, c_int, _P, c_int64, c_float]),
    "gm_std_all": (c_int, [_P, _P, c_int64, c_int, c_int, _P, _P]),
    "gm_dragan_xhat": (c_int, [_P, _P, c_int64, _P, Slot, _P, Slot, _P, c_float, _P, c_int64, c_int,
                               c_int]),
    "gm_dragan_rows": (c_int, [_P, _P, _P, c_int64, _P, c_int64, _P, _P, c_float, c_float, c_float,
                               c_int, c_int]),
    "gm_dragan_head_bwd": (c_int, [_P, _P, c_int64, _P, c_int64, _P, _P, _P, _P, _P, c_int64, c_int,
                                   c_int]),
    "gm_fisher_commit": (c_int, [_P, _P]),
    "gm_dragan_head_bwd_store": (c_int, [_P, _P, c_int64, _P, c_int64, _P, _P, _P, _P, _P, c_int64, c_int,
                                   c_int]),
    "gm_info_q_loss": (c_int, [_P, _P, c_int64, _P, Slot, c_int64, c_int, c_int, c_int, c_int, c_float,
                               _P, c_int64, _P, Slot]),
    "gm_info_q_loss_dp": (c_int, [_P, _P, c_int64, _P, Slot, c_int64, c_int, c_int, c_int, c_int, c_int, c_float,
                                  _P, c_int64, _P, Slot]),
    "gm_act_bwd": (c_int, [_P, _P, _P, _P, c_int64, c_int]),
    "gm_randperm_prefix": (c_int, [ctypes.c_uint64, c_int64, c_int, _P]),
    "gm_mt19937_skip": (c_int, [_P, c_int64, ctypes.c_uint64]),
    "gm_comm_create": (c_int, [c_int, c_int, c_int64, POINTER(c_void_p), _P]),
    "gm_comm_connect": (c_int, [_P, _P]),
    "gm_comm_destroy": (c_int, [_P]),
    "gm_comm_error": (c_int, [_P, POINTER(c_int)]),
    "gm_comm_info": (c_int, [_P, POINTER(c_int)]),
    "gm_rccl_available": (c_int, []),
    "gm_rccl_unique_id": (c_int, [_P]),
    "gm_rccl_comm_create": (c_int, [c_int, c_int, _P, POINTER(c_void_p)]),
    "gm_rccl_allreduce_f32": (c_int, [_P, _P, _P, c_int64]),
    "gm_rccl_comm_destroy": (c_int, [_P]),
    "gm_comm_set_exchange": (c_int, [_P, c_int]),
    "gm_comm_set_max_blocks": (c_int, [_P, c_int]),
    "gm_comm_set_wait_seconds": (c_int, [_P, ctypes.c_double]),
    "gm_comm_buffer": (c_int, [_P, POINTER(c_void_p), POINTER(c_int64)]),
    "gm_allreduce_f32": (c_int, [_P, _P, _P, c_int64]),
    "gm_allreduce_adam_f32": (c_int, [_P, _P, _P, c_int64, _P, _P, _P, _P, Slot, ctypes.c_double,
                                      ctypes.c_double, ctypes.c_double, ctypes.c_double, c_float, _P]),
    "gm_allreduce_scalars": (c_int, [_P, _P, _P, c_int]),
    "gm_stage_in": (c_int, [_P, POINTER(StageSeg), c_int, Slot, c_int]),
    "gm_stage_in_gated": (c_int, [_P, POINTER(StageSeg), c_int, Slot, c_int, _P, Slot, ctypes.c_double, _P,
                                  c_int]),
    "gm_host_device_ptr": (c_int, [_P, POINTER(c_void_p)]),
    "gm_host_replay": (c_int, [_P, c_int64, POINTER(DrawOp), c_int, c_int]),
    "gm_host_replay_threads": (c_int, [c_int]),
    "gm_numpy_legacy_normal_f32": (c_int, [_P, _P, _P, _P, ctypes.c_double, ctypes.c_double, c_int64, _P, c_int]),
    "gm_fill_submit": (c_int64, [_P, c_int64, POINTER(DrawOp), c_int, c_int, _P, c_int64]),
    "gm_fill_wait": (c_int, [c_int64]),
    "gm_fill_completed": (c_int64, []),
    "gm_fill_reset": (c_int, []),
    "gm_host_replay_flavour": (c_int, [c_int]),
    "gm_graph_begin": (c_int, [_P]),
    "gm_graph_end": (c_int, [_P, POINTER(c_void_p)]),
    "gm_graph_launch": (c_int, [_P, _P]),
    "gm_graph_destroy": (c_int, [_P]),
    "gm_event_create": (c_int, [POINTER(c_void_p)]),
    "gm_event_record": (c_int, [_P, _P]),
    "gm_event_sync": (c_int, [_P]),
    "gm_event_elapsed_ms": (c_int, [_P, _P, POINTER(c_float)]),
    "gm_event_destroy": (c_int, [_P]),
}

_lib = None


class GMError(RuntimeError):
    pass


def load():
    """Load libgm_hip.so; raise loudly if it is not there (build with __graft_entry__.build())."""
    global _lib
    if _lib is not None:
        return _lib
    if not os.path.isfile(LIB_PATH):
        raise GMError("HIP extension %s is missing -- run `python __graft_entry__.py build` "
                      "(hipcc --offload-arch=gfx950); there is no CPU fallback." % LIB_PATH)
    lib = ctypes.CDLL(LIB_PATH)
    for name, (res, args) in _SIGNATURES.items():
        fn = getattr(lib, name)          # AttributeError if a declared symbol is not exported
        fn.restype = res
        fn.argtypes = args
    _lib = lib
    return lib


def check(rc, what):
    if rc != 0:
        msg = load().gm_last_error()
        raise GMError("%s failed (rc=%d): %s" % (what, rc, msg.decode() if msg else "?"))


def call(name, *args):
    check(getattr(load(), name)(*args), name)


def declared_symbols():
    """Every function declared in include/gm_hip.h (parsed), for the export test."""
    import re
    hdr = os.path.join(os.path.dirname(HERE), "include", "gm_hip.h")
    text = open(hdr).read()
    text = re.sub(r"/\*.*?\*/", "", text, flags=re.S)
    return sorted(set(re.findall(r"\b(gm_[a-z0-9_]+)\s*\(", text)))
