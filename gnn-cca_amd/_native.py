"""ctypes binding of include/gnncca_mpn.h (libgnncca_mpn.so).  No compute here and no fallback: if the library is
missing or a call fails, the caller gets an exception."""
import ctypes as C
import os

HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.environ.get("GNNCCA_LIB") or os.path.join(HERE, "lib", "libgnncca_mpn.so")  # GNNCCA_LIB: diagnostic builds

ABI_VERSION = 2
MAX_LAYERS = 8
OK, ERR_INVALID_ARG, ERR_UNSUPPORTED, ERR_WORKSPACE, ERR_HIP, ERR_NO_DEVICE = range(6)
AGG = {"sum": 0, "mean": 1, "max": 2}
GRAPH_UNSORTED, GRAPH_BAD_INDEX = 1, 2


class Layer(C.Structure):
    _fields_ = [("in_dim", C.c_int32), ("out_dim", C.c_int32), ("has_bn", C.c_int32), ("relu", C.c_int32)]


class Mlp(C.Structure):
    _fields_ = [("n_layers", C.c_int32), ("layers", Layer * MAX_LAYERS)]


class MpnDims(C.Structure):
    _fields_ = [("abi_version", C.c_int32), ("node_in", C.c_int32), ("edge_in", C.c_int32), ("node_dim", C.c_int32),
                ("edge_dim", C.c_int32), ("agg", C.c_int32), ("num_enc_steps", C.c_int32),
                ("num_class_steps", C.c_int32), ("reattach_nodes", C.c_int32), ("reattach_edges", C.c_int32),
                ("enc_node", Mlp), ("enc_edge", Mlp), ("edge_mlp", Mlp), ("node_mlp", Mlp), ("cls_edge", Mlp)]


class Frames(C.Structure):
    _fields_ = [("xw", C.c_void_p), ("yw", C.c_void_p), ("max_dist", C.c_void_p), ("person_id", C.c_void_p),
                ("cam", C.c_void_p), ("graph_of", C.c_void_p), ("graph_ptr", C.c_void_p), ("src_order", C.c_void_p),
                ("edge_ptr", C.c_void_p)]


class FramesIO(C.Structure):
    _fields_ = [("staged_dev", C.c_void_p), ("n_nodes", C.c_int64), ("n_frames", C.c_int64), ("n_edges", C.c_int64),
                ("node_embeds", C.c_void_p), ("reid_embeds", C.c_void_p), ("reid_dim", C.c_int32), ("mode", C.c_int32), ("normalize", C.c_int32),
                ("node_norm", C.c_void_p), ("reid_norm", C.c_void_p), ("edge_index", C.c_void_p), ("edge_attr", C.c_void_p),
                ("edge_labels", C.c_void_p), ("logits", C.c_void_p), ("probs", C.c_void_p), ("predictions", C.c_void_p), ("pruned", C.c_void_p),
                ("counters", C.c_void_p), ("labels", C.c_void_p), ("counters_len", C.c_int64)]


class PostBatch(C.Structure):
    """gnncca_post_batch (include/gnncca_mpn.h): one batch's HOST copies for the asynchronous finalizer pool."""
    _fields_ = [("src", C.c_void_p), ("dst", C.c_void_p), ("node_ptr", C.c_void_p), ("edge_ptr", C.c_void_p), ("n_frames", C.c_int32),
                ("switches", C.c_int32), ("triggers", C.c_void_p), ("probs", C.c_void_p), ("predictions", C.c_void_p), ("labels", C.c_void_p),
                ("n_clusters", C.c_void_p), ("ready_event", C.c_void_p), ("device", C.c_int32)]


class Trace(C.Structure):
    _fields_ = [("h_enc", C.c_void_p), ("e_enc", C.c_void_p), ("h_steps", C.c_void_p), ("e_steps", C.c_void_p)]


PROFILE_MAX = 64
KERNEL_KINDS = ["plan", "plan_sort_unused", "enc_gemm", "enc_reduce", "enc_tail", "step", "step_last"]


class Profile(C.Structure):
    _fields_ = [("options", C.c_uint32), ("count", C.c_int32), ("kind", C.c_int32 * PROFILE_MAX), ("ms", C.c_float * PROFILE_MAX)]


class Dropout(C.Structure):
    """gnncca_dropout (include/gnncca_mpn.h): Dropout probabilities per MLP group + a DEVICE seed word."""
    _fields_ = [("p_enc", C.c_float), ("p_edge", C.c_float), ("p_node", C.c_float), ("p_cls", C.c_float), ("seed_dev", C.c_void_p)]


OPT_EDGE_STATE_BF16 = 1
OPT_ENC_SPLIT3 = 2
OPT_ENC_UNSPLIT = 4
OPT_COLUMN_RANGES = 8
BWD_GRADS_ZEROED = 1
POST_ROUNDING, POST_PRUNING, POST_SPLITTING = 1, 2, 4        # gnncca_post_finalize_frame_host switches (config_inference.yaml:6-8)
POST_TRIGGER_ROUNDING, POST_TRIGGER_SPLITTING = 1, 2         # trigger bits of gnncca_post_prune_cluster_frames_ex


_SIGNATURES = {
    "gnncca_abi_version": (C.c_int, []),
    "gnncca_status_string": (C.c_char_p, [C.c_int]),
    "gnncca_last_hip_error": (C.c_int, []),
    "gnncca_param_count": (C.c_int, [C.POINTER(MpnDims)]),
    "gnncca_packed_weights_bytes": (C.c_size_t, [C.POINTER(MpnDims)]),
    "gnncca_pack_weights": (C.c_int, [C.POINTER(MpnDims), C.POINTER(C.c_void_p), C.c_int, C.c_void_p, C.c_size_t]),
    "gnncca_pack_program_bytes": (C.c_size_t, []),
    "gnncca_pack_program": (C.c_int, [C.POINTER(MpnDims), C.c_void_p, C.c_size_t]),
    "gnncca_pack_weights_device": (C.c_int, [C.POINTER(MpnDims), C.POINTER(C.c_void_p), C.c_int, C.c_void_p, C.c_void_p,
                                             C.c_size_t, C.c_void_p]),
    "gnncca_workspace_bytes": (C.c_size_t, [C.POINTER(MpnDims), C.c_int64, C.c_int64]),
    "gnncca_supported": (C.c_int, [C.POINTER(MpnDims)]),
    "gnncca_num_outputs": (C.c_int, [C.POINTER(MpnDims)]),
    "gnncca_mpn_forward": (C.c_int, [C.POINTER(MpnDims), C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_int64,
                                     C.c_int64, C.c_void_p, C.c_size_t, C.c_void_p, C.POINTER(Trace), C.c_void_p]),
    "gnncca_mpn_forward_ex": (C.c_int, [C.POINTER(MpnDims), C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_int64,
                                        C.c_int64, C.c_void_p, C.c_size_t, C.c_void_p, C.POINTER(Trace), C.c_uint32, C.c_void_p]),
    "gnncca_mpn_forward_profiled": (C.c_int, [C.POINTER(MpnDims), C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p,
                                              C.c_int64, C.c_int64, C.c_void_p, C.c_size_t, C.c_void_p, C.c_void_p,
                                              C.POINTER(Profile)]),
    "gnncca_normalize_columns": (C.c_int, [C.c_void_p, C.c_int64, C.c_int64, C.c_void_p, C.c_void_p, C.c_void_p]),
    "gnncca_build_edges": (C.c_int, [C.POINTER(Frames), C.c_void_p, C.c_int32, C.c_int64, C.c_int64, C.c_int32,
                                     C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p]),
    "gnncca_post_threshold": (C.c_int, [C.c_void_p, C.c_int64, C.c_void_p, C.c_void_p, C.c_void_p]),
    "gnncca_post_workspace_bytes": (C.c_size_t, [C.c_int64, C.c_int64]),
    "gnncca_post_prune_cluster": (C.c_int, [C.c_void_p, C.c_void_p, C.c_int64, C.c_int64, C.c_void_p, C.c_size_t,
                                            C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p]),
    "gnncca_post_prune_cluster_frames": (C.c_int, [C.c_void_p, C.c_void_p, C.c_int64, C.c_int64, C.c_void_p, C.c_void_p, C.c_int32,
                                                   C.c_void_p, C.c_size_t, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p,
                                                   C.c_void_p, C.c_void_p]),
    "gnncca_post_prune_cluster_frames_ex": (C.c_int, [C.c_void_p, C.c_void_p, C.c_int64, C.c_int64, C.c_void_p, C.c_void_p, C.c_int32,
                                                      C.c_void_p, C.c_size_t, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p,
                                                      C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p]),
    "gnncca_post_finalize_frame_host": (C.c_int, [C.c_void_p, C.c_void_p, C.c_int64, C.c_int64, C.c_int64, C.c_void_p, C.c_void_p, C.c_int32,
                                                  C.c_void_p, C.c_void_p, C.c_void_p]),
    "gnncca_post_finalize_frames_host": (C.c_int, [C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_int32, C.c_void_p, C.c_void_p,
                                                   C.c_int32, C.c_void_p, C.c_void_p, C.c_int32]),
    "gnncca_post_pool_create": (C.c_void_p, [C.c_int32]),
    "gnncca_post_pool_threads": (C.c_int32, [C.c_void_p]),
    "gnncca_post_pool_destroy": (None, [C.c_void_p]),
    "gnncca_post_pool_submit": (C.c_int64, [C.c_void_p, C.POINTER(PostBatch)]),
    "gnncca_post_pool_submit_copy": (C.c_int64, [C.c_void_p, C.POINTER(PostBatch), C.c_void_p, C.c_void_p, C.c_size_t, C.c_int32, C.c_void_p]),
    "gnncca_post_pool_wait": (C.c_int, [C.c_void_p, C.c_int64, C.c_void_p, C.c_void_p]),
    "gnncca_post_pool_wait_timed": (C.c_int, [C.c_void_p, C.c_int64, C.c_void_p, C.c_void_p, C.c_void_p]),
    "gnncca_backward_supported": (C.c_int, [C.POINTER(MpnDims)]),
    "gnncca_backward_workspace_bytes": (C.c_size_t, [C.POINTER(MpnDims), C.c_int64, C.c_int64]),
    "gnncca_mpn_backward": (C.c_int, [C.POINTER(MpnDims), C.POINTER(C.c_void_p), C.c_int, C.c_void_p, C.c_void_p, C.c_void_p,
                                      C.c_int64, C.c_int64, C.POINTER(Trace), C.c_void_p, C.c_void_p, C.POINTER(C.c_void_p),
                                      C.c_void_p, C.c_size_t, C.c_void_p]),
    "gnncca_mpn_backward_ex": (C.c_int, [C.POINTER(MpnDims), C.POINTER(C.c_void_p), C.c_int, C.c_void_p, C.c_void_p, C.c_void_p,
                                         C.c_int64, C.c_int64, C.POINTER(Trace), C.c_void_p, C.c_void_p, C.POINTER(C.c_void_p),
                                         C.c_void_p, C.c_size_t, C.c_uint32, C.c_void_p]),
    "gnncca_classifier_train": (C.c_int, [C.POINTER(MpnDims), C.POINTER(C.c_void_p), C.c_int, C.c_void_p, C.c_int64, C.c_void_p,
                                          C.c_void_p, C.c_void_p, C.c_void_p]),
    "gnncca_mpn_forward_train": (C.c_int, [C.POINTER(MpnDims), C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_int64,
                                           C.c_int64, C.c_void_p, C.c_size_t, C.c_void_p, C.POINTER(Trace), C.POINTER(Dropout),
                                           C.c_void_p]),
    "gnncca_classifier_train_dropout": (C.c_int, [C.POINTER(MpnDims), C.POINTER(C.c_void_p), C.c_int, C.c_void_p, C.c_int64,
                                                  C.c_void_p, C.c_void_p, C.c_void_p, C.POINTER(Dropout), C.c_void_p]),
    "gnncca_mpn_backward_train": (C.c_int, [C.POINTER(MpnDims), C.POINTER(C.c_void_p), C.c_int, C.c_void_p, C.c_void_p, C.c_void_p,
                                            C.c_int64, C.c_int64, C.POINTER(Trace), C.c_void_p, C.c_void_p, C.POINTER(C.c_void_p),
                                            C.c_void_p, C.c_size_t, C.c_uint32, C.POINTER(Dropout), C.c_void_p]),
    "gnncca_mlp_eval_workspace_bytes": (C.c_size_t, [C.POINTER(Mlp), C.c_int64]),
    "gnncca_mlp_eval": (C.c_int, [C.POINTER(Mlp), C.POINTER(C.c_void_p), C.c_int, C.c_void_p, C.c_int64, C.c_void_p, C.c_void_p,
                                  C.c_size_t, C.c_void_p]),
    "gnncca_gather_cat": (C.c_int, [C.c_void_p, C.c_void_p, C.c_int, C.c_int64, C.c_void_p, C.c_void_p, C.c_int, C.c_int64,
                                    C.c_void_p, C.c_void_p, C.c_int, C.c_int64, C.c_int64, C.c_void_p, C.c_void_p]),
    "gnncca_aggregate_workspace_bytes": (C.c_size_t, [C.c_int64, C.c_int64]),
    "gnncca_aggregate": (C.c_int, [C.c_void_p, C.c_void_p, C.c_int64, C.c_int64, C.c_int, C.c_int, C.c_void_p, C.c_void_p, C.c_size_t,
                                   C.c_void_p]),
    "gnncca_train_tape_bytes": (C.c_size_t, [C.POINTER(MpnDims), C.c_int64, C.c_int64]),
    "gnncca_train_tape_latents": (C.c_int, [C.POINTER(MpnDims), C.c_int64, C.c_int64, C.POINTER(C.c_int64), C.c_int]),
    "gnncca_train_forward": (C.c_int, [C.POINTER(MpnDims), C.POINTER(C.c_void_p), C.c_int, C.c_void_p, C.c_void_p, C.c_void_p,
                                       C.c_int64, C.c_int64, C.c_void_p, C.c_size_t, C.c_void_p, C.POINTER(Dropout), C.c_void_p]),
    "gnncca_train_backward": (C.c_int, [C.POINTER(MpnDims), C.POINTER(C.c_void_p), C.c_int, C.c_void_p, C.c_void_p, C.c_void_p,
                                        C.c_int64, C.c_int64, C.c_void_p, C.c_size_t, C.c_void_p, C.POINTER(C.c_void_p),
                                        C.POINTER(Dropout), C.c_void_p]),
    "gnncca_normalize_columns2": (C.c_int, [C.c_void_p, C.c_int64, C.c_void_p, C.c_void_p, C.c_int64, C.c_void_p, C.c_int64, C.c_void_p]),
    "gnncca_frames_forward": (C.c_int, [C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_size_t, C.c_void_p, C.c_size_t, C.c_uint32, C.c_void_p]),
    "gnncca_plan_frames_bytes": (C.c_size_t, [C.c_int64, C.c_int64]),
    "gnncca_plan_frames": (C.c_int64, [C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_int64, C.c_void_p, C.c_void_p, C.c_int64, C.c_void_p,
                                       C.c_size_t]),
    "gnncca_pad_frame": (C.c_int, [C.c_void_p, C.c_int64, C.c_void_p, C.c_void_p, C.c_int64, C.c_void_p, C.c_int64, C.c_int, C.c_void_p, C.c_void_p,
                                 C.c_int64, C.c_int, C.c_int, C.c_void_p]),
    "gnncca_read_graph_flags": (C.c_int, [C.c_void_p, C.POINTER(C.c_uint32), C.c_void_p]),
    "gnncca_read_graph_flags2": (C.c_int, [C.c_void_p, C.POINTER(C.c_uint32), C.c_void_p]),
}

_lib = None


class NativeError(RuntimeError):
    """A libgnncca_mpn call returned a non-zero status (the int status of SURVEY.md 8b, as a RuntimeError)."""


def exported_symbols():
    return sorted(_SIGNATURES)


def lib():
    """Load libgnncca_mpn.so (once).  Raises if it has not been built: there is no other compute path."""
    global _lib
    if _lib is None:
        if not os.path.exists(LIB_PATH):
            raise NativeError(f"{LIB_PATH} is missing: build it with `python gnn-cca_amd/build.py` "
                              "(the MI355X HIP path is the only implementation; there is no CPU fallback)")
        handle = C.CDLL(LIB_PATH)
        for name, (res, args) in _SIGNATURES.items():
            fn = getattr(handle, name)  # AttributeError if the .so does not export a declared symbol
            fn.restype, fn.argtypes = res, args
        if handle.gnncca_abi_version() != ABI_VERSION:
            raise NativeError("libgnncca_mpn.so ABI version mismatch; rebuild")
        _lib = handle
    return _lib


def check(status, what):
    if status != OK:
        l = lib()
        msg = l.gnncca_status_string(status).decode()
        if status == ERR_HIP:
            msg += f" (hipError_t {l.gnncca_last_hip_error()})"
        if status == ERR_UNSUPPORTED:
            raise NotImplementedError(f"{what}: {msg}")
        raise NativeError(f"{what}: {msg}")
