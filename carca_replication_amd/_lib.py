"""Build and load libcarca_hip.so (the C ABI declared in include/carca_hip.h) with ctypes.

The library is built IN-TREE (``carca_replication_amd/libcarca_hip.so``) by ``build()`` with
``hipcc --offload-arch=gfx950``; the built file travels to the GPU box with the repo snapshot.
There is no CPU fallback: if the library cannot be built or loaded, every op raises.
"""
from __future__ import annotations

import ctypes as C
import hashlib
import os
import shutil
import subprocess
import threading

_HERE = os.path.dirname(os.path.abspath(__file__))
_CSRC = os.path.join(_HERE, "csrc")
_INCLUDE = os.path.join(os.path.dirname(_HERE), "include")
LIB_PATH = os.path.join(_HERE, "libcarca_hip.so")
_STAMP = LIB_PATH + ".srchash"
SOURCES = ["api.hip", "gemm.hip", "gemm_split.hip", "gemm_stream.hip", "wgrad_cu.hip", "decoders.hip", "batch_build.hip", "embed.hip", "sa_block.hip", "sa_eval.hip", "cross_score.hip", "cross_stream.hip", "loss_metrics.hip", "backward.hip", "block_bwd.hip", "row_chain.hip", "optim.hip"]
HEADERS = ["carca_common.h", "attn_common.h", "gemm_epilogue.h", "cross_fold.h"]

MAX_SEGS = 4
MAX_GROUPS = 3
MAX_L = 64

_lock = threading.Lock()
_lib = None


class CarcaHipError(RuntimeError):
    pass


def _source_hash() -> str:
    h = hashlib.sha256()
    for name in SOURCES + HEADERS:
        p = os.path.join(_CSRC, name)
        if os.path.exists(p):
            with open(p, "rb") as f:
                h.update(name.encode() + b"\0" + f.read())
    with open(os.path.join(_INCLUDE, "carca_hip.h"), "rb") as f:
        h.update(f.read())
    return h.hexdigest()


def _hipcc() -> str | None:
    for cand in (shutil.which("hipcc"), "/opt/rocm/bin/hipcc"):
        if cand and os.path.exists(cand):
            return cand
    return None


def is_built() -> bool:
    if not os.path.exists(LIB_PATH) or not os.path.exists(_STAMP):
        return False
    with open(_STAMP) as f:
        return f.read().strip() == _source_hash()


def build(force: bool = False, verbose: bool = False) -> str:
    """Compile every HIP source for gfx950 into one shared library (cross-compiles without a GPU)."""
    import fcntl

    # one builder at a time ACROSS processes too (torchrun ranks, pytest workers on a fresh checkout): the objects go to
    # fixed paths under build/, and a rank that links another rank's half-written object gets a corrupt library
    with _lock, open(os.path.join(_HERE, ".build.lock"), "w") as lockf:
        fcntl.flock(lockf, fcntl.LOCK_EX)
        if not force and is_built():  # (re-checked under the lock: another process may just have finished)
            return LIB_PATH
        hipcc = _hipcc()
        if hipcc is None:
            raise CarcaHipError("hipcc not found: cannot build libcarca_hip.so (no CPU fallback exists)")
        srcs = [os.path.join(_CSRC, s) for s in SOURCES if os.path.exists(os.path.join(_CSRC, s))]
        objs = []
        procs = []
        os.makedirs(os.path.join(_HERE, "build"), exist_ok=True)
        for s in srcs:
            o = os.path.join(_HERE, "build", os.path.basename(s) + ".o")
            objs.append(o)
            cmd = [hipcc, "-O3", "--offload-arch=gfx950", "-fPIC", "-std=c++17", "-c", s, "-o", o]
            procs.append((cmd, subprocess.Popen(cmd, stdout=subprocess.PIPE, stderr=subprocess.STDOUT, text=True)))
        for cmd, p in procs:
            out, _ = p.communicate()
            if p.returncode != 0:
                raise CarcaHipError("hipcc failed: " + " ".join(cmd) + "\n" + out)
            if verbose and out.strip():
                print(out)
        tmp = LIB_PATH + ".tmp"
        cmd = [hipcc, "--offload-arch=gfx950", "-shared", "-fPIC", "-o", tmp] + objs
        r = subprocess.run(cmd, stdout=subprocess.PIPE, stderr=subprocess.STDOUT, text=True)
        if r.returncode != 0:
            raise CarcaHipError("link failed: " + " ".join(cmd) + "\n" + r.stdout)
        os.replace(tmp, LIB_PATH)
        with open(_STAMP, "w") as f:
            f.write(_source_hash())
        global _lib
        _lib = None
        return LIB_PATH


# ---- ctypes mirrors of the structs in include/carca_hip.h ------------------------------------------
_fp = C.c_void_p  # device pointers travel as integers


class PackDesc(C.Structure):
    _fields_ = [("src", _fp), ("dst", _fp), ("rows", C.c_int32), ("cols", C.c_int32), ("src_ld", C.c_int32),
                ("dst_rows", C.c_int32), ("dst_cols", C.c_int32), ("row_dh", C.c_int32), ("row_dhp", C.c_int32),
                ("col_dh", C.c_int32), ("col_dhp", C.c_int32), ("transposed", C.c_int32), ("frag16", C.c_int32),
                ("fold_vec", _fp), ("fold_H", C.c_int32)]


class RowSeg(C.Structure):
    _fields_ = [("ids", _fp), ("attrs", _fp), ("ctx", _fp), ("e_out", _fp), ("rows", C.c_int32), ("T", C.c_int32),
                ("add_pos", C.c_int32), ("attrs_bstride", C.c_int64), ("ctx_bstride", C.c_int64),
                ("attrs_table", _fp), ("attrs_table_rows", C.c_int32)]


class GemmSeg(C.Structure):
    _fields_ = [("a0", _fp), ("a1", _fp), ("c", _fp), ("ids", _fp), ("add", _fp), ("gate", _fp), ("rowscale", _fp),
                ("rows", C.c_int32), ("T", C.c_int32), ("add_pos", C.c_int32), ("a0_bstride", C.c_int64),
                ("a1_bstride", C.c_int64), ("a0_gather", C.c_int32)]


class GemmDesc(C.Structure):
    _fields_ = [("seg", GemmSeg * MAX_SEGS), ("nseg", C.c_int32), ("lda0", C.c_int32), ("lda1", C.c_int32),
                ("K0", C.c_int32), ("K1", C.c_int32), ("bt0", _fp), ("bt1", _fp), ("ldb0", C.c_int32),
                ("ldb1", C.c_int32), ("N", C.c_int32), ("ldc", C.c_int32), ("ncols_out", C.c_int32), ("bias", _fp),
                ("pos", _fp), ("colvec", _fp), ("ld_add", C.c_int32), ("ld_gate", C.c_int32),
                ("gate_slope", C.c_float), ("mask_rows", C.c_int32), ("alpha", C.c_float), ("gate_scale", C.c_float),
                ("gate_zero_drops", C.c_int32), ("add_table", _fp), ("ld_add_table", C.c_int32)]


class WgradSeg(C.Structure):
    _fields_ = [("dy", _fp), ("x", _fp), ("x1", _fp), ("ids", _fp), ("rows", C.c_int32), ("T", C.c_int32),
                ("x_bstride", C.c_int64), ("x1_bstride", C.c_int64), ("x_gather", C.c_int32)]


class WgradDesc(C.Structure):
    _fields_ = [("seg", WgradSeg * MAX_SEGS), ("nseg", C.c_int32), ("ld_dy", C.c_int32), ("ld_x", C.c_int32),
                ("ld_x1", C.c_int32), ("N", C.c_int32), ("K", C.c_int32), ("K1", C.c_int32), ("dw", _fp),
                ("ldw", C.c_int32), ("db", _fp), ("mask_rows", C.c_int32)]


class Dropout(C.Structure):
    _fields_ = [("p", C.c_float), ("seed", C.c_uint64), ("site", C.c_uint32), ("seed_offset", C.c_void_p)]


class SaSave(C.Structure):
    _fields_ = [(n, _fp) for n in ("qn", "qh", "kh", "vh", "r", "s2", "h1", "m_attn", "m_ffn1", "m_ffn2")]


class CaSave(C.Structure):
    _fields_ = [("kh", _fp), ("vh", _fp), ("qh", _fp * MAX_GROUPS), ("m_attn", _fp * MAX_GROUPS)]


class CrossBwdGroup(C.Structure):
    _fields_ = [("qh", _fp), ("y", _fp), ("dy", _fp), ("ids", _fp), ("dqh", _fp), ("dlogit", _fp), ("m_attn", _fp),
                ("N", C.c_int32), ("ld_y", C.c_int32)]


class AdamTensor(C.Structure):
    _fields_ = [("p", _fp), ("g", _fp), ("m", _fp), ("v", _fp), ("n", C.c_int64), ("row_mask", _fp), ("row_len", C.c_int64)]


class SaBwdDesc(C.Structure):
    _fields_ = ([(n, C.c_int32) for n in ("B", "L", "d", "H", "residual")] + [("drop_p", C.c_float), ("ids", _fp), ("dy", _fp)]
                + [(n, _fp) for n in ("x_in", "qn", "qh", "kh", "vh", "r", "s2", "h1", "m_attn", "m_ffn2", "wq_t", "wk_t",
                                      "wv_t", "w1_t", "w2_t", "ln1_w", "ln2_w", "g_w1", "g_b1", "g_w2", "g_b2", "g_wq",
                                      "g_wk", "g_wv", "g_bq", "g_bk", "g_bv", "g_ln1_w", "g_ln1_b", "g_ln2_w", "g_ln2_b",
                                      "workspace", "dx")])


class CrossBwdIn(C.Structure):
    _fields_ = [(n, _fp) for n in ("qh", "y", "dy", "ids", "o", "m_attn", "de")] + [("N", C.c_int32), ("ld_y", C.c_int32)]


class CrossBwdDesc(C.Structure):
    _fields_ = ([(n, C.c_int32) for n in ("B", "L", "d", "H", "ngroups", "residual", "training")] + [("drop_p", C.c_float),
                ("group", CrossBwdIn * MAX_GROUPS)]
                + [(n, _fp) for n in ("p_ids", "kh", "vh", "p_normed", "enc_out", "wq_t", "wk_t", "wv_t", "ffn_w_pad", "ffn_w",
                                      "norm_w", "g_ffn_w", "g_ffn_b", "g_ffn_w_pad", "g_wq", "g_wk", "g_wv", "g_bq", "g_bk",
                                      "g_bv", "g_norm_w", "g_norm_b", "workspace", "dx")])


class EmbedBwdSeg(C.Structure):
    _fields_ = [(n, _fp) for n in ("de", "ids", "attrs", "ctx", "attrs_table")] + [
        ("attrs_bstride", C.c_int64), ("ctx_bstride", C.c_int64), ("rows", C.c_int32), ("T", C.c_int32),
        ("attrs_table_rows", C.c_int32), ("joint_only", C.c_int32)]


class EmbedBwdDesc(C.Structure):
    _fields_ = ([("seg", EmbedBwdSeg * MAX_SEGS)] + [(n, C.c_int32) for n in ("nseg", "d", "g", "n_attrs", "n_ctx", "ld_de", "L")]
                + [("zq", _fp), ("joint_wt", _fp), ("ld_joint_wt", C.c_int32)]
                + [(n, _fp) for n in ("g_items", "g_feats_w", "g_feats_b", "g_joint_w", "g_joint_b", "g_pos", "workspace",
                                      "ev_early")] + [("skip_joint", C.c_int32), ("only_joint", C.c_int32), ("table_stream", _fp)])


class SaWeights(C.Structure):
    _fields_ = [(n, _fp) for n in ("ln1_w", "ln1_b", "ln2_w", "ln2_b", "wq", "wk", "wv", "bq", "bk", "bv", "w1", "w2",
                                   "b1", "b2")]


class CaWeights(C.Structure):
    _fields_ = [(n, _fp) for n in ("ln_w", "ln_b", "wq", "wk", "wv", "bq", "bk", "bv", "ffn_w_pad", "ffn_w", "ffn_b", "wu", "cu")]


class TargetGroup(C.Structure):
    _fields_ = [("o", _fp), ("ids", _fp), ("y", _fp), ("N", C.c_int32), ("ldy", C.c_int32)]


MAX_BLOCKS = 8


class ForwardDesc(C.Structure):
    _fields_ = [("segs", RowSeg * MAX_SEGS), ("ngroups", C.c_int32), ("B", C.c_int32), ("L", C.c_int32),
                ("d", C.c_int32), ("g", C.c_int32), ("H", C.c_int32), ("n_attrs", C.c_int32), ("n_ctx", C.c_int32),
                ("n_blocks", C.c_int32), ("ld_e", C.c_int32), ("items_w", _fp), ("feats_w", _fp), ("feats_b", _fp),
                ("joint_w", _fp), ("joint_b", _fp), ("pos", _fp), ("zq", _fp), ("x_work", _fp * 2),
                ("sa", SaWeights * MAX_BLOCKS), ("sa_residual", C.c_int32 * MAX_BLOCKS), ("ca", CaWeights),
                ("ca_residual", C.c_int32), ("training", C.c_int32), ("y", _fp * MAX_GROUPS),
                ("N", C.c_int32 * MAX_GROUPS), ("ldy", C.c_int32), ("p_normed", _fp), ("fold_wc", _fp), ("fold_bias", _fp),
                ("fold_ldwc", C.c_int32), ("x_out", _fp * MAX_BLOCKS), ("sa_save", SaSave * MAX_BLOCKS), ("ca_save", CaSave),
                ("save_blocks", C.c_int32), ("save_cross", C.c_int32), ("p_embed", C.c_float), ("p_block", C.c_float),
                ("p_cross", C.c_float), ("seed", C.c_uint64), ("m_embed", _fp), ("seed_offset", C.c_void_p),
                ("n_events", C.c_int32), ("z_table", _fp), ("ld_z_table", C.c_int32)]


# name -> (restype, argtypes); every symbol include/carca_hip.h declares
_i, _f = C.c_int, C.c_float
SIGNATURES = {
    "carca_abi_version": (_i, []),
    "carca_set_tuning": (_i, [_i, _i]),
    "carca_det_begin": (_i, [_fp, C.c_longlong, _fp, _fp]),
    "carca_det_flush": (_i, [_fp, _fp, C.c_longlong, C.c_longlong, _fp]),
    "carca_set_debug_buffer": (_i, [_fp]),
    "carca_last_error": (C.c_char_p, []),
    "carca_poll_errors": (_i, []),
    "carca_capture_scope": (_i, [_fp, C.POINTER(C.c_ulonglong)]),
    "carca_capture_bytes": (C.c_longlong, [C.c_ulonglong]),
    "carca_capture_release": (_i, [C.c_ulonglong]),
    "carca_release_stream_scratch": (_i, [_fp]),
    "carca_padded_dims": (_i, [_i, _i, C.POINTER(_i), C.POINTER(_i), C.POINTER(_i)]),
    "carca_pack_weights": (_i, [C.POINTER(PackDesc), _i, _fp]),
    "carca_embed_fwd": (_i, [C.POINTER(RowSeg), _i, _i, _i, _i, _i, _fp, _fp, _fp, _fp, _fp, _fp, _fp, _i, _i, _fp]),
    "carca_gemm_rows": (_i, [C.POINTER(GemmDesc), _fp]),
    "carca_split_bytes": (C.c_longlong, [_i, _i, _i]),
    "carca_split_pack": (_i, [_fp, _i, _i, _i, _i, _fp, _fp]),
    "carca_split_bind": (_i, [_fp, _fp, _i, _i, _i]),
    "carca_split_launch_count": (C.c_longlong, []),
    "carca_gemm_rows_group": (_i, [C.POINTER(GemmDesc), _i, _fp]),
    "carca_gemm_wgrad": (_i, [C.POINTER(WgradDesc), _fp]),
    "carca_gemm_wgrad_group": (_i, [C.POINTER(WgradDesc), _i, _fp]),
    "carca_sa_block_fwd": (_i, [_fp, _i, _fp, _fp, _i, _i, _i, _i, _i, C.POINTER(SaWeights), _i, C.POINTER(SaSave),
                                C.POINTER(Dropout), _fp]),
    "carca_sa_block_eval": (_i, [_fp, _i, _fp, _fp, _i, _i, _i, _i, _i, C.POINTER(SaWeights), _i, _i, _fp]),
    "carca_cross_score_fwd": (_i, [_fp, _i, _fp, _fp, C.POINTER(TargetGroup), _i, _i, _i, _i, _i, _i,
                                   C.POINTER(CaWeights), _i, _i, C.POINTER(CaSave), C.POINTER(Dropout), _fp]),
    "carca_dropout_fwd": (_i, [_fp, _i, _i, _i, C.POINTER(Dropout), _fp, _fp]),
    "carca_mask_mul": (_i, [_fp, _i, _fp, _i, _f, _fp, _i, _i, _i, _i, _fp]),
    "carca_layernorm_bwd": (_i, [_fp, _i, _fp, _i, _fp, _i, _i, _fp, _i, _fp, _i, _i, _fp, _fp, _fp]),
    "carca_embed_scatter": (_i, [_fp, _i, _fp, _i, _i, _f, _fp, _fp]),
    "carca_colsum": (_i, [_fp, _i, _i, _i, _fp, _fp, _i, _fp, _fp]),
    "carca_sa_attn_bwd": (_i, [_fp, _fp, _fp, _fp, _i, _fp, _fp, _fp, _fp, _i, _i, _i, _i, _fp, _f, _fp]),
    "carca_cross_attn_bwd": (_i, [_fp, _fp, _fp, C.POINTER(CrossBwdGroup), _i, _fp, _fp, _fp, _fp, _i, _i, _i, _i, _i,
                                  _f, _fp]),
    "carca_unpack_grads": (_i, [C.POINTER(PackDesc), _i, _i, _fp]),
    "carca_forward": (_i, [C.POINTER(ForwardDesc), C.POINTER(_fp), _fp]),
    "carca_event_create": (_i, [C.POINTER(_fp)]),
    "carca_event_destroy": (_i, [_fp]),
    "carca_event_elapsed_ms": (_i, [_fp, _fp, C.POINTER(C.c_float)]),
    "carca_stream_wait_event": (_i, [_fp, _fp]),
    "carca_bce_fwd": (_i, [_fp, _fp, _fp, _i, _f, _fp, _fp, _fp, _fp, _fp]),
    "carca_rank_metrics": (_i, [_fp, _i, _i, _i, _fp, _fp, _fp, _fp]),
    "carca_eval_metrics": (_i, [_fp, _fp, _fp, _i, _i, _i, _f, _fp, _fp]),
    "carca_layernorm_fwd": (_i, [_fp, _i, _fp, _i, _i, _i, _fp, _fp, _fp]),
    "carca_dot_score_fwd": (_i, [_fp, _i, _fp, _i, _fp, _i, _i, _i, _i, _i, _i, _fp]),
    "carca_dot_score_bwd": (_i, [_fp, _i, _fp, _i, _fp, _fp, _fp, _i, _fp, _i, _i, _i, _i, _i, _i, _i, _fp]),
    "carca_slot_decay_scale": (_i, [_fp, _i, _fp, _i, _i, _i, _i, _f, _fp]),
    "carca_l2norm_fwd": (_i, [_fp, _i, _fp, _i, _i, _i, _fp]),
    "carca_l2norm_bwd": (_i, [_fp, _i, _fp, _i, _fp, _i, _i, _i, _fp]),
    "carca_add_positions": (_i, [_fp, _i, _fp, _fp, _i, _i, _i, _i, _fp]),
    "carca_mha_core": (_i, [_fp, _i, _fp, _fp, _i, _fp, _fp, _i, _i, _i, _i, _i, _i, _i, _fp, _i, _fp, _fp]),
    "carca_mha_core_bwd": (_i, [_fp, _i, _fp, _fp, _i, _fp, _fp, _i, _i, _i, _i, _i, _i, _i, _fp, _i, _fp, _fp, _fp, _fp,
                                _fp]),
    "carca_gemm_rows_log": (_i, [C.c_char_p, _i]),
    "carca_mha_core_drop": (_i, [_fp, _i, _fp, _fp, _i, _fp, _fp, _i, _i, _i, _i, _i, _i, _i, _fp, _i, _fp, C.POINTER(Dropout),
                                 _fp, _fp]),
    "carca_mha_core_bwd_drop": (_i, [_fp, _i, _fp, _fp, _i, _fp, _fp, _i, _i, _i, _i, _i, _i, _i, _fp, _i, _fp, _fp, _fp, _fp,
                                     _fp, C.c_float, _fp]),
    "carca_knn_score": (_i, [_fp, C.c_int64, _fp, C.c_int64, _fp, _fp, _i, _fp, _i, _i, _i, _i, _fp]),
    "carca_sa_block_bwd_workspace": (C.c_size_t, [_i, _i, _i, _i]),
    "carca_sa_block_bwd": (_i, [C.POINTER(SaBwdDesc), C.POINTER(WgradDesc), C.POINTER(_i), _fp]),
    "carca_cross_score_bwd_workspace": (C.c_size_t, [_i, _i, _i, _i, C.POINTER(C.c_int32), _i]),
    "carca_cross_score_bwd": (_i, [C.POINTER(CrossBwdDesc), C.POINTER(WgradDesc), C.POINTER(_i), _fp]),
    "carca_embed_bwd_workspace": (C.c_size_t, [C.POINTER(C.c_int32), _i, _i, _i]),
    "carca_embed_bwd": (_i, [C.POINTER(EmbedBwdDesc), _fp]),
    "carca_early_event_recorded": (_i, []),
    "carca_adam_step": (_i, [C.POINTER(AdamTensor), _i, C.c_double, C.c_double, C.c_double, C.c_double, C.c_double, _i,
                        _fp]),
    "carca_mark_rows": (_i, [_fp, C.c_int64, _fp, C.c_int64, _fp]),
    "carca_zero_rows": (_i, [_fp, C.c_int64, _i, C.POINTER(_fp), C.POINTER(C.c_int64), _i, _fp]),
    "carca_concat_ids": (_i, [C.POINTER(_fp), C.POINTER(C.c_int64), _i, _fp, _fp]),
    "carca_build_eval_batch": (_i, [_fp, _fp, _fp, _fp, _i, _i, _i, _i, _i, _i, _i, _i, C.c_uint64, _fp, _fp, _fp, _fp, _fp,
                                    _fp]),
    "carca_build_train_batch": (_i, [_fp, _fp, _fp, _fp, _i, _i, _i, _i, _i, _i, _i, C.c_uint64, _fp, _fp, _fp, _fp, _fp,
                                     _fp]),
}


def declared_symbols() -> list[str]:
    """Function names declared in include/carca_hip.h (parsed from the header text)."""
    import re

    with open(os.path.join(_INCLUDE, "carca_hip.h")) as f:
        text = f.read()
    text = re.sub(r"/\*.*?\*/", "", text, flags=re.S)
    return sorted(set(re.findall(r"\b(carca_[a-z0-9_]+)\s*\(", text)))


def load():
    """Load the library (building it first if the sources changed) and type its entry points."""
    global _lib
    if _lib is not None:
        return _lib
    # torch must load ITS libamdhip64.so.7 first: the library below has the same soname, and a process
    # that resolves it to /opt/rocm's copy before torch starts ends up with a runtime that sees no device
    import torch  # noqa: F401

    if not is_built():
        build()
    with _lock:
        if _lib is not None:
            return _lib
        try:
            lib = C.CDLL(LIB_PATH)
        except OSError as e:  # fail loudly: there is no other implementation to fall back to
            raise CarcaHipError(f"cannot load {LIB_PATH}: {e}") from e
        for name, (res, args) in SIGNATURES.items():
            fn = getattr(lib, name)
            fn.restype = res
            fn.argtypes = args
        if lib.carca_abi_version() != 2:
            raise CarcaHipError("libcarca_hip.so ABI version mismatch")
        _lib = lib
        return lib


def check(rc: int, what: str) -> None:
    if rc == 0:
        return
    msg = load().carca_last_error().decode(errors="replace")
    kind = {-1: "unsupported configuration", -2: "bad argument"}.get(rc, f"hipError {rc}")
    raise CarcaHipError(f"{what}: {kind}: {msg}")
