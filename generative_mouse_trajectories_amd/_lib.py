"""ctypes binding of libgmt_hip.so (include/gmt_hip.h).  The library is built in-tree by
`_build.build()`; there is NO fallback: if it is missing or its ABI version differs, every
entry point raises."""
from __future__ import annotations

import ctypes
import os

import torch

_PKG = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.path.join(_PKG, 'libgmt_hip.so')
ABI_VERSION = 9
_lib = None

ERRORS = {-1: 'bad argument (null pointer / shape / alignment)', -2: 'unsupported configuration',
          -3: 'HIP launch failed', -4: 'workspace too small'}

c_p = ctypes.c_void_p
c_i = ctypes.c_int
c_ll = ctypes.c_longlong
c_f = ctypes.c_float
c_d = ctypes.c_double



class GenFwdArgs(ctypes.Structure):
    _fields_ = [(n, c_p) for n in ('g0', 'w_u_pk', 'w_hh0_pk', 'w_ih1_pk', 'w_hh1_pk', 'bias1', 'w1_pk', 'b1', 'w2',
                                   'b2', 'starts', 'ends', 'init_in', 'target', 'tf_mask', 'out_seq', 'gates0',
                                   'c0_all', 'h0_all', 'gates1', 'c1_all', 'h1_all', 'u_all', 'y_all', 'oraw_all', 'h0d_all')] \
        + [('T', c_i), ('B', c_i), ('H', c_i), ('dropout_p', c_f), ('seed', ctypes.c_uint),
           ('stop_distance', c_f), ('rows_valid', c_i), ('lengths_out', c_p)]


class GenBwdArgs(ctypes.Structure):
    _fields_ = [(n, c_p) for n in ('gates0', 'c0_all', 'gates1', 'c1_all', 'y_all', 'oraw_all', 'u_all', 'd_out',
                                   'w1T_pk', 'w_l1T_pk', 'w_hh0T_pk', 'w_uP', 'w2', 'tf_mask', 'd_ypre', 'd_o',
                                   'd_in0', 'd_g0')] + [('T', c_i), ('B', c_i), ('H', c_i), ('dropout_p', c_f),
                                                ('seed', ctypes.c_uint)]


class GenBf16Args(ctypes.Structure):
    _fields_ = [(n, c_p) for n in ('cond_pre', 'w0', 'w1', 'bias1', 'wh1', 'bh1', 'wh2', 'bh2', 'ends', 'target', 'tf_mask',
                                   'out_seq', 'x0', 'x1', 'c0', 'c1', 'pos', 'gates0', 'gates1', 'y_all', 'oraw_all')] \
        + [('T', c_i), ('B', c_i), ('H', c_i), ('dropout_p', c_f), ('seed', ctypes.c_uint)]


class GenBf16BwdArgs(ctypes.Structure):
    _fields_ = [(n, c_p) for n in ('gates0', 'gates1', 'c0', 'c1', 'x0', 'y_all', 'oraw_all', 'd_out', 'w1T', 'whh1T', 'wih1T',
                                   'whh0T', 'w_u', 'wh2', 'dypre_all', 'd_o', 'd_in0', 'dc0', 'dc1', 'p_carry',
                                   'tf_mask_host')] + [('T', c_i), ('B', c_i), ('H', c_i), ('dropout_p', c_f), ('seed', ctypes.c_uint)]


class SnFwdLayer(ctypes.Structure):
    _fields_ = [(n, c_p) for n in ('w_orig', 'u', 'v', 'w_out', 'sigma', 'uv_saved')] + [('out_features', c_i), ('in_features', c_i)]


class SnBwdLayer(ctypes.Structure):
    _fields_ = [(n, c_p) for n in ('d_w_sn', 'w_orig', 'uv_saved', 'sigma', 'd_w_orig')] + [('out_features', c_i), ('in_features', c_i)]


class CondFwdArgs(ctypes.Structure):
    _fields_ = [(n, c_p) for n in ('start', 'end', 'z')] + [(n, c_ll) for n in ('ld_start', 'ld_end', 'ld_z')] \
        + [(n, c_p) for n in ('w1', 'b1', 'ln1_g', 'ln1_b', 'w2', 'b2', 'ln2_g', 'ln2_b', 'w_g')] + [('ld_wg', c_ll)] \
        + [(n, c_p) for n in ('bias_a', 'bias_b', 'cond', 'g0', 'c_in', 'xhat1', 'rstd1', 'a1', 'xhat2', 'rstd2')] \
        + [(n, c_i) for n in ('B', 'Bp', 'H', 'latent', 'c_pitch', 'g0_unit_major')] \
        + [(n, c_f) for n in ('slope', 'ln_eps', 'dist_eps')]


class CondBwdArgs(ctypes.Structure):
    _fields_ = [(n, c_p) for n in ('d_g0', 'd_cond', 'w_g')] + [('ld_wg', c_ll)] \
        + [(n, c_p) for n in ('cond', 'xhat2', 'rstd2', 'ln2_g', 'w2', 'a1', 'xhat1', 'rstd1', 'ln1_g', 'dy2', 'dy1', 'ln_part')] \
        + [(n, c_i) for n in ('B', 'Bp', 'H', 'g0_unit_major')] + [('slope', c_f)]


class HeadFwdArgs(ctypes.Structure):
    _fields_ = [(n, c_p) for n in ('x', 'w1', 'b1', 'w2', 'b2', 'w3', 'b3', 'score', 'a1', 'a2')] \
        + [(n, c_i) for n in ('Bp', 'H', 'a2_pitch')] + [('slope', c_f), ('dropout_p', c_f), ('seed', ctypes.c_uint)] \
        + [('m1', c_p), ('m2', c_p)]


class HeadBwdArgs(ctypes.Structure):
    _fields_ = [(n, c_p) for n in ('d_score', 'a1', 'a2', 'w1', 'w2', 'w3', 'dx', 'dy1', 'dy2', 'ds4')] \
        + [(n, c_i) for n in ('Bp', 'H', 'a2_pitch')] + [('slope', c_f), ('dropout_p', c_f), ('seed', ctypes.c_uint)]


SN_MAX_LAYERS = 8

_SIGS = {
    'gmt_version': [],
    'gmt_kinematics_fwd': [c_p, c_p, c_p, c_i, c_i, c_f, c_p],
    'gmt_kinematics_bwd': [c_p, c_p, c_p, c_p, c_p, c_i, c_i, c_f, c_p],
    'gmt_kinematics_jvp': [c_p, c_p, c_p, c_p, c_i, c_i, c_f, c_p],
    'gmt_lrelu_mask_mul': [c_p, c_p, c_p, c_f, c_ll, c_p],
    'gmt_telemetry_kinematics_f64': [c_p, c_p, c_p, c_p, c_p, c_p, c_p, c_ll, c_i, c_i, c_d, c_d, c_p],
    'gmt_telemetry_derived_f64': [c_p, c_p, c_p, c_p, c_p, c_p, c_p, c_ll, c_i, c_p],
    'gmt_adam_step': [c_p, c_p, c_p, c_p, c_ll, c_f, c_f, c_f, c_f, c_i, c_f, c_p],
    'gmt_pack_weight': [c_p, c_p, c_i, c_i, c_i, c_i, c_i, c_i, c_ll, c_ll, c_p],
    'gmt_lstm_layer_fwd': [c_p] * 10 + [c_i] * 5 + [c_f, ctypes.c_uint, c_p],
    'gmt_lstm_layer_bwd': [c_p] * 11 + [c_i] * 4 + [c_f, ctypes.c_uint, c_p],
    'gmt_lstm_layer_bwd_fx': [c_p] * 8 + [c_i] * 5 + [c_f, ctypes.c_uint, c_p],
    'gmt_lstm_layer_bwd2': [c_p] * 12 + [c_i] * 5 + [c_p],
    'gmt_lstm16_supported': [c_i, c_i],
    'gmt_lstm16_layer_fwd': [c_p] * 7 + [c_i] * 4 + [c_p],
    'gmt_lstm16_layer_bwd': [c_p] * 11 + [c_i] * 4 + [c_p],
    'gmt_lstm16_layer_bwd2': [c_p] * 11 + [c_i] * 4 + [c_p],
    'gmt_lstm_xproj': [c_p, c_p, c_p, c_p, c_ll, c_i, c_i, c_i, c_p],
    'gmt_wgrad_workspace_floats': [c_i, c_i, c_ll, c_p, c_p],
    'gmt_wgrad': [c_p, c_ll, c_p, c_ll, c_ll, c_i, c_i, c_p, c_p, c_i, c_p, c_ll, c_f, ctypes.c_uint, c_p],
    'gmt_wgrad_gates': [c_p, c_ll, c_p, c_ll, c_ll, c_i, c_i, c_p, c_p, c_i, c_p, c_ll, c_f, ctypes.c_uint, c_i, c_p],
    'gmt_slab_sum': [c_p, c_p, c_ll, c_i, c_i, c_p],
    'gmt_cond_encoder_fwd': [c_p, c_p],
    'gmt_cond_encoder_bwd': [c_p, c_p],
    'gmt_critic_head_fwd': [c_p, c_p],
    'gmt_critic_head_bwd': [c_p, c_p],
    'gmt_wgrad_noalloc_init': [],
    'gmt_wgrad_noalloc': [c_p, c_ll, c_p, c_ll, c_ll, c_i, c_i, c_p, c_p, c_i, c_f, ctypes.c_uint, c_p],
    'gmt_encoder_fwd': [c_p] * 9 + [c_i, c_i, c_i, c_f, c_f, c_p],
    'gmt_bf16_gemm_nt': [c_p, c_ll, c_p, c_ll, c_p, c_ll, c_p, c_p, c_ll, c_i, c_f, c_ll, c_i, c_i, c_p],
    'gmt_bf16_lstm_layer_fwd': [c_p, c_p, c_p, c_p, c_p, c_p, c_i, c_i, c_i, c_i, c_i, c_f, ctypes.c_uint, c_p],
    'gmt_bf16_lstm_layer_bwd': [c_p, c_p, c_p, c_p, c_p, c_p, c_p, c_p, c_i, c_i, c_i, c_i, c_i, c_f, ctypes.c_uint, c_p],
    'gmt_bf16_wgrad_tn': [c_p, c_ll, c_p, c_ll, c_p, c_ll, c_i, c_i, c_i, c_p],
    'gmt_bf16_lstm_layer_bwd_gp': [c_p] * 11 + [c_i] * 4 + [c_p],
    'gmt_bf16_lstm_layer_bwd2': [c_p] * 11 + [c_i] * 4 + [c_p],
    'gmt_bf16_set_row_groups': [c_i],
    'gmt_dx_gemm_resident': [c_i, c_i],
    'gmt_dx_gemm': [c_p, c_p, c_p, c_p, c_f, c_ll, c_i, c_i, c_i, c_p],
    'gmt_traj_from_seq': [c_p, c_ll, c_ll, c_p, c_p, c_p, c_i, c_i, c_i, c_f, c_p],
    'gmt_traj_from_seq_bwd': [c_p, c_p, c_p, c_ll, c_ll, c_i, c_i, c_i, c_i, c_p],
    'gmt_rollout_to_xyt_f64': [c_p, c_ll, c_ll, c_p, c_p, c_p, c_p, c_i, c_ll, c_d, c_d, c_p],
    'gmt_gen_loss_workspace_floats': [c_i, c_p],
    'gmt_gen_loss_fwd': [c_p, c_ll, c_ll, c_p, c_p, c_p, c_p, c_p, c_p, c_p, c_i, c_i, c_p],
    'gmt_gen_loss_bwd': [c_p, c_ll, c_ll, c_p, c_p, c_p, c_p, c_p, c_p, c_ll, c_ll, c_i, c_i, c_i, c_p],
    'gmt_critic_loss_fwd': [c_p, c_p, c_i, c_p, c_p, c_p],
    'gmt_critic_loss_bwd': [c_p, c_i, c_i, c_p, c_p, c_p],
    'gmt_gen_step_loss_finish': [c_p, c_p, c_f, c_i, c_f, c_f, c_p, c_p, c_p, c_p],
    'gmt_gen_step_loss_bwd': [c_p, c_ll, c_ll, c_p, c_p, c_p, c_p, c_p, c_p, c_p, c_ll, c_ll, c_p, c_i, c_i, c_i, c_p],
    'gmt_spectral_norm_fwd': [c_p, c_i, c_f, c_i, c_p],
    'gmt_spectral_norm_bwd': [c_p, c_i, c_p],
    'gmt_gen_rollout_fwd': [c_p, c_p],
    'gmt_bf16_gen_rollout_fwd': [c_p, c_p],
    'gmt_bf16_gen_rollout_bwd': [c_p, c_p],
    'gmt_gen_rollout_bwd': [c_p, c_p],
}


class GmtError(RuntimeError):
    pass


def lib():
    global _lib
    if _lib is None:
        if not os.path.exists(LIB_PATH):
            raise GmtError(f'{LIB_PATH} is missing: run `python -c "import __graft_entry__ as g; g.build()"` '
                           '(there is no non-HIP fallback)')
        L = ctypes.CDLL(LIB_PATH)
        for name, sig in _SIGS.items():
            fn = getattr(L, name)
            fn.argtypes = sig
            fn.restype = c_i
        if L.gmt_version() != ABI_VERSION:
            raise GmtError(f'libgmt_hip.so ABI {L.gmt_version()} != expected {ABI_VERSION}: rebuild')
        _lib = L
    return _lib


def exported_symbols():
    return list(_SIGS.keys())


def ptr(t):
    """Device pointer of a tensor (None -> NULL).  The tensor must be contiguous."""
    if t is None:
        return None
    if not t.is_contiguous():
        raise GmtError('non-contiguous tensor passed to a HIP kernel')
    return t.data_ptr()


def stream():
    return torch.cuda.current_stream().cuda_stream


# Optional per-kernel timing with events recorded on the launch stream (bench.py turns it
# on for the timed region: roofline.achieved needs each kernel's own duration).  A call may state the
# ALGORITHMIC work of its launch (flops= matrix FLOPs as 2 x MACs, nbytes= bytes an HBM-bound kernel must move);
# profile_stop() then yields per-label totals from which the roofline table is made.
PROFILE = None      # None, or dict: label -> list of (start_event, end_event, flops, nbytes)
PROFILE_ONLY = None # None = every labelled call, or the set of labels to time (an event pair costs ~10 us of GPU
                    # time and serialises neighbouring launches: timing all ~300 calls of an iteration slows it by 8 %)


def profile_start(only=None):
    global PROFILE, PROFILE_ONLY
    PROFILE = {}
    PROFILE_ONLY = set(only) if only is not None else None


def profile_stop():
    """Returns {label: (launches, total_ms, total_flops, total_bytes)}; call after the stream has been synchronised."""
    global PROFILE
    out = {}
    if PROFILE:
        for k, evs in PROFILE.items():
            out[k] = (len(evs), sum(a.elapsed_time(b) for a, b, _, _ in evs), sum(e[2] for e in evs), sum(e[3] for e in evs))
    PROFILE = None
    return out


def call(name, *args, label=None, flops=0.0, nbytes=0.0):
    fn = getattr(lib(), name)
    if PROFILE is not None and label != 'skip' and (PROFILE_ONLY is None or (label or name) in PROFILE_ONLY):
        a = torch.cuda.Event(enable_timing=True)
        b = torch.cuda.Event(enable_timing=True)
        a.record()
        rc = fn(*args)
        b.record()
        PROFILE.setdefault(label or name, []).append((a, b, float(flops), float(nbytes)))
    else:
        rc = fn(*args)
    if rc != 0:
        raise GmtError(f'{name} failed: {ERRORS.get(rc, rc)}')


class timed:
    """`with timed(label, flops=...)`: the same per-label event timing as call() for work that is not one of the library's
    entry points (a plain GEMM handed to hipBLASLt through torch)."""

    def __init__(self, label, flops=0.0, nbytes=0.0):
        self.label, self.flops, self.nbytes, self.a = label, float(flops), float(nbytes), None

    def __enter__(self):
        if PROFILE is not None and (PROFILE_ONLY is None or self.label in PROFILE_ONLY):
            self.a = torch.cuda.Event(enable_timing=True)
            self.a.record()
        return self

    def __exit__(self, *exc):
        if self.a is not None and PROFILE is not None:
            b = torch.cuda.Event(enable_timing=True)
            b.record()
            PROFILE.setdefault(self.label, []).append((self.a, b, self.flops, self.nbytes))
        return False


def require_cuda(*tensors):
    for t in tensors:
        if t is not None and not t.is_cuda:
            raise GmtError('this path runs on the GPU only (hand-written HIP kernels, no CPU fallback); '
                           f'got a tensor on {t.device}')
