"""ctypes binding of libomnisafe_amd.so -- the C-ABI boundary (include/omnisafe_amd.h).

There is NO CPU fallback: if the library (or a GPU) is missing the product fails loudly here.
"""
from __future__ import annotations

import ctypes as C
import os

import torch

from . import build as _build

_LIB = None
OSA_OK, OSA_EUNSUPPORTED = 0, -3  # include/omnisafe_amd.h

c_f32p = C.c_void_p  # device pointers travel as integers
c_ptr = C.c_void_p

# name -> (restype, argtypes); mirrors include/omnisafe_amd.h one to one.
_I, _L, _F, _D, _P, _U = C.c_int, C.c_long, C.c_float, C.c_double, C.c_void_p, C.c_ulonglong
SIGNATURES: dict[str, tuple] = {
    'osa_strerror': (C.c_char_p, [_I]),
    'osa_version': (_I, []),
    'osa_build_arch': (C.c_char_p, []),
    'osa_abi_digest': (C.c_char_p, []),
    'osa_buffer_store_step': (_I, [_I, _I, _I, _I, _P, _I, _P, _I, _P, _P, _P, _P, _P, _P, _I, _P, _I,
                                   _P, _P, _P, _P, _P, _P]),
    'osa_gae_scan': (_I, [_P, _P, _P, _P, _P, _P, _P, _I, _I, _D, _D, _D, _F, _I, _P, _P, _P, _P, _P, _P]),
    'osa_gae_scan_tiled': (_I, [_P, _P, _P, _P, _P, _P, _P, _I, _I, _D, _D, _D, _F, _I, _P, _P, _P, _P, _P, _P]),
    'osa_gae_chained_ws_doubles': (C.c_size_t, [_I, _I]),
    'osa_gae_chained_timed_out': (_I, [_P, _I, _I, _P]),
    'osa_gae_scan_chained': (_I, [_P, _P, _P, _P, _P, _P, _P, _I, _I, _D, _D, _D, _F, _I, _P, _P, _P, _P, _P, _P, _P]),
    'osa_reduce_ws_bytes': (C.c_size_t, []),
    'osa_adv_stats_phase1': (_I, [_P, _P, _L, _P, _P, _P]),
    'osa_adv_stats_phase2': (_I, [_P, _L, _P, _P, _P]),
    'osa_buffer_get': (_I, [_I, _I, _I, _I, _P, _I, _P, _I, _P, _P, _P, _P, _P, _P, _P, _I, _I,
                            _P, _I, _P, _I, _P, _P, _P, _P, _P, _P, _P]),
    'osa_mlp_layout': (_I, [_I, _I, _I, _P]),
    'osa_policy_step': (_I, [_I, _I, _I, _P, _P, _I, _I, _P, _U, _U, _P, _I, _I, _P, _I, _P, _P, _P, _P, _I,
                             _P]),
    'osa_policy_step_scaled': (_I, [_I, _I, _I, _P, _P, _I, _I, _P, _U, _U, _P, _I, _I, _P, _I, _P, _P, _P, _P, _I,
                                    _P, _I, _P, _P, _F, _F, _P]),
    'osa_minibatch_ws_floats': (C.c_size_t, [_I, _I, _I, _I]),
    # general networks (csrc/general_mlp.hip): shapes in an osa_gmlp_desc (models.GmlpDesc, passed by reference)
    'osa_gmlp_layout': (_I, [_P, _P]),
    'osa_gmlp_ws_floats': (C.c_size_t, [_P, _L]),
    'osa_gmlp_policy_step': (_I, [_P, _P, _P, _I, _L, _P, _U, _U, _P, _I, _I, _P, _I, _P, _P, _P, _P, _I, _P, _I, _P,
                                  _P, _F, _F, _P, C.c_size_t, _P]),
    'osa_gmlp_minibatch': (_I, [_P, _P, _P, _P, _P, _P, _P, _I, _P, _I, _P, _P, _P, _P, _P, _P, _L, _P, _P, _I, _I,
                                _I, _P, _F, _P, C.c_size_t, _P, _P]),
    'osa_gmlp_minibatch_ext': (_I, [_P, _P, _P, _P, _P, _P, _P, _I, _P, _I, _P, _P, _P, _P, _P, _P, _L, _P, _P, _I, _I,
                                    _I, _P, _F, _P, C.c_size_t, _P, _P, _P]),
    'osa_gmlp_adam_apply': (_I, [_P, _P, _P, _P, _P, _P, _P, _I, _P, _P]),
    'osa_gmlp_actor_stats': (_I, [_P, _P, _P, _I, _L, _P, _I, _P, _I, _I, _P, _I, _P, _P, _P, _P, _P, _I, _P,
                                  C.c_size_t, _P, _P]),
    'osa_ppo_minibatch': (_I, [_I, _I, _I, _P, _P, _P, _P, _P, _P, _I, _P, _I, _P, _P, _P, _P, _P, _P,
                               _I, _P, _P, _I, _I, _I, _I, _P, _P, _P]),
    'osa_ppo_minibatch_ext': (_I, [_I, _I, _I, _P, _P, _P, _P, _P, _P, _I, _P, _I, _P, _P, _P, _P, _P, _P,
                                   _I, _P, _P, _I, _I, _I, _I, _P, _P, _P, _P]),
    'osa_episode_flush_ws_doubles': (C.c_size_t, [_L]),
    'osa_episode_flush': (_I, [_P, _P, _P, _P, _P, _L, _P, _P, _P, _P, _P, _P, _P, _P]),
    'osa_gather_mean': (_I, [_P, _P, _L, _P, _P]),
    'osa_saute_step': (_I, [_I, _P, _P, _P, _P, _P, _P, _F, _F, _P, _P, _P, _I, _P, _I, _I, _P, _P, _P]),
    'osa_debug_set_clock_buffer': (_I, [_P]),
    'osa_debug_set_pass_clock_buffer': (_I, [_P]),
    'osa_debug_set_part_clock_buffer': (_I, [_P]),
    'osa_ppo_pass_supported': (_I, [_I, _I, _I]),
    'osa_ppo_dp_end_pass': (_I, [_P, _I, _I, _P]),
    'osa_ppo_dp_ws_floats': (C.c_size_t, [_I, _I, _I, _I]),
    'osa_ppo_dp_pass_ws_floats': (C.c_size_t, [_I, _I, _I, _I]),
    'osa_shuffle_rows': (_I, [_P, _I, _L, _P, _P]),
    'osa_dp_exchange_alloc': (_I, [C.c_size_t, C.POINTER(C.c_void_p)]),
    'osa_dp_exchange_free': (_I, [_P]),
    'osa_ppo_dp_pass': (_I, [_I, _I, _I, _P, _P, _P, _P, _P, _I, _P, _I, _P, _P, _P, _P, _P, _P, _L, _I, _I,
                        _P, _P, _I, _I, _P, _P, _P, _P]),
    'osa_ppo_dp_pass_placed': (_I, [_I, _I, _I, _P, _P, _P, _P, _P, _I, _P, _I, _P, _P, _P, _P, _P, _P, _L, _I, _I,
                               _P, _P, _I, _I, _P, _P, _I, _P, _P]),
    'osa_ppo_chunked_pass': (_I, [_I, _I, _I, _P, _P, _P, _P, _P, _I, _P, _I, _P, _P, _P, _P, _P, _P, _L, _I,
                                  _P, _P, _I, _I, _P, _P, _I, _P, _P]),
    'osa_ppo_dp_chunked_pass_ws_floats': (C.c_size_t, [_I, _I, _I, _I, _I]),
    'osa_ppo_dp_chunked_pass': (_I, [_I, _I, _I, _P, _P, _P, _P, _P, _I, _P, _I, _P, _P, _P, _P, _P, _P, _L, _I, _I,
                                     _P, _P, _I, _I, _P, _P, _I, _P, _P]),
    # one-shot peer exchange (csrc/p2p_pass_kernel.hip)
    'osa_p2p_exchange_floats': (C.c_size_t, [_I, _I, _I, _I]),
    'osa_p2p_exchange_alloc': (_I, [C.c_size_t, C.POINTER(C.c_void_p), _P]),
    'osa_p2p_exchange_open': (_I, [_P, C.POINTER(C.c_void_p)]),
    'osa_p2p_exchange_release': (_I, [_P]),
    'osa_p2p_exchange_timed_out': (_I, [_P, _P]),
    'osa_ppo_p2p_pass': (_I, [_I, _I, _I, _P, _P, _P, _P, _P, _I, _P, _I, _P, _P, _P, _P, _P, _P, _L, _I, _I, _I,
                              _P, C.c_uint, _D, _P, _P, _I, _I, _P, _P]),
    'osa_ppo_dp_step': (_I, [_I, _I, _I, _P, _P, _P, _P, _P, _I, _P, _I, _P, _P, _P, _P, _P, _P, _L, _I, _I,
                             _I, _P, _P, _P, _I, _I, _P, _P, _P]),
    'osa_ppo_dp_step_phase': (_I, [_I, _I, _I, _P, _P, _P, _P, _P, _I, _P, _I, _P, _P, _P, _P, _P, _P, _L, _I, _I, _I, _P, _P, _P, _I, _I, _P, _P, _I, _P]),
    'osa_ppo_pass': (_I, [_I, _I, _I, _P, _P, _P, _P, _P, _I, _P, _I, _P, _P, _P, _P, _P, _P, _L, _I,
                          _P, _P, _I, _I, _P, _P]),
    'osa_ppo_wide_pass_supported': (_I, [_I, _I, _I]),
    'osa_ppo_wide_pass_ws_floats': (C.c_size_t, [_I, _I, _I]),
    'osa_ppo_wide_pass': (_I, [_I, _I, _I, _P, _P, _P, _P, _P, _I, _P, _I, _P, _P, _P, _P, _P, _P, _L, _I,
                               _P, _P, _I, _I, _P, _P, _P]),
    'osa_ppo_split_pass_supported': (_I, [_I, _I, _I]),
    'osa_ppo_split_pass_xch_floats': (C.c_size_t, [_I, _I, _I]),
    'osa_ppo_split_pass': (_I, [_I, _I, _I, _P, _P, _P, _P, _P, _I, _P, _I, _P, _P, _P, _P, _P, _P, _L, _I,
                                _P, _P, _I, _I, _P, _I, _P, _P]),
    'osa_ppo_split_pass_timed_out': (_I, [_P, _P]),
    'osa_ppo_split_pass_clear_flag': (_I, [_P]),
    'osa_ppo_split_dp_xch_floats': (C.c_size_t, [_I, _I, _I, _I]),
    'osa_ppo_split_dp_dpx_floats': (C.c_size_t, [_I, _I, _I, _I]),
    'osa_ppo_split_dp_pass': (_I, [_I, _I, _I, _P, _P, _P, _P, _P, _I, _P, _I, _P, _P, _P, _P, _P, _P, _L, _I, _I,
                                   _P, _P, _I, _I, _P, _P, _I, _P, _P]),
    'osa_ppo_pass_ext': (_I, [_I, _I, _I, _P, _P, _P, _P, _P, _I, _P, _I, _P, _P, _P, _P, _P, _P, _L, _I,
                              _P, _P, _I, _I, _P, _P, _P]),
    'osa_adam_apply': (_I, [_I, _I, _I, _P, _P, _P, _P, _P, _P, _I, _P]),
    'osa_actor_fvp_raw': (_I, [_I, _I, _I, _P, _P, _P, _I, _L, _P, _I, _P, _P, _P]),
    'osa_fvp_finish': (_I, [_I, _P, _P, _F, _I, _I, _F, _P, _P]),
    'osa_cg_init': (_I, [_I, _P, _P, _P, _P, _P, _P]),
    'osa_cg_step': (_I, [_I, _P, _P, _P, _P, _P, _F, _F, _P]),
    'osa_vec_lincomb': (_I, [_I, _F, _P, _F, _P, _P, _P]),
    'osa_vec_dot': (_I, [_I, _P, _P, _P, _P]),
    'osa_actor_eval': (_I, [_I, _I, _I, _P, _P, _I, _L, _P, _I, _P, _P, _P, _P, _P, _I, _P, _P, _P, _P]),
    'osa_actor_kl': (_I, [_I, _I, _I, _P, _P, _I, _L, _P, _I, _P, _I, _P, _I, _P, _P, _P]),
    'osa_normalizer_ws_doubles': (C.c_size_t, [_I, _I]),
    'osa_normalizer_push': (_I, [_P, _I, _I, _I, _P, _P, _P, _P, _P, _P, _P, _P]),
    'osa_normalizer_apply': (_I, [_P, _I, _P, _I, _I, _I, _P, _P, _P, _P, _F, _P]),
    'osa_action_scale': (_I, [_P, _I, _P, _I, _I, _I, _P, _P, _F, _F, _P]),
    'osa_rollout_post_step': (_I, [_I, _I] + [_P] * 21),
    'osa_synth_env_step': (_I, [_U, _U, _P, _I, _I, _I, _F, _P, _P, _I, _P, _P, _P, _P, _P, _I, _I, _P]),
    'osa_reach_env_step': (_I, [_U, _U, _P, _I, _I, _I, _P, _P, _P, _I, _P, _I, _P, _P, _P, _P, _P, _I, _I, _P]),
}


class OsaError(RuntimeError):
    pass


def lib_path() -> str:
    # OSA_LIB_PATH: an alternative build of the same library (tools/build_variant_lib.sh: phase clocks)
    return os.environ.get('OSA_LIB_PATH') or _build.LIB_PATH


def load(require_gpu: bool = False):
    """Load (building if necessary) the shared library and declare every prototype."""
    global _LIB
    if _LIB is None:
        path = lib_path()
        own = path == _build.LIB_PATH  # OSA_LIB_PATH variants (tools/) are taken as they are

        def build(force: bool) -> None:
            try:
                _build.build_library(force=force, verbose=False)
            except Exception as exc:  # pragma: no cover
                raise OsaError(
                    f'libomnisafe_amd.so is missing or stale and could not be built ({exc}); '
                    'run `python -m omnisafe_amd.build` -- there is no CPU fallback') from exc

        def digest(lib) -> str:
            try:
                lib.osa_abi_digest.restype = C.c_char_p
                return lib.osa_abi_digest().decode()
            except AttributeError:
                return 'none'

        if own and not os.path.exists(path):
            build(False)
        lib = C.CDLL(path)
        if own:
            # ctypes cannot check prototypes, and the library is git-ignored but travels with the working
            # tree: refuse to bind one that was not built from the sources next to it (content digest, not
            # file times) -- rebuild it where hipcc exists, fail loudly where it does not
            want = _build.source_digest()
            if digest(lib) != want:
                import _ctypes

                _ctypes.dlclose(lib._handle)  # noqa: SLF001
                build(True)
                lib = C.CDLL(path)
                if digest(lib) != want:
                    raise OsaError(f'{path} was built from other sources (digest {digest(lib)}, expected '
                                   f'{want}); run `python -m omnisafe_amd.build --force`')
        for name, (res, args) in SIGNATURES.items():
            fn = getattr(lib, name)  # AttributeError if the symbol is not exported
            fn.restype = res
            fn.argtypes = args
        _LIB = lib
    if require_gpu and not torch.cuda.is_available():
        raise OsaError('omnisafe_amd needs an AMD GPU visible to torch (ROCm); there is no CPU fallback')
    return _LIB


def check(code: int, what: str = '') -> None:
    if code != 0:
        msg = load().osa_strerror(code).decode()
        raise OsaError(f'{what or "libomnisafe_amd"} failed: {msg} ({code})')


def ptr(t: torch.Tensor | None) -> int | None:
    """Device pointer of a tensor (None -> NULL)."""
    if t is None:
        return None
    assert t.is_cuda, 'omnisafe_amd kernels take device tensors'
    return t.data_ptr()


_RAW_STREAM = getattr(torch._C, '_cuda_getCurrentRawStream', None)  # pylint: disable=protected-access


def stream_ptr() -> int:
    """hipStream_t of torch's current stream, so kernels order with torch ops and RCCL.
    (torch.cuda.current_stream() builds a Python Stream object through five layers of device-index helpers: 9 us per
    call, ~40 calls per epoch, all of them on the host path between a synchronisation and the next launch; the raw
    getter is the same value in 0.2 us.)"""
    if _RAW_STREAM is not None:
        return _RAW_STREAM(torch.cuda.current_device())
    return torch.cuda.current_stream().cuda_stream
