"""ctypes binding of libblvm_hip.so (the C ABI declared in include/blvm_hip.h).

No torch types cross this boundary: tensors are passed as raw device pointers (`tensor.data_ptr()`), sizes as ints,
the stream as the current torch HIP stream handle.  There is NO CPU fallback: every wrapper raises if the library
is missing or if it is handed a tensor that does not live on a HIP device.
"""
import ctypes
import os

import torch

# BLVM_HIP_LIB: an alternative build of the same library (kernel experiments); the default is the in-tree build
_LIB_PATH = os.environ.get("BLVM_HIP_LIB") or os.path.join(os.path.dirname(os.path.abspath(__file__)), "lib", "libblvm_hip.so")
_lib = None

c_int, c_float, c_void_p, c_size_t = ctypes.c_int, ctypes.c_float, ctypes.c_void_p, ctypes.c_size_t


class BlvmHipError(RuntimeError):
    pass


class VrnnWeights(ctypes.Structure):
    """struct BlvmVrnnWeights / BlvmVrnnGrads (identical field order)."""

    _fields_ = [
        ("prior_w", c_void_p * 3), ("prior_b", c_void_p * 3), ("prior_hw", c_void_p), ("prior_hb", c_void_p),
        ("post_w", c_void_p * 3), ("post_b", c_void_p * 3), ("post_hw", c_void_p), ("post_hb", c_void_p),
        ("phi_w", c_void_p * 4), ("phi_b", c_void_p * 4),
        ("gru_wih", c_void_p), ("gru_whh", c_void_p), ("gru_bih", c_void_p), ("gru_bhh", c_void_p),
    ]  # fmt: skip


class VrnnDecodeWeights(ctypes.Structure):
    """struct BlvmVrnnDecodeWeights."""

    _fields_ = [
        ("enc_w", c_void_p * 3), ("enc_b", c_void_p * 3), ("cell", ctypes.POINTER(VrnnWeights)),
        ("dec_w", c_void_p * 3), ("dec_b", c_void_p * 3), ("lik_w", c_void_p), ("lik_b", c_void_p),
    ]  # fmt: skip


class SrnnWeights(ctypes.Structure):
    """struct BlvmSrnnWeights / BlvmSrnnGrads."""

    _fields_ = [
        ("prior_w", c_void_p * 3), ("prior_b", c_void_p * 3), ("prior_hw", c_void_p), ("prior_hb", c_void_p),
        ("post_w", c_void_p * 3), ("post_b", c_void_p * 3), ("post_hw", c_void_p), ("post_hb", c_void_p),
    ]  # fmt: skip


class SrnnDecodeWeights(ctypes.Structure):
    """struct BlvmSrnnDecodeWeights."""

    _fields_ = [
        ("enc_w", c_void_p * 3), ("enc_b", c_void_p * 3), ("gru_wih", c_void_p), ("gru_whh", c_void_p), ("gru_bih", c_void_p),
        ("gru_bhh", c_void_p), ("chain", ctypes.POINTER(SrnnWeights)), ("dec_w", c_void_p * 3), ("dec_b", c_void_p * 3),
        ("lik_w", c_void_p), ("lik_b", c_void_p),
    ]  # fmt: skip


class RssmWeights(ctypes.Structure):
    """struct BlvmRssmWeights / BlvmRssmGrads."""

    _fields_ = [
        ("gin_w", c_void_p), ("gin_b", c_void_p), ("gru_wih", c_void_p), ("gru_whh", c_void_p), ("gru_bih", c_void_p),
        ("gru_bhh", c_void_p),
        ("prior_w", c_void_p * 3), ("prior_b", c_void_p * 3), ("prior_hw", c_void_p), ("prior_hb", c_void_p),
        ("post_w", c_void_p * 3), ("post_b", c_void_p * 3), ("post_hw", c_void_p), ("post_hb", c_void_p),
    ]  # fmt: skip


_SIGNATURES = {
    "blvm_upload_i32": (c_int, [c_void_p, c_int, c_void_p, c_void_p]),
    "blvm_version": (c_int, []),
    "blvm_last_error": (ctypes.c_char_p, []),
    "blvm_device_ok": (c_int, []),
    "blvm_async_errors": (c_int, [c_void_p]),
    "blvm_async_errors_take": (c_int, [c_void_p]),
    "blvm_pchain_configure": (c_int, [c_int, c_int]),
    "blvm_pchain_max_batch": (c_int, []),
    "blvm_set_operand_dtype": (c_int, [c_int]),
    "blvm_get_operand_dtype": (c_int, []),
    "blvm_pchain_profile": (c_int, [c_void_p]),
    "blvm_pchain_tune": (c_int, [c_int]),
    "blvm_pchain_chain_probe": (c_int, [c_void_p, c_void_p, c_void_p, c_void_p, c_int, c_int, c_int, c_int, c_void_p]),
    "blvm_pchain_rows_to_t16": (c_int, [c_void_p, c_int, c_int, c_int, c_void_p, c_void_p]),
    "blvm_gemm_f32": (c_int, [c_int, c_int, c_int, c_int, c_int, c_void_p, c_int, c_void_p, c_int, c_void_p, c_int,
                              c_void_p, c_int, c_float, c_void_p, c_int, c_int, c_int, c_void_p]),
    "blvm_act_bwd_f32": (c_int, [c_void_p, c_void_p, c_float, c_void_p, c_size_t, c_void_p]),
    "blvm_colsum_f32": (c_int, [c_int, c_int, c_void_p, c_int, c_void_p, c_int, c_void_p]),
    "blvm_wgrad_f32": (c_int, [c_int, c_int, c_int, c_void_p, c_int, c_void_p, c_int, c_void_p, c_int, c_void_p, c_int, c_void_p]),
    "blvm_wgrad_group_f32": (c_int, [c_int, c_void_p, c_void_p, c_int, c_void_p, c_void_p, c_void_p, c_void_p, c_void_p, c_void_p, c_void_p, c_void_p]),
    "blvm_dmol_fwd": (c_int, [c_void_p, c_int, c_void_p, c_void_p, c_void_p, c_void_p, c_int, c_int, c_int, c_int,
                              c_int, c_int, c_float, c_void_p, c_void_p, c_void_p]),
    "blvm_dmol_bwd": (c_int, [c_void_p, c_int, c_void_p, c_void_p, c_void_p, c_void_p, c_void_p, c_int, c_int, c_int,
                              c_int, c_int, c_int, c_float, c_void_p, c_void_p, c_void_p]),
    "blvm_kl_fwd": (c_int, [c_void_p, c_void_p, c_void_p, c_void_p, c_int, c_void_p, c_int, c_int, c_int, c_int,
                            c_float, c_void_p, c_void_p, c_void_p]),
    "blvm_kl_bwd": (c_int, [c_void_p, c_void_p, c_void_p, c_void_p, c_int, c_void_p, c_void_p, c_void_p, c_int, c_int,
                            c_int, c_int, c_float, c_void_p, c_void_p, c_void_p, c_void_p, c_void_p]),
    "blvm_vrnn_decode_scratch_floats": (c_size_t, [c_int] * 4),
    "blvm_vrnn_decode": (c_int, [ctypes.POINTER(VrnnDecodeWeights)] + [c_void_p] * 5 + [c_int] * 7 + [c_float] * 3 + [c_void_p] * 4),
    "blvm_srnn_generate_scratch_floats": (c_size_t, [c_int] * 6),
    "blvm_srnn_generate": (c_int, [ctypes.POINTER(SrnnDecodeWeights)] + [c_void_p] * 6 + [c_int] * 7 + [c_float] * 3 + [c_void_p] * 5),
    "blvm_vrnn_generate_scratch_floats": (c_size_t, [c_int] * 6),
    "blvm_vrnn_generate": (c_int, [ctypes.POINTER(VrnnDecodeWeights)] + [c_void_p] * 5 + [c_int] * 7 + [c_float] * 3 + [c_void_p] * 4),
    "blvm_vrnn_reserve_floats": (c_size_t, [c_int] * 6),
    "blvm_vrnn_bwd_workspace_floats": (c_size_t, [c_int] * 6),
    "blvm_vrnn_seq_fwd": (c_int, [ctypes.POINTER(VrnnWeights), c_void_p, c_void_p, c_void_p, c_int, c_int, c_int, c_int,
                                  c_int, c_int, c_int, c_float, c_void_p, c_void_p, c_void_p, c_void_p, c_void_p,
                                  c_void_p, c_void_p, c_void_p]),
    "blvm_vrnn_seq_bwd": (c_int, [ctypes.POINTER(VrnnWeights), c_void_p, c_void_p, c_void_p, c_void_p, c_void_p,
                                  c_void_p, c_void_p, c_void_p, c_void_p, c_void_p, c_void_p, c_void_p, c_void_p,
                                  c_int, c_float, c_int, c_int, c_int, c_int, c_int, c_int, c_int, c_float, c_void_p,
                                  c_void_p, ctypes.POINTER(VrnnWeights), c_void_p, c_void_p]),
    "blvm_lstm_reserve_floats": (c_size_t, [c_int] * 3),
    "blvm_lstm_bwd_workspace_floats": (c_size_t, [c_int] * 3),
    "blvm_lstm_seq_fwd": (c_int, [c_void_p] * 8 + [c_int] * 4 + [c_void_p] * 5),
    "blvm_lstm_seq_bwd": (c_int, [c_void_p] * 5 + [c_int] * 4 + [c_void_p] * 9),
    "blvm_gru_reserve_floats": (c_size_t, [c_int] * 3),
    "blvm_gru_bwd_workspace_floats": (c_size_t, [c_int] * 3),
    "blvm_gru_seq_fwd": (c_int, [c_void_p] * 5 + [c_int, c_void_p, c_void_p] + [c_int] * 5 + [c_void_p, ctypes.c_longlong, c_int]
                         + [c_void_p] * 3),
    "blvm_gru_seq_bwd": (c_int, [c_void_p] * 3 + [c_int, c_void_p, c_int, c_void_p, c_void_p, ctypes.c_longlong, c_int]
                         + [c_int] * 4 + [c_void_p, c_int, c_int] + [c_void_p] * 7),
    "blvm_srnn_reserve_floats": (c_size_t, [c_int] * 5),
    "blvm_srnn_bwd_workspace_floats": (c_size_t, [c_int] * 5),
    "blvm_srnn_latent_fwd": (c_int, [ctypes.POINTER(SrnnWeights)] + [c_void_p] * 4 + [c_int] * 6 + [c_float, c_float]
                             + [c_void_p] * 7),
    "blvm_srnn_latent_bwd": (c_int, [ctypes.POINTER(SrnnWeights)] + [c_void_p] * 13 + [c_int, c_float] + [c_int] * 6
                             + [c_float, c_float] + [c_void_p] * 3 + [ctypes.POINTER(SrnnWeights), c_void_p, c_void_p]),
    "blvm_scale_act_f32": (c_int, [c_void_p, c_float, c_float, c_void_p, c_size_t, c_void_p]),
    "blvm_conv1d_k2_workspace_floats": (c_size_t, [c_int] * 2),
    "blvm_conv1d_k2_fwd": (c_int, [c_void_p] * 3 + [c_int] * 5 + [c_void_p] * 3),
    "blvm_conv1d_k2_bwd": (c_int, [c_void_p] * 3 + [c_int] * 5 + [c_void_p] * 5),
    "blvm_wavenet_block_reserve_floats": (c_size_t, [c_int] * 4),
    "blvm_wavenet_block_workspace_floats": (c_size_t, [c_int] * 5),
    "blvm_wavenet_block_fwd": (c_int, [c_void_p] * 5 + [c_int] * 6 + [c_float] + [c_void_p] * 5),
    "blvm_wavenet_block_bwd": (c_int, [c_void_p] * 6 + [c_int] * 6 + [c_float] + [c_void_p] * 7),
    "blvm_wavenet_stack_floats": (c_int, [c_int] * 3 + [c_void_p, c_int, c_void_p, c_void_p]),
    "blvm_wavenet_stack_fwd": (c_int, [c_void_p] * 4 + [c_int] * 6 + [c_float] + [c_void_p] * 5),
    "blvm_wavenet_stack_bwd": (c_int, [c_void_p] * 4 + [c_int] * 6 + [c_float] + [c_void_p] * 8),
    "blvm_wavenet_decode_pack_floats": (c_size_t, [c_int] * 4),
    "blvm_wavenet_decode_scratch_floats": (c_size_t, [c_void_p] + [c_int] * 4),
    "blvm_wavenet_decode": (c_int, [c_void_p] * 2 + [c_int] * 7 + [c_float] * 3 + [c_void_p] * 5),
    "blvm_rssm_reserve_floats": (c_size_t, [c_int] * 4),
    "blvm_rssm_bwd_workspace_floats": (c_size_t, [c_int] * 4),
    "blvm_rssm_seq_fwd": (c_int, [ctypes.POINTER(RssmWeights)] + [c_void_p] * 5 + [c_int] * 7 + [c_float] + [c_void_p] * 8),
    "blvm_rssm_seq_bwd": (c_int, [ctypes.POINTER(RssmWeights)] + [c_void_p] * 15 + [c_int, c_float] + [c_int] * 7 + [c_float]
                          + [c_void_p] * 4 + [ctypes.POINTER(RssmWeights), c_void_p, c_void_p]),
    "blvm_gmm_fwd": (c_int, [c_void_p, c_int] + [c_void_p] * 4 + [c_int] * 5 + [c_float] * 2 + [c_void_p] * 3),
    "blvm_gmm_bwd": (c_int, [c_void_p, c_int] + [c_void_p] * 5 + [c_int] * 5 + [c_float] * 2 + [c_void_p] * 3),
    "blvm_gauss_head_fwd": (c_int, [c_void_p, c_int] + [c_void_p] * 4 + [c_int] * 4 + [c_float] * 2 + [c_void_p] * 3),
    "blvm_gauss_head_bwd": (c_int, [c_void_p, c_int] + [c_void_p] * 5 + [c_int] * 4 + [c_float] * 2 + [c_void_p] * 3),
    "blvm_mix_sample": (c_int, [c_void_p] * 3 + [ctypes.c_longlong] + [c_int] * 2 + [c_float] * 3 + [c_void_p] * 2),
    "blvm_gauss_latent_fwd": (c_int, [c_void_p] * 5 + [c_size_t] + [c_float] * 3 + [c_int] + [c_void_p] * 5),
    "blvm_gauss_latent_bwd": (c_int, [c_void_p] * 9 + [c_size_t] + [c_float] * 3 + [c_int] + [c_void_p] * 5),
    "blvm_chan_norm_workspace_doubles": (c_size_t, [c_int]),
    "blvm_chan_norm_stats": (c_int, [c_void_p] + [c_int] * 3 + [c_void_p] * 2 + [c_float] + [c_void_p] * 4),
    "blvm_chan_norm_fwd": (c_int, [c_void_p] + [c_int] * 3 + [c_void_p] * 2 + [c_float] + [c_void_p] * 4),
    "blvm_chan_norm_bwd": (c_int, [c_void_p] * 4 + [c_int] * 4 + [c_void_p] * 6),
    "blvm_dwconv_out_length": (c_int, [c_int] * 5),
    "blvm_dwconv_fwd": (c_int, [c_void_p] * 5 + [c_int] * 8 + [c_void_p] * 2),
    "blvm_dwconv_bwd": (c_int, [c_void_p] * 6 + [c_int] * 8 + [c_void_p] * 4),
    "blvm_resample_add_fwd": (c_int, [c_void_p] * 2 + [c_int] * 3 + [c_void_p] * 2),
    "blvm_resample_add_bwd": (c_int, [c_void_p] + [c_int] * 3 + [c_void_p] * 2),
}  # fmt: skip

EXPORTS = tuple(_SIGNATURES)


def lib_path() -> str:
    return _LIB_PATH


def load():
    """Load libblvm_hip.so (once) and attach argument/return types.  Raises BlvmHipError if it is not built."""
    global _lib
    if _lib is None:
        if not os.path.exists(_LIB_PATH):
            raise BlvmHipError(
                f"libblvm_hip.so not found at {_LIB_PATH}: build it with benchmarking-lvms_amd/csrc/build.sh "
                "(or __graft_entry__.build()).  There is no CPU/PyTorch fallback for the blvm hot path."
            )
        lib = ctypes.CDLL(_LIB_PATH)
        for name, (res, args) in _SIGNATURES.items():
            fn = getattr(lib, name)
            fn.restype, fn.argtypes = res, args
        _lib = lib
    return _lib


def check(rc: int, what: str):
    if rc != 0:
        msg = load().blvm_last_error().decode(errors="replace")
        raise BlvmHipError(f"{what} failed (code {rc}): {msg}")


def take_async_errors():
    """(number of persistent launches of this process that gave up on a bounded spin since the previous take, code of the last one).
    Read-and-clear: an abort is reported once.  A read of pinned host memory — meaningful after the host has synchronised."""
    code = ctypes.c_uint(0)
    n = load().blvm_async_errors_take(ctypes.byref(code))
    return int(n), int(code.value)


def check_async(what: str = "a persistent recurrent launch", group=None):
    """Raise if a persistent chain launch of this process gave up on a bounded spin since the last check (its results are garbage).
    Cheap (a read of pinned host memory): called wherever the host has just synchronised with the device to read results back.
    With a `torch.distributed` process group (`group=True`: the default group) the count is all-reduced (MAX) first, so that every
    rank raises together instead of one rank leaving the others blocked in their next collective."""
    n, code = take_async_errors()
    local = n
    if group is not None and group is not False:
        import torch.distributed as dist

        if dist.is_available() and dist.is_initialized():
            pg = None if group is True else group
            on_gpu = dist.get_backend(pg) == "nccl"
            t = torch.tensor([n], dtype=torch.int32, device="cuda" if on_gpu else "cpu")
            dist.all_reduce(t, op=dist.ReduceOp.MAX, group=pg)
            n = int(t.item())
    if n:
        where = f"last at step {code >> 4}, link {code & 15}" if local else "on another rank"
        raise BlvmHipError(
            f"{what}: {n} persistent launch(es) aborted on a bounded spin ({where}): "
            "a workgroup of the launch was not resident (is another process using this GPU?); results are invalid"
        )


DTYPES = {"f32": 0, "bf16": 1}


def set_operand_dtype(name: str):
    """Operand type of the matrix products: "f32" (default) or "bf16" (bf16 operands, fp32 accumulation, for the persistent
    recurrent chains and the K6 GEMMs) — what the reference's `--use_amp True` selects with torch.autocast
    (`experiments/experiment_vrnn_audio.py:219-230`).  Process-wide."""
    if name not in DTYPES:
        raise ValueError(f"operand dtype {name!r}: one of {sorted(DTYPES)}")
    check(load().blvm_set_operand_dtype(DTYPES[name]), "blvm_set_operand_dtype")


def get_operand_dtype() -> str:
    return "bf16" if load().blvm_get_operand_dtype() == 1 else "f32"


_raw_stream = getattr(torch._C, "_cuda_getCurrentRawStream", None)
_raw_device = getattr(torch._C, "_cuda_getDevice", None)


def stream_ptr() -> int:
    """hipStream_t of torch's current stream on the current device, as an integer.  Through the raw accessors when this torch has
    them (0.1 us; `torch.cuda.current_stream().cuda_stream` builds a Stream object per call: 2.7 us, a third of a small launch's host
    cost — the CW-VAE step makes ~1 000 such calls)."""
    if _raw_stream is not None and _raw_device is not None:
        return _raw_stream(_raw_device())
    return torch.cuda.current_stream().cuda_stream


def ptr(t):
    """Device pointer of a tensor (None -> NULL).  Refuses anything that is not a contiguous HIP tensor."""
    if t is None:
        return None
    if not t.is_cuda:
        raise BlvmHipError(
            "blvm HIP kernels were handed a CPU tensor: this build runs the hot path on gfx950 only "
            "(no CPU fallback); move the model and inputs to a HIP device."
        )
    if not t.is_contiguous():
        raise BlvmHipError("blvm HIP kernels need contiguous tensors")
    return t.data_ptr()
