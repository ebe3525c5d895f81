"""ctypes binding of libvitsom_hip.so (the C-ABI declared in include/vitsom_hip.h).

There is NO fallback: if the shared library is missing the import fails loudly, and every
wrapper raises ``VsomError`` on a non-zero status.
"""
import ctypes as C
import os

import torch

_HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.path.join(_HERE, "libvitsom_hip.so")

c_fp = C.c_void_p      # float* / int64_t* / void*  (device pointers travel as integers)
c_stream = C.c_void_p

# name -> (restype, argtypes); mirrors include/vitsom_hip.h one to one (tests/test_abi.py checks)
SIGNATURES = {
    "vsom_version": (C.c_int, []),
    "vsom_last_error_string": (C.c_char_p, []),
    "vsom_linear_fwd": (C.c_int, [c_fp, C.c_long, c_fp, c_fp, c_fp, C.c_long, C.c_int, C.c_int, C.c_int, c_stream]),
    "vsom_linear_gelu_fwd": (C.c_int, [c_fp, C.c_long, c_fp, c_fp, c_fp, c_fp, C.c_int, C.c_int, C.c_int, c_stream]),
    "vsom_linear_relu_fwd": (C.c_int, [c_fp, C.c_long, c_fp, c_fp, c_fp, c_fp, C.c_int, C.c_int, C.c_int, c_stream]),
    "vsom_l1_loss_workspace_bytes": (C.c_size_t, [C.c_long]),
    "vsom_l1_loss": (C.c_int, [c_fp, c_fp, c_fp, c_fp, C.c_float, C.c_long, c_fp, C.c_size_t, c_stream]),
    "vsom_linear_residual_fwd": (C.c_int, [c_fp, C.c_long, c_fp, c_fp, c_fp, C.c_long, C.c_int, c_fp, C.c_long,
                                           C.c_int, C.c_int, C.c_int, c_stream]),
    "vsom_linear_bwd_input": (C.c_int, [c_fp, C.c_long, c_fp, c_fp, C.c_long, C.c_int, C.c_int, C.c_int, C.c_int,
                                        c_fp, c_stream]),
    "vsom_linear_bwd_input_t": (C.c_int, [c_fp, C.c_long, c_fp, c_fp, C.c_long, C.c_int, C.c_int, C.c_int, C.c_int,
                                          c_fp, c_stream]),
    "vsom_transpose_many": (C.c_int, [c_fp, c_fp, C.c_void_p, C.c_int, C.c_int, C.c_int, c_stream]),
    "vsom_set_gemm_mode": (C.c_int, [C.c_int]),
    "vsom_get_gemm_mode": (C.c_int, []),
    "vsom_linear_bwd_weight_workspace_bytes": (C.c_size_t, [C.c_int, C.c_int, C.c_int]),
    "vsom_linear_bwd_weight": (C.c_int, [c_fp, C.c_long, c_fp, C.c_long, c_fp, c_fp, C.c_int, C.c_int, C.c_int,
                                         c_fp, C.c_size_t, c_stream]),
    "vsom_patch_embed_fwd": (C.c_int, [c_fp, c_fp, c_fp, c_fp, c_fp, c_fp, c_fp, C.c_int, C.c_int, C.c_int, C.c_int,
                                       C.c_int, c_stream]),
    "vsom_patch_embed_bwd_workspace_bytes": (C.c_size_t, [C.c_int] * 5),
    "vsom_patch_embed_bwd": (C.c_int, [c_fp, c_fp, c_fp, c_fp, c_fp, C.c_int, C.c_int, C.c_int, C.c_int, C.c_int,
                                       c_fp, C.c_size_t, c_stream]),
    "vsom_layernorm_fwd": (C.c_int, [c_fp, c_fp, c_fp, c_fp, c_fp, c_fp, C.c_int, C.c_int, C.c_float, c_stream]),
    "vsom_layernorm_bwd_workspace_bytes": (C.c_size_t, [C.c_int, C.c_int]),
    "vsom_layernorm_bwd": (C.c_int, [c_fp] * 9 + [C.c_int, C.c_int, c_fp, C.c_size_t, c_stream]),
    "vsom_layernorm_bwd_deferrable": (C.c_int, [C.c_int, C.c_int]),
    "vsom_layernorm_bwd_partial": (C.c_int, [c_fp] * 7 + [C.c_int, C.c_int, c_fp, C.c_size_t, c_stream]),
    "vsom_layernorm_bwd_finish_many": (C.c_int, [c_fp, C.c_int, C.c_int, C.c_int, c_stream]),
    "vsom_attention_fwd": (C.c_int, [c_fp, c_fp, c_fp, C.c_int, C.c_int, C.c_int, C.c_int, c_stream]),
    "vsom_attention_bwd": (C.c_int, [c_fp] * 6 + [C.c_int] * 4 + [c_stream]),
    "vsom_set_attention_fused": (C.c_int, [C.c_int]),
    "vsom_attention_probs": (C.c_int, [c_fp, c_fp, c_fp, C.c_int, C.c_int, C.c_int, C.c_int, c_stream]),
    "vsom_tape_begin": (C.c_int, []),
    "vsom_tape_cut": (C.c_int, []),
    "vsom_tape_pause": (C.c_int, [C.c_int]),
    "vsom_tape_end": (C.c_int, []),
    "vsom_tape_recording": (C.c_int, []),
    "vsom_tape_segment_ops": (C.c_int, [C.c_int, C.c_int]),
    "vsom_tape_replay": (C.c_int, [C.c_int, C.c_int]),
    "vsom_tape_destroy": (C.c_int, [C.c_int]),
    "vsom_event_record": (C.c_int, [C.c_int, c_stream]),
    "vsom_stream_wait_event": (C.c_int, [c_stream, C.c_int]),
    "vsom_comm_unique_id": (C.c_int, [C.c_void_p]),
    "vsom_comm_init": (C.c_int, [C.c_void_p, C.c_int, C.c_int]),
    "vsom_comm_allreduce_sum": (C.c_int, [c_fp, C.c_long, c_stream]),
    "vsom_comm_info": (C.c_int, [C.POINTER(C.c_int), C.POINTER(C.c_int)]),
    "vsom_comm_destroy": (C.c_int, []),
    "vsom_row_inv_norm": (C.c_int, [c_fp, C.c_long, C.c_int, C.c_int, C.c_float, c_fp, c_stream]),
    "vsom_row_sqnorm": (C.c_int, [c_fp, C.c_long, C.c_int, C.c_int, c_fp, c_stream]),
    "vsom_bmu_euclid_fwd": (C.c_int, [c_fp, C.c_long, c_fp, c_fp, c_fp, c_fp, c_fp, C.c_int, C.c_int, C.c_int, c_fp,
                                      C.c_size_t, c_stream]),
    "vsom_bmu_manhattan_workspace_bytes": (C.c_size_t, [C.c_int, C.c_int, C.c_int]),
    "vsom_bmu_manhattan_fwd": (C.c_int, [c_fp, C.c_long, c_fp, c_fp, c_fp, C.c_int, C.c_int, C.c_int, c_fp, C.c_size_t,
                                         c_stream]),
    "vsom_som_bwd_manhattan": (C.c_int, [c_fp, C.c_long, c_fp, c_fp, c_fp, c_fp, C.c_long, C.c_int, C.c_int, C.c_int,
                                         C.c_int, c_stream]),
    "vsom_bmu_cosine_workspace_bytes": (C.c_size_t, [C.c_int, C.c_int, C.c_int]),
    "vsom_bmu_cosine_fwd": (C.c_int, [c_fp, C.c_long, c_fp, c_fp, c_fp, c_fp, c_fp, C.c_int, C.c_int, C.c_int, c_fp,
                                      C.c_size_t, c_stream]),
    "vsom_bmu_cosine_dots": (C.c_int, [c_fp, C.c_long, c_fp, C.c_int, C.c_int, C.c_int, c_fp, C.c_size_t, c_stream]),
    "vsom_bmu_cosine_finalize": (C.c_int, [c_fp, C.c_size_t, c_fp, c_fp, c_fp, c_fp, C.c_int, C.c_int, C.c_int, c_stream]),
    "vsom_bmu_cosine_x3_workspace_bytes": (C.c_size_t, [C.c_int, C.c_int, C.c_int]),
    "vsom_bmu_cosine_x3_dots": (C.c_int, [c_fp, C.c_long, c_fp, C.c_int, C.c_int, C.c_int, c_fp, C.c_size_t, c_stream]),
    "vsom_bmu_cosine_x3_finalize": (C.c_int, [c_fp, C.c_long, c_fp, c_fp, C.c_size_t, c_fp, c_fp, c_fp, c_fp, c_fp, C.c_int,
                                              C.c_int, C.c_int, c_stream]),
    "vsom_bmu_planes_bytes": (C.c_size_t, [C.c_int, C.c_int]),
    "vsom_bmu_planes_from": (C.c_int, [c_fp, C.c_long, C.c_int, C.c_int, c_fp, C.c_size_t, c_stream]),
    "vsom_adamw_step_planes": (C.c_int, [c_fp, c_fp, c_fp, c_fp, c_fp, C.c_long, C.c_float, C.c_float, C.c_float, C.c_float,
                                         C.c_int, C.c_float, C.c_int, C.c_long, C.c_int, C.c_int, c_fp, C.c_size_t, c_stream]),
    "vsom_bmu_cosine_x3_planes_supported": (C.c_int, [C.c_int, C.c_int, C.c_int]),
    "vsom_bmu_cosine_x3_planes_workspace_bytes": (C.c_size_t, [C.c_int, C.c_int, C.c_int]),
    "vsom_bmu_cosine_x3_planes_dots": (C.c_int, [c_fp, c_fp, C.c_int, C.c_int, C.c_int, c_fp, C.c_size_t, c_stream]),
    "vsom_bmu_cosine_x3_planes_finalize": (C.c_int, [c_fp, C.c_long, c_fp, c_fp, c_fp, c_fp, C.c_size_t, c_fp, c_fp, c_fp, c_fp,
                                                     c_fp, C.c_int, C.c_int, C.c_int, c_stream]),
    "vsom_som_neigh_workspace_bytes": (C.c_size_t, [C.c_int, C.c_int]),
    "vsom_som_neigh_loss": (C.c_int, [c_fp, c_fp, c_fp, C.c_float, c_fp, c_fp, C.c_float, c_fp, c_fp, c_fp, c_fp,
                                      c_fp, C.c_int, C.c_int, C.c_int, c_fp, C.c_size_t, c_stream]),
    "vsom_som_bwd": (C.c_int, [c_fp, C.c_long, c_fp, c_fp, c_fp, c_fp, c_fp, c_fp, C.c_long, C.c_int, C.c_int,
                               C.c_int, C.c_int, c_stream]),
    "vsom_l1_unpatchify_workspace_bytes": (C.c_size_t, [C.c_int] * 4),
    "vsom_l1_unpatchify": (C.c_int, [c_fp, c_fp, c_fp, c_fp, c_fp, C.c_float, C.c_int, C.c_int, C.c_int, C.c_int,
                                     c_fp, C.c_size_t, c_stream]),
    "vsom_cross_entropy_ls_workspace_bytes": (C.c_size_t, [C.c_int]),
    "vsom_cross_entropy_ls": (C.c_int, [c_fp, c_fp, C.c_float, c_fp, c_fp, C.c_float, C.c_int, C.c_int, c_fp,
                                        C.c_size_t, c_stream]),
    "vsom_adamw_step": (C.c_int, [c_fp, c_fp, c_fp, c_fp, c_fp, C.c_long, C.c_float, C.c_float, C.c_float, C.c_float,
                                  C.c_int, C.c_float, C.c_int, c_stream]),
    "vsom_contingency": (C.c_int, [c_fp, c_fp, C.c_long, C.c_int, C.c_int, c_fp, c_fp, c_stream]),
    "vsom_argmax_rows": (C.c_int, [c_fp, C.c_long, C.c_int, C.c_int, c_fp, c_stream]),
    "vsom_fill": (C.c_int, [c_fp, C.c_long, C.c_float, c_stream]),
    "vsom_scaled_mul": (C.c_int, [c_fp, c_fp, c_fp, C.c_long, c_fp, C.c_float, c_stream]),
    "vsom_som_weighted_loss": (C.c_int, [c_fp, c_fp, c_fp, c_fp, C.c_float, c_fp, c_fp, c_fp, c_fp, C.c_int, C.c_int, C.c_int,
                                         c_fp, C.c_size_t, c_stream]),
    "vsom_lincomb2": (C.c_int, [c_fp, c_fp, C.c_float, c_fp, C.c_float, C.c_void_p, c_stream]),
    "vsom_loss_parts": (C.c_int, [c_fp, c_fp, C.c_float, c_fp, C.c_float, C.c_float, C.c_void_p, c_stream]),
    "vsom_scale_by": (C.c_int, [c_fp, C.c_long, c_fp, c_stream]),
    "vsom_reduce_slabs": (C.c_int, [c_fp, C.c_long, C.c_int, c_fp, C.c_long, c_stream]),
}


class VsomError(RuntimeError):
    pass


def _load():
    if not os.path.exists(LIB_PATH):
        raise ImportError(
            f"{LIB_PATH} is missing: build it with `python -c 'import __graft_entry__ as g; g.build()'` "
            "(hipcc --offload-arch=gfx950).  vit_som_amd has no CPU/eager fallback.")
    lib = C.CDLL(LIB_PATH)
    for name, (res, args) in SIGNATURES.items():
        fn = getattr(lib, name)          # AttributeError if the .so does not export it
        fn.restype, fn.argtypes = res, args
    return lib


lib = _load()


def last_error() -> str:
    return lib.vsom_last_error_string().decode()


def check(rc: int, what: str):
    if rc != 0:
        raise VsomError(f"{what} failed with status {rc}: {last_error()}")


def ptr(t):
    """Device pointer of a tensor (None -> NULL)."""
    return None if t is None else t.data_ptr()


_raw_stream = getattr(torch._C, "_cuda_getCurrentRawStream", None)
_cur_device = getattr(torch._C, "_cuda_getDevice", None)


_forced_stream = None


class on_stream:
    """Launch the ops inside the block on the given torch stream WITHOUT making it torch's current
    stream (``with torch.cuda.stream(s)`` costs ~10 us of host time per entry; this costs none).
    Only for code that launches library kernels exclusively: torch's own ops still go to torch's
    current stream."""

    def __init__(self, torch_stream):
        self.handle = torch_stream.cuda_stream
        self.torch_stream = torch_stream

    def __enter__(self):
        global _forced_stream, _forced_torch_stream
        self.prev, _forced_stream = _forced_stream, self.handle
        self.prev_t, _forced_torch_stream = _forced_torch_stream, self.torch_stream

    def __exit__(self, *exc):
        global _forced_stream, _forced_torch_stream
        _forced_stream = self.prev
        _forced_torch_stream = self.prev_t


_forced_torch_stream = None


def launch_torch_stream():
    """The torch Stream object of the stream the next launch goes to (for torch.cuda.Event timing of a library kernel: an
    event recorded on torch's current stream would not see a kernel launched under ``on_stream``)."""
    return _forced_torch_stream if _forced_torch_stream is not None else torch.cuda.current_stream()


def stream():
    """Raw hipStream_t the next launch goes to: the stream set by ``on_stream`` if any, else torch's
    current stream on the current device (called once per launch: the private fast accessors cost
    ~0.3 us, the public Stream object ~8 us)."""
    if _forced_stream is not None:
        return _forced_stream
    if _raw_stream is not None and _cur_device is not None:
        return _raw_stream(_cur_device())
    return torch.cuda.current_stream().cuda_stream


class Event:
    """A library-owned HIP event (vsom_event_record / vsom_stream_wait_event): an ordering edge between two of the step's
    streams that a launch tape can hold (torch.cuda.Event records would be invisible to it).  Ids come from a process-wide
    pool of 512; `Event.pooled()` hands out round-robin events for record-then-wait edges that are consumed at once."""
    _next_id = 0
    _pool, _pool_pos = [], 0
    POOL = 192

    def __init__(self):
        if Event._next_id >= 512:
            raise RuntimeError("vit_som_amd: out of library events (512)")
        self.id = Event._next_id
        Event._next_id += 1

    @classmethod
    def pooled(cls):
        if len(cls._pool) < cls.POOL:
            cls._pool.append(cls())
        cls._pool_pos = (cls._pool_pos + 1) % cls.POOL
        return cls._pool[cls._pool_pos % len(cls._pool)]

    def record(self, torch_stream=None):
        """Record on `torch_stream` (default: the stream launches currently go to)."""
        check(lib.vsom_event_record(self.id, torch_stream.cuda_stream if torch_stream is not None else stream()), "vsom_event_record")
        return self

    def wait(self, torch_stream=None):
        """Make `torch_stream` (default: the launch stream) wait for the last record of this event."""
        check(lib.vsom_stream_wait_event(torch_stream.cuda_stream if torch_stream is not None else stream(), self.id),
              "vsom_stream_wait_event")


def stream_wait_stream(waiter, other):
    """`waiter` (a torch stream or None = the launch stream) waits for everything enqueued on `other` so far."""
    ev = Event.pooled().record(other)
    ev.wait(waiter)
