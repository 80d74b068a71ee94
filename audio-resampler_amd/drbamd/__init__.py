"""ctypes binding of libdrb_audio_converter_amd.so for the tests and the
benchmark.  It is plumbing, not the product: every call goes straight through
the C ABI declared in include/drb_audio_converter_amd.h.  There is no Python
or CPU implementation behind it - if the shared library is missing, importing
this module fails.
"""
import ctypes
import os

HERE = os.path.dirname(os.path.abspath(__file__))
LIB_DIR = os.path.join(os.path.dirname(HERE), "lib")
LIB_PATH = os.environ.get("DRB_AMD_LIB") or os.path.join(LIB_DIR, "libdrb_audio_converter_amd.so")
HARNESS_PATH = os.path.join(LIB_DIR, "libdrb_api_harness.so")

PUSH, PULL = 0, 1
POOR, FINE, GOOD, BEST = 0, 1, 2, 3
F32, I16, I32 = 0, 1, 2
PLANAR, INTERLEAVED = 0, 1
EXACT, FAST = 0, 1

STATUS = ["ok", "invalid_conversion", "unsupported", "no_device", "out_of_memory",
          "bad_argument", "capacity", "device"]

if not os.path.exists(LIB_PATH):
    raise ImportError("%s is missing: build it with `make -C audio-resampler_amd/csrc` "
                      "(or __graft_entry__.build())" % LIB_PATH)

# One HIP runtime per process: torch wheels bundle their own libamdhip64 (same
# SONAME as the system one).  Loading torch first makes this library bind to
# that copy instead of bringing a second runtime, which would see no device.
try:
    import torch  # noqa: F401
except ImportError:  # plain C users link the system runtime
    pass

lib = ctypes.CDLL(LIB_PATH)


class BatchConfig(ctypes.Structure):
    _fields_ = [(name, ctypes.c_int) for name in (
        "source_sampling_rate", "target_sampling_rate", "channel_count", "block_size",
        "max_block_count", "direction", "quality", "sample_type", "layout", "arithmetic",
        "stream_count", "device")]


class PlanInfo(ctypes.Structure):
    _fields_ = [("stage_count", ctypes.c_int), ("stage_kind", ctypes.c_int * 8),
                ("stage_arg", ctypes.c_int * 8), ("throttling", ctypes.c_int),
                ("buffer_length", ctypes.c_long), ("converter_size", ctypes.c_long),
                ("history_frames", ctypes.c_int), ("tile_frames", ctypes.c_int),
                ("flops_per_output_sample", ctypes.c_double),
                ("bytes_per_output_sample", ctypes.c_double),
                ("fast_path", ctypes.c_int), ("fast_tile_frames", ctypes.c_int), ("fast_launches", ctypes.c_int),
                ("fast_issued_flops_per_output_sample", ctypes.c_double), ("fast_collapsed", ctypes.c_int),
                ("fast_issued_half_flops_per_output_sample", ctypes.c_double), ("kernel_name", ctypes.c_char * 48),
                ("fast_why_not", ctypes.c_char * 96)]

    def stages(self):
        kinds = {1: "up2x", 2: "down2x", 3: "sinc", 4: "slicer"}
        return [(kinds[self.stage_kind[i]], self.stage_arg[i]) for i in range(self.stage_count)]


c_int_p = ctypes.POINTER(ctypes.c_int)
c_void_pp = ctypes.POINTER(ctypes.c_void_p)

lib.drb_amd_last_error.restype = ctypes.c_char_p
lib.drb_amd_batch_create.argtypes = [ctypes.POINTER(BatchConfig), ctypes.POINTER(ctypes.c_void_p)]
lib.drb_amd_batch_destroy.argtypes = [ctypes.c_void_p]
lib.drb_amd_batch_destroy.restype = None
lib.drb_amd_batch_reset.argtypes = [ctypes.c_void_p, ctypes.c_int, ctypes.c_int]
lib.drb_amd_batch_query.argtypes = [ctypes.c_void_p, c_int_p, c_int_p]
lib.drb_amd_batch_process.argtypes = [ctypes.c_void_p, c_void_pp, ctypes.c_longlong, c_int_p,
                                      c_void_pp, ctypes.c_longlong, c_int_p, c_int_p,
                                      ctypes.c_void_p]
lib.drb_amd_batch_process_dense.argtypes = [ctypes.c_void_p,
                                            ctypes.c_void_p, ctypes.c_longlong, ctypes.c_longlong,
                                            ctypes.c_void_p, ctypes.c_longlong, ctypes.c_longlong,
                                            ctypes.c_int, ctypes.c_int, c_int_p, ctypes.c_void_p]
lib.drb_amd_batch_process_dense_static.argtypes = lib.drb_amd_batch_process_dense.argtypes
lib.drb_amd_batch_process_host_dense.argtypes = [ctypes.c_void_p,
                                                 ctypes.c_void_p, ctypes.c_longlong, ctypes.c_longlong,
                                                 ctypes.c_void_p, ctypes.c_longlong, ctypes.c_longlong,
                                                 ctypes.c_int, ctypes.c_int, c_int_p]
lib.drb_amd_batch_synchronize.argtypes = [ctypes.c_void_p, ctypes.c_void_p]
lib.drb_amd_batch_time_kernels.argtypes = [ctypes.c_void_p, ctypes.c_int]
lib.drb_amd_batch_kernel_ms.argtypes = [ctypes.c_void_p, ctypes.POINTER(ctypes.c_float)]
lib.drb_amd_batch_step_ms.argtypes = [ctypes.c_void_p, ctypes.POINTER(ctypes.c_float)]
# (diagnostics that a build of the library named by DRB_AMD_LIB for comparison may lack)
if hasattr(lib, "drb_amd_batch_step_groups"):
    lib.drb_amd_batch_step_groups.argtypes = [ctypes.c_void_p]
if hasattr(lib, "drb_amd_batch_step_walkers"):
    lib.drb_amd_batch_step_walkers.argtypes = [ctypes.c_void_p]
if hasattr(lib, "drb_amd_host_times"):
    lib.drb_amd_host_times.argtypes = [ctypes.POINTER(ctypes.c_double), ctypes.c_int]
    lib.drb_amd_host_times.restype = ctypes.c_long
if hasattr(lib, "drb_amd_allocation_calls"):
    lib.drb_amd_allocation_calls.restype = ctypes.c_longlong
lib.drb_amd_plan_info.argtypes = [ctypes.POINTER(BatchConfig), ctypes.POINTER(PlanInfo)]
lib.drb_audio_converter_alignment_and_size.restype = ctypes.c_bool
lib.drb_audio_converter_alignment_and_size.argtypes = [ctypes.c_int] * 7 + [
    ctypes.POINTER(ctypes.c_long), ctypes.POINTER(ctypes.c_long)]


class DrbError(RuntimeError):
    def __init__(self, status):
        self.status = STATUS[status] if 0 <= status < len(STATUS) else str(status)
        super().__init__("%s: %s" % (self.status, lib.drb_amd_last_error().decode()))


def check(status):
    if status != 0:
        raise DrbError(status)


def make_config(src, dst, channels, block_size, max_block_count, direction, quality,
                sample_type=F32, layout=PLANAR, arithmetic=EXACT, streams=1, device=-1):
    return BatchConfig(src, dst, channels, block_size, max_block_count, direction, quality,
                       sample_type, layout, arithmetic, streams, device)


def plan_info(config):
    info = PlanInfo()
    check(lib.drb_amd_plan_info(ctypes.byref(config), ctypes.byref(info)))
    return info


def converter_size(src, dst, channels, block_size, max_block_count, direction, quality):
    """The four-function API's size query; None when the request is rejected."""
    alignment, size = ctypes.c_long(), ctypes.c_long()
    ok = lib.drb_audio_converter_alignment_and_size(src, dst, channels, block_size, max_block_count,
                                                    direction, quality, ctypes.byref(alignment),
                                                    ctypes.byref(size))
    return (alignment.value, size.value) if ok else None


def host_times(start_over=False):
    """(steps, [microseconds of frame arithmetic, step preparation, jobs and descriptors, launches, waiting for a step slot])
    spent inside the process calls so far; None where the library has no such call."""
    if not hasattr(lib, "drb_amd_host_times"):
        return None
    spent = (ctypes.c_double * 5)()
    steps = lib.drb_amd_host_times(spent, 1 if start_over else 0)
    return steps, list(spent)


def plan_describe(config):
    """(passes, edges, throttling, buffer_length) in upstream's planning order, or None."""
    passes = (ctypes.c_int * 20)()
    edges = (ctypes.c_int * 55)()
    throttling, buffer_length = ctypes.c_int(), ctypes.c_long()
    n = lib.drb_amd_plan_describe(ctypes.byref(config), passes, edges, ctypes.byref(throttling),
                                  ctypes.byref(buffer_length))
    if n < 0:
        return None
    names = ["interleaver", "deinterleaver", "up", "down", "sinc", "slicer"]
    out = []
    for i in range(n):
        kind, arg = passes[2 * i], passes[2 * i + 1]
        if kind in (2, 3):
            out.append("%s%d" % (names[kind], arg))
        elif kind == 4:
            out.append("sinc%d%s" % (arg % 100, "c" if arg >= 100 else "l"))
        else:
            out.append(names[kind])
    rows = [[edges[5 * i + j] for j in range(5)] for i in range(n + 1)]
    return out, rows, throttling.value, buffer_length.value


class Batch:
    """A batch of independent streams on one GPU.  Buffers are device memory,
    passed as raw addresses (e.g. torch.Tensor.data_ptr())."""

    def __init__(self, config):
        self.config = config
        self.handle = ctypes.c_void_p()
        check(lib.drb_amd_batch_create(ctypes.byref(config), ctypes.byref(self.handle)))
        self.streams = config.stream_count

    def close(self):
        if self.handle:
            lib.drb_amd_batch_destroy(self.handle)
            self.handle = ctypes.c_void_p()

    def __del__(self):
        self.close()

    def reset(self, first=0, count=None):
        check(lib.drb_amd_batch_reset(self.handle, first, self.streams - first if count is None else count))

    def query(self, frames):
        frames = (ctypes.c_int * self.streams)(*frames)
        counts = (ctypes.c_int * self.streams)()
        check(lib.drb_amd_batch_query(self.handle, frames, counts))
        return list(counts)

    def process(self, source_ptrs, source_channel_stride, source_frames,
                target_ptrs, target_channel_stride, target_frames, stream=None):
        n = self.streams
        src = (ctypes.c_void_p * n)(*source_ptrs)
        dst = (ctypes.c_void_p * n)(*target_ptrs)
        sf = (ctypes.c_int * n)(*source_frames)
        tf = (ctypes.c_int * n)(*target_frames)
        produced = (ctypes.c_int * n)()
        check(lib.drb_amd_batch_process(self.handle, src, source_channel_stride, sf, dst,
                                        target_channel_stride, tf, produced,
                                        ctypes.c_void_p(stream or 0)))
        return list(produced)

    def process_arrays(self, src, source_channel_stride, sf, dst, target_channel_stride, tf, produced, stream=None):
        """drb_amd_batch_process on ctypes arrays the caller keeps ((c_void_p * n), (c_int * n)):
        nothing is built per call."""
        check(lib.drb_amd_batch_process(self.handle, src, source_channel_stride, sf, dst,
                                        target_channel_stride, tf, produced,
                                        ctypes.c_void_p(stream or 0)))

    def process_dense(self, source_ptr, source_stream_stride, source_channel_stride,
                      target_ptr, target_stream_stride, target_channel_stride,
                      frames, capacity, stream=None, produced=None):
        check(lib.drb_amd_batch_process_dense(self.handle, ctypes.c_void_p(source_ptr),
                                              source_stream_stride, source_channel_stride,
                                              ctypes.c_void_p(target_ptr), target_stream_stride,
                                              target_channel_stride, frames, capacity, produced,
                                              ctypes.c_void_p(stream or 0)))

    def process_dense_static(self, source_ptr, source_stream_stride, source_channel_stride,
                             target_ptr, target_stream_stride, target_channel_stride,
                             frames, capacity, stream=None, produced=None):
        """process_dense without any allocation (the library's static device workspace)."""
        check(lib.drb_amd_batch_process_dense_static(self.handle, ctypes.c_void_p(source_ptr),
                                                     source_stream_stride, source_channel_stride,
                                                     ctypes.c_void_p(target_ptr), target_stream_stride,
                                                     target_channel_stride, frames, capacity, produced,
                                                     ctypes.c_void_p(stream or 0)))

    def process_host_dense(self, source_ptr, source_stream_stride, source_channel_stride,
                           target_ptr, target_stream_stride, target_channel_stride,
                           frames, capacity, produced=None):
        """Buffers in host memory (pinned for speed); returns when the targets are filled."""
        check(lib.drb_amd_batch_process_host_dense(self.handle, ctypes.c_void_p(source_ptr),
                                                   source_stream_stride, source_channel_stride,
                                                   ctypes.c_void_p(target_ptr), target_stream_stride,
                                                   target_channel_stride, frames, capacity, produced))

    def time_kernels(self, enable=True):
        check(lib.drb_amd_batch_time_kernels(self.handle, 1 if enable else 0))

    def kernel_ms(self):
        ms = ctypes.c_float()
        check(lib.drb_amd_batch_kernel_ms(self.handle, ctypes.byref(ms)))
        return ms.value

    def step_ms(self):
        ms = ctypes.c_float()
        check(lib.drb_amd_batch_step_ms(self.handle, ctypes.byref(ms)))
        return ms.value

    def step_groups(self):
        return lib.drb_amd_batch_step_groups(self.handle)

    def step_walkers(self):
        return lib.drb_amd_batch_step_walkers(self.handle)

    def synchronize(self, stream=None):
        check(lib.drb_amd_batch_synchronize(self.handle, ctypes.c_void_p(stream or 0)))
