"""ctypes bindings for the checkers used by the tests: the CPU restatement
(oracle/libdrb_oracle.so) and, in the build container only, the upstream
converter compiled from its own source (oracle/_ref/libdrb_ref.so).

Only tests, the smoke check and the benchmark's cpu_baseline leg import this.
"""
import ctypes
import os

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
ORACLE_SO = os.path.join(ROOT, "oracle", "libdrb_oracle.so")
REF_SO = os.path.join(ROOT, "oracle", "_ref", "libdrb_ref.so")

PUSH, PULL = 0, 1
POOR, FINE, GOOD, BEST = 0, 1, 2, 3
RATES = [8000, 11025, 16000, 22050, 44100, 48000, 60000,
         88200, 96000, 120000, 176400, 192000, 240000]
BLOCK_SIZES = [1, 4, 16, 64, 256, 1024, 4096]

c_int_p = ctypes.POINTER(ctypes.c_int)
c_float_p = ctypes.POINTER(ctypes.c_float)


def _fp(a):
    return a.ctypes.data_as(c_float_p)


def _ip(a):
    return a.ctypes.data_as(c_int_p)


class OraclePlan(ctypes.Structure):
    _fields_ = [
        ("source_rate", ctypes.c_int), ("target_rate", ctypes.c_int),
        ("channels", ctypes.c_int), ("block_size", ctypes.c_int),
        ("max_block_count", ctypes.c_int), ("direction", ctypes.c_int),
        ("quality", ctypes.c_int), ("stage_count", ctypes.c_int),
        ("stage_kind", ctypes.c_int * 8), ("stage_arg", ctypes.c_int * 8),
        ("stage_in_rate", ctypes.c_int * 8), ("stage_out_rate", ctypes.c_int * 8),
        ("throttling", ctypes.c_int), ("consistent", ctypes.c_int),
    ]

    def stages(self):
        kinds = {1: "up2x", 2: "down2x", 3: "sinc", 4: "slicer"}
        return [(kinds[self.stage_kind[i]], self.stage_arg[i]) for i in range(self.stage_count)]


_oracle = None


def oracle_lib():
    global _oracle
    if _oracle is None:
        lib = ctypes.CDLL(ORACLE_SO)
        lib.drb_oracle_push_count.restype = ctypes.c_long
        lib.drb_oracle_push_count.argtypes = [ctypes.POINTER(OraclePlan), ctypes.c_long]
        lib.drb_oracle_pull_count.restype = ctypes.c_long
        lib.drb_oracle_pull_count.argtypes = [ctypes.POINTER(OraclePlan), ctypes.c_long]
        lib.drb_oracle_push.restype = ctypes.c_long
        lib.drb_oracle_push.argtypes = [ctypes.POINTER(OraclePlan), c_float_p, ctypes.c_long,
                                        ctypes.c_long, c_float_p, ctypes.c_long, ctypes.c_long]
        lib.drb_oracle_pull.restype = ctypes.c_long
        lib.drb_oracle_pull.argtypes = [ctypes.POINTER(OraclePlan), c_float_p, ctypes.c_long,
                                        ctypes.c_long, c_float_p, ctypes.c_long, ctypes.c_long]
        _oracle = lib
    return _oracle


def oracle_plan(src, dst, ch, bs, mbc, direction, quality):
    """Returns (status, plan); status 1 ok, 0 rejected, -1 unsupported."""
    plan = OraclePlan()
    status = oracle_lib().drb_oracle_plan(src, dst, ch, bs, mbc, direction, quality,
                                          ctypes.byref(plan))
    return status, plan


def oracle_push(plan, x):
    """x: float32 [channels, frames] -> float32 [channels, produced]."""
    lib = oracle_lib()
    x = np.ascontiguousarray(x, np.float32)
    ch, n = x.shape
    produced = lib.drb_oracle_push_count(ctypes.byref(plan), n)
    out = np.zeros((ch, max(produced, 1)), np.float32)
    got = lib.drb_oracle_push(ctypes.byref(plan), _fp(x), n, n, _fp(out), out.shape[1], out.shape[1])
    assert got == produced, (got, produced)
    return out[:, :produced]


def oracle_pull(plan, x, target_frames):
    """x: producer data float32 [channels, frames]; returns (out, consumed)."""
    lib = oracle_lib()
    x = np.ascontiguousarray(x, np.float32)
    ch, n = x.shape
    out = np.zeros((ch, max(target_frames, 1)), np.float32)
    consumed = lib.drb_oracle_pull(ctypes.byref(plan), _fp(x), n, n, _fp(out),
                                   target_frames, out.shape[1])
    assert consumed >= 0
    return out[:, :target_frames], consumed


def oracle_callback_counts(plan, chunks):
    lib = oracle_lib()
    chunks = np.ascontiguousarray(chunks, np.int32)
    counts = np.zeros(1 << 16, np.int32)
    n = lib.drb_oracle_callback_counts(ctypes.byref(plan), _ip(chunks), len(chunks),
                                       _ip(counts), len(counts))
    assert n <= len(counts)
    return counts[:n].copy()


def oracle_convert_samples(values, to):
    """Integer sample conventions of the batched entry point."""
    lib = oracle_lib()
    flat = np.ascontiguousarray(values).reshape(-1)
    fn, dtype = {
        ("int16", "f32"): (lib.drb_oracle_i16_to_f32, np.float32),
        ("int32", "f32"): (lib.drb_oracle_i32_to_f32, np.float32),
        ("float32", "i16"): (lib.drb_oracle_f32_to_i16, np.int16),
        ("float32", "i32"): (lib.drb_oracle_f32_to_i32, np.int32),
    }[(str(flat.dtype), to)]
    out = np.zeros(flat.size, dtype)
    fn(flat.ctypes.data_as(ctypes.c_void_p), out.ctypes.data_as(ctypes.c_void_p),
       ctypes.c_long(flat.size))
    return out.reshape(np.shape(values))


# ---- four-function API harness (works on any implementation) ---------------

class ApiHarness:
    """Wraps a shared object that exports harness_push / harness_pull /
    harness_sizes (oracle/api_harness.c linked to some implementation of the
    four-function converter API)."""

    def __init__(self, path):
        self.lib = ctypes.CDLL(path)
        for name in ("harness_push", "harness_pull"):
            fn = getattr(self.lib, name)
            fn.restype = ctypes.c_long
            fn.argtypes = [ctypes.c_int] * 6 + [c_float_p, ctypes.c_long, c_int_p, ctypes.c_int,
                                                c_float_p, ctypes.c_long, c_int_p, ctypes.c_int,
                                                c_int_p, ctypes.POINTER(ctypes.c_double),
                                                ctypes.c_double]
        self.lib.harness_sizes.restype = ctypes.c_long
        self.lib.harness_sizes.argtypes = [ctypes.c_int] * 7 + [ctypes.POINTER(ctypes.c_long)]

    def sizes(self, src, dst, ch, bs, mbc, direction, quality):
        work = ctypes.c_long(0)
        size = self.lib.harness_sizes(src, dst, ch, bs, mbc, direction, quality, ctypes.byref(work))
        return size, work.value

    def push(self, src, dst, ch, bs, mbc, quality, x, chunks, out_capacity=None, timestamp=0.0):
        x = np.ascontiguousarray(x, np.float32)
        assert x.shape[0] == ch
        n = x.shape[1]
        chunks = np.ascontiguousarray(chunks, np.int32)
        assert chunks.sum() <= n
        cap = out_capacity or int(n * (dst / src) * 1.02) + 8192
        out = np.zeros((ch, cap), np.float32)
        counts = np.zeros(1 << 18, np.int32)
        stamps = np.zeros(1 << 18, np.float64)
        total = ctypes.c_int(0)
        produced = self.lib.harness_push(src, dst, ch, bs, mbc, quality, _fp(x), n, _ip(chunks),
                                         len(chunks), _fp(out), cap, _ip(counts), len(counts),
                                         ctypes.byref(total),
                                         stamps.ctypes.data_as(ctypes.POINTER(ctypes.c_double)), timestamp)
        if produced < 0:
            return None
        assert total.value <= len(counts)
        self.stamps = stamps[:total.value].copy()   # what the callbacks were told the time was
        return out[:, :produced].copy(), counts[:total.value].copy()

    def pull(self, src, dst, ch, bs, mbc, quality, x, chunks, timestamp=0.0):
        x = np.ascontiguousarray(x, np.float32)
        assert x.shape[0] == ch
        n = x.shape[1]
        chunks = np.ascontiguousarray(chunks, np.int32)
        cap = int(chunks.sum())
        out = np.zeros((ch, max(cap, 1)), np.float32)
        counts = np.zeros(1 << 18, np.int32)
        stamps = np.zeros(1 << 18, np.float64)
        total = ctypes.c_int(0)
        consumed = self.lib.harness_pull(src, dst, ch, bs, mbc, quality, _fp(x), n, _ip(chunks),
                                         len(chunks), _fp(out), out.shape[1], _ip(counts),
                                         len(counts), ctypes.byref(total),
                                         stamps.ctypes.data_as(ctypes.POINTER(ctypes.c_double)), timestamp)
        if consumed < 0:
            return None
        self.stamps = stamps[:total.value].copy()
        return out[:, :cap].copy(), counts[:total.value].copy(), consumed


_ref = None


def have_ref():
    return os.path.exists(REF_SO)


def ref_harness():
    global _ref
    if _ref is None:
        _ref = ApiHarness(REF_SO)
    return _ref


PASS_TAGS = ["interleaver", "deinterleaver",
             "up16", "up32", "up48", "up64", "up80", "up96",
             "down16", "down32", "down48", "down64", "down80", "down96",
             "sinc8l", "sinc16l", "sinc24l", "sinc32l",
             "sinc8c", "sinc16c", "sinc24c", "sinc32c", "slicer"]


def ref_plan(src, dst, ch, bs, mbc, direction, quality):
    """Planner known answers straight from the upstream build.  Returns None
    when validation rejects, "unsupported" past the filter-order tables."""
    lib = ref_harness().lib
    passes = (ctypes.c_int * 10)()
    formats = (ctypes.c_int * 55)()
    throttling = ctypes.c_int()
    buffer_length = ctypes.c_long()
    lib.ref_probe_plan.restype = ctypes.c_int
    n = lib.ref_probe_plan(src, dst, ch, bs, mbc, direction, quality, passes, formats,
                           ctypes.byref(throttling), ctypes.byref(buffer_length))
    if n == -2:
        return None
    if n == -1:
        return "unsupported"
    return {
        "passes": [PASS_TAGS[passes[i]] for i in range(n)],
        "formats": [[formats[i * 5 + j] for j in range(5)] for i in range(n + 1)],
        "throttling": throttling.value,
        "buffer_length": buffer_length.value,
    }


# ---- synthetic signals ------------------------------------------------------

def white_noise(channels, frames, seed):
    rng = np.random.default_rng(seed)
    return rng.uniform(-1.0, 1.0, size=(channels, frames)).astype(np.float32)


def log_sweep(channels, frames, rate, top_hz, amplitude=0.5):
    t = np.arange(frames, dtype=np.float64) / rate
    duration = max(frames / rate, 1e-9)
    f0, f1 = 20.0, float(top_hz)
    k = np.log(f1 / f0)
    phase = 2 * np.pi * f0 * duration / k * (np.exp(t / duration * k) - 1.0)
    base = amplitude * np.sin(phase)
    return np.stack([np.roll(base, 37 * c) for c in range(channels)]).astype(np.float32)
