#!/usr/bin/env python3
"""Headline benchmark: output Msamples/s of the batched resampling path.

One step = every stream of the batch pushes `--seconds` of synthetic white
noise through the cascade (state carried from step to step), with the source
and target buffers already resident in HBM.  The default, 10 s, is the stream
length of the workload definition, converted in one call per step.

`python3 bench.py --gpus N` starts N processes itself, one per GPU, before it
touches a GPU (under `python -m torch.distributed.run --nproc-per-node N` the
ranks are there already and nothing more is started).  Streams are independent:
every rank owns its own streams and no data moves between ranks;
torch.distributed (RCCL) serves the barrier and the max-over-ranks of the wall
time only.  `--scaling weak` (default) gives every rank the workload's
streams-per-GPU, `--scaling strong` splits `--total-streams` over the ranks.

Prints ONE JSON line on rank 0.  At one GPU the line also carries `configs`
(the other benchmark configurations and two variants of the headline, each
with its own value, kernel time and roofline), `host_fed` (the PCIe-inclusive
rate) and `cpu_baseline`.
"""
import argparse
import ctypes
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
for path in (os.path.join(ROOT, "audio-resampler_amd"), os.path.join(ROOT, "tests")):
    if path not in sys.path:
        sys.path.insert(0, path)

WORKLOADS = {
    # name: (src, dst, channels, block, max_blocks, direction, quality, sample, layout, streams per GPU)
    "headline": (44100, 48000, 2, 256, 16, 0, 2, "f32", "planar", 1024),
    "headline_poor": (44100, 48000, 2, 256, 16, 0, 0, "f32", "planar", 1024),
    "headline_best": (44100, 48000, 2, 256, 16, 0, 3, "f32", "planar", 1024),
    "cfg2_i16": (48000, 44100, 2, 256, 16, 0, 2, "i16", "interleaved", 1024),
    "cfg2_f32": (48000, 44100, 2, 256, 16, 0, 2, "f32", "planar", 1024),
    "cfg2_i16_planar": (48000, 44100, 2, 256, 16, 0, 2, "i16", "planar", 1024),
    "cfg3_8ch": (96000, 48000, 8, 256, 16, 0, 2, "f32", "planar", 512),
    "cfg4_mono_pull": (16000, 48000, 1, 256, 16, 1, 2, "f32", "planar", 8192),
    "cfg5_i32_best": (192000, 44100, 2, 256, 16, 0, 3, "i32", "interleaved", 256),
}

# what a default run measures beside the headline: (name, workload, options)
SIDE_RUNS = [
    ("cfg2", "cfg2_i16", {}),
    ("cfg3", "cfg3_8ch", {}),
    ("cfg4", "cfg4_mono_pull", {"ragged": (2.0, 6.0)}),
    ("cfg5", "cfg5_i32_best", {}),                       # int32: the exact kernels, whatever --arithmetic says
    ("headline_4096_frame_calls", "headline", {"call_frames": 4096}),
    ("headline_phases_apart", "headline", {"phases_apart": True}),
    ("headline_sweep", "headline", {"signal": "sweep"}),
    ("headline_poor", "headline_poor", {}),
    ("headline_best", "headline_best", {}),
    ("headline_exact", "headline", {"arithmetic": "exact"}),
    ("headline_exact_1s_steps", "headline", {"arithmetic": "exact", "seconds": 1.0}),
]

FP32_PEAK_TFLOPS = 157.3    # MI355X float32, vector and matrix instruction alike
HALF_PEAK_TFLOPS = 2500.0   # MI355X dense half-precision matrix instruction
HBM_PEAK_GBS = 8000.0       # MI355X HBM3E spec


def cpu_baseline(workload, seconds_budget=12.0):
    """The upstream converter (oracle/_ref, compiled from its own source) or,
    where that build is absent, the CPU restatement, one instance per thread,
    4096-frame process() calls, scalar build flags."""
    import concurrent.futures

    import support

    src, dst, ch, bs, mbc, direction, quality = workload[:7]
    threads = max(1, min(int(os.environ.get("OMP_NUM_THREADS", "8")), os.cpu_count() or 1))
    frames = src  # one second per conversion
    x = support.white_noise(ch, frames, 1)
    if support.have_ref():
        kind, harness = "reference", support.ref_harness()

        def convert():
            if direction == 0:
                y, _ = harness.push(src, dst, ch, bs, mbc, quality, x, [4096] * (frames // 4096))
                return y.size
            want = int(frames * dst / src * 0.9)
            y, _, _ = harness.pull(src, dst, ch, bs, mbc, quality, x, [4096] * (want // 4096))
            return y.size
    else:
        kind = "port"
        status, plan = support.oracle_plan(src, dst, ch, bs, mbc, direction, quality)
        assert status == 1

        def convert():
            if direction == 0:
                return support.oracle_push(plan, x).size
            return support.oracle_pull(plan, x, int(frames * dst / src * 0.9))[0].size

    def worker(_):
        done, start = 0, time.perf_counter()
        while time.perf_counter() - start < seconds_budget:
            done += convert()
        return done, time.perf_counter() - start

    convert()  # page in
    with concurrent.futures.ThreadPoolExecutor(threads) as pool:
        results = list(pool.map(worker, range(threads)))
    rate = sum(done / elapsed for done, elapsed in results) / 1e6
    what = ("the upstream converter compiled from its own source" if kind == "reference"
            else "this repository's CPU restatement of the upstream converter (oracle/)")
    return {"value": round(rate, 3), "unit": "Msamples/s", "cores": threads, "kind": kind,
            "sample": "%s, %d threads x repeated 1 s conversions of the headline shape for %.0f s, "
                      "4096-frame process() calls" % (what, threads, seconds_budget)}


def host_fed(batch, source, ch, target_capacity, torch_type, frames, capacity, layout, direction, steps=3):
    """The same step with sources and targets in pinned host memory: copies in,
    conversion and copies out overlap in slices of streams.  A side measurement
    (never `value`): the link, not the kernel, sets this rate."""
    import torch

    streams = source.shape[0]
    # pinned from the start: a pageable copy on the way would take the memory next to the GPU
    host_source = torch.empty(source.shape, dtype=source.dtype, pin_memory=True)
    host_source.copy_(source)
    source_frames = host_source.shape[2] if layout == "planar" else host_source.shape[1]
    host_target = torch.zeros(streams * ch * target_capacity, dtype=torch_type, pin_memory=True)
    produced = (ctypes.c_int * streams)()

    def step():
        batch.process_host_dense(host_source.data_ptr(), ch * source_frames, source_frames,
                                 host_target.data_ptr(), ch * target_capacity, target_capacity,
                                 frames, capacity, produced=produced)
        return (sum(produced) if direction == 0 else frames * streams) * ch

    step()
    start = time.perf_counter()
    samples = sum(step() for _ in range(steps))
    elapsed = time.perf_counter() - start
    moved = (host_source.numel() * host_source.element_size() + host_target.numel() * host_target.element_size()) * steps
    return {"value": round(samples / elapsed / 1e6, 2), "unit": "Msamples/s", "ms_per_step": round(elapsed / steps * 1e3, 3),
            "link_GBps": round(moved / elapsed / 1e9, 2),
            "note": "pinned host buffers, H2D + convert + D2H overlapped in slices (drb_amd_batch_process_host_dense)"}


def roofline_of(info, fast, samples_per_launch, kernel_ms, step_ms):
    """Prices one step's filter kernels against the roofs of what they run on.

    Memory side: the algorithmic bytes (source + target samples) over the kernel
    time against the HBM spec.  Compute side: the flops the kernels hand to the
    hardware, each part against the peak of the instruction it runs on
    (half-precision matrix instruction, float32 matrix instruction, or float32
    vector arithmetic for the exact kernels), as the time those flops need at
    peak over the kernel time.  The bound is whichever fraction is larger."""
    seconds = kernel_ms * 1e-3
    nbytes = info.bytes_per_output_sample * samples_per_launch
    hbm_frac = nbytes / seconds / 1e9 / HBM_PEAK_GBS
    if fast:
        issued = info.fast_issued_flops_per_output_sample
        half = info.fast_issued_half_flops_per_output_sample
        unit_name = "mfma"
    else:
        # the exact kernels round every product and every sum on its own, in
        # upstream's order: two vector instructions per multiply-add of the cascade
        issued, half, unit_name = info.flops_per_output_sample, 0.0, "valu"
    floor = (half / (HALF_PEAK_TFLOPS * 1e12) + (issued - half) / (FP32_PEAK_TFLOPS * 1e12)) * samples_per_launch
    compute_frac = floor / seconds
    out = {}
    if hbm_frac >= compute_frac:
        out.update({"bound": "hbm", "achieved": round(nbytes / seconds / 1e9, 1), "peak": HBM_PEAK_GBS, "unit": "GB/s",
                    "frac": round(hbm_frac, 4)})
    else:
        peak = HALF_PEAK_TFLOPS if half >= issued - half and half > 0 else FP32_PEAK_TFLOPS
        out.update({"bound": unit_name, "achieved": round(compute_frac * peak, 3), "peak": peak, "unit": "TFLOP/s",
                    "frac": round(compute_frac, 4)})
    out.update({"hbm_frac": round(hbm_frac, 4), "compute_frac": round(compute_frac, 4),
                "kernel": info.kernel_name.decode(), "kernel_ms": round(kernel_ms, 4), "device_step_ms": round(step_ms, 4),
                "step_frac": round(out["frac"] * kernel_ms / step_ms, 4) if step_ms > 0 else None,
                "algorithmic_bytes_per_sample": round(info.bytes_per_output_sample, 3),
                "issued_flops_per_sample": round(issued, 1),
                "issued_half_precision_flops_per_sample": round(half, 1),
                "upstream_cascade_flops_per_sample": round(info.flops_per_output_sample, 1),
                # the rate the upstream cascade's own flop count would stand for (not a rate the hardware ran)
                "upstream_cascade_equivalent_tflops": round(info.flops_per_output_sample * samples_per_launch / seconds / 1e12, 1)})
    return out


def synthetic_source(torch, signal, streams, ch, frames, rate, limit, device, seed):
    """[streams, ch, frames] float32 on the device.  noise: uniform white noise in
    [-1, 1); sweep: a logarithmic sine sweep from 20 Hz to `limit` Hz over the
    buffer, amplitude 0.5, every stream and channel starting at a phase of its own."""
    generator = torch.Generator(device=device)
    generator.manual_seed(seed)
    if signal == "noise":
        return torch.rand((streams, ch, frames), device=device, generator=generator) * 2 - 1
    import math
    t = torch.arange(frames, device=device, dtype=torch.float64) / rate
    span = frames / rate
    k = math.log(limit / 20.0)
    phase = 2 * math.pi * 20.0 * span / k * (torch.exp(t / span * k) - 1.0)
    start = torch.rand((streams, ch, 1), device=device, generator=generator, dtype=torch.float64) * 2 * math.pi
    out = torch.empty((streams, ch, frames), device=device, dtype=torch.float32)
    for first in range(0, streams, 64):   # (float64 temporaries of a few hundred MB at a time)
        out[first:first + 64] = (0.5 * torch.sin(phase + start[first:first + 64])).float()
    return out


def measure(name, arithmetic, streams, seconds, steps, warmup, rank, world, local_rank,
            call_frames=0, ragged=None, phases_apart=False, want_host_fed=False, kernel_probe=6, host_fed_seconds=1.0,
            signal="noise"):
    """Runs one workload on this rank's GPU and returns its measurements.  The
    timed region is bracketed by barriers over all ranks and device
    synchronisation on both sides; the time is the slowest rank's."""
    import numpy as np
    import torch

    import drbamd
    from drbamd import sharding

    device = torch.device("cuda", local_rank)
    src, dst, ch, bs, mbc, direction, quality, sample, layout, _ = WORKLOADS[name]
    sample_type = {"f32": drbamd.F32, "i16": drbamd.I16, "i32": drbamd.I32}[sample]
    torch_type = {"f32": torch.float32, "i16": torch.int16, "i32": torch.int32}[sample]
    size = 2 if sample == "i16" else 4
    config = drbamd.make_config(src, dst, ch, bs, mbc, direction, quality, sample_type=sample_type,
                                layout=drbamd.INTERLEAVED if layout == "interleaved" else drbamd.PLANAR,
                                arithmetic=drbamd.EXACT if arithmetic == "exact" else drbamd.FAST,
                                streams=streams, device=local_rank)
    info = drbamd.plan_info(config)
    fast = arithmetic == "fast" and bool(info.fast_path)
    batch = drbamd.Batch(config)

    # user-side frames per stream per step: push = source frames, pull = target frames
    user_rate = dst if direction == 1 else src
    whole = int(round(seconds * user_rate / bs)) * bs
    rng = np.random.default_rng(99 + rank)
    if ragged:
        lengths = [int(rng.uniform(ragged[0], ragged[1]) * user_rate) // bs * bs for _ in range(streams)]
    else:
        lengths = [whole] * streams
    longest = max(lengths)
    per_call = call_frames if call_frames else longest
    calls_per_step = (longest + per_call - 1) // per_call
    lead = 0
    if phases_apart:
        # every stream is pushed an odd number of frames of its own before anything is timed
        lead = 10001
    other_whole = batch.query([longest] * streams)[0]
    source_frames = (other_whole + 8192 if direction == 1 else longest + lead) + int(os.environ.get("BENCH_SOURCE_PAD", "0"))
    target_capacity = longest + lead if direction == 1 else other_whole + 8192 + lead

    noise = synthetic_source(torch, signal, streams, ch, source_frames, src, 0.45 * min(src, dst), device, 1234 + rank)
    if layout == "interleaved":
        noise = noise.transpose(1, 2).contiguous()
    if sample == "i16":
        source = (noise * 32767).round().to(torch.int16)
    elif sample == "i32":
        source = (noise.double() * 2147483647).round().to(torch.int32)
    else:
        source = noise
    del noise
    target = torch.zeros(streams * ch * target_capacity, dtype=torch_type, device=device)
    stream = torch.cuda.current_stream().cuda_stream
    produced = (ctypes.c_int * streams)()
    stream_bytes = ch * source_frames * size
    target_bytes = ch * target_capacity * size
    frame_bytes = size * (ch if layout == "interleaved" else 1)

    if phases_apart:
        first = [int(rng.integers(0, 5000)) * 2 + 1 for _ in range(streams)]
        batch.process([source.data_ptr() + s * stream_bytes for s in range(streams)], source_frames, first,
                      [target.data_ptr() + s * target_bytes for s in range(streams)], target_capacity,
                      [target_capacity] * streams, stream=stream)

    uniform = not ragged and not call_frames
    assert not (ragged and call_frames)

    if ragged:
        # whole clips of different lengths, one call per step: the per-stream tables are made once
        void_n, int_n = ctypes.c_void_p * streams, ctypes.c_int * streams
        table_source = void_n(*[source.data_ptr() + s * stream_bytes for s in range(streams)])
        table_target = void_n(*[target.data_ptr() + s * target_bytes for s in range(streams)])
        table_lengths = int_n(*lengths)
        table_room = int_n(*([source_frames if direction == 1 else target_capacity] * streams))

    def ragged_call():
        if direction == 0:
            batch.process_arrays(table_source, source_frames, table_lengths, table_target, target_capacity, table_room,
                                 produced, stream=stream)
            return sum(produced)
        batch.reset()   # a pull stream keeps its place in the producer's data: every step converts fresh clips
        batch.process_arrays(table_source, source_frames, table_room, table_target, target_capacity, table_lengths,
                             produced, stream=stream)
        return sum(lengths)

    def dense_call(done, moved, now):
        """Every stream converts `now` frames from where it stands (`done` user-side
        frames in, `moved` frames on the other side)."""
        at_source, at_target = (moved, done) if direction == 1 else (done, moved)
        batch.process_dense(source.data_ptr() + at_source * frame_bytes, ch * source_frames, source_frames,
                            target.data_ptr() + at_target * frame_bytes, ch * target_capacity, target_capacity,
                            now, (source_frames - at_source) if direction == 1 else (target_capacity - at_target),
                            stream=stream, produced=produced)
        return produced[0]

    def step():
        """One step: every stream converts its whole length (in calls of
        `per_call` frames where asked for); returns the target frames made."""
        if ragged:
            return ragged_call()
        done = moved = made = 0
        while done < longest:
            now = min(per_call, longest - done)
            count = dense_call(done, moved, now)
            made += (count if direction == 0 else now) * streams
            done += now
            moved += count
        if direction == 1:
            batch.reset()
        return made

    def barrier():
        torch.cuda.synchronize()
        if world > 1:
            import torch.distributed as dist
            dist.barrier()
        torch.cuda.synchronize()

    # The chip's clocks take some 25 ms of work to settle after the idle set-up above
    # (the first launches of a run take 2.4 instead of 1.65 ms).  Not steps: nothing of
    # the converter runs here, the target buffer is cleared a few times.
    settle = time.perf_counter()
    while time.perf_counter() - settle < 0.06:
        target.zero_()
        torch.cuda.synchronize()
    for _ in range(warmup):
        step()
    barrier()
    start = time.perf_counter()
    out_frames = 0
    for _ in range(steps):
        out_frames += step()
    barrier()
    elapsed = time.perf_counter() - start
    batch.synchronize(stream)

    elapsed, all_frames = sharding.reduce_step_stats(elapsed, out_frames, device=device)
    result = {"value": round(all_frames * ch / elapsed / 1e6, 2), "unit": "Msamples/s",
              "ms_per_step": round(elapsed / steps * 1e3, 4)}

    # what the host spends on a step (frame arithmetic, descriptors, launches), with
    # the batch's ring of four step slots empty so that no call waits for the device
    torch.cuda.synchronize()
    start = time.perf_counter()
    if uniform:
        for _ in range(3):
            step()
        result["host_ms_per_step"] = round((time.perf_counter() - start) / 3 * 1e3, 4)
    else:
        drbamd.host_times(start_over=True)
        step()
        result["host_ms_per_step"] = round((time.perf_counter() - start) * 1e3, 4)   # includes waits for the device
        spent = drbamd.host_times()
        if spent:
            # what the library's calls themselves took on the host, and what of it was waiting for the device
            result["host_work_ms_per_step"] = round(sum(spent[1][:4]) * 1e-3, 4)
            result["host_wait_ms_per_step"] = round(spent[1][4] * 1e-3, 4)
    torch.cuda.synchronize()

    # the filter kernels and the whole device side of a step: HIP events on the launch stream
    batch.time_kernels(True)
    kernel_ms, step_ms, samples = [], [], []
    if uniform:
        for _ in range(kernel_probe):
            n = step()
            kernel_ms.append(batch.kernel_ms())
            step_ms.append(batch.step_ms())
            samples.append(n * ch)
    else:
        # one call of the step's many: the first one
        for _ in range(kernel_probe):
            if ragged:
                n = ragged_call()
            else:
                if direction == 1:
                    batch.reset()
                count = dense_call(0, 0, min(per_call, longest))
                n = (count if direction == 0 else min(per_call, longest)) * streams
            kernel_ms.append(batch.kernel_ms())
            step_ms.append(batch.step_ms())
            samples.append(n * ch)
    batch.time_kernels(False)
    kernel_ms, step_ms, samples = kernel_ms[1:], step_ms[1:], samples[1:]   # the first one warms the timing events up
    mean_kernel = sum(kernel_ms) / len(kernel_ms)
    mean_step = sum(step_ms) / len(step_ms)
    mean_samples = sum(samples) / len(samples)
    result["roofline"] = roofline_of(info, fast, mean_samples, mean_kernel, mean_step)
    if fast and info.fast_collapsed and mean_samples / (streams * ch) <= 2 * info.fast_tile_frames:
        # steps of a tile or two per stream: the split kernel with workgroups that go from stream
        # to stream where the streams run in step, else the staged kernels (drb_batch.c)
        walkers = batch.step_walkers()
        result["roofline"]["kernel"] = ("drb_split_kernel (%d workgroups going from stream to stream)" % walkers if walkers
                                        else "drb_fast_kernel (steps of at most two tiles per stream)")
    result["dtype"] = ("f32 products from pairs of f16 halves" if fast and info.fast_issued_half_flops_per_output_sample > 0
                       else "f32")
    result["config"] = {
        "workload": "%d x %d-ch %s %d->%d Hz per GPU, quality %s, %s, %s layout, %s, %s arithmetic%s"
                    % (streams, ch, sample, src, dst, ["poor", "fine", "good", "best"][quality],
                       "pull" if direction else "push", layout,
                       ("clips of %.1f to %.1f s" % ragged) if ragged else "%.2f s per stream per step" % seconds,
                       arithmetic if fast or arithmetic == "exact" else "exact (int32 samples: fast arithmetic runs the exact kernels)",
                       (", in calls of %d frames" % call_frames if call_frames else "")
                       + (", every stream pushed an odd number of frames of its own beforehand" if phases_apart else "")),
        "signal": "uniform white noise in [-1, 1)" if signal == "noise"
                  else "logarithmic sine sweep, 20 Hz to 0.45 x the lower rate, amplitude 0.5, a start phase per stream and channel",
        "name": name, "streams_per_gpu": streams, "frames_per_step": longest, "calls_per_step": calls_per_step,
        "tile_frames": info.fast_tile_frames if fast else info.tile_frames, "history_frames": info.history_frames}

    if want_host_fed:
        # one second per stream per call, whatever the step above was (1.5 GB of pinned memory instead of 15)
        fed = int(round(min(seconds, host_fed_seconds) * user_rate / bs)) * bs
        fed_other = batch.query([fed] * streams)[0]
        fed_source = fed_other + 4096 if direction == 1 else fed
        fed_target = fed if direction == 1 else fed_other + 4096
        fed_slice = source[:, :fed_source, :] if layout == "interleaved" else source[:, :, :fed_source]
        result["host_fed"] = host_fed(batch, fed_slice.contiguous(), ch, fed_target, torch_type, fed,
                                      fed_source if direction == 1 else fed_target, layout, direction)
        result["host_fed"]["seconds_per_step"] = min(seconds, host_fed_seconds)

    batch.close()
    del source, target
    torch.cuda.empty_cache()
    return result


def traffic_of(name, arithmetic, streams, seconds):
    """HBM bytes per launch cannot be counted from inside this process: they come
    from the committed rocprofv3 --pmc passes of this same command, if any."""
    try:
        with open(os.path.join(ROOT, "profiles", "traffic.json")) as handle:
            entry = json.load(handle).get("%s/%s/%d/%.1f" % (name, arithmetic, streams, seconds))
        return entry["bytes_per_launch"] if entry else None
    except OSError:
        return None


def main():
    parser = argparse.ArgumentParser()
    parser.add_argument("--gpus", type=int, default=1)
    parser.add_argument("--steps", type=int, default=50)
    parser.add_argument("--warmup", type=int, default=20,
                        help="untimed steps first (the clocks take some 15 launches to settle after an idle spell)")
    parser.add_argument("--workload", default="headline", choices=sorted(WORKLOADS))
    parser.add_argument("--streams", type=int, default=0, help="streams per GPU (0 = workload default)")
    parser.add_argument("--scaling", default="weak", choices=["weak", "strong"])
    parser.add_argument("--total-streams", type=int, default=0,
                        help="strong scaling: streams of the whole job (0 = 8 x the workload's streams per GPU)")
    parser.add_argument("--seconds", type=float, default=10.0,
                        help="audio seconds per stream per step (10 s = a whole stream of the survey's workload in one call)")
    parser.add_argument("--call-frames", type=int, default=0, help="convert every step in calls of this many frames")
    parser.add_argument("--phases-apart", action="store_true", help="push every stream an odd number of frames of its own first")
    parser.add_argument("--arithmetic", default="fast", choices=["exact", "fast"])
    parser.add_argument("--no-cpu-baseline", action="store_true")
    parser.add_argument("--no-host-fed", action="store_true", help="skip the PCIe-inclusive side measurement")
    parser.add_argument("--host-fed-seconds", type=float, default=1.0,
                        help="seconds per stream per call of the PCIe-inclusive side measurement (pinned memory grows with it)")
    parser.add_argument("--no-configs", action="store_true", help="skip the other configurations and headline variants")
    args = parser.parse_args()

    from drbamd import sharding

    if "WORLD_SIZE" not in os.environ and args.gpus > 1:
        # outside a launcher: one child per GPU, started before this process makes any GPU call
        sys.exit(sharding.spawn_ranks(args.gpus, __file__, sys.argv[1:]))

    import torch

    rank, local_rank, world = sharding.rank_of_environment()
    assert torch.cuda.is_available(), "bench.py needs a GPU"
    torch.cuda.set_device(local_rank)
    device = torch.device("cuda", local_rank)

    if world > 1:
        import torch.distributed as dist
        dist.init_process_group("nccl", device_id=device)

    per_gpu = args.streams or WORKLOADS[args.workload][9]
    streams, _, total = sharding.streams_of_rank(args.scaling, per_gpu, args.total_streams, rank, world)

    single = world == 1
    main_run = measure(args.workload, args.arithmetic, streams, args.seconds, args.steps, args.warmup, rank, world, local_rank,
                       call_frames=args.call_frames, phases_apart=args.phases_apart,
                       want_host_fed=single and not args.no_host_fed, host_fed_seconds=args.host_fed_seconds)

    if rank == 0:
        src, dst = WORKLOADS[args.workload][:2]
        roofline = main_run["roofline"]
        # (HBM bytes per launch from the committed rocprofv3 --pmc passes of this command,
        # profiles/: counters cannot be read from inside the timed process; null where none is kept)
        roofline["traffic"] = traffic_of(args.workload, args.arithmetic, streams, args.seconds)
        roofline["traffic_source"] = "profiles/traffic.json: TCC_EA0_RDREQ / WRREQ passes of this command, committed with their listing"

        line = {
            "metric": "output Msamples/s, %d stereo 44.1->48k streams" % per_gpu if args.workload == "headline"
                      else "output Msamples/s, %s" % args.workload,
            "value": main_run["value"], "unit": "Msamples/s", "n_gpus": world,
            "steps": args.steps, "warmup": args.warmup,
            "ms_per_step": main_run["ms_per_step"], "host_ms_per_step": main_run["host_ms_per_step"],
            "higher_is_better": True, "scaling": args.scaling, "vs_baseline": None,
            "dtype": main_run["dtype"], "data": "synthetic white noise, uniform [-1, 1) (configs[headline_sweep]: the same on a logarithmic sine sweep)",
            "config": dict(main_run["config"], total_streams=total),
            "roofline": roofline,
        }
        if "host_fed" in main_run:
            line["host_fed"] = main_run["host_fed"]
        for key in ("host_work_ms_per_step", "host_wait_ms_per_step"):
            if key in main_run:
                line[key] = main_run[key]

    if single and not args.no_configs and args.workload == "headline" and not args.call_frames and not args.phases_apart:
        configs = []
        for label, name, options in SIDE_RUNS:
            options = dict(options)
            side_steps = 3 if options.get("call_frames") else 10
            side_arithmetic = options.pop("arithmetic", args.arithmetic)
            side_seconds = options.pop("seconds", 10.0)
            try:
                run = measure(name, side_arithmetic, WORKLOADS[name][9], side_seconds, side_steps, 1 if options.get("call_frames") else 4, rank, world, local_rank, **options)
            except Exception as error:  # a side measurement never takes the headline down with it
                configs.append({"name": label, "error": "%s: %s" % (type(error).__name__, error)})
                continue
            run["roofline"]["traffic"] = traffic_of(label, side_arithmetic, WORKLOADS[name][9], side_seconds)
            configs.append(dict(run, name=label))
        line["configs"] = configs

    if rank == 0:
        if single and not args.no_cpu_baseline:
            line["cpu_baseline"] = cpu_baseline(WORKLOADS[args.workload])
        print(json.dumps(line), flush=True)

    if world > 1:
        import torch.distributed as dist
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
