#!/usr/bin/env python3
"""Headline benchmark: output Msamples/s of the batched resampling path.

One step = every stream of the batch pushes `--seconds` of synthetic white
noise through the cascade (state carried from step to step), with the source
and target buffers already resident in HBM.  The default, 10 s, is the stream
length of the workload definition, converted in one call per step; with
`--seconds 1` the per-step costs (coefficient sets, descriptor uploads, the
first and last tiles of every stream) weigh ten times as much.  N > 1 runs one process per GPU
(torch.distributed over RCCL is used for the barrier and the max-over-ranks
only; streams are independent, there is no collective on the data path), each
rank owning its own `--streams` streams: weak scaling.

Prints ONE JSON line on rank 0.
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
    # name: (src, dst, channels, block, max_blocks, direction, quality, sample, layout, streams)
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

FP32_PEAK_TFLOPS = 157.3   # MI355X fp32 vector = fp32 MFMA peak
HBM_PEAK_GBS = 8000.0      # MI355X HBM3E spec


def cpu_baseline(workload, seconds_budget=12.0):
    """The upstream converter (oracle/_ref, compiled from its own source) or,
    where that build is absent, the CPU restatement, one instance per thread,
    4096-frame process() calls, scalar build flags."""
    import concurrent.futures

    import numpy as np

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
    return {"value": round(rate, 3), "unit": "Msamples/s", "cores": threads, "kind": kind,
            "sample": "%d threads x repeated 1 s conversions of the same workload shape for %.0f s, "
                      "4096-frame process() calls" % (threads, seconds_budget)}


def host_fed(batch, source, ch, source_frames, target_capacity, torch_type, frames, capacity, layout, direction, steps=3):
    """The same step with sources and targets in pinned host memory: copies in,
    conversion and copies out overlap in slices of streams.  A side measurement
    (never `value`): the link, not the kernel, sets this rate."""
    import torch

    streams = source.shape[0]
    host_source = source.cpu().pin_memory()
    source_frames = host_source.shape[2] if layout == "planar" else host_source.shape[1]
    host_target = torch.zeros(streams * ch * target_capacity, dtype=torch_type).pin_memory()
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


def main():
    parser = argparse.ArgumentParser()
    parser.add_argument("--gpus", type=int, default=1)
    parser.add_argument("--steps", type=int, default=20)
    parser.add_argument("--warmup", type=int, default=3)
    parser.add_argument("--workload", default="headline", choices=sorted(WORKLOADS))
    parser.add_argument("--streams", type=int, default=0, help="streams per GPU (0 = workload default)")
    parser.add_argument("--seconds", type=float, default=10.0,
                        help="audio seconds per stream per step (10 s = a whole stream of the survey's workload in one call)")
    parser.add_argument("--arithmetic", default="fast", choices=["exact", "fast"])
    parser.add_argument("--no-cpu-baseline", action="store_true")
    parser.add_argument("--no-host-fed", action="store_true", help="skip the PCIe-inclusive side measurement")
    args = parser.parse_args()

    import torch

    import drbamd

    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    assert world == args.gpus, "launch with --nproc-per-node equal to --gpus"
    assert torch.cuda.is_available(), "bench.py needs a GPU"
    torch.cuda.set_device(local_rank)
    device = torch.device("cuda", local_rank)

    if world > 1:
        import torch.distributed as dist
        dist.init_process_group("nccl", device_id=device)

    src, dst, ch, bs, mbc, direction, quality, sample, layout, default_streams = WORKLOADS[args.workload]
    streams = args.streams or default_streams
    sample_type = {"f32": drbamd.F32, "i16": drbamd.I16, "i32": drbamd.I32}[sample]
    torch_type = {"f32": torch.float32, "i16": torch.int16, "i32": torch.int32}[sample]
    config = drbamd.make_config(src, dst, ch, bs, mbc, direction, quality, sample_type=sample_type,
                                layout=drbamd.INTERLEAVED if layout == "interleaved" else drbamd.PLANAR,
                                arithmetic=drbamd.EXACT if args.arithmetic == "exact" else drbamd.FAST,
                                streams=streams, device=local_rank)
    info = drbamd.plan_info(config)
    batch = drbamd.Batch(config)

    # user-side frames per step: push = source frames, pull = target frames
    user_rate = dst if direction == 1 else src
    frames = int(round(args.seconds * user_rate / bs)) * bs
    other = batch.query([frames] * streams)[0]
    source_frames = (other + 4096 if direction == 1 else frames) + int(os.environ.get("BENCH_SOURCE_PAD", "0"))
    target_capacity = frames if direction == 1 else other + 4096

    generator = torch.Generator(device=device)
    generator.manual_seed(1234 + rank)
    noise = torch.rand((streams, ch, source_frames), device=device, generator=generator) * 2 - 1
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
    produced = (ctypes.c_int * streams)()
    stream = torch.cuda.current_stream().cuda_stream

    def step():
        batch.process_dense(source.data_ptr(), ch * source_frames, source_frames,
                            target.data_ptr(), ch * target_capacity, target_capacity,
                            frames, source_frames if direction == 1 else target_capacity,
                            stream=stream, produced=produced)
        return sum(produced) if direction == 0 else frames * streams

    def barrier():
        torch.cuda.synchronize()
        if world > 1:
            dist.barrier()
        torch.cuda.synchronize()

    for _ in range(args.warmup):
        step()
    barrier()
    start = time.perf_counter()
    out_frames = 0
    for _ in range(args.steps):
        out_frames += step()
    host_elapsed = time.perf_counter() - start   # the host's share: descriptors, planning, launches
    barrier()
    elapsed = time.perf_counter() - start
    batch.synchronize(stream)

    from drbamd import sharding

    elapsed, all_frames = sharding.reduce_step_stats(elapsed, out_frames, device=device)
    samples = all_frames * ch
    value = samples / elapsed / 1e6

    # dominant kernel: duration from HIP events on the launch stream
    batch.time_kernels(True)
    kernel_ms, kernel_samples = [], []
    for _ in range(max(3, min(args.steps, 10))):
        n = step()
        kernel_ms.append(batch.kernel_ms())
        kernel_samples.append(n * ch)
    batch.time_kernels(False)
    mean_ms = sum(kernel_ms) / len(kernel_ms)
    mean_samples = sum(kernel_samples) / len(kernel_samples)
    flops = info.flops_per_output_sample * mean_samples
    nbytes = info.bytes_per_output_sample * mean_samples
    compute_frac = flops / (mean_ms * 1e-3) / 1e12 / FP32_PEAK_TFLOPS
    memory_frac = nbytes / (mean_ms * 1e-3) / 1e9 / HBM_PEAK_GBS
    if compute_frac >= memory_frac:
        roofline = {"bound": "mfma", "achieved": round(flops / (mean_ms * 1e-3) / 1e12, 3),
                    "peak": FP32_PEAK_TFLOPS, "unit": "TFLOP/s", "frac": round(compute_frac, 4)}
    else:
        roofline = {"bound": "hbm", "achieved": round(nbytes / (mean_ms * 1e-3) / 1e9, 1),
                    "peak": HBM_PEAK_GBS, "unit": "GB/s", "frac": round(memory_frac, 4)}
    fast = args.arithmetic == "fast" and bool(info.fast_path)

    # HBM bytes per launch cannot be counted from inside this process: they come
    # from the committed rocprofv3 --pmc passes of this same command, if any
    traffic = None
    try:
        with open(os.path.join(ROOT, "profiles", "traffic.json")) as handle:
            entry = json.load(handle).get("%s/%s/%d/%.1f" % (args.workload, args.arithmetic, streams, args.seconds))
        if entry:
            traffic = entry["bytes_per_launch"]
    except OSError:
        pass

    kernel = "drb_cascade_kernel"
    if fast:
        kinds = set(kind for kind in info.stage_kind[:info.stage_count] if kind != 4)
        kernel = "drb_fast_kernel" if 3 in kinds else "drb_decimate_kernel" if kinds == {2} else "drb_interpolate_kernel"
        if info.fast_collapsed:
            kernel = "drb_collapsed_plain_kernel" if (sample, layout) == ("f32", "planar") else "drb_collapsed_kernel"
        if info.fast_launches > 1:
            kernel += " + %d half-band launches" % (info.fast_launches - 1)
    roofline.update({"traffic": traffic, "kernel": kernel,
                     "kernel_ms": round(mean_ms, 4),
                     "algorithmic_flops_per_sample": round(info.flops_per_output_sample, 1),
                     "algorithmic_bytes_per_sample": round(info.bytes_per_output_sample, 3),
                     "other_roofline_frac": round(min(compute_frac, memory_frac), 4)})
    if fast:
        # the matrix-core kernel merges upstream's two (or four) dot products per
        # output into one and pads its banded products: what it really issues
        issued = info.fast_issued_flops_per_output_sample
        if info.fast_launches == 1 and 3 in kinds and streams * ch >= 2048 and not info.fast_collapsed:
            # one workgroup then walks over all tiles of a stream and only its
            # first tile spends a segment on warming up: 16 useful segments per
            # tile instead of the 15 the figure is stated for
            issued *= 15.0 / 16.0
        roofline.update({"issued_flops_per_sample": round(issued, 1),
                         "issued_mfma_frac": round(issued * mean_samples / (mean_ms * 1e-3) / 1e12 / FP32_PEAK_TFLOPS, 4)})

    if rank == 0:
        line = {
            "metric": "output Msamples/s, 1024 stereo 44.1->48k streams",
            "value": round(value, 2), "unit": "Msamples/s", "n_gpus": world,
            "steps": args.steps, "warmup": args.warmup,
            "ms_per_step": round(elapsed / args.steps * 1e3, 4),
            "host_ms_per_step": round(host_elapsed / args.steps * 1e3, 4),
            "higher_is_better": True, "scaling": "weak", "vs_baseline": None,
            "dtype": "f32", "data": "synthetic white noise, uniform [-1, 1)",
            "config": {"workload": "%d x %d-ch %s %d->%d Hz per GPU, quality %s, %s, %s layout, %.2f s per step, %s arithmetic"
                                   % (streams, ch, sample, src, dst, ["poor", "fine", "good", "best"][quality],
                                      "pull" if direction else "push", layout, args.seconds, args.arithmetic),
                       "name": args.workload, "streams_per_gpu": streams, "frames_per_step": frames,
                       "tile_frames": info.fast_tile_frames if fast else info.tile_frames,
                       "history_frames": info.history_frames},
            "roofline": roofline,
        }
        if world == 1 and not args.no_host_fed:
            # one second per stream per call, whatever the step above was (1.5 GB of
            # pinned memory instead of 15)
            fed = int(round(min(args.seconds, 1.0) * user_rate / bs)) * bs
            fed_other = batch.query([fed] * streams)[0]
            fed_source = fed_other + 4096 if direction == 1 else fed
            fed_target = fed if direction == 1 else fed_other + 4096
            fed_slice = source[:, :fed_source, :] if layout == "interleaved" else source[:, :, :fed_source]
            line["host_fed"] = host_fed(batch, fed_slice.contiguous(), ch, fed_source, fed_target, torch_type, fed,
                                        fed_source if direction == 1 else fed_target, layout, direction)
            line["host_fed"]["seconds_per_step"] = min(args.seconds, 1.0)
        if world == 1 and not args.no_cpu_baseline:
            line["cpu_baseline"] = cpu_baseline(WORKLOADS[args.workload])
        print(json.dumps(line), flush=True)

    batch.close()
    if world > 1:
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
