#!/usr/bin/env python3
"""Randomised parity run on the GPU: conversions drawn from everything the
upstream planner accepts (13 x 13 rates, 4 qualities, 2 directions, 1-8
channels, 7 block sizes), with random sample type, layout, arithmetic, stream
count and step pattern, each stream against the oracle (the CPU restatement of
upstream under oracle/): exact arithmetic bit for bit, fast arithmetic within
1e-6 + 1e-5 |want| (float32) or one step (int16).

usage: python3 tools/soak.py [--seconds 120] [--seed 1] [--arithmetic exact|fast|both] [--api | --dense | --scale]
Prints one line per failing case and a summary; exit status 1 on any failure."""
import argparse
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "tests"))
sys.path.insert(0, os.path.join(ROOT, "audio-resampler_amd"))

import support  # noqa: E402
import drbamd  # noqa: E402
import gpu_support  # noqa: E402


MOST_STREAMS = 5


def draw(rng, arithmetic):
    while True:
        src, dst = (int(v) for v in rng.choice(support.RATES, 2))
        ch = int(rng.integers(1, 9))
        bs = int(rng.choice(support.BLOCK_SIZES))
        mbc = int(rng.choice([1, 2, 16, 64]))
        direction = int(rng.integers(0, 2))
        quality = int(rng.integers(0, 4))
        status, plan = support.oracle_plan(src, dst, ch, bs, mbc, direction, quality)
        if status != 1:
            continue
        sample = int(rng.choice([drbamd.F32, drbamd.F32, drbamd.I16, drbamd.I32]))
        layout = int(rng.choice([drbamd.PLANAR, drbamd.INTERLEAVED]))
        which = arithmetic if arithmetic != "both" else str(rng.choice(["exact", "fast"]))
        if arithmetic == "fast" and rng.integers(0, 8):
            # (most draws have no matrix-core kernel and would run the exact ones: draw again, most of the time)
            try:
                if not drbamd.plan_info(drbamd.make_config(src, dst, ch, bs, mbc, direction, quality, sample_type=sample,
                                                           layout=layout, arithmetic=drbamd.FAST)).fast_path:
                    continue
            except drbamd.DrbError:
                continue
        return dict(src=src, dst=dst, ch=ch, bs=bs, mbc=mbc, direction=direction, quality=quality,
                    sample=sample, layout=layout, arithmetic=which, streams=int(rng.integers(1, MOST_STREAMS + 1))), plan


def run_case(case, plan, rng, even):
    arithmetic = drbamd.FAST if case["arithmetic"] == "fast" else drbamd.EXACT
    try:
        config = drbamd.make_config(case["src"], case["dst"], case["ch"], case["bs"], case["mbc"], case["direction"],
                                    case["quality"], sample_type=case["sample"], layout=case["layout"],
                                    arithmetic=arithmetic, streams=case["streams"])
        info = drbamd.plan_info(config)
    except drbamd.DrbError:
        return "unsupported"
    n, ch = case["streams"], case["ch"]
    pull = case["direction"] == support.PULL
    # user-side frames per stream: a few tiles' worth, now and then a few hundred (the
    # split kernel's coefficient sets change every 64 tiles); batches of many streams: short ones
    most = int(rng.integers(2000, 60000)) if rng.integers(0, 8) else int(rng.integers(200000, 500000))
    if n > 16:
        most = int(rng.integers(500, 9000))
    ratio = case["src"] / case["dst"]
    source_most = int(most * ratio) + 4096 if pull else most
    noise = [support.white_noise(ch, source_most + 8192, int(rng.integers(1, 1 << 30))) for _ in range(n)]
    if case["sample"] == drbamd.I16:
        inputs = [np.round(x * 32767).astype(np.int16) for x in noise]
    elif case["sample"] == drbamd.I32:
        inputs = [np.round(x.astype(np.float64) * (2 ** 24 - 1)).astype(np.int32) * 128 for x in noise]
    else:
        inputs = noise
    totals = [int(v) for v in rng.integers(0, most + 1, n)]
    # (a step whose frame count reaches a 2x decimator odd is refused: such chains get multiples of 16)
    totals = [t // even * even for t in totals]
    runner = gpu_support.StreamRunner(config, inputs)
    left = list(totals)
    again = int(rng.integers(0, n)) if rng.integers(0, 4) == 0 else -1   # a stream that is started over after its second step
    taken = 0
    try:
        while any(left):
            limit = int(rng.choice([64, 700, 5000, 70000])) if most < 100000 else int(rng.choice([5000, 70000, 600000]))
            steps = [min(l, int(rng.integers(0, limit)) // even * even) for l in left]
            if not any(steps):
                steps = [min(l, even * 8) for l in left]
            runner.step(steps)
            left = [l - s for l, s in zip(left, steps)]
            taken += 1
            if taken == 2 and again >= 0:
                runner.batch.reset(again, 1)
                runner.read[again] = 0
                runner.outputs[again] = []
                left[again] = totals[again]
        for s in (range(n) if n <= 16 else sorted(int(v) for v in rng.choice(n, 16, replace=False))):
            xf = inputs[s] if case["sample"] == drbamd.F32 else support.oracle_convert_samples(inputs[s], "f32").reshape(inputs[s].shape)
            if pull:
                want_f, _ = support.oracle_pull(plan, xf, totals[s])
            else:
                want_f = support.oracle_push(plan, xf[:, :totals[s]])
            got = runner.result(s) if totals[s] or runner.outputs[s] else np.zeros((ch, 0))
            if case["sample"] == drbamd.F32:
                want = want_f
            else:
                to = "i16" if case["sample"] == drbamd.I16 else "i32"
                want = support.oracle_convert_samples(np.ascontiguousarray(want_f), to).reshape(want_f.shape)
            if got.shape != want.shape:
                return "shape %s against %s (stream %d)" % (got.shape, want.shape, s)
            if not want.size:
                continue
            exact = case["arithmetic"] == "exact" or not info.fast_path
            if case["sample"] == drbamd.F32:
                if exact:
                    if not np.array_equal(got.view(np.uint32), want.view(np.uint32)):
                        return "stream %d differs (exact)" % s
                else:
                    excess = np.abs(got.astype(np.float64) - want) - (1e-6 + 1e-5 * np.abs(want))
                    if excess.max() > 0:
                        return "stream %d: |err| %.3g beyond the bound by %.3g" % (s, np.abs(got - want).max(), excess.max())
            else:
                diff = np.abs(got.astype(np.int64) - want.astype(np.int64)).max()
                if diff > (0 if exact else 1):
                    return "stream %d: %d steps off" % (s, diff)
    finally:
        runner.close()
    return "fast" if (case["arithmetic"] == "fast" and info.fast_path) else "exact"


def run_dense_case(case, plan, rng):
    """The dense entry points (uniform batches): device-resident, the one that
    allocates nothing, and host-fed, drawn at random, several steps each."""
    import ctypes
    import torch
    arithmetic = drbamd.FAST if case["arithmetic"] == "fast" else drbamd.EXACT
    n, ch = case["streams"], case["ch"]
    try:
        config = drbamd.make_config(case["src"], case["dst"], ch, case["bs"], case["mbc"], case["direction"],
                                    case["quality"], sample_type=case["sample"], layout=case["layout"],
                                    arithmetic=arithmetic, streams=n)
        info = drbamd.plan_info(config)
    except drbamd.DrbError:
        return "unsupported"
    pull = case["direction"] == support.PULL
    entry = str(rng.choice(["dense", "static", "host"]))
    steps = [int(v) // 16 * 16 for v in rng.integers(0, int(rng.choice([600, 9000, 50000])), int(rng.integers(1, 5)))]
    total = sum(steps)
    if total == 0:
        return "unsupported"
    count = support.oracle_lib().drb_oracle_pull_count if pull else support.oracle_lib().drb_oracle_push_count
    other_total = int(count(plan, total))
    source_total = other_total if pull else total
    np_type = gpu_support.NP_TYPES[case["sample"]]
    noise = [support.white_noise(ch, source_total + 64, int(rng.integers(1, 1 << 30))) for _ in range(n)]
    if case["sample"] == drbamd.I16:
        inputs = [np.round(x * 32767).astype(np.int16) for x in noise]
    elif case["sample"] == drbamd.I32:
        inputs = [np.round(x.astype(np.float64) * (2 ** 24 - 1)).astype(np.int32) * 128 for x in noise]
    else:
        inputs = noise
    interleaved = case["layout"] == drbamd.INTERLEAVED
    batch = drbamd.Batch(config)
    got = [[] for _ in range(n)]
    read = 0
    try:
        for frames in steps:
            other = batch.query([frames] * n)[0]
            take, make = (other, frames) if pull else (frames, other)
            capacity = (take if pull else make) + int(rng.integers(0, 40))
            room = max(make, capacity if not pull else make, 1)
            chunk = np.stack([x[:, read:read + take] for x in inputs])                     # [n, ch, take]
            chunk = np.concatenate([chunk, np.zeros((n, ch, max(1, take) - take), np_type)], 2) if take == 0 else chunk
            width = chunk.shape[2]
            host_source = np.ascontiguousarray(chunk.transpose(0, 2, 1) if interleaved else chunk)
            host_target = np.full((n, room, ch) if interleaved else (n, ch, room), 7, np_type)
            produced = (ctypes.c_int * n)()
            source_strides = (width * ch, 0 if interleaved else width)
            target_strides = (room * ch, 0 if interleaved else room)
            bound = take if pull else room
            if entry == "host":
                source = torch.from_numpy(host_source).pin_memory()
                target = torch.from_numpy(host_target).pin_memory()
                batch.process_host_dense(source.data_ptr(), *source_strides, target.data_ptr(), *target_strides,
                                         frames, bound, produced=produced)
                result = target.numpy()
            else:
                source = torch.from_numpy(host_source).cuda()
                target = torch.from_numpy(host_target).cuda()
                stream = torch.cuda.current_stream().cuda_stream
                call = batch.process_dense_static if entry == "static" else batch.process_dense
                try:
                    call(source.data_ptr(), *source_strides, target.data_ptr(), *target_strides,
                         frames, bound, stream=stream, produced=produced)
                except drbamd.DrbError as error:
                    if entry == "static" and error.status == "capacity":
                        return "unsupported"   # (a chain of several launches that has not run a step this long yet)
                    raise
                batch.synchronize(stream)
                result = target.cpu().numpy()
            if list(produced) != [other] * n:
                return "%s: produced %s against %d" % (entry, list(produced)[:4], other)
            for s in range(n):
                got[s].append(np.ascontiguousarray(result[s, :make, :].T) if interleaved else result[s, :, :make].copy())
            read += take
        for s in range(n):
            xf = inputs[s] if case["sample"] == drbamd.F32 else support.oracle_convert_samples(inputs[s], "f32").reshape(inputs[s].shape)
            want_f = support.oracle_pull(plan, xf, total)[0] if pull else support.oracle_push(plan, xf[:, :total])
            want = want_f if case["sample"] == drbamd.F32 else support.oracle_convert_samples(
                np.ascontiguousarray(want_f), "i16" if case["sample"] == drbamd.I16 else "i32").reshape(want_f.shape)
            result = np.concatenate(got[s], axis=1)
            if result.shape != want.shape:
                return "%s: shape %s against %s (stream %d)" % (entry, result.shape, want.shape, s)
            if not want.size:
                continue
            exact = case["arithmetic"] == "exact" or not info.fast_path
            if case["sample"] == drbamd.F32:
                if exact and not np.array_equal(result.view(np.uint32), want.view(np.uint32)):
                    return "%s: stream %d differs (exact)" % (entry, s)
                excess = np.abs(result.astype(np.float64) - want) - (1e-6 + 1e-5 * np.abs(want))
                if excess.max() > 0:
                    return "%s: stream %d beyond the bound by %.3g" % (entry, s, excess.max())
            else:
                diff = np.abs(result.astype(np.int64) - want.astype(np.int64)).max()
                if diff > (0 if exact else 1):
                    return "%s: stream %d: %d steps off" % (entry, s, diff)
    finally:
        batch.close()
    return "fast" if (case["arithmetic"] == "fast" and info.fast_path) else "exact"


def run_scale_case(case, plan, rng):
    """Batches of hundreds to thousands of streams and steps of many tiles (what the
    benchmark runs, at configurations drawn at random): samples made on the
    device, a few streams checked."""
    import ctypes
    import torch
    if case["sample"] == drbamd.I32:
        return "unsupported"
    arithmetic = drbamd.FAST if case["arithmetic"] == "fast" else drbamd.EXACT
    ch = case["ch"]
    pull = case["direction"] == support.PULL
    steps = [int(v) // 16 * 16 for v in rng.integers(20000, 160000, int(rng.integers(1, 4)))]
    total = sum(steps)
    count = support.oracle_lib().drb_oracle_pull_count if pull else support.oracle_lib().drb_oracle_push_count
    other_total = int(count(plan, total))
    source_total, target_total = (other_total, total) if pull else (total, other_total)
    n = int(rng.integers(300, 2000))
    n = max(16, min(n, int(6e9 / (4.0 * ch * max(source_total, target_total, 1)))))
    try:
        config = drbamd.make_config(case["src"], case["dst"], ch, case["bs"], case["mbc"], case["direction"],
                                    case["quality"], sample_type=case["sample"], layout=case["layout"],
                                    arithmetic=arithmetic, streams=n)
        info = drbamd.plan_info(config)
    except drbamd.DrbError:
        return "unsupported"
    generator = torch.Generator(device="cuda:0")
    generator.manual_seed(int(rng.integers(1, 1 << 30)))
    interleaved = case["layout"] == drbamd.INTERLEAVED
    width = source_total + 64
    room = target_total + 4096
    shape = (n, width, ch) if interleaved else (n, ch, width)
    if case["sample"] == drbamd.I16:
        source = torch.randint(-32767, 32768, shape, device="cuda:0", dtype=torch.int16, generator=generator)
    else:
        source = torch.rand(shape, device="cuda:0", generator=generator) * 2.0 - 1.0
    target = torch.zeros((n, room, ch) if interleaved else (n, ch, room), device="cuda:0", dtype=source.dtype)
    # (bytes from one frame to the next of a stream's buffer)
    hop = ch if interleaved else 1
    size = source.element_size()
    batch = drbamd.Batch(config)
    stream = torch.cuda.current_stream().cuda_stream
    read = made = 0
    try:
        for frames in steps:
            other = batch.query([frames] * n)[0]
            take, make = (other, frames) if pull else (frames, other)
            produced = (ctypes.c_int * n)()
            batch.process_dense(source.data_ptr() + size * hop * read, ch * width, 0 if interleaved else width,
                                target.data_ptr() + size * hop * made, ch * room, 0 if interleaved else room,
                                frames, take if pull else room - made, stream=stream, produced=produced)
            if list(produced) != [other] * n:
                return "produced %s against %d" % (list(produced)[:4], other)
            read += take
            made += make
        batch.synchronize(stream)
        exact = case["arithmetic"] == "exact" or not info.fast_path
        for s in sorted(set([0, n - 1] + [int(v) for v in rng.integers(0, n, 3)])):
            x = np.ascontiguousarray(source[s].cpu().numpy().T) if interleaved else source[s].cpu().numpy()
            xf = x if case["sample"] == drbamd.F32 else support.oracle_convert_samples(x, "f32").reshape(x.shape)
            want_f = support.oracle_pull(plan, xf, total)[0] if pull else support.oracle_push(plan, xf[:, :total])
            want = want_f if case["sample"] == drbamd.F32 else support.oracle_convert_samples(np.ascontiguousarray(want_f), "i16").reshape(want_f.shape)
            got = np.ascontiguousarray(target[s, :made, :].cpu().numpy().T) if interleaved else target[s, :, :made].cpu().numpy()
            if got.shape != want.shape:
                return "shape %s against %s (stream %d of %d)" % (got.shape, want.shape, s, n)
            if not want.size:
                continue   # (too few frames for one block of the slicer's)
            if case["sample"] == drbamd.F32:
                if exact and not np.array_equal(got.view(np.uint32), want.view(np.uint32)):
                    return "stream %d of %d differs (exact)" % (s, n)
                excess = np.abs(got.astype(np.float64) - want) - (1e-6 + 1e-5 * np.abs(want))
                if excess.max() > 0:
                    return "stream %d of %d: |err| %.3g beyond the bound by %.3g (%s)" % (
                        s, n, np.abs(got.astype(np.float64) - want).max(), excess.max(), info.kernel_name.decode())
            else:
                diff = np.abs(got.astype(np.int64) - want.astype(np.int64)).max()
                if diff > (0 if exact else 1):
                    return "stream %d of %d: %d steps off" % (s, n, diff)
    finally:
        batch.close()
    return "fast" if (case["arithmetic"] == "fast" and info.fast_path) else "exact"


_harness = None


def run_api_case(case, plan, rng):
    """The four-function drop-in on the GPU (through the harness that also drives
    the upstream build): samples bit for bit and the callbacks' frame counts."""
    global _harness
    if _harness is None:
        _harness = support.ApiHarness(drbamd.HARNESS_PATH)
    if not plan.consistent:
        return "unsupported"
    src, dst, ch, bs, mbc, direction, quality = (case[k] for k in ("src", "dst", "ch", "bs", "mbc", "direction", "quality"))
    chunks = [int(v) for v in rng.integers(0, int(rng.choice([40, 900, 6000])), int(rng.integers(1, 7)))]
    if _harness.sizes(src, dst, ch, bs, mbc, direction, quality) == (0, 0):
        return "unsupported"
    # Upstream plans a few chains whose sub-block does not fit its own work buffer (e.g.
    # mono 11025 -> 44100, block size 1: 64 frames in, 256 out, 128 floats of buffer) and
    # writes past it; the drop-in stops the process there instead.  Nothing to compare.
    info = drbamd.plan_info(drbamd.make_config(src, dst, ch, bs, mbc, direction, quality))
    count = support.oracle_lib().drb_oracle_pull_count if direction == support.PULL else support.oracle_lib().drb_oracle_push_count
    most = max(int(count(plan, info.throttling * k)) - int(count(plan, info.throttling * (k - 1))) for k in range(1, 40))
    if max(most, info.throttling) * ch > info.buffer_length:
        return "unsupported"
    if direction == support.PUSH:
        x = support.white_noise(ch, max(sum(chunks), 1), int(rng.integers(1, 1 << 30)))
        result = _harness.push(src, dst, ch, bs, mbc, quality, x, chunks)
        if result is None:
            return "unsupported"
        got, counts = result
        want = support.oracle_push(plan, x[:, :sum(chunks)])
    else:
        need = support.oracle_lib().drb_oracle_pull_count(plan, sum(chunks))
        x = support.white_noise(ch, max(int(need), 1), int(rng.integers(1, 1 << 30)))
        result = _harness.pull(src, dst, ch, bs, mbc, quality, x, chunks)
        if result is None:
            return "unsupported"
        got, counts, consumed = result
        want, used = support.oracle_pull(plan, x, sum(chunks))
        if used != consumed:
            return "consumed %d against %d" % (consumed, used)
    if got.shape != want.shape:
        return "shape %s against %s" % (got.shape, want.shape)
    if want.size and not np.array_equal(np.ascontiguousarray(got).view(np.uint32), np.ascontiguousarray(want).view(np.uint32)):
        return "samples differ"
    if support.oracle_callback_counts(plan, chunks).tolist() != counts.tolist():
        return "callback counts differ"
    return "api"


def run_one(case, plan, rng):
    """run_case, over again with frame counts in multiples of 16 where a step was
    refused because an odd count reached a 2x decimator"""
    state = rng.bit_generator.state
    try:
        return run_case(case, plan, rng, 1)
    except drbamd.DrbError as error:
        if "odd frame count" not in str(error):
            raise
    rng.bit_generator.state = state
    return run_case(case, plan, rng, 16)


def main():
    parser = argparse.ArgumentParser()
    parser.add_argument("--seconds", type=float, default=120.0)
    parser.add_argument("--seed", type=int, default=1)
    parser.add_argument("--arithmetic", default="both", choices=["exact", "fast", "both"])
    parser.add_argument("--streams", type=int, default=5, help="streams per batch, at most (beyond 16: short streams, 16 of them checked)")
    parser.add_argument("--api", action="store_true", help="the four-function drop-in instead of the batched entry point")
    parser.add_argument("--scale", action="store_true", help="batches of hundreds of streams and steps of many tiles, a few streams checked")
    parser.add_argument("--dense", action="store_true", help="the dense entry points (device-resident, static workspace, host-fed)")
    args = parser.parse_args()
    global MOST_STREAMS
    MOST_STREAMS = args.streams
    rng = np.random.default_rng(args.seed)
    started = time.time()
    tally, failures, last_print = {}, 0, started
    while time.time() - started < args.seconds:
        case, plan = draw(rng, args.arithmetic)
        if os.environ.get("DRB_SOAK_TRACE"):
            print("case", case, flush=True)   # (the drop-in aborts where upstream has no error to return)
        try:
            outcome = (run_api_case(case, plan, rng) if args.api else run_dense_case(case, plan, rng) if args.dense
                       else run_scale_case(case, plan, rng) if args.scale else run_one(case, plan, rng))
        except Exception as error:  # a step the library refused, or a helper's assertion
            outcome = "error: %s" % (str(error)[:200] or type(error).__name__)
        if outcome in ("fast", "exact", "api", "unsupported"):
            tally[outcome] = tally.get(outcome, 0) + 1
        else:
            failures += 1
            print("FAIL", case, outcome, flush=True)
        if time.time() - last_print > 30:
            last_print = time.time()
            print("... %d s: %s, %d failures" % (last_print - started, tally, failures), flush=True)
    print("cases: %s; failures: %d" % (tally, failures))
    return 1 if failures else 0


if __name__ == "__main__":
    sys.exit(main())
