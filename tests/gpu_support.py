"""Helpers for the GPU tests: device buffers come from torch, every conversion
goes through the C ABI (drbamd.Batch)."""
import numpy as np
import torch

import drbamd

NP_TYPES = {drbamd.F32: np.float32, drbamd.I16: np.int16, drbamd.I32: np.int32}
TORCH_TYPES = {drbamd.F32: torch.float32, drbamd.I16: torch.int16, drbamd.I32: torch.int32}


def to_layout(x, layout):
    """x: [channels, frames] -> flat array in the requested layout."""
    return np.ascontiguousarray(x.T if layout == drbamd.INTERLEAVED else x).reshape(-1)


def from_layout(flat, channels, frames, layout):
    if layout == drbamd.INTERLEAVED:
        return np.ascontiguousarray(flat.reshape(frames, channels).T)
    return flat.reshape(channels, frames)


class StreamRunner:
    """Feeds a batch step by step from host arrays and gathers the outputs.

    inputs: list (one per stream) of arrays [channels, frames] in the batch's
    sample type.  Each step takes, per stream, a user-side frame count (push:
    source frames, pull: target frames)."""

    def __init__(self, config, inputs, device="cuda:0"):
        self.config = config
        self.batch = drbamd.Batch(config)
        self.device = device
        self.channels = config.channel_count
        self.layout = config.layout
        self.np_type = NP_TYPES[config.sample_type]
        self.torch_type = TORCH_TYPES[config.sample_type]
        self.inputs = [np.ascontiguousarray(x, self.np_type) for x in inputs]
        self.read = [0] * len(inputs)
        self.outputs = [[] for _ in inputs]
        self.counts = [[] for _ in inputs]

    def step(self, frames):
        n = len(self.inputs)
        pull = self.config.direction == drbamd.PULL
        other = self.batch.query(frames)
        source_frames = other if pull else list(frames)
        target_frames = list(frames) if pull else other
        src_tensors, dst_tensors, src_ptrs, dst_ptrs = [], [], [], []
        src_stride = max(max(source_frames), 1)
        dst_stride = max(max(target_frames), 1)
        for s in range(n):
            take = source_frames[s]
            chunk = self.inputs[s][:, self.read[s]:self.read[s] + take]
            if chunk.shape[1] < take:  # producer ran dry: silence
                chunk = np.concatenate([chunk, np.zeros((self.channels, take - chunk.shape[1]), self.np_type)], 1)
            if self.layout == drbamd.INTERLEAVED:
                host = to_layout(chunk, self.layout)
            else:
                padded = np.zeros((self.channels, src_stride), self.np_type)
                padded[:, :take] = chunk
                host = padded.reshape(-1)
            src = torch.from_numpy(host.copy()).to(self.device) if host.size else torch.zeros(1, dtype=self.torch_type, device=self.device)
            dst = torch.full((self.channels * dst_stride + 1,), 77, dtype=self.torch_type, device=self.device)
            src_tensors.append(src)
            dst_tensors.append(dst)
            src_ptrs.append(src.data_ptr())
            dst_ptrs.append(dst.data_ptr())
            self.read[s] += take
        torch.cuda.synchronize()
        stream = torch.cuda.current_stream().cuda_stream
        produced = self.batch.process(src_ptrs, src_stride, source_frames, dst_ptrs, dst_stride,
                                      target_frames, stream=stream)
        self.batch.synchronize(stream)
        assert produced == other
        for s in range(n):
            got = target_frames[s]
            flat = dst_tensors[s].cpu().numpy()
            assert flat[-1] == 77  # nothing written past the buffer
            if self.layout == drbamd.INTERLEAVED:
                out = from_layout(flat[:got * self.channels], self.channels, got, self.layout)
            else:
                out = flat[:-1].reshape(self.channels, dst_stride)[:, :got]
            self.outputs[s].append(out.copy())
            self.counts[s].append(produced[s])
        return produced

    def result(self, s=0):
        return np.concatenate(self.outputs[s], axis=1)

    def close(self):
        self.batch.close()


def chunk_schedule(total, pattern):
    """Cuts `total` frames following `pattern` cyclically."""
    out, i = [], 0
    while total > 0:
        step = min(pattern[i % len(pattern)], total)
        out.append(step)
        total -= step
        i += 1
    return out
