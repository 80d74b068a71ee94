"""Wall time of each of the first steps of the headline batch (enqueue + synchronize),
to see what the first steps pay beyond the steady state.
Run on the GPU box: python3 tools/step_times.py"""
import sys, time
sys.path.insert(0, "audio-resampler_amd")
import torch, drbamd
streams, ch, frames = 1024, 2, 441088
config = drbamd.make_config(44100, 48000, ch, 256, 16, drbamd.PUSH, drbamd.GOOD, arithmetic=drbamd.FAST, streams=streams)
batch = drbamd.Batch(config)
cap = 484096
src = torch.rand(streams, ch, frames, device="cuda") * 2 - 1
dst = torch.zeros(streams, ch, cap, device="cuda")
stream = torch.cuda.current_stream().cuda_stream
torch.cuda.synchronize()
for i in range(12):
    t0 = time.perf_counter()
    batch.process_dense(src.data_ptr(), ch * frames, frames, dst.data_ptr(), ch * cap, cap, frames, cap, stream=stream)
    t1 = time.perf_counter()
    torch.cuda.synchronize()
    t2 = time.perf_counter()
    print("step %2d: enqueue %.3f ms, until done %.3f ms" % (i, (t1 - t0) * 1e3, (t2 - t0) * 1e3), flush=True)
