"""What the host link gives two overlapping copy streams (H2D and D2H of 64 MB
pieces) as a function of how much pinned memory the pieces are spread over.
Run on the GPU box: python3 tools/microbench/link_probe.py"""
import time
import torch

piece = 64 << 20
dev_in = [torch.empty(piece, dtype=torch.uint8, device="cuda") for _ in range(3)]
dev_out = [torch.empty(piece, dtype=torch.uint8, device="cuda") for _ in range(3)]
s_in, s_out = torch.cuda.Stream(), torch.cuda.Stream()
for total_gb in (0.5, 2.0, 4.0):
    pieces = int(total_gb * (1 << 30)) // piece
    host_in = torch.empty(pieces * piece, dtype=torch.uint8).pin_memory()
    host_out = torch.empty(pieces * piece, dtype=torch.uint8).pin_memory()
    host_in.fill_(1); host_out.fill_(2)
    for label, both in (("H2D alone", False), ("H2D and D2H together", True)):
        torch.cuda.synchronize()
        start = time.perf_counter()
        for i in range(pieces):
            with torch.cuda.stream(s_in):
                dev_in[i % 3].copy_(host_in[i * piece:(i + 1) * piece], non_blocking=True)
            if both:
                with torch.cuda.stream(s_out):
                    host_out[i * piece:(i + 1) * piece].copy_(dev_out[i % 3], non_blocking=True)
        torch.cuda.synchronize()
        elapsed = time.perf_counter() - start
        moved = pieces * piece * (2 if both else 1)
        print("%.1f GB pinned each way, %s: %.1f GB/s" % (total_gb, label, moved / elapsed / 1e9), flush=True)
    del host_in, host_out

# as the benchmark allocates them: a 3.6 GB source and a 3.9 GB target, pinned copies of pageable tensors
src = torch.rand(3_600_000_000 // 4).pin_memory()
dst = torch.zeros(3_900_000_000 // 4).pin_memory()
for label, host, to_host in (("D2H alone into the 3.9 GB target", dst, True), ("H2D alone from the 3.6 GB source", src, False)):
    view = host.view(torch.uint8)
    pieces = view.numel() // piece
    torch.cuda.synchronize()
    start = time.perf_counter()
    for i in range(pieces):
        with torch.cuda.stream(s_out if to_host else s_in):
            if to_host:
                view[i * piece:(i + 1) * piece].copy_(dev_out[i % 3], non_blocking=True)
            else:
                dev_in[i % 3].copy_(view[i * piece:(i + 1) * piece], non_blocking=True)
    torch.cuda.synchronize()
    print("%s: %.1f GB/s" % (label, pieces * piece / (time.perf_counter() - start) / 1e9), flush=True)
