"""Which pairs of HIP streams run kernels concurrently?  (streams share hardware queues: GPU_MAX_HW_QUEUES, default 4)"""
import os, sys, time, torch
dev = torch.device("cuda:0")
n = int(sys.argv[1]) if len(sys.argv) > 1 else 10
streams = [torch.cuda.Stream() for _ in range(n)]
cyc = 2_000_000
def t_pair(a, b):
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    with torch.cuda.stream(a): torch.cuda._sleep(cyc)
    if b is not None:
        with torch.cuda.stream(b): torch.cuda._sleep(cyc)
    torch.cuda.synchronize()
    return time.perf_counter() - t0
t_pair(streams[0], streams[1])
one = min(t_pair(streams[0], None) for _ in range(3))
print(f"one sleep: {one*1e3:.2f} ms; matrix: 1 = serialised (same queue), . = concurrent")
for i in range(n):
    row = ""
    for j in range(n):
        if i == j: row += " -"; continue
        t = min(t_pair(streams[i], streams[j]) for _ in range(2))
        row += " 1" if t > 1.6 * one else " ."
    print(f"stream {i}: {row}")

cur = torch.cuda.current_stream()
row = ""
for j in range(n):
    t = min(t_pair(cur, streams[j]) for _ in range(2))
    row += " 1" if t > 1.6 * one else " ."
print(f"default : {row}")
