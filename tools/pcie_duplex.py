"""Does this box move H2D and D2H at the same time?  Pinned copies on two streams, alone and together."""
import time, json, torch
n = 1 << 30
h1 = torch.empty(n, dtype=torch.uint8, pin_memory=True); h2 = torch.empty(n, dtype=torch.uint8, pin_memory=True)
d1 = torch.empty(n, dtype=torch.uint8, device="cuda"); d2 = torch.empty(n, dtype=torch.uint8, device="cuda")
s1, s2 = torch.cuda.Stream(), torch.cuda.Stream()
def run(h2d, d2h, reps=3):
    torch.cuda.synchronize(); t0 = time.perf_counter()
    for _ in range(reps):
        if h2d:
            with torch.cuda.stream(s1): d1.copy_(h1, non_blocking=True)
        if d2h:
            with torch.cuda.stream(s2): h2.copy_(d2, non_blocking=True)
    torch.cuda.synchronize(); return reps * n / (time.perf_counter() - t0) / 1e9
run(True, True, 1)
out = {"h2d_alone_GBps": run(True, False), "d2h_alone_GBps": run(False, True), "both_each_GBps": run(True, True)}
# chunked: 64 MiB pieces interleaved on the two streams
def run_chunked(reps=3, piece=1 << 26):
    torch.cuda.synchronize(); t0 = time.perf_counter()
    for _ in range(reps):
        for o in range(0, n, piece):
            with torch.cuda.stream(s1): d1[o:o + piece].copy_(h1[o:o + piece], non_blocking=True)
            with torch.cuda.stream(s2): h2[o:o + piece].copy_(d2[o:o + piece], non_blocking=True)
    torch.cuda.synchronize(); return reps * n / (time.perf_counter() - t0) / 1e9
out["both_each_GBps_64MiB_pieces"] = run_chunked()
print(json.dumps(out))
