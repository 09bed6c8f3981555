"""Host <-> device copy rates of the box by copy size (page-locked host memory, hipMemcpyAsync through torch): what bounds
felics_compress_batch's end-to-end figure.  Runs ON THE GPU BOX:  python3 profiles/tools/micro/link_rate.py"""
import time, torch
dev = torch.device("cuda", 0)
total = 512 << 20
h = torch.empty(total, dtype=torch.uint8).pin_memory()
d = torch.empty(total, dtype=torch.uint8, device=dev)
s2 = torch.cuda.Stream()
d.copy_(h); h.copy_(d); torch.cuda.synchronize()  # (first touch of the pinned pages)
for chunk_mb in (512, 64, 8, 4, 1):
    c = chunk_mb << 20
    for name, fn in (("H2D", lambda a, b: d[a:b].copy_(h[a:b], non_blocking=True)), ("D2H", lambda a, b: h[a:b].copy_(d[a:b], non_blocking=True))):
        torch.cuda.synchronize(); t = time.perf_counter()
        for a in range(0, total, c):
            fn(a, a + c)
        torch.cuda.synchronize(); dt = time.perf_counter() - t
        print("%s %4d MB copies: %.1f GB/s" % (name, chunk_mb, total / dt / 1e9))
# both directions at once on two streams
h2 = torch.empty(total, dtype=torch.uint8).pin_memory(); d2 = torch.empty(total, dtype=torch.uint8, device=dev)
h2.copy_(d2); torch.cuda.synchronize()  # (first touch of the pinned pages)
for c_mb in (64, 8, 4):
    c = c_mb << 20
    for rep in range(2):
        torch.cuda.synchronize(); t = time.perf_counter()
        for a in range(0, total, c):
            d[a:a + c].copy_(h[a:a + c], non_blocking=True)
            with torch.cuda.stream(s2):
                h2[a:a + c].copy_(d2[a:a + c], non_blocking=True)
        torch.cuda.synchronize(); dt = time.perf_counter() - t
    print("both directions at once, %d MB copies: %.1f GB/s each way" % (c_mb, total / dt / 1e9))
