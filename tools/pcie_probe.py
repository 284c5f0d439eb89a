"""Host link probe: page-locked H2D / D2H rate for one 512 MB copy, for 32 MB slices on one stream and on two / four streams,
and both directions at once. Input of the host-I/O budget of gmrfx_refactorize_solve (csrc/device.cpp host_upload / host_download)."""
import time
import torch

n = 64 * 1000 * 1000
h = torch.empty(n, dtype=torch.float64).pin_memory()
h2 = torch.empty(n, dtype=torch.float64).pin_memory()
d = torch.empty(n, dtype=torch.float64, device="cuda")
d2 = torch.empty(n, dtype=torch.float64, device="cuda")
h.fill_(1.0)
torch.cuda.synchronize()
GB = n * 8 / 1e9


def timed(fn, reps=5):
    fn(); torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(reps):
        fn()
    torch.cuda.synchronize()
    return (time.perf_counter() - t0) / reps


def sliced(dst, src, nstreams, slice_elems=4 * 1000 * 1000):
    streams = [torch.cuda.Stream() for _ in range(nstreams)]
    def fn():
        for k, j0 in enumerate(range(0, n, slice_elems)):
            with torch.cuda.stream(streams[k % nstreams]):
                dst[j0:j0 + slice_elems].copy_(src[j0:j0 + slice_elems], non_blocking=True)
    return fn


print(f"one copy      H2D {GB / timed(lambda: d.copy_(h, non_blocking=True)):6.1f} GB/s   D2H {GB / timed(lambda: h2.copy_(d, non_blocking=True)):6.1f} GB/s")
for ns in (1, 2, 4):
    print(f"32 MB slices on {ns} stream(s)  H2D {GB / timed(sliced(d, h, ns)):6.1f} GB/s   D2H {GB / timed(sliced(h2, d, ns)):6.1f} GB/s")
s1, s2 = torch.cuda.Stream(), torch.cuda.Stream()
def both():
    with torch.cuda.stream(s1):
        d.copy_(h, non_blocking=True)
    with torch.cuda.stream(s2):
        h2.copy_(d2, non_blocking=True)
print(f"both directions at once: {2 * GB / timed(both):6.1f} GB/s in total")
