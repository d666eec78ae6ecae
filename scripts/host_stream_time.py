"""PCIe-inclusive rate of the host frame stream (imgfd_stream_*): 4K u8 frames in host memory -> Harris + FAST-9 +
Canny, upload of batch i+1 overlapped with the kernels of batch i.  Same detector parameters as bench.py."""
import json, os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
from image_amd import _lib, framestream, synth
NX, NY, BATCH, NB = 3840, 2160, 32, 8
ctx = _lib.Context(0)
base = np.stack([synth.frame(50000 + f, NX, NY) for f in range(4)])
def fill(dst):
    for f in range(dst.shape[0]):
        dst[f] = base[f % 4]
pinned = [framestream.PinnedFrames(BATCH, NY, NX) for _ in range(3)]
for p in pinned: fill(p.array)
pageable = [np.empty((BATCH, NY, NX), np.uint8) for _ in range(3)]
for a in pageable: fill(a)
out = {"frame": f"{NX}x{NY}", "batch": BATCH, "batches": NB}
def run(fs, bufs):
    counts = []
    t = time.perf_counter()
    for r in fs.run(bufs[i % 3] for i in range(NB)):
        counts.append((int(r["harris_counts"].sum()), int(r["fast9_counts"].sum()), int(r["canny_counts"].sum())))
    return time.perf_counter() - t, counts
kw = dict(fast9_threshold=20, suppress_non_max=1)
with framestream.FrameStream(NX, NY, BATCH, ctx=ctx, **kw) as fs:
    run(fs, [a.array for a in pinned])   # warm-up (allocations, first launches)
    for name, bufs in (("pinned", [a.array for a in pinned]), ("pageable", pageable)):
        ts = []
        for _ in range(3):
            t, counts = run(fs, bufs); ts.append(t)
        assert len(set(counts)) == 1, counts   # same frames in every batch -> same counts
        out[name] = {"s": round(min(ts), 4), "Mpix_s": round(NB * BATCH * NX * NY / min(ts) / 1e6, 1),
                     "GBps_up": round(NB * BATCH * NX * NY / min(ts) / 1e9, 2)}
    out["counts_per_batch"] = counts[0]
print(json.dumps(out))
