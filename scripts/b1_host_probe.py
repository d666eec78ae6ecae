"""Is a single-frame pass bound by the host (queueing ~20 launches) or by the device?  N calls of imgfd_detect_dev without a
wait: wall time until the last call RETURNS (host) and until the device is idle (total)."""
import os, sys, time, json
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from image_amd.device import DeviceDetector
B = int(os.environ.get("BATCH", 1)); N = int(os.environ.get("N", 300))
det = DeviceDetector(0)
frames = det.synth_frames(B, 3840, 2160, seed0=50000)
corners = torch.empty(B, 20000, 3, device="cuda"); points = torch.empty(B, 60000, 2, dtype=torch.int32, device="cuda")
edges = torch.empty_like(frames); counts = torch.zeros(3, B, dtype=torch.int64, device="cuda")
kw = dict(fast9_threshold=20, suppress_non_max=1)
for which in ("all", "canny", "harris", "fast9"):
    p = dict(kw)
    if which != "all": p.update(harris=int(which == "harris"), fast9=int(which == "fast9"), canny=int(which == "canny"))
    for _ in range(20): det.detect_all(frames, corners, points, edges, counts, **p)
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(N): det.detect_all(frames, corners, points, edges, counts, **p)
    t1 = time.perf_counter(); torch.cuda.synchronize(); t2 = time.perf_counter()
    # one call at a time, waiting for each: the device-side latency of a lone pass
    t3 = time.perf_counter()
    for _ in range(100):
        det.detect_all(frames, corners, points, edges, counts, **p); torch.cuda.synchronize()
    t4 = time.perf_counter()
    print(json.dumps({"which": which, "batch": B, "host_us_per_call": round((t1 - t0) / N * 1e6, 1), "total_us_per_call": round((t2 - t0) / N * 1e6, 1),
                      "lone_call_with_wait_us": round((t4 - t3) / 100 * 1e6, 1)}), flush=True)
