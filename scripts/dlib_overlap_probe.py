"""fHOG beside SURF: config 4's two functions on the same tiles, one after the other on one stream (the bench's step) against fHOG on a
second context / stream beside imgfd_surf_dev -- whole batch at once, or chunk by chunk."""
import ctypes as C, json, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from image_amd.device import DeviceDetector
T, S = int(os.environ.get("TILES", "64")), 4096
det = DeviceDetector(0)
s2 = torch.cuda.Stream()
with torch.cuda.stream(s2):
    det2 = DeviceDetector(0)
tiles = torch.empty((T, S, S, 3), dtype=torch.uint8, device="cuda")
for t in range(T):
    tiles[t] = det.synth_frames(3, S, S, seed0=3 * (3 + t)).permute(1, 2, 0)
nr, nc = C.c_int(), C.c_int()
det.lib.imgfd_fhog_size(S, S, 8, 1, 1, C.byref(nr), C.byref(nc))
hog = torch.empty((T, 31, nc.value, nr.value), dtype=torch.float32, device="cuda")
feat = torch.zeros((T, 1000, 70), dtype=torch.float64, device="cuda")
counts = torch.zeros((T,), dtype=torch.int64, device="cuda")
torch.cuda.synchronize()

def serial():
    det.fhog(tiles, hog); det.surf(tiles, feat, counts, redo=False)

def beside(chunk):
    def run():
        main = torch.cuda.current_stream()
        s2.wait_stream(main)
        for a in range(0, T, chunk):
            b = min(T, a + chunk)
            with torch.cuda.stream(s2):
                det2.fhog(tiles[a:b], hog[a:b])
            det.surf(tiles[a:b], feat[a:b], counts[a:b], redo=False)
        main.wait_stream(s2)
    return run

def timed(fn, reps=int(os.environ.get("REPS", "3"))):
    for _ in range(2): fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(reps): fn()
    e1.record(); e1.synchronize()
    return round(e0.elapsed_time(e1) / reps / T, 4)

out = {"tiles": T, "serial_ms_per_tile": timed(serial)}
ref_hog, ref_feat, ref_counts = hog.clone(), feat.clone(), counts.clone()
for chunk in (T, 64, 16):
    hog.zero_(); feat.zero_(); counts.zero_()
    out[f"beside_chunk{chunk}_ms_per_tile"] = timed(beside(chunk))
    out[f"beside_chunk{chunk}_same_bits"] = bool(torch.equal(hog, ref_hog) and torch.equal(counts, ref_counts) and torch.equal(feat, ref_feat))
out["gpixel_per_s"] = {k: round(S * S / v / 1e6, 1) for k, v in out.items() if k.endswith("ms_per_tile")}
print(json.dumps(out))
