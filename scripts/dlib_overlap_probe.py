"""fHOG and SURF of the same tiles one after the other on one context (what bench.py --config 4 times) against fHOG on a second
context while SURF runs on the first: per tile, whole-call timings (wall clock around a device-wide wait)."""
import ctypes as C, json, os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from image_amd.device import DeviceDetector
T, S = int(os.environ.get("TILES", 32)), 4096
det, det2 = DeviceDetector(0), DeviceDetector(0)
tiles = torch.empty((T, S, S, 3), dtype=torch.uint8, device="cuda")
for t in range(T):
    tiles[t] = det.synth_frames(3, S, S, seed0=3 * (3 + t)).permute(1, 2, 0)
nr, nc = C.c_int(), C.c_int()
det.lib.imgfd_fhog_size(S, S, 8, 1, 1, C.byref(nr), C.byref(nc))
hog = torch.empty((T, 31, nc.value, nr.value), dtype=torch.float32, device="cuda")
feat = torch.zeros((T, 1000, 70), dtype=torch.float64, device="cuda")
counts = torch.zeros((T,), dtype=torch.int64, device="cuda")
torch.cuda.synchronize()


def one_after_the_other():
    det.fhog(tiles, hog)
    det.surf(tiles, feat, counts, max_points=1000, threshold=30.0)


def side_by_side():
    det2.fhog(tiles, hog)
    det.surf(tiles, feat, counts, max_points=1000, threshold=30.0)


def only_surf():
    det.surf(tiles, feat, counts, max_points=1000, threshold=30.0)


def only_fhog():
    det.fhog(tiles, hog)


out = {}
for name, fn in (("one_after_the_other", one_after_the_other), ("side_by_side", side_by_side), ("only_surf", only_surf), ("only_fhog", only_fhog),
                 ("side_by_side_again", side_by_side), ("one_after_the_other_again", one_after_the_other)):
    fn(); torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(3): fn()
    torch.cuda.synchronize()
    out[name + "_ms_per_tile"] = round((time.perf_counter() - t0) * 1e3 / 3 / T, 4)
out["surf_points"] = int(counts.sum())
print(json.dumps(out))
