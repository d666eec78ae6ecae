"""examples/stream_counts.c: the C ABI used from plain C99 (no C++, no Python).  It must compile and link against
libimgfd.so with gcc everywhere; on a GPU box it is run and its counts are compared with the Python mirror."""
import os
import re
import subprocess

import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
SRC = os.path.join(ROOT, "examples", "stream_counts.c")
LIB = os.path.join(ROOT, "image_amd", "libimgfd.so")


def _build(out, src=SRC, extra=()):
    cmd = ["gcc", "-std=c99", "-O2", "-Wall", "-Werror", *extra, "-I", os.path.join(ROOT, "include"), src, "-o", out, LIB,
           "-Wl,-rpath," + os.path.join(ROOT, "image_amd"), "-Wl,-rpath-link,/opt/rocm/lib"]
    r = subprocess.run(cmd, capture_output=True, text=True)
    assert r.returncode == 0, r.stderr


def test_example_compiles_as_c99_and_links(tmp_path):
    if not os.path.exists(LIB):
        pytest.skip("libimgfd.so not built yet")
    _build(str(tmp_path / "stream_counts"))


@pytest.mark.gpu
def test_example_runs_and_agrees_with_the_batch_entry_points(tmp_path):
    exe = str(tmp_path / "stream_counts")
    _build(exe)
    r = subprocess.run([exe, "320", "200", "7"], capture_output=True, text=True, timeout=120)
    assert r.returncode == 0, r.stderr
    rows = re.findall(r"frame (\d+): harris (\d+) fast9 (\d+) canny (\d+)", r.stdout)
    assert [int(a) for a, *_ in rows] == list(range(7))
    # the same frames through the Python mirror of the batch entry points
    import backends
    be = backends.GpuBackend()

    def draw(f, nx=320, ny=200):
        y, x = np.mgrid[0:ny, 0:nx]
        p = (((x + y) >> 3) & 31).astype(np.uint8)
        for k in range(12):
            x0, y0 = (37 * k + 11 * f) % (nx - 40), (53 * k + 7 * f) % (ny - 30)
            p[y0:y0 + 20 + k, x0:x0 + 30 + k] = 90 + 12 * k
        return p
    frames = np.stack([draw(f) for f in range(7)])
    _, hc = be.harris_dev(frames)
    _, fc = be.fast9_dev(frames, 50, False)
    _, cc = be.canny_dev(frames)
    for f, (_, h, f9, c) in enumerate(rows):
        assert (int(h), int(f9), int(c)) == (int(hc[f]), int(fc[f]), int(cc[f])), f


MULTI = os.path.join(ROOT, "examples", "multi_gpu_counts.c")


def test_multi_gpu_example_compiles_as_c99_and_links(tmp_path):
    if not os.path.exists(LIB):
        pytest.skip("libimgfd.so not built yet")
    _build(str(tmp_path / "multi_gpu_counts"), MULTI, ("-pthread",))


@pytest.mark.gpu
def test_multi_gpu_example_totals_equal_the_single_stream_example(tmp_path):
    """examples/multi_gpu_counts.c shards the frames over every visible device (one context + one frame stream + one thread per
    device) and sums the counts on the host: the totals equal the sum of examples/stream_counts.c's per-frame lines"""
    one, many = str(tmp_path / "stream_counts"), str(tmp_path / "multi_gpu_counts")
    _build(one); _build(many, MULTI, ("-pthread",))
    r1 = subprocess.run([one, "320", "200", "9"], capture_output=True, text=True, timeout=120)
    r2 = subprocess.run([many, "320", "200", "9"], capture_output=True, text=True, timeout=120)
    assert r1.returncode == 0 and r2.returncode == 0, (r1.stderr, r2.stderr)
    rows = re.findall(r"frame (\d+): harris (\d+) fast9 (\d+) canny (\d+)", r1.stdout)
    want = [sum(int(r[k]) for r in rows) for k in (1, 2, 3)]
    m = re.search(r"total over (\d+) device\(s\), 9 frames: harris (\d+) fast9 (\d+) canny (\d+)", r2.stdout)
    assert m and [int(m.group(k)) for k in (2, 3, 4)] == want, (r2.stdout, want)
