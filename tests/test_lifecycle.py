"""Contexts and streams come and go without leaving device memory behind (GPU only): every imgfd_ctx_destroy /
imgfd_stream_close gives back the workspace, the companion contexts of the two-stream and four-lane paths, the fHOG gradient
table, the events, the pinned staging buffers."""
import numpy as np
import pytest

from image_amd import synth

pytestmark = pytest.mark.gpu


def _free_bytes():
    import torch
    torch.cuda.synchronize()
    return torch.cuda.mem_get_info(0)[0]


def test_context_cycles_leave_no_device_memory_behind():
    import torch
    from image_amd import api, _lib, framestream
    gray = synth.frame(1, 640, 480)
    rgb = np.ascontiguousarray(synth.frame_rgb(2, 512, 384).transpose(2, 1, 0)).astype(np.int32)   # R's (3, width, height)
    frames = np.stack([synth.frame(3 + f, 640, 480) for f in range(4)])

    def cycle():
        ctx = _lib.Context(0)
        try:
            api.image_harris(gray.T.astype(np.float64), ctx=ctx)
            api.image_detect_corners(gray.T, ctx=ctx)
            api.image_canny_edge_detector(gray.T, ctx=ctx)
            api.image_fhog(rgb, ctx=ctx)
            api.image_surf(rgb, ctx=ctx)
            with framestream.FrameStream(640, 480, batch=4, ctx=ctx, corner_cap=1024, point_cap=1024, keep_edges=True) as fs:
                fs.submit(frames)
                assert fs.collect()["n_frames"] == 4
        finally:
            ctx.close()

    for _ in range(3):   # first uses: HIP runtime pools, module loading, torch's own caches
        cycle()
    import psutil
    proc = psutil.Process()
    before, rss0 = _free_bytes(), proc.memory_info().rss
    for _ in range(25):
        cycle()
    after, rss1 = _free_bytes(), proc.memory_info().rss
    assert before - after < 8 << 20, f"{(before - after) / 2**20:.1f} MiB of device memory lost over 25 context cycles"
    # host side: result vectors (imgfd_free), pinned staging buffers, companion contexts
    assert rss1 - rss0 < 64 << 20, f"{(rss1 - rss0) / 2**20:.1f} MiB of host memory gained over 25 context cycles"


def test_contexts_on_concurrent_host_threads():
    """a context is thread-compatible: four host threads, each with its own context, run the five functions at once (ctypes
    drops the GIL inside the calls) -- every result as if it had run alone"""
    import threading
    import oracle
    from image_amd import api, _lib
    imgs = [synth.frame(20 + t, 320 + 16 * t, 240) for t in range(4)]
    rgbs = [synth.frame_rgb(30 + t, 264, 200 + 8 * t) for t in range(4)]
    errors = []

    def work(t):
        try:
            ctx = _lib.Context(0)
            ctx.set_fir_mode(0)
            for rep in range(3):
                g = imgs[t]
                h = api.detect_corners(g.astype(np.float64), g.shape[1], g.shape[0], threshold=1.0, gaussian=1, precision=1, verbose=0, ctx=ctx)
                ref = oracle.harris(g.astype(np.float32), threshold=1.0, gaussian=1, precision=1)
                assert np.array_equal(np.asarray(h["x"], np.float32), ref[:, 0]) and np.array_equal(np.asarray(h["strength"], np.float32), ref[:, 2])
                c = api.image_canny_edge_detector(g.T, ctx=ctx)
                assert c["pixels_nonzero"] == oracle.canny(g)[1]
                f = api.image_detect_corners(g.T, threshold=20, suppress_non_max=True, ctx=ctx)
                assert len(f["x"]) == len(oracle.fast9(g, 20, True))
                x = np.ascontiguousarray(rgbs[t].transpose(2, 1, 0)).astype(np.int32)
                fh = api.image_fhog(x, ctx=ctx)
                assert np.array_equal(fh["fhog"].astype(np.float32).view(np.uint32), oracle.fhog(rgbs[t], 8, 1, 1).view(np.uint32))
            ctx.close()
        except Exception as e:   # noqa: BLE001 -- reported by the main thread
            errors.append((t, repr(e)))

    threads = [threading.Thread(target=work, args=(t,)) for t in range(4)]
    for th in threads: th.start()
    for th in threads: th.join()
    assert not errors, errors
