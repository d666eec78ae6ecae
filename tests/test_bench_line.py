"""bench.py's reporting helpers (no device): the trailing summary the driver's tail always holds, the hash-tied traffic files, and -- through
the C ABI on the test backend -- the shader-clock probe and the explicit second step of imgfd_surf_dev."""
import ctypes as C
import json
import os
import sys

import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import bench  # noqa: E402


def test_line_summary_holds_every_configuration_and_verdict():
    par0 = {"harris_coordinates_match": True, "harris_strengths_differing_in_any_bit": 0, "fast9_coordinates_match": True, "canny_mismatching_pixels": 0}
    res = {"value": 70000.0, "ms_per_step": 150.0,
           "config": {"shader_clock": {"mean_GHz": 2.2}},
           "roofline": {"frac": 0.47, "avg_launch_us": 1400.0, "shader_clock_GHz": 1.8, "traffic": 5, "in_pipeline": {"avg_launch_us": 1230.0}},
           "cpu_baseline": {"value": 7.1, "cores": 32, "reference_only_value": 100.0, "parity_frame0": par0,
                            "parity_frames": {"16": par0, "31": dict(par0, canny_mismatching_pixels=2)}},
           "configs": {"2_batch1": {"value": 46000.0, "ms_per_step": 9.0, "config": {"passes_per_step": 50}},
                       "3": {"value": 133000.0, "ms_per_step": 16.0, "cpu_baseline": {"parity_sample": {"frames_checked": 11, "canny_mismatching_pixels_total": 0}}},
                       "4": {"value": 61000.0, "ms_per_step": 70.0, "roofline": {"surf": {"ms_per_tile": 0.2}, "fhog_ms_per_tile": 0.07},
                             "cpu_baseline": {"parity_sample": {"tiles_checked": 3, "fhog_bit_equal": True, "surf_points_equal": True}}},
                       "5": {"error": "RuntimeError: out of memory"},
                       "5_h2d": {"value": 53000.0, "frac_of_pcie_floor": 0.93},
                       "host_api": {"calls": {"imgfd_surf_i32": {"ms_best": 5.2, "floor_over_best": 0.7}}}}}
    s = bench.line_summary(res)
    assert s["default"] == {"Mpx_s": 70000.0, "ms_step": 150.0, "clock_GHz": 2.2}
    assert s["k3"]["frac"] == 0.47 and s["k3"]["clock_GHz"] == 1.8 and s["k3"]["pipe_us"] == 1230.0
    assert s["2_batch1"]["ms_frame"] == 0.18 and s["4"]["surf_ms_tile"] == 0.2 and s["5"] == {"error": "RuntimeError: out of memory"}
    assert s["5_h2d"]["of_pcie_floor"] == 0.93 and s["host_api_ms"] == {"surf_i32": [5.2, 0.7]}
    assert s["parity"]["default"] == {"frames": 3, "harris_xy": True, "harris_R_bits_differing": 0, "fast9_xy": True, "canny_px_differing": 2}
    assert s["parity"]["3"]["frames_checked"] == 11 and s["parity"]["4"]["fhog_bit_equal"] is True
    assert len(json.dumps(s)) < 2000   # short enough for the tail of stdout the driver keeps


def test_traffic_files_are_tied_to_the_kernel_sources(tmp_path, monkeypatch):
    """a traffic number measured on other kernel sources is not reported: profiles/*.json carry the SHA-1 of the sources they were
    measured on, and bench.py answers None when it differs"""
    prof = tmp_path / "profiles"; prof.mkdir()
    csrc = tmp_path / "image_amd" / "csrc"; csrc.mkdir(parents=True)
    for f in set(sum(bench.FUNCTION_SOURCES.values(), ())) | {"fir_tensor.hip", "fir_tensor_device.h", "fir_device.h"}:
        (csrc / f).write_text("// " + f)
    monkeypatch.setattr(bench, "ROOT", str(tmp_path))
    good = bench.kernel_source_hash(bench.FUNCTION_SOURCES["surf"])
    (prof / "function_traffic.json").write_text(json.dumps({"surf": {"traffic_bytes_per_unit": 1000.0, "kernel_source_sha1": good},
                                                            "fhog": {"traffic_bytes_per_unit": 10.0, "kernel_source_sha1": "stale"}}))
    assert bench.function_traffic("surf", 256) == 256000 and bench.function_traffic("fhog", 4) is None and bench.function_traffic("canny", 1) is None
    (csrc / "surf.hip").write_text("// edited")
    assert bench.function_traffic("surf", 256) is None
    (prof / "k3_traffic.json").write_text(json.dumps({"kernel_source_sha1": bench.kernel_source_hash(), "batch": 32, "traffic_bytes_per_launch": 64, "traffic_bytes_per_frame": 2.0}))
    assert bench.traffic_for("fir_tensor", 32) == 64 and bench.traffic_for("fir_tensor", 8) == 16
    (csrc / "fir_tensor.hip").write_text("// edited")
    assert bench.traffic_for("fir_tensor", 32) is None


def test_committed_traffic_files_match_the_tree():
    """the files bench.py will read on the GPU box were measured on the sources that are in the tree"""
    tr = json.load(open(os.path.join(ROOT, "profiles", "function_traffic.json")))
    for name, files in bench.FUNCTION_SOURCES.items():
        assert tr[name]["kernel_source_sha1"] == bench.kernel_source_hash(files), f"profiles/function_traffic.json: {name} is stale (scripts/gpu_pmc_functions.sh)"
    k3 = json.load(open(os.path.join(ROOT, "profiles", "k3_traffic.json")))
    assert k3["kernel_source_sha1"] == bench.kernel_source_hash(), "profiles/k3_traffic.json is stale (scripts/gpu_pmc_k3.sh)"


def test_clock_probe_through_the_c_abi(be):
    """imgfd_clock_probe queues one wavefront on a stream of its own; imgfd_clock_probe_read waits for the probes queued since the
    last read: mean / min / max GHz and their number (the emulator's two counters tick 21 : 1, the device's at the shader clock)"""
    a, b, c, n = C.c_double(0), C.c_double(0), C.c_double(0), C.c_int(0)
    assert be.lib.imgfd_clock_probe_read(be.ctx, C.byref(a), C.byref(b), C.byref(c), C.byref(n)) == 0 and n.value == 0   # none queued: zeros
    assert be.lib.imgfd_clock_probe(be.ctx, 0) == 1 and be.lib.imgfd_clock_probe(be.ctx, 200000) == 1            # span out of range
    for span in (50, 200, 200):
        assert be.lib.imgfd_clock_probe(be.ctx, span) == 0
    assert be.lib.imgfd_clock_probe_read(be.ctx, C.byref(a), C.byref(b), C.byref(c), C.byref(n)) == 0
    assert n.value == 3 and 0.5 < b.value <= a.value <= c.value < 3.5, (a.value, b.value, c.value)
    assert be.lib.imgfd_clock_probe_read(be.ctx, C.byref(a), C.byref(b), C.byref(c), C.byref(n)) == 0 and n.value == 0   # the read reset them


def test_surf_dev_redo_argument_errors(be):
    d = be.to_dev(np.zeros((1, 64, 64, 3), np.uint8))
    feat = be.empty((1, 8, 70), np.float64); cnt = be.empty((1,), np.int64)
    ok = (be.ctx, be.ptr(d), 1, 64, 64, 64 * 64 * 3, 8, 30.0, be.ptr(feat), 8, be.ptr(cnt))
    assert be.lib.imgfd_surf_dev_redo(*ok, None) == 0                              # nothing queued, counts 0: nothing to redo
    assert be.lib.imgfd_surf_dev_redo(be.ctx, None, 1, 64, 64, 64 * 64 * 3, 8, 30.0, be.ptr(feat), 8, be.ptr(cnt), None) == 1
    assert be.lib.imgfd_surf_dev_redo(be.ctx, be.ptr(d), 1, 0, 64, 64 * 64 * 3, 8, 30.0, be.ptr(feat), 8, be.ptr(cnt), None) == 1
    assert b"imgfd_surf_dev_redo" in (be.lib.imgfd_last_error(be.ctx) or b"")
    n = C.c_int(-1)
    assert be.lib.imgfd_surf_dev_redo(be.ctx, be.ptr(d), 0, 64, 64, 64 * 64 * 3, 8, 30.0, be.ptr(feat), 8, be.ptr(cnt), C.byref(n)) == 0 and n.value == 0
