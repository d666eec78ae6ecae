#!/usr/bin/env python3
"""bench.py -- benchmarks of the MI355X feature-detection backend (driver contract: see the task description).

Default (no --config): BASELINE.json's metric, Mpixels/s of Harris + FAST-9 + Canny on 3840x2160 gray frames.
A *step* is `--inner` passes of the hot path -- image_harris() defaults, FAST-9 (threshold 20, non-max suppression) and
Canny (s=2, 3/10, accGrad) -- over one batch of `--batch` synthetic frames that is already resident in HBM (generated on
the device by imgfd_synth_frames) when the timed region starts.

  --config 2   configs[1] (+ Canny): the default above; `--batch 1` gives the literal single-frame case
  --config 3   configs[2]: image_canny_edge_detector() on a 1024-frame 1920x1080 batch
  --config 4   configs[3]: image.dlib fHOG + SURF on 4096x4096 RGB tiles, batch 256
  --config 5   configs[4]: 10 000-frame 3840x2160 stream, Harris + Canny, sharded over the ranks (frames pre-staged in
               HBM, per-frame counts gathered, a sample re-checked against the oracle)

N>1: `--gpus N` without WORLD_SIZE in the environment launches N ranks itself (one process per GPU, rank r on GPU r);
under torchrun (WORLD_SIZE set) the process is one of the ranks.  Every rank owns its own frames (weak scaling, no
data-path collective); RCCL only gathers feature counts after the timed region and takes the max of the elapsed time.

Prints ONE JSON line on rank 0, including
  roofline     -- the Harris structure-tensor kernel (20 algorithmic B/px: reads Ix, Iy, writes A, B, C), timed with
                  HIP events in this very run; configs 3/4 report the whole-function entries of SURVEY.md 8(d)
  cpu_baseline -- the reference's own code (oracle/_ref) timed on this host, rank 0, N=1 only
"""
import argparse
import json
import os
import socket
import subprocess
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

HBM_PEAK_GBS = 8000.0           # MI355X_MICROARCH.md: 8 TB/s spec
MEASURED_COPY_GBS = 6300.0      # float4 device-to-device copy on this part (profiles/r01/ubench.txt; the guide quotes the same)
F64_LANE_OPS_PER_S = 34.0e12    # sustained f64 VALU lane-op/s of the add+fmac pattern at 8 waves/SIMD (profiles/r01/ubench2.txt)
F64_LANE_OPS_HW = 256 * 4 * 16 * 2.4e9   # the hardware's issue peak: 256 CUs x 4 SIMDs x 16 lanes x 2.4 GHz = 39.3 T lane-op/s
RESPONSE_BYTES_PER_PX = 12.25   # the kernel the product launches: reads Ix, Iy (8 B), writes R (4 B) and a threshold byte per 4 pixels
TENSOR_BYTES_PER_PX = 20        # read Ix,Iy (8 B), write A,B,C (12 B)
TENSOR_F64_OPS_PER_PX = 90      # 2 passes x 3 planes x (1 mul + 7 add + 7 fma), the reference's own arithmetic
TENSOR_F64_RATE_OPS_PER_PX = 107.25   # + the f32<->f64 conversions of the two register windows (3 planes x (30 + 16 + 30 + 16) / 16): they issue at the f64 rate (ubench7)
F64_LANE_OPS_3_WAVES = 27.0e12  # what three waves per SIMD -- the kernel's occupancy: its 30-value double windows need 150-168 registers -- issue of ANY f64
                                # instruction with 8 independent destinations each: 25.4-29.0 T lane-op/s (profiles/r04/ubench7_f64_instruction_rates.txt)


# ------------------------------------------------------------------------------------------------ CPU legs
def _median(fn, reps=5):
    """one warm-up run (the first run of a binary is several times slower in a VM), then the median of >= 5"""
    fn()
    ts = []
    for _ in range(reps):
        t = time.perf_counter(); fn(); ts.append(time.perf_counter() - t)
    ts.sort()
    return ts[len(ts) // 2]


def _avail_cores():
    return len(os.sched_getaffinity(0)) if hasattr(os, "sched_getaffinity") else (os.cpu_count() or 1)


def cpu_harris_fast9_canny(img, want_fast9=True, gpu_frame0=None):
    """Reference CPU path on this host for one gray frame: Harris = reference sources + OpenMP (best of a few team
    sizes), FAST-9 = reference f9.cpp (single-threaded code), Canny = oracle restatement (the reference's blur needs
    FFTW3, absent here; the restatement is pinned against the reference sources over a stand-in DFT in
    tests/test_oracle.py, but that build's O(n^3) DFT is no timing baseline).
    gpu_frame0: the device results for the same frame: the CPU outputs computed here double as the metric's
    "feature-coordinate match vs CPU" check."""
    import numpy as np

    import oracle
    ny, nx = img.shape
    avail = _avail_cores()
    out = {"unit": "Mpixels/s"}
    px = img.size
    have_ref = oracle.have_ref("harris") and oracle.have_ref("f9")
    f32 = img.astype(np.float32)
    cores = 1
    t_f = 0.0
    if have_ref:
        t_h = None
        sweep = {}
        for th in sorted({min(avail, 64), min(avail, 32), min(avail, 16), min(avail, 8), avail}):   # nproc threads too
            t = _median(lambda: oracle.ref_harris(f32, threads=th))
            sweep[th] = round(1e3 * t, 2)
            if t_h is None or t < t_h:
                t_h, cores = t, th
        t_h1 = _median(lambda: oracle.ref_harris(f32, threads=1))   # state the single-thread time too
        if want_fast9:
            t_f = _median(lambda: oracle.ref_fast9(img, 20, True))
        kind = "reference+restatement(canny)"   # Harris and FAST-9: the reference's own sources; Canny: the pinned restatement (the reference needs FFTW3)
    else:
        sweep = None
        t_h = _median(lambda: oracle.harris(f32))
        if want_fast9:
            t_f = _median(lambda: oracle.fast9(img, 20, True))
        kind = "port"
    parts = {"harris_ms": round(1e3 * t_h, 2)}
    if want_fast9:
        parts["fast9_ms"] = round(1e3 * t_f, 2)
    if have_ref:
        parts["harris_1_thread_ms"] = round(1e3 * t_h1, 2)
        parts["harris_ms_by_threads"] = sweep
    # Canny: the reference multiplies three 2-D FFTs (tools.c:166-185, FFTW3 -- absent here); the leg quoted beside the GPU is
    # that ALGORITHM (numpy's pocketfft for the blur, the restatement's stages behind it); the restatement's own direct 25-tap
    # f64 convolution is timed next to it (VERDICT r04, weak point 8)
    t_c_direct = _median(lambda: oracle.canny(img))
    t_c = _median(lambda: oracle.canny_fft(img))
    parts["canny_fft_blur_ms"] = round(1e3 * t_c, 2)
    parts["canny_restatement_direct_convolution_ms"] = round(1e3 * t_c_direct, 2)
    if gpu_frame0 is not None:
        out["parity_frame0"] = parity_of_frame(img, gpu_frame0, threads=cores)
    total = t_h + t_f + t_c
    out.update({"value": round(px / total / 1e6, 3), "kind": kind, "cores": cores,
                "reference_only_value": round(px / (t_h + t_f) / 1e6, 3),   # top level: the driver's parsed line keeps it
                "reference_only": {"value": round(px / (t_h + t_f) / 1e6, 3), "unit": "Mpixels/s",
                                   "what": "Harris" + (" + FAST-9" if want_fast9 else "") + ": the reference's own sources only (no restated leg)"},
                "sample": f"1 frame {nx}x{ny}, median of 5 after a warm-up; Harris: reference src + OpenMP x{cores} (the fastest of 8/16/32/64/{avail} threads, "
                          f"{avail} available); " + ("FAST-9: reference f9.cpp (1 thread); " if want_fast9 else "") +
                          "Canny: the reference's algorithm -- FFT-product blur through numpy's pocketfft (the reference needs FFTW3, absent here) + the "
                          "restated gradient / maxima / hysteresis stages (pinned against the reference sources), 1 thread",
                "parts": parts})
    return out


def parity_of_frame(img, gpu, threads=1):
    """the metric's "feature-coordinate match vs CPU" for one gray frame: gpu = (corner list, FAST-9 point list or None, edge map)
    as the device left them; the CPU side is the reference's own code (Harris, FAST-9) and the pinned restatement (Canny)"""
    import numpy as np

    import oracle
    have_ref = oracle.have_ref("harris") and oracle.have_ref("f9")
    f32 = img.astype(np.float32)
    rh = oracle.ref_harris(f32, threads=threads) if have_ref else oracle.harris(f32)
    gh, gf, ge = gpu
    same_h = gh.shape == rh.shape and bool(np.array_equal(gh[:, :2], rh[:, :2]))
    par = {"harris_corners": int(len(rh)), "harris_coordinates_match": same_h,
           # the timed mode (fir_mode 1: fused f64 accumulate) promises coordinates + 1e-4; how many strengths of this sample differ in a bit at all
           "harris_strengths_differing_in_any_bit": int(np.count_nonzero(gh[:, 2].astype(np.float32).view(np.uint32) != rh[:, 2].astype(np.float32).view(np.uint32))) if same_h else None,
           "harris_strength_max_rel_err": float(np.max(np.abs(gh[:, 2] - rh[:, 2]) / np.maximum(1.0, np.abs(rh[:, 2])))) if same_h and len(rh) else None}
    if gf is not None:
        rf = oracle.ref_fast9(img, 20, True) if have_ref else oracle.fast9(img, 20, True)
        par.update({"fast9_corners": int(len(rf)), "fast9_coordinates_match": bool(gf.shape == rf.shape and np.array_equal(gf, rf))})
    re_, rn = oracle.canny(img)
    par.update({"canny_edge_pixels": int(rn), "canny_mismatching_pixels": int(np.count_nonzero(ge != re_))})
    return par


# ------------------------------------------------------------------------------------------------ workloads
class Workload:
    """One configuration of BASELINE.json: resident inputs, a step, counts, parity sample, roofline, CPU leg."""
    metric = ""
    unit = "Mpixels/s"

    def __init__(self, args, det, rank, world):
        self.args, self.det, self.rank, self.world = args, det, rank, world

    def default_steps(self):
        return 20


class Detect4K(Workload):
    """configs[1] + Canny (the metric's workload), or configs[4] (stream=True: Harris + Canny, every frame once)."""
    NX, NY = 3840, 2160

    def __init__(self, args, det, rank, world, stream_mode=False):
        super().__init__(args, det, rank, world)
        self.stream_mode = stream_mode
        self.metric = "Mpixels/s Harris+FAST9+Canny on 3840x2160 gray" if not stream_mode else \
            "Mpixels/s Harris+Canny on a 3840x2160 synthetic stream (configs[4])"

    def prepare(self):
        import torch

        from image_amd import stream
        a, det = self.args, self.det
        NX, NY = self.NX, self.NY
        self.B = B = a.batch if a.batch else 32
        if self.stream_mode:
            self.first, self.n_local = stream.rank_block(a.frames, self.rank, self.world)
            self.inner = 1
        else:
            self.first, self.n_local = stream.rank_block(B * self.world, self.rank, self.world)  # weak scaling: B frames per rank
            self.inner = a.inner
        gen = 64
        self.frames = torch.empty((self.n_local, NY, NX), dtype=torch.uint8, device="cuda")
        for f0 in range(0, self.n_local, gen):   # pre-stage every frame of this rank in HBM (untimed)
            n = min(gen, self.n_local - f0)
            self.frames[f0:f0 + n] = det.synth_frames(n, NX, NY, seed0=stream.frame_seed(50000, self.first + f0))
        self.h2d = bool(getattr(a, "h2d", False))
        if self.h2d:
            # delivery included: a ring of up to 4 batches in pinned host memory (distinct frames; a longer stream re-reads the
            # ring -- what is measured is the rate of upload + kernels, the per-frame counts are those of the ring's frames), two
            # device input buffers, a copy stream of its own
            # the whole share of this rank when it fits 12 GB of pinned memory (a 10 000-frame stream on 8 ranks: 10.4 GB each),
            # else a ring of four batches that a longer stream re-reads (the rate is upload + kernels either way; with a re-read ring
            # the per-frame counts are the ring's, and the sampled parity check compares only frames that were streamed as themselves)
            ring = self.n_local if self.n_local * NX * NY <= (12 << 30) else min(4 * B, self.n_local)
            self.host_ring = torch.empty((ring, NY, NX), dtype=torch.uint8, pin_memory=True)
            self.host_ring.copy_(self.frames[:ring])
            self.dev_in = [torch.empty((B, NY, NX), dtype=torch.uint8, device="cuda") for _ in range(2)]
            self.copy_stream = torch.cuda.Stream()
            self.ev_copied = [torch.cuda.Event() for _ in range(2)]
            self.ev_done = [torch.cuda.Event() for _ in range(2)]
            self.h2d_pass = 0
            self.h2d_pending = None
        self.cap_h, self.cap_f = (0, 0) if self.stream_mode else (65536, 262144)
        dev = "cuda"
        self.corners = torch.empty((B, max(1, self.cap_h), 3), dtype=torch.float32, device=dev)
        self.points = torch.empty((B, max(1, self.cap_f), 2), dtype=torch.int32, device=dev)
        self.edges = torch.empty((B, NY, NX), dtype=torch.uint8, device=dev)
        self.counts = torch.zeros((3, B), dtype=torch.int64, device=dev)
        self.frame_counts = torch.zeros((2, max(1, self.n_local)), dtype=torch.int64, device=dev)  # stream mode: harris, canny per frame
        self.cursor = 0
        self._counts_n = {}
        self.params = dict(fast9_threshold=20, suppress_non_max=1)
        if self.stream_mode:
            self.params.update(fast9=0)

    def default_steps(self):
        return -(-self.n_local // self.B) if self.stream_mode else 20

    def px_per_step(self):
        return self.inner * self.B * self.NX * self.NY   # stream mode: the last batch may be short; total handled in px_total

    def px_total(self, steps):
        if self.stream_mode:
            return min(self.n_local, steps * self.B) * self.NX * self.NY
        return steps * self.px_per_step()

    # ---- --h2d: batch `f0 .. f0+n` of the host ring into device buffer `slot`, on the copy stream, behind the kernels that last read that buffer
    def _upload(self, slot, f0, n):
        import torch
        ring = self.host_ring.shape[0]
        with torch.cuda.stream(self.copy_stream):
            self.copy_stream.wait_event(self.ev_done[slot])
            k = 0
            while k < n:   # the ring may wrap inside a batch
                r0 = (f0 + k) % ring
                m = min(n - k, ring - r0)
                self.dev_in[slot][k:k + m].copy_(self.host_ring[r0:r0 + m], non_blocking=True)
                k += m
            self.ev_copied[slot].record(self.copy_stream)

    def _delivered(self, f0, n, next_f0=None, next_n=0):
        """the device buffer holding frames f0 .. f0+n of the stream (uploaded if nobody queued it yet), with the NEXT batch's
        upload queued before this batch's kernels so that the two overlap"""
        import torch
        p = self.h2d_pass
        if self.h2d_pending != (p, f0, n):
            self._upload(p % 2, f0, n)
        if next_n > 0:
            self._upload((p + 1) % 2, next_f0, next_n)
            self.h2d_pending = (p + 1, next_f0, next_n)
        torch.cuda.current_stream().wait_event(self.ev_copied[p % 2])
        return self.dev_in[p % 2][:n]

    def _consumed(self):
        import torch
        self.ev_done[self.h2d_pass % 2].record(torch.cuda.current_stream())
        self.h2d_pass += 1

    def reset(self):
        self.cursor = 0
        # BASELINE.md section 3 asks for a median over >= 100 event-timed iterations: an event pair around EVERY pass of the timed
        # region (800 by default) beside the per-step pairs.  (Not for small batches: a pass of one frame takes 0.23 ms.)
        self.pass_ev = [] if (not self.stream_mode and self.B >= 8) else None

    def step(self):
        det = self.det
        if self.stream_mode:
            f0 = self.cursor
            n = min(self.B, self.n_local - f0)
            if n <= 0:
                return
            c = self._counts_n.get(n)
            if c is None:
                import torch
                c = self._counts_n[n] = torch.zeros((3, n), dtype=torch.int64, device="cuda")
            if self.h2d:
                n2 = min(self.B, self.n_local - (f0 + n))
                src = self._delivered(f0, n, f0 + n, n2)
            else:
                src = self.frames[f0:f0 + n]
            det.detect_all(src, self.corners, self.points, self.edges[:n], c, corner_cap=self.cap_h,
                           point_cap=self.cap_f, **self.params)
            if self.h2d:
                self._consumed()
            self.frame_counts[0, f0:f0 + n] = c[0]
            self.frame_counts[1, f0:f0 + n] = c[2]
            self.cursor += n
            return
        import torch
        pe = getattr(self, "pass_ev", None)
        if pe is not None:
            e0 = torch.cuda.Event(enable_timing=True); e0.record(); pe.append(e0)
        for it in range(self.inner):
            frames = self._delivered(0, self.B, 0, self.B) if self.h2d else self.frames
            if self.args.no_overlap:
                det.harris(frames, out=(self.corners, self.counts[0]))
                det.fast9(frames, threshold=20, suppress_non_max=True, out=(self.points, self.counts[1]))
                det.canny(frames, out=(self.edges, self.counts[2]))
            else:
                det.detect_all(frames, self.corners, self.points, self.edges, self.counts, **self.params)
            if self.h2d:
                self._consumed()
            if pe is not None:
                e1 = torch.cuda.Event(enable_timing=True); e1.record(); pe.append(e1)

    def pass_times_ms(self):
        """durations of the individual passes of the timed region (inner + 1 events per step)"""
        pe = getattr(self, "pass_ev", None)
        if not pe:
            return []
        n = self.inner + 1
        return [pe[i].elapsed_time(pe[i + 1]) for i in range(len(pe) - 1) if (i + 1) % n != 0]

    def count_vector(self):
        """int64 totals of this rank: harris corners, fast9 corners, canny edge pixels"""
        import torch
        if self.stream_mode:
            z = torch.zeros((), dtype=torch.int64, device="cuda")
            return torch.stack([self.frame_counts[0, :self.cursor].sum(), z, self.frame_counts[1, :self.cursor].sum()])
        return self.counts.sum(dim=1)

    def describe(self, counts):
        a = self.args
        c = {"workload": ("configs[1]+Canny: image_harris() defaults + FAST-9 thr 20 nonmax + Canny s=2 3/10 accGrad" if not self.stream_mode else
                          f"configs[4]: {a.frames}-frame stream, image_harris() defaults + Canny s=2 3/10 accGrad, contiguous blocks of frames per rank") +
             f" on {self.NX}x{self.NY} u8 frames " + ("resident in HBM" if not self.h2d else
                                                         f"in pinned host memory (ring of {self.host_ring.shape[0]} frames), every batch uploaded (1 B/px) on a copy stream beside the kernels of the batch before"),
             "delivery": "h2d (PCIe included)" if self.h2d else "resident",
             "frames_per_step_per_gpu": self.B, "passes_per_step": self.inner,
             "schedule": "one stream" if a.no_overlap else "two streams (imgfd_detect_dev)",
             "fir_mode": "fused-accumulate" if a.fir_mode else "strict",
             "feature_counts": {"harris_corners": int(counts[0]), "fast9_corners": int(counts[1]), "canny_edge_pixels": int(counts[2])}}
        if self.stream_mode:
            c["frames_total"] = a.frames
            c["frames_this_rank"] = self.n_local
        return c

    # ---- roofline: the 20 B/px structure-tensor kernel, timed in this run on this batch's own gradients
    def roofline(self, k3_pipe_us, k3_pipe_n, steps):
        import torch
        det, NX, NY = self.det, self.NX, self.NY
        B = min(self.B, self.n_local)
        ix = torch.empty((B, NY, NX), dtype=torch.float32, device="cuda")
        iy = torch.empty_like(ix)
        for f in range(B):
            det.gradients_of(self.frames[f], ix[f], iy[f])
        # warm-up covers the first touch of the freshly allocated A, B, C; a 40 ms clock probe (one wavefront of 6 registers: it fits
        # beside the kernel's 3 x 168 per SIMD) spans the warm-up and the first timed launches: the clock this kernel alone runs at
        planes = tuple(torch.empty_like(ix) for _ in range(3))
        us = det.time_structure_tensor_batch(ix, iy, warmup=12, iters=max(20, min(60, steps)), probe_us=40000, out=planes)
        k3_clock = det.clock_probe_read()
        # The same launch, one at a time with the device idle for 5 ms in front of each: what the kernel takes when the chip is not
        # at the power limit its own back-to-back launches drive it to (an f64 FMA stream clocks an MI355X down by a fifth).  Reported
        # BESIDE the sustained number, never instead of it.
        spaced = []
        for _ in range(8):
            torch.cuda.synchronize(); time.sleep(0.005)
            spaced.append(det.time_structure_tensor_batch(ix, iy, warmup=0, iters=1, probe_us=1000, out=planes))
        spaced_clock = det.clock_probe_read()
        spaced.sort()
        del ix, iy, planes
        k3_bytes = TENSOR_BYTES_PER_PX * NX * NY * B
        achieved = k3_bytes / (us * 1e-6) / 1e9
        f64_floor_us = TENSOR_F64_OPS_PER_PX * NX * NY * B / F64_LANE_OPS_PER_S * 1e6
        r = {"kernel": "fir_tensor (Harris structure-tensor pass: Ix,Iy -> A,B,C)", "bound": "f64-valu",
             "achieved": round(achieved, 1), "peak": HBM_PEAK_GBS, "unit": "GB/s", "frac": round(achieved / HBM_PEAK_GBS, 4),
             "frac_of_measured_copy": round(achieved / MEASURED_COPY_GBS, 4), "measured_copy_GBps": MEASURED_COPY_GBS,
             "frac_of_f64_issue_roof": round(f64_floor_us / us, 4),
             "frac_of_f64_hardware_issue_peak": round(TENSOR_F64_OPS_PER_PX * NX * NY * B / F64_LANE_OPS_HW * 1e6 / us, 4),
             "f64_hardware_issue_peak": "256 CUs x 4 SIMDs x 16 lanes x 2.4 GHz = 39.3 T lane-op/s (34 T measured for dependent add+fmac chains)",
             "f64_issue_roof": f"{TENSOR_F64_OPS_PER_PX} f64 FIR lane-ops/px (the reference's double accumulation) at the {F64_LANE_OPS_PER_S / 1e12:.0f} T lane-op/s "
                               "the vector pipe sustains for add+fmac chains (profiles/r01/ubench2.txt): the kernel is bound by f64 issue, HBM second",
             "f64_floor": {"f64_rate_lane_ops_per_px": TENSOR_F64_RATE_OPS_PER_PX, "lane_ops_per_s_at_the_kernels_occupancy": F64_LANE_OPS_3_WAVES,
                           "floor_us_per_launch": round(TENSOR_F64_RATE_OPS_PER_PX * NX * NY * B / F64_LANE_OPS_3_WAVES * 1e6, 1),
                           "frac": round(TENSOR_F64_RATE_OPS_PER_PX * NX * NY * B / F64_LANE_OPS_3_WAVES * 1e6 / us, 4),
                           "what": "the kernel's f64-rate instructions alone (90 FIR operations + 17.25 conversions per pixel) at the rate three waves per SIMD issue them on this "
                                   "part, measured (scripts/ubench/ubench7.hip); the phase split and the barrier-free experiment behind this number: profiles/r04/k3_phase_split.txt, "
                                   "k3_wave_experiment.txt; DESIGN.md section 4, LOG.md"},
             "avg_launch_us": round(us, 2), "frames_per_launch": B, "algorithmic_bytes_per_launch": k3_bytes,
             "shader_clock_GHz": k3_clock["mean_GHz"],   # while these launches ran (imgfd_clock_probe, 40 ms): tells a slow box from a slow kernel
             "single_launches_from_idle": {"median_us": round(spaced[len(spaced) // 2], 2), "min_us": round(spaced[0], 2), "launches": len(spaced),
                                           "frac": round(k3_bytes / (spaced[len(spaced) // 2] * 1e-6) / 1e9 / HBM_PEAK_GBS, 4), "shader_clock_GHz": spaced_clock["mean_GHz"],
                                           "what": "one launch at a time, 5 ms of idle device in front of each: the kernel below the power limit its own back-to-back launches "
                                                   "reach -- NOT the roofline number (that is the sustained one above)"},
             "timed": "HIP events on the context's stream around back-to-back launches of the stage doorway on this batch's gradients, after the timed region"}
        tr = traffic_for("fir_tensor", B)
        r["traffic"], r["traffic_unit"] = tr, "bytes/launch (PMC FETCH_SIZE x2 + WRITE_SIZE, profiles/k3_traffic.json; refreshed by scripts/gpu_pmc_k3.sh)"
        if k3_pipe_n:
            per = k3_pipe_us / k3_pipe_n
            launches_per_pass = max(1, round(k3_pipe_n / max(1, steps * self.inner)))
            fpl = max(1, self.B // launches_per_pass)
            ach = RESPONSE_BYTES_PER_PX * NX * NY * fpl / (per * 1e-6) / 1e9
            r["in_pipeline"] = {"kernel": det.tensor_kernel_name(), "avg_launch_us": round(per, 2), "launches": k3_pipe_n,
                                "frames_per_launch": fpl, "algorithmic_bytes_per_px": RESPONSE_BYTES_PER_PX,
                                "achieved": round(ach, 1), "unit": "GB/s", "frac": round(ach / HBM_PEAK_GBS, 4),
                                "frac_of_f64_issue_roof": round(TENSOR_F64_OPS_PER_PX * NX * NY * fpl / F64_LANE_OPS_PER_S * 1e6 / per, 4),
                                "note": "the structure-tensor launches of the timed region itself (HIP events on the context's stream)"}
        return r

    def parity_and_cpu(self, want_cpu):
        """frame 0 (default) or a 1 % sample of this rank's frames (stream mode) against the CPU path"""
        import numpy as np

        from image_amd import stream, synth
        if not self.stream_mode:
            def dev(f):   # what the last pass of the timed region left for frame f of the batch
                n_h, n_f = int(self.counts[0, f]), int(self.counts[1, f])
                return (self.corners[f, :min(n_h, self.cap_h)].cpu().numpy(), self.points[f, :min(n_f, self.cap_f)].cpu().numpy(), self.edges[f].cpu().numpy())
            host = synth.frame(stream.frame_seed(50000, self.first), self.NX, self.NY)   # host twin of device frame 0
            assert np.array_equal(self.frames[0].cpu().numpy(), host), "device and host frame generators diverged"
            out = cpu_harris_fast9_canny(host, True, dev(0)) if want_cpu else {"parity_frame0": parity_of_frame(host, dev(0))}
            # frame 0 is not the batch: the middle and the last frame too ("check >= 2 frames of the default batch")
            more = sorted({self.B // 2, self.B - 1} - {0})[: max(0, self.args.max_parity_frames - 1)]
            out["parity_frames"] = {}
            for f in more:
                img = synth.frame(stream.frame_seed(50000, self.first + f), self.NX, self.NY)
                out["parity_frames"][str(f)] = parity_of_frame(img, dev(f), threads=min(16, _avail_cores()))
            return out
        return stream_sample_check(self, want_cpu)


def _oracle_frame(job):
    """worker of the sampled parity check (its own process): oracle outputs of one synthetic frame"""
    seed, nx, ny = job
    import numpy as np

    import oracle
    from image_amd import synth
    img = synth.frame(seed, nx, ny)
    h = oracle.ref_harris(img.astype(np.float32), threads=1) if oracle.have_ref("harris") else oracle.harris(img.astype(np.float32))
    e, n = oracle.canny(img)
    return h, np.packbits(e != 0), int(n)


def stream_sample_check(wl, want_cpu):
    """configs[4]: re-run 1 % of this rank's frames through the device path with full outputs and compare corner lists,
    strengths and edge maps with the oracle (one process per frame on the host's cores)."""
    import concurrent.futures as cf

    import numpy as np
    import torch

    from image_amd import stream
    limit = min(wl.cursor, wl.host_ring.shape[0]) if getattr(wl, "h2d", False) else wl.cursor   # --h2d: frames beyond a re-read ring were not streamed as themselves
    n_s = min(limit, max(1, round(0.01 * limit)), wl.args.max_parity_frames)
    if n_s <= 0:
        return None
    idx = np.unique(np.linspace(0, limit - 1, n_s).round().astype(int))
    cap = 65536
    res = {"frames_checked": int(len(idx)), "frame_indices": [int(wl.first + i) for i in idx[:8]] + (["..."] if len(idx) > 8 else [])}
    t0 = time.perf_counter()
    with cf.ProcessPoolExecutor(max_workers=min(len(idx), max(1, _avail_cores() // 2), 48)) as ex:
        futs = [ex.submit(_oracle_frame, (stream.frame_seed(50000, wl.first + int(i)), wl.NX, wl.NY)) for i in idx]
        # the device side of the sample, while the host works
        corners = torch.empty((1, cap, 3), dtype=torch.float32, device="cuda")
        edges = torch.empty((1, wl.NY, wl.NX), dtype=torch.uint8, device="cuda")
        got = []
        for i in idx:
            (c, n) = wl.det.harris(wl.frames[int(i):int(i) + 1], out=(corners, torch.zeros(1, dtype=torch.int64, device="cuda")))
            (e, m) = wl.det.canny(wl.frames[int(i):int(i) + 1], out=(edges, torch.zeros(1, dtype=torch.int64, device="cuda")))
            got.append((c[0, :int(n[0])].cpu().numpy(), np.packbits(e[0].cpu().numpy() != 0), int(m[0]),
                        int(wl.frame_counts[0, int(i)]), int(wl.frame_counts[1, int(i)])))
        bad_xy = bad_R = bad_edges = bad_counts = 0
        worst_rel = 0.0
        for (gc, ge, gm, sc_h, sc_c), f in zip(got, futs):
            rh, re_, rn = f.result()
            if gc.shape != rh.shape or not np.array_equal(gc[:, :2], rh[:, :2]):
                bad_xy += 1
            elif len(rh):
                rel = float(np.max(np.abs(gc[:, 2] - rh[:, 2]) / np.maximum(1.0, np.abs(rh[:, 2]))))
                worst_rel = max(worst_rel, rel)
                bad_R += rel > 1e-4
            bad_edges += int(np.count_nonzero(np.unpackbits(ge ^ re_)))
            bad_counts += (sc_h != len(rh)) + (sc_c != rn and gm != sc_c)
    res.update({"frames_with_different_corner_coordinates": bad_xy, "frames_with_strength_rel_err_above_1e-4": int(bad_R),
                "harris_strength_max_rel_err": worst_rel, "canny_mismatching_pixels_total": bad_edges,
                "frames_whose_streamed_counts_differ": int(bad_counts), "oracle_seconds": round(time.perf_counter() - t0, 1)})
    out = {"parity_sample": res}
    if want_cpu:
        from image_amd import synth
        cpu = cpu_harris_fast9_canny(synth.frame(stream.frame_seed(50000, wl.first), wl.NX, wl.NY), want_fast9=False)
        out.update(cpu)
    return out


class Canny1080p(Workload):
    """configs[2]: image_canny_edge_detector() on a 1024-frame 1920x1080 batch."""
    NX, NY = 1920, 1080
    metric = "Mpixels/s image_canny_edge_detector() on a 1920x1080 batch (configs[2])"

    def prepare(self):
        import torch

        from image_amd import stream
        a, det = self.args, self.det
        self.F = a.batch if a.batch else 1024
        self.first, _ = stream.rank_block(self.F * self.world, self.rank, self.world)
        self.frames = torch.empty((self.F, self.NY, self.NX), dtype=torch.uint8, device="cuda")
        for f0 in range(0, self.F, 128):
            n = min(128, self.F - f0)
            self.frames[f0:f0 + n] = det.synth_frames(n, self.NX, self.NY, seed0=stream.frame_seed(1000, self.first + f0))
        self.edges = torch.empty_like(self.frames)
        self.counts = torch.zeros((self.F,), dtype=torch.int64, device="cuda")
        self.inner = 1

    def px_per_step(self):
        return self.F * self.NX * self.NY

    def px_total(self, steps):
        return steps * self.px_per_step()

    def reset(self):
        pass

    def step(self):
        self.det.canny(self.frames, out=(self.edges, self.counts))

    def count_vector(self):
        import torch
        z = torch.zeros((), dtype=torch.int64, device="cuda")
        return torch.stack([z, z, self.counts.sum()])

    def describe(self, counts):
        return {"workload": f"configs[2]: image_canny_edge_detector() s=2 3/10 accGrad on {self.F} frames {self.NX}x{self.NY} u8 resident in HBM, one imgfd_canny_dev call per step",
                "frames_per_step_per_gpu": self.F, "passes_per_step": 1, "feature_counts": {"canny_edge_pixels": int(counts[2])}}

    def roofline(self, k3_us, k3_n, steps, ms_per_step=None):
        alg = 2 * self.px_per_step()   # Canny compulsory traffic, u8 in + u8 edge map out
        ach = alg / (ms_per_step * 1e-3) / 1e9
        return {"kernel": "imgfd_canny_dev (whole function: blur, gradient+NMS, hysteresis, expansion)", "bound": "f64-valu",
                "achieved": round(ach, 1), "peak": HBM_PEAK_GBS, "unit": "GB/s", "frac": round(ach / HBM_PEAK_GBS, 4),
                "algorithmic_bytes_per_step": alg, "traffic": function_traffic("canny", self.F * self.NX * self.NY / (3840 * 2160)),
                "traffic_unit": "bytes/step (PMC FETCH_SIZE x2 + WRITE_SIZE over every kernel of imgfd_canny_dev on 4K frames, scaled by the pixels; profiles/function_traffic.json)",
                "note": "2 B/px compulsory traffic over the whole function's time; the f64 blur and gradient arithmetic of the reference bound it, not HBM"}

    def parity_and_cpu(self, want_cpu):
        import numpy as np

        import oracle
        from image_amd import stream, synth
        idx = sorted(set(np.linspace(0, self.F - 1, max(3, -(-self.F // 100))).astype(int).tolist()))[: self.args.max_parity_frames]   # >= 1 % of the batch
        mism = 0
        cnt_ok = True
        ts = []
        for i in idx:
            img = synth.frame(stream.frame_seed(1000, self.first + i), self.NX, self.NY)
            t = time.perf_counter(); e, n = oracle.canny(img); ts.append(time.perf_counter() - t)
            mism += int(np.count_nonzero(self.edges[i].cpu().numpy() != e))
            cnt_ok &= int(self.counts[i]) == n
        out = {"parity_sample": {"frames_checked": len(idx), "canny_mismatching_pixels_total": mism, "pixels_nonzero_equal": bool(cnt_ok)}}
        if want_cpu:
            t = min(ts)
            img = synth.frame(stream.frame_seed(1000, self.first + idx[0]), self.NX, self.NY)
            tf = []
            for _ in range(3):   # the reference's algorithm: FFT-product blur (pocketfft here, FFTW3 there) + the restated stages
                t0 = time.perf_counter(); oracle.canny_fft(img); tf.append(time.perf_counter() - t0)
            out.update({"value": round(self.NX * self.NY / min(tf) / 1e6, 3), "unit": "Mpixels/s", "cores": 1, "kind": "port",
                        "direct_convolution_value": round(self.NX * self.NY / t / 1e6, 3),
                        "sample": f"1 frame {self.NX}x{self.NY}, best of 3: canny_edge_detector() with the reference's algorithm for the blur (three 2-D FFTs, "
                                  f"numpy's pocketfft standing in for the absent FFTW3) + the restated stages, 1 thread as the reference is; direct_convolution_value: "
                                  f"the restatement's own separable f64 convolution, best of {len(idx)} frames"})
        return out


class DlibTiles(Workload):
    """configs[3]: image_fhog() + image_surf() on 4096x4096 RGB tiles, batch 256."""
    S = 4096
    metric = "Mpixels/s image.dlib fHOG + SURF on 4096x4096 RGB tiles (configs[3])"

    def prepare(self):
        import ctypes as C

        import torch

        from image_amd import stream
        a, det = self.args, self.det
        S = self.S = a.tile
        self.T = a.batch if a.batch else 256
        self.first, _ = stream.rank_block(self.T * self.world, self.rank, self.world)
        self.tiles = torch.empty((self.T, S, S, 3), dtype=torch.uint8, device="cuda")
        for t in range(self.T):   # channel c of tile t is G(3*seed + c), seed = 3 + global tile index (image_amd.synth.frame_rgb)
            g = det.synth_frames(3, S, S, seed0=3 * (3 + self.first + t))
            self.tiles[t] = g.permute(1, 2, 0)
        nr, nc = C.c_int(), C.c_int()
        det.lib.imgfd_fhog_size(S, S, 8, 1, 1, C.byref(nr), C.byref(nc))
        self.hog = torch.empty((self.T, 31, nc.value, nr.value), dtype=torch.float32, device="cuda")
        self.cap = 1000
        self.feat = torch.zeros((self.T, self.cap, 70), dtype=torch.float64, device="cuda")
        self.counts = torch.zeros((self.T,), dtype=torch.int64, device="cuda")
        self.ev = [torch.cuda.Event(enable_timing=True) for _ in range(3)]
        self.t_fhog = self.t_surf = 0.0
        self.n_timed = 0
        self.inner = 1

    def px_per_step(self):
        return self.T * self.S * self.S

    def px_total(self, steps):
        return steps * self.px_per_step()

    def reset(self):
        self.t_fhog = self.t_surf = 0.0
        self.n_timed = 0

    def step(self):
        det = self.det
        self.ev[0].record()
        det.fhog(self.tiles, self.hog)
        self.ev[1].record()
        det.surf(self.tiles, self.feat, self.counts, max_points=1000, threshold=30.0)
        self.ev[2].record()
        self.ev[2].synchronize()
        self.t_fhog += self.ev[0].elapsed_time(self.ev[1])
        self.t_surf += self.ev[1].elapsed_time(self.ev[2])
        self.n_timed += 1

    def count_vector(self):
        import torch
        z = torch.zeros((), dtype=torch.int64, device="cuda")
        return torch.stack([self.counts.sum(), z, z])

    def describe(self, counts):
        return {"workload": f"configs[3]: image_fhog() cell 8 padding 1/1 + image_surf() max_points 1000 threshold 30 on {self.T} RGB tiles {self.S}x{self.S} resident in HBM",
                "tiles_per_step_per_gpu": self.T, "passes_per_step": 1, "feature_counts": {"surf_points": int(counts[0])}}

    def roofline(self, k3_us, k3_n, steps, ms_per_step=None):
        px = self.px_per_step()
        n = max(1, self.n_timed)
        tf, ts = self.t_fhog / n, self.t_surf / n
        hog_bytes = 3 * px + self.hog.numel() * 4
        surf_bytes = 43 * px   # RGB in, int32 integral write+read, f64 pyramid write + one read
        # what the code is DESIGNED to move since round 6 (SURF): RGB read twice by the band / strip scans (6), the table
        # written once (4) and read once per pyramid kernel (8), four of six intervals of the f64 pyramid written and read once
        # (2 x 4 x 8 x (1/4 + 1/16 + 1/64 + 1/256) = 21.25): 39.25 B/px -- beside SURVEY's staged 43 with all six intervals
        surf_design = 39.25 * px
        af, asf = hog_bytes / (tf * 1e-3) / 1e9, surf_bytes / (ts * 1e-3) / 1e9
        scale = self.S * self.S / (4096 * 4096)
        unit = "bytes/step (PMC FETCH_SIZE x2 + WRITE_SIZE over every kernel of the function on 4096^2 tiles, profiles/function_traffic.json)"
        return {"kernel": "imgfd_fhog_dev (K13-K15, whole function)", "bound": "hbm", "achieved": round(af, 1), "peak": HBM_PEAK_GBS, "unit": "GB/s",
                "frac": round(af / HBM_PEAK_GBS, 4), "algorithmic_bytes_per_step": hog_bytes, "ms_per_step": round(tf, 3),
                "traffic": function_traffic("fhog", self.T * scale), "traffic_unit": unit,
                "surf": {"kernel": "imgfd_surf_dev (K16-K19, whole function)", "bound": "hbm", "achieved": round(asf, 1), "peak": HBM_PEAK_GBS,
                         "unit": "GB/s", "frac": round(asf / HBM_PEAK_GBS, 4), "algorithmic_bytes_per_step": surf_bytes, "ms_per_step": round(ts, 3),
                         "ms_per_tile": round(ts / self.T, 4), "traffic": function_traffic("surf", self.T * scale), "traffic_unit": unit,
                         "designed_bytes_per_step": int(surf_design), "frac_of_designed_bytes": round(surf_design / (ts * 1e-3) / 1e9 / HBM_PEAK_GBS, 4),
                         "designed_bytes": "39.25 B/px: RGB read twice (band / strip scans), the int32 table written once and read by both pyramid kernels, "
                                           "four of the six intervals per octave of the f64 pyramid written and read once (intervals 0 and 5 are never built)"},
                "fhog_ms_per_tile": round(tf / self.T, 4)}

    def parity_and_cpu(self, want_cpu):
        import numpy as np

        import oracle
        from image_amd import synth
        S = self.S
        use_ref = oracle.have_ref("dlib")
        key = lambda x, y, sc: sorted(zip(sc.tolist(), x.tolist(), y.tolist()))
        sample = sorted(set(np.linspace(0, self.T - 1, max(1, -(-self.T // 100))).astype(int).tolist()))   # >= 1 % of the batch (3 of 256 tiles)
        per_tile, t_h, t_s = [], None, None
        for ti in sample:
            rgb = synth.frame_rgb(3 + self.first + ti, S, S)
            assert np.array_equal(self.tiles[ti].cpu().numpy(), rgb), "device and host tile generators diverged"
            t = time.perf_counter(); rh = oracle.ref_fhog(rgb) if use_ref else oracle.fhog(rgb); th_ = time.perf_counter() - t
            t = time.perf_counter(); rs = oracle.surf(rgb, 1000, 30.0, use_ref=use_ref); ts_ = time.perf_counter() - t
            if t_h is None:
                t_h, t_s = th_, ts_
            gh = np.ascontiguousarray(self.hog[ti].cpu().numpy().transpose(2, 1, 0))
            n = int(self.counts[ti])
            gs = self.feat[ti, :n].cpu().numpy()
            # exact score ties (the synthetic tiles have them) may come in a different order: the reference leaves it to std::sort
            same_pts = n == len(rs["x"]) and bool(np.array_equal(gs[:, 4], rs["score"])) and key(gs[:, 0], gs[:, 1], gs[:, 4]) == key(rs["x"], rs["y"], rs["score"])
            desc_err = None
            if same_pts and n:
                order_g = np.lexsort((gs[:, 1], gs[:, 0], gs[:, 4])); order_r = np.lexsort((rs["y"], rs["x"], rs["score"]))
                desc_err = float(np.max(np.abs(gs[order_g, 6:] - np.nan_to_num(rs["surf"])[order_r])))
            per_tile.append({"tile": int(self.first + ti), "fhog_bit_equal": bool(gh.shape == rh.shape and np.array_equal(gh, rh)),
                             "fhog_max_abs_err": float(np.max(np.abs(gh - rh))) if gh.shape == rh.shape else None,
                             "surf_points": n, "surf_points_equal": same_pts, "surf_descriptor_max_abs_err": desc_err})
        errs = [p["surf_descriptor_max_abs_err"] for p in per_tile if p["surf_descriptor_max_abs_err"] is not None]
        out = {"parity_sample": {"tiles_checked": len(per_tile), "tiles": [p["tile"] for p in per_tile],
                                 "fhog_bit_equal": all(p["fhog_bit_equal"] for p in per_tile),
                                 "fhog_max_abs_err": max((p["fhog_max_abs_err"] for p in per_tile if p["fhog_max_abs_err"] is not None), default=None),
                                 "surf_points": [p["surf_points"] for p in per_tile], "surf_points_equal": all(p["surf_points_equal"] for p in per_tile),
                                 "surf_descriptor_max_abs_err": max(errs) if errs else None}}
        if want_cpu:
            out.update({"value": round(S * S / (t_h + t_s) / 1e6, 3), "unit": "Mpixels/s", "cores": 1, "kind": "reference" if use_ref else "port",
                        "sample": f"1 tile {S}x{S}, one run each: dlib's own extract_fhog_features ({1e3 * t_h:.0f} ms) and get_surf_points ({1e3 * t_s:.0f} ms) "
                                  "compiled in place (oracle/_ref/libref_dlib.so), single-threaded as the reference is"})
        return out


def traffic_for(kernel, batch):
    """HBM bytes per launch from the PMC passes (collected offline: PMC and timing must not share a run).  The file
    records the hash of the kernel sources it was measured on; a stale file yields None."""
    try:
        tr = json.load(open(os.path.join(ROOT, "profiles", "k3_traffic.json")))
        if tr.get("kernel_source_sha1") != kernel_source_hash():
            return None
        per_frame = tr.get("traffic_bytes_per_frame")
        if tr.get("batch") == batch:
            return int(tr["traffic_bytes_per_launch"])
        return int(per_frame * batch) if per_frame else None
    except Exception:
        return None


def kernel_source_hash(files=("fir_tensor.hip", "fir_tensor_device.h", "fir_device.h")):
    import hashlib
    h = hashlib.sha1()
    for f in files:
        with open(os.path.join(ROOT, "image_amd", "csrc", f), "rb") as fh:
            h.update(fh.read())
    return h.hexdigest()


# whole functions whose HBM traffic (PMC, all their kernels) is kept beside their roofline entries: name -> sources the number is tied to
FUNCTION_SOURCES = {"canny": ("canny.hip",), "fhog": ("fhog.hip", "fhog_fused.hip", "fhog_device.h"),
                    "surf": ("surf.hip", "surf_describe.hip", "surf_describe.h")}


def function_traffic(name, units):
    """HBM bytes of `units` frames / tiles through one whole function (every kernel it launches): PMC FETCH_SIZE x 2 + WRITE_SIZE
    per unit from profiles/function_traffic.json (scripts/gpu_pmc_functions.sh; separate passes, never beside a timing run), or
    None when the function's kernel sources have changed since it was measured."""
    try:
        tr = json.load(open(os.path.join(ROOT, "profiles", "function_traffic.json")))[name]
        if tr.get("kernel_source_sha1") != kernel_source_hash(FUNCTION_SOURCES[name]):
            return None
        return int(tr["traffic_bytes_per_unit"] * units)
    except Exception:
        return None


# ------------------------------------------------------------------------------------------------ launch
def pin_to_gpu_numa(local):
    """Bind this rank's host threads (the launch thread, the staging copies of --h2d) to the CPUs of the NUMA node its GPU hangs
    off: eight ranks that all run on node 0 share one memory controller and one PCIe root for their uploads.  Best effort:
    returns a description for the per-rank table, never fails the run."""
    try:
        import torch
        pr = torch.cuda.get_device_properties(local)
        bdf = f"{getattr(pr, 'pci_domain_id', 0):04x}:{pr.pci_bus_id:02x}:{pr.pci_device_id:02x}.0"
        node = int(open(f"/sys/bus/pci/devices/{bdf}/numa_node").read().strip())
        if node < 0:
            return {"gpu_pci": bdf, "numa_node": None, "pinned": False, "note": "the platform reports no NUMA node for this device"}
        cpus = set()
        for part in open(f"/sys/devices/system/node/node{node}/cpulist").read().strip().split(","):
            lo, _, hi = part.partition("-")
            cpus.update(range(int(lo), int(hi or lo) + 1))
        cpus &= os.sched_getaffinity(0)
        if not cpus:
            return {"gpu_pci": bdf, "numa_node": node, "pinned": False, "note": "none of that node's CPUs is in this process's affinity mask"}
        os.sched_setaffinity(0, cpus)
        return {"gpu_pci": bdf, "numa_node": node, "pinned": True, "cpus": len(cpus)}
    except Exception as e:
        return {"pinned": False, "note": f"{type(e).__name__}: {e}"}


def free_port():
    s = socket.socket(); s.bind(("127.0.0.1", 0)); p = s.getsockname()[1]; s.close()
    return p


def spawn_ranks(n, argv, rendezvous_timeout_s=None):
    """`--gpus N` outside torchrun: start the N ranks ourselves, rank r on GPU r, rendezvous on 127.0.0.1.  The first rank
    that exits with an error ends the job: its siblings are terminated (a rank that died before the rendezvous would leave
    the others waiting for it) and its return code is the launcher's."""
    port = free_port()
    procs = []
    for r in range(n):
        env = dict(os.environ, RANK=str(r), LOCAL_RANK=str(r), WORLD_SIZE=str(n), MASTER_ADDR="127.0.0.1",
                   MASTER_PORT=str(port), IMGFD_BENCH_CHILD="1")
        env.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
        if rendezvous_timeout_s:
            env["IMGFD_BENCH_RENDEZVOUS_S"] = str(int(rendezvous_timeout_s))
        procs.append(subprocess.Popen([sys.executable, os.path.abspath(__file__)] + argv, env=env))
    rc = 0
    live = list(procs)
    while live and rc == 0:
        time.sleep(0.05)
        for p in list(live):
            code = p.poll()
            if code is None:
                continue
            live.remove(p)
            if code != 0:
                rc = code
                print(f"bench.py: rank {procs.index(p)} exited with code {code}; stopping the other ranks", file=sys.stderr)
                break
    for p in live:   # only our own children, by handle
        p.terminate()
    for p in live:
        try:
            p.wait(timeout=10)
        except subprocess.TimeoutExpired:
            p.kill(); p.wait()
    return rc


def parse(argv):
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=None)
    ap.add_argument("--warmup", type=int, default=3)
    ap.add_argument("--config", type=int, default=2, choices=[2, 3, 4, 5], help="BASELINE.json configs[] entry (1-based as the judge counts them)")
    ap.add_argument("--batch", type=int, default=0, help="frames (tiles) per step per GPU; default 32 (config 2/5), 1024 (config 3), 256 (config 4)")
    ap.add_argument("--inner", type=int, default=40, help="config 2: passes over the batch inside one step (default: a timed region of ~3 s)")
    ap.add_argument("--frames", type=int, default=10000, help="config 5: frames of the whole stream")
    ap.add_argument("--tile", type=int, default=4096, help="config 4: tile edge")
    ap.add_argument("--max-parity-frames", type=int, default=100)
    ap.add_argument("--no-cpu", action="store_true", help="skip the cpu_baseline leg")
    ap.add_argument("--no-extra", action="store_true",
                    help="default run only: skip the short real-size runs of the other BASELINE configs that fill the line's \"configs\" entry")
    ap.add_argument("--no-dist", action="store_true", help="N=1: do not create the (one-rank) RCCL process group for the count reductions")
    ap.add_argument("--fir-mode", type=int, default=1)
    ap.add_argument("--no-overlap", action="store_true",
                    help="run the three detectors back to back on one stream instead of imgfd_detect_dev's two-stream schedule")
    ap.add_argument("--backend", default="nccl", choices=["nccl", "gloo"],
                    help="collective backend; nccl (= RCCL) is the product path, gloo is a functional check")
    ap.add_argument("--h2d", action="store_true",
                    help="config 2 / 5: the frames live in PINNED HOST memory and every pass uploads its batch (double-buffered on a copy stream, "
                         "overlapped with the kernels of the previous batch): the rate with delivery included (SURVEY.md 8e asks for both)")
    ap.add_argument("--share-device", action="store_true",
                    help="test hook: every rank uses cuda:0 (functional check of the N>1 path on a 1-GPU box; not a measurement)")
    ap.add_argument("--dry-run", action="store_true",
                    help="test hook: no device work at all -- launch, rendezvous and count reduction only (CPU, gloo)")
    ap.add_argument("--crash-rank", type=int, default=-1, help="test hook (with --dry-run): this rank exits with code 3 before the rendezvous")
    return ap.parse_args(argv)


def measure(args, det, rank, world, dist, want_cpu, light=False):
    """One configuration: stage the inputs, W warm-up steps, EXACTLY K timed steps between barriers (wall clock, max over
    ranks; a HIP event pair around every step gives the per-step median beside it), counts reduced over the ranks."""
    import torch

    from image_amd import stream
    wl = {2: lambda: Detect4K(args, det, rank, world), 3: lambda: Canny1080p(args, det, rank, world),
          4: lambda: DlibTiles(args, det, rank, world), 5: lambda: Detect4K(args, det, rank, world, stream_mode=True)}[args.config]()
    wl.prepare()
    steps = args.steps if args.steps is not None else wl.default_steps()
    on = dist is not None and dist.is_initialized()

    def barrier():
        torch.cuda.synchronize()
        if on and world > 1:
            dist.all_reduce(torch.zeros(1))   # a host tensor: gloo; no RCCL communicator exists before the timed region ends
        torch.cuda.synchronize()

    det.clock_probe(50); det.clock_probe_read()   # the probe's stream and ring exist before the clock starts
    for _ in range(args.warmup):
        wl.step()
    barrier()
    wl.reset()
    det.lib.imgfd_profile_k3(det.ctx.handle, 1)
    ev = [(torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)) for _ in range(steps)]
    t0 = time.perf_counter()
    for k in range(steps):
        det.clock_probe(200)   # one wavefront on a stream of its own, 200 us: the shader clock WHILE this step's kernels run
        ev[k][0].record()
        wl.step()
        ev[k][1].record()
    barrier()
    dt = time.perf_counter() - t0
    clock = det.clock_probe_read()
    k3_us, k3_n = det.profile_k3_read()
    det.lib.imgfd_profile_k3(det.ctx.handle, 0)
    step_ms = sorted(a.elapsed_time(b) for a, b in ev)
    pass_ms = sorted(wl.pass_times_ms()) if hasattr(wl, "pass_times_ms") else []

    # the path's only collectives: feature counts (sum; per-frame vectors are gathered in stream mode) and the elapsed time (max)
    dt_local = dt
    local_counts = wl.count_vector()
    args.local_counts = local_counts.clone()   # main(): the one RCCL all_gather, when every clock has stopped
    counts, dt = stream.reduce_counts(local_counts, dt, dist if on else None, on_host=True)
    args.reduced_counts = [int(v) for v in counts.tolist()]
    # the per-rank table (host objects over the process group's gloo side: no device collective needed)
    mine = {"rank": rank, "device": int(torch.cuda.current_device()), "pixels": int(wl.px_total(steps)), "s": round(dt_local, 4),
            "Mpixels_per_s": round(wl.px_total(steps) / dt_local / 1e6, 1), "numa": getattr(args, "numa_note", None)}
    if hasattr(wl, "n_local"):
        mine["frames"] = int(wl.n_local)
    per_rank = [mine]
    if on and world > 1:
        per_rank = [None] * world
        dist.all_gather_object(per_rank, mine)
    px_local = torch.tensor([wl.px_total(steps)], dtype=torch.int64, device="cuda")
    px_all, _ = stream.reduce_counts(px_local, 0.0, dist if on else None, on_host=True)
    per_frame = None
    if args.config == 5:
        per_frame = stream.gather_frame_counts(wl.frame_counts[:, :wl.cursor], dist if on else None, on_host=True)
    if (counts < 0).any():
        raise SystemExit("bench.py: a detector reported a negative feature count (record buffer overflow)")
    res = None
    if rank == 0:
        ms_per_step = 1e3 * dt / steps
        res = {
            "metric": wl.metric, "value": round(int(px_all[0]) / dt / 1e6, 2), "unit": "Mpixels/s",
            "n_gpus": world, "steps": steps, "warmup": args.warmup, "ms_per_step": round(ms_per_step, 4),
            "ms_per_step_median_hip_events": round(step_ms[len(step_ms) // 2], 4), "timed_region_s": round(dt, 3),
            "higher_is_better": True, "scaling": "weak" if args.config != 5 else "strong", "vs_baseline": None,
            "dtype": {2: "f64-accumulate/f32 (Harris), u8 (FAST-9), f64 (Canny)", 5: "f64-accumulate/f32 (Harris), f64 (Canny)",
                      3: "f64 (Canny)", 4: "f32 (fHOG), int32/f64 (SURF)"}[args.config],
            "data": "synthetic",
            "config": {**({"note": "functional check only: ranks share one device / gloo collectives"} if (args.share_device or (world > 1 and args.backend != "nccl")) else {}),
                       **wl.describe(counts)},
        }
        res["config"]["per_rank"] = per_rank
        res["config"]["shader_clock"] = {**clock, "how": "imgfd_clock_probe: one wavefront per step on a stream of its own reads s_memtime (shader cycles) and "
                                                         "s_memrealtime (constant rate) 200 us apart while the step's kernels run; rank 0"}
        if pass_ms:
            res["ms_per_pass_median_hip_events"] = round(pass_ms[len(pass_ms) // 2], 4)
            res["passes_event_timed"] = len(pass_ms)
        if on:
            res["config"]["collectives"] = (f"feature counts, elapsed time and per-frame count vectors reduced / gathered over the group's gloo side (host copies of a few bytes), "
                                            f"world size {dist.get_world_size()}; the barriers of the timed region are host barriers (gloo) around device synchronises")
        if per_frame is not None:
            res["config"]["per_frame_counts_gathered"] = int(per_frame.shape[1])
            res["config"]["per_frame_counts_checksum"] = {"harris": int(per_frame[0].sum()), "canny": int(per_frame[1].sum())}
        if args.config in (2, 5):
            res["roofline"] = wl.roofline(k3_us, k3_n, steps)
        else:
            res["roofline"] = wl.roofline(k3_us, k3_n, steps, ms_per_step=ms_per_step)
        if world == 1:
            cb = wl.parity_and_cpu(want_cpu)
            if cb:
                res["cpu_baseline" if want_cpu else "parity"] = cb
    del wl
    torch.cuda.empty_cache()
    return res


def _pcie_rates(device):
    """what a plain copy of a host vector achieves on this box (GB/s): pageable and pinned, up and down -- the floor the host
    entry points are priced against (torch copies: hipMemcpyAsync on the current stream)"""
    import torch
    n = 64 << 20
    out = {}
    for kind in ("pageable", "pinned"):
        h = torch.empty(n, dtype=torch.uint8, pin_memory=(kind == "pinned"))
        h.fill_(7)
        d = torch.empty(n, dtype=torch.uint8, device=f"cuda:{device}")
        for name, fn in (("h2d", lambda: d.copy_(h)), ("d2h", lambda: h.copy_(d))):
            fn(); torch.cuda.synchronize()
            ts = []
            for _ in range(5):
                t = time.perf_counter(); fn(); torch.cuda.synchronize(); ts.append(time.perf_counter() - t)
            out[f"{kind}_{name}_GBps"] = round(n / min(ts) / 1e9, 1)
    return out


def host_api_config(device):
    """configs.host_api: the boundary an R user hits -- the five R-facing entry points of include/imgfd.h on host vectors as
    the .Call glue passes them (REAL(x) / INTEGER(x): pageable memory), one 3840x2160 frame / one 4096x4096 RGB tile per
    call, PCIe in both directions included, results in the vectors R would own (the edge map and the fHOG features as
    doubles, widened on the device).  Reference: rcpp_harris.cpp:19-60, f9_rcpp.cpp:8-35, rcpp_canny.cpp:122-244,
    rcpp_fhog.cpp:10-46, rcpp_surf.cpp:10-53."""
    import ctypes as C

    import numpy as np

    from image_amd import _binding, _lib, synth
    ctx = _lib.Context(device)
    lib, h = ctx.lib, ctx.handle
    NX, NY, S = 3840, 2160, 4096
    img = synth.frame(2, NX, NY)
    d64 = np.ascontiguousarray(img.astype(np.float64)); i32 = np.ascontiguousarray(img.astype(np.int32))
    rgb = np.ascontiguousarray(synth.frame_rgb(3, S, S).astype(np.int32))
    edges = np.zeros((NY, NX), np.float64); nz = C.c_int64(0)
    nr, nc = C.c_int(0), C.c_int(0)
    ctx.check(lib.imgfd_fhog_size(S, S, 8, 1, 1, C.byref(nr), C.byref(nc)), "fhog_size")
    hog = np.zeros((31 * nr.value * nc.value,), np.float64)
    counts = {}

    def harris():
        out = _binding.Corners()
        ctx.check(lib.imgfd_harris_f64(h, d64.ctypes.data_as(C.c_void_p), NX, NY, 0.06, 1.0, 2.5, 130.0, 0, 0, 0, 1, 0, 1, 0, 10, 0, C.byref(out)), "harris_f64")
        counts["harris_corners"] = int(out.n)
        if out.n: lib.imgfd_free(out.corners)

    def fast9():
        out = _binding.Points()
        ctx.check(lib.imgfd_fast9_i32(h, i32.ctypes.data_as(C.c_void_p), NX, NY, NX, 20, 1, C.byref(out)), "fast9_i32")
        counts["fast9_points"] = int(out.n)
        if out.n: lib.imgfd_free(out.points)

    def canny():
        ctx.check(lib.imgfd_canny_f64out(h, i32.ctypes.data_as(C.c_void_p), NX, NY, 2.0, 3.0, 10.0, 1, edges.ctypes.data_as(C.c_void_p), C.byref(nz)), "canny_f64out")
        counts["canny_pixels_nonzero"] = int(nz.value)

    def fhog():
        ctx.check(lib.imgfd_fhog_f64out(h, rgb.ctypes.data_as(C.c_void_p), S, S, 8, 1, 1, hog.ctypes.data_as(C.c_void_p), hog.size, C.byref(nr), C.byref(nc)), "fhog_f64out")

    def surf():
        o = _binding.SurfOut()
        ctx.check(lib.imgfd_surf_i32(h, rgb.ctypes.data_as(C.c_void_p), S, S, 1000, 30.0, C.byref(o)), "surf_i32")
        counts["surf_points"] = int(o.n)
        if o.n: lib.imgfd_free(o.data)

    rates = _pcie_rates(device)
    up, down = rates["pageable_h2d_GBps"] * 1e9, rates["pageable_d2h_GBps"] * 1e9
    px4k, pxt = NX * NY, S * S
    plan = [("imgfd_harris_f64", harris, px4k, 8 * px4k, 0, "image_harris(): 8 B/px up (the NumericMatrix), a corner list down"),
            ("imgfd_fast9_i32", fast9, px4k, 4 * px4k, 0, "image_detect_corners(): 4 B/px up (as.integer(x)), a point list down"),
            ("imgfd_canny_f64out", canny, px4k, 4 * px4k, 8 * px4k, "image_canny_edge_detector(): 4 B/px up, 8 B/px down (the NumericMatrix of edges, widened on the device)"),
            ("imgfd_fhog_f64out", fhog, pxt, 12 * pxt, 8 * hog.size, "image_fhog(): 12 B/px up (3 ints), 31 doubles per 8x8 cell down (widened on the device)"),
            ("imgfd_surf_i32", surf, pxt, 12 * pxt, 0, "image_surf(): 12 B/px up, <= 1000 points x 70 doubles down")]
    out = {"what": "one call per entry point on pageable host vectors, results in host vectors; best and median of 7 calls after 2 warm-up calls",
           "frame": f"{NX}x{NY} gray", "tile": f"{S}x{S} RGB", "copy_rates_this_box": rates, "calls": {}}
    for name, fn, px, b_up, b_down, note in plan:
        fn(); fn()
        ts = []
        for _ in range(7):
            t = time.perf_counter(); fn(); ts.append(time.perf_counter() - t)
        ts.sort()
        floor = b_up / up + b_down / down
        out["calls"][name] = {"ms_best": round(1e3 * ts[0], 3), "ms_median": round(1e3 * ts[len(ts) // 2], 3), "Mpixels_per_s": round(px / ts[0] / 1e6, 1),
                              "bytes_up": int(b_up), "bytes_down": int(b_down), "pcie_floor_ms": round(1e3 * floor, 3),
                              "floor_over_best": round(floor / ts[0], 3), "note": note}
    out["results"] = counts
    three = sum(out["calls"][k]["ms_best"] for k in ("imgfd_harris_f64", "imgfd_fast9_i32", "imgfd_canny_f64out"))
    out["harris_fast9_canny_Mpixels_per_s"] = round(px4k / three / 1e3, 1)
    ctx.close()
    return out


def stream_h2d_config(device, n_batches=12, batch=32):
    """configs.5_h2d: configs[4] with delivery included -- 3840x2160 u8 frames in PINNED host memory through imgfd_stream_*
    (upload of batch i+1 overlapped with the kernels of batch i), Harris + Canny as configs[4] runs them; the resident rate
    of the same stream is configs.5.  Floor: 1 B/px up at the pinned copy rate of this box."""
    import numpy as np

    from image_amd import _lib, framestream, synth
    NX, NY = 3840, 2160
    ctx = _lib.Context(device)
    base = np.stack([synth.frame(50000 + f, NX, NY) for f in range(4)])
    pinned = [framestream.PinnedFrames(batch, NY, NX) for _ in range(3)]
    for p in pinned:
        for f in range(batch):
            p.array[f] = base[f % 4]
    out = {"what": f"{n_batches} batches of {batch} frames {NX}x{NY} from pinned host memory, two batches in flight, Harris + Canny (image_harris() defaults, Canny s=2 3/10 accGrad)"}
    try:
        with framestream.FrameStream(NX, NY, batch, ctx=ctx, harris=True, fast9=False, canny=True) as fs:
            def run():
                counts = []
                t = time.perf_counter()
                for r in fs.run(pinned[i % 3].array for i in range(n_batches)):
                    counts.append((int(r["harris_counts"].sum()), int(r["canny_counts"].sum())))
                return time.perf_counter() - t, counts
            run()
            ts = []
            for _ in range(3):
                t, counts = run(); ts.append(t)
            px = n_batches * batch * NX * NY
            rates = _pcie_rates(device)
            out.update({"s_best_of_3": round(min(ts), 4), "value": round(px / min(ts) / 1e6, 1), "unit": "Mpixels/s", "GBps_up": round(px / min(ts) / 1e9, 2),
                        "pinned_h2d_GBps_this_box": rates["pinned_h2d_GBps"], "pcie_floor_Mpixels_per_s": round(rates["pinned_h2d_GBps"] * 1e3, 1),
                        "frac_of_pcie_floor": round(px / min(ts) / 1e9 / rates["pinned_h2d_GBps"], 3),
                        "counts_per_batch": {"harris": counts[0][0], "canny": counts[0][1]}, "counts_equal_across_batches": len(set(counts)) == 1})
    finally:
        for p in pinned:
            p.free()
        ctx.close()
    return out


def extra_configs(args, det, dist):
    """The other BASELINE.json configurations at their real sizes, a few steps each, in this same process (N = 1): the
    driver's one line then carries all five.  configs[4] stages all of its 10 000 frames (83 GB of the 288) and streams every one
    of them once."""
    import copy
    out = {}
    plan = [("2_batch1", dict(config=2, batch=1, inner=50, steps=20, warmup=3), False),
            ("3", dict(config=3, batch=0, steps=3, warmup=1), True),
            ("4", dict(config=4, batch=0, steps=2, warmup=1), True),
            ("5", dict(config=5, batch=0, frames=10000, steps=None, warmup=1, max_parity_frames=100), False)]
    for name, over, cpu in plan:
        a = copy.copy(args)
        for k, v in over.items():
            setattr(a, k, v)
        t0 = time.perf_counter()
        try:
            r = measure(a, det, 0, 1, dist, want_cpu=cpu and not args.no_cpu)
        except Exception as e:   # one configuration must not take the headline number with it
            out[name] = {"error": f"{type(e).__name__}: {e}"}
            continue
        keep = {k: r[k] for k in ("metric", "value", "unit", "steps", "warmup", "ms_per_step", "ms_per_step_median_hip_events", "ms_per_pass_median_hip_events",
                                   "passes_event_timed", "dtype", "config", "roofline") if k in r}
        for k in ("cpu_baseline", "parity"):
            if k in r:
                keep[k] = r[k]
        if name == "5":
            keep["config"]["frames_staged"] = 10000
        keep["wall_s_incl_staging_and_checks"] = round(time.perf_counter() - t0, 1)
        out[name] = keep
    # the boundary R users hit, and configs[4] with delivery: host vectors, PCIe included
    for name, fn in (("host_api", lambda: host_api_config(det.device)), ("5_h2d", lambda: stream_h2d_config(det.device))):
        t0 = time.perf_counter()
        try:
            out[name] = fn()
            out[name]["wall_s"] = round(time.perf_counter() - t0, 1)
        except Exception as e:
            out[name] = {"error": f"{type(e).__name__}: {e}"}
    return out


def line_summary(res):
    """the numbers a reader of the driver's record needs, in one short object: every configuration's rate and time, the
    structure-tensor kernel's roofline fraction and clock, the parity verdicts"""
    def g(d, *path, default=None):
        for k in path:
            if not isinstance(d, dict) or k not in d:
                return default
            d = d[k]
        return d
    rf = res.get("roofline", {})
    out = {"default": {"Mpx_s": res.get("value"), "ms_step": res.get("ms_per_step"), "clock_GHz": g(res, "config", "shader_clock", "mean_GHz")}}
    if "frac" in rf and "avg_launch_us" in rf:
        out["k3"] = {"frac": rf.get("frac"), "us": rf.get("avg_launch_us"), "clock_GHz": rf.get("shader_clock_GHz"), "traffic": rf.get("traffic"),
                     "pipe_us": g(rf, "in_pipeline", "avg_launch_us"), "from_idle_us": g(rf, "single_launches_from_idle", "median_us"),
                     "from_idle_clock_GHz": g(rf, "single_launches_from_idle", "shader_clock_GHz")}
    cb = res.get("cpu_baseline") or res.get("parity") or {}
    par = {}
    p0 = [cb.get("parity_frame0")] + list((cb.get("parity_frames") or {}).values())
    p0 = [p for p in p0 if p]
    if p0:
        par["default"] = {"frames": len(p0), "harris_xy": all(p.get("harris_coordinates_match") for p in p0),
                          "harris_R_bits_differing": sum(p.get("harris_strengths_differing_in_any_bit") or 0 for p in p0),
                          "fast9_xy": all(p.get("fast9_coordinates_match", True) for p in p0), "canny_px_differing": sum(p.get("canny_mismatching_pixels") or 0 for p in p0)}
    if cb.get("value") is not None:
        out["cpu"] = {"Mpx_s": cb.get("value"), "cores": cb.get("cores"), "ref_only_Mpx_s": cb.get("reference_only_value")}
    cfgs = res.get("configs") or {}
    for name in ("2_batch1", "3", "4", "5"):
        c = cfgs.get(name)
        if not c:
            continue
        if "error" in c:
            out[name] = {"error": c["error"][:80]}
            continue
        e = {"Mpx_s": c.get("value"), "ms_step": c.get("ms_per_step")}
        if name == "2_batch1":
            e["ms_frame"] = round(c["ms_per_step"] / max(1, g(c, "config", "passes_per_step", default=1)), 4) if c.get("ms_per_step") else None
        if name == "4":
            e.update({"surf_ms_tile": g(c, "roofline", "surf", "ms_per_tile"), "fhog_ms_tile": g(c, "roofline", "fhog_ms_per_tile")})
        out[name] = e
        ps = g(c, "cpu_baseline", "parity_sample") or g(c, "parity", "parity_sample")
        if ps:
            keep = ("frames_checked", "tiles_checked", "canny_mismatching_pixels_total", "frames_with_different_corner_coordinates", "frames_with_strength_rel_err_above_1e-4",
                    "frames_whose_streamed_counts_differ", "fhog_bit_equal", "surf_points_equal", "surf_descriptor_max_abs_err", "pixels_nonzero_equal")
            par[name] = {k: ps[k] for k in keep if k in ps}
    h = cfgs.get("5_h2d")
    if h and "value" in h:
        out["5_h2d"] = {"Mpx_s": h.get("value"), "of_pcie_floor": h.get("frac_of_pcie_floor")}
    ha = g(cfgs, "host_api", "calls")
    if ha:
        out["host_api_ms"] = {k.replace("imgfd_", ""): [v.get("ms_best"), v.get("floor_over_best")] for k, v in ha.items()}
    if par:
        out["parity"] = par
    return out


def main(argv=None):
    argv = list(sys.argv[1:] if argv is None else argv)
    args = parse(argv)
    if args.gpus > 1 and "WORLD_SIZE" not in os.environ:
        sys.exit(spawn_ranks(args.gpus, argv))

    rank = int(os.environ.get("RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    local = int(os.environ.get("LOCAL_RANK", "0"))
    if args.dry_run and rank == args.crash_rank:
        sys.exit(3)
    # ONE line on stdout: libraries that write to file descriptor 1 (RCCL prints a version banner there) go to stderr
    sys.stdout.flush()
    result_fd = os.dup(1)
    os.dup2(2, 1)

    def emit(obj):
        os.write(result_fd, (json.dumps(obj) + "\n").encode())

    import datetime

    import torch
    import torch.distributed as dist

    from image_amd import stream
    rdv = datetime.timedelta(seconds=int(os.environ.get("IMGFD_BENCH_RENDEZVOUS_S", "600")))
    if args.dry_run:
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        if world > 1:
            dist.init_process_group("gloo", timeout=rdv)
        counts = torch.tensor([rank + 1, 10 * (rank + 1), 100 * (rank + 1)], dtype=torch.int64)
        counts, dt = stream.reduce_counts(counts, 1.0 + rank, dist if world > 1 else None)
        if rank == 0:
            emit({"metric": "dry run (launch + reduction only)", "value": None, "n_gpus": world, "dry_run": True,
                  "feature_counts": counts.tolist(), "max_elapsed_s": dt})
        if world > 1:
            dist.destroy_process_group()
        return

    if not torch.cuda.is_available():
        raise SystemExit("bench.py needs an MI355X: the backend has no CPU path")
    if args.share_device:
        local = 0
    elif local >= torch.cuda.device_count():
        raise SystemExit(f"bench.py: rank {rank} needs GPU {local} but only {torch.cuda.device_count()} are visible: --gpus {world} is more than this node has "
                         "(--share-device runs a functional check of the N>1 path on one GPU)")
    torch.cuda.set_device(local)
    args.numa_note = pin_to_gpu_numa(local) if world > 1 or os.environ.get("IMGFD_BENCH_PIN") else {"pinned": False, "note": "one rank: left to the scheduler"}
    dist_note = None
    os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
    # The process group serves host tensors through gloo and device tensors through RCCL.  The RCCL communicator is created by
    # the first device collective -- the count reduction AFTER the timed region -- on purpose: with a communicator alive the
    # same passes ran 12 % slower (58.1 against 66.3 Gpixel/s at N = 1, same kernels' device times; profiles/r03): the barrier
    # in front of the timed region is therefore a host barrier (gloo) after a device synchronise.
    mixed = "cpu:gloo,cuda:nccl" if args.backend == "nccl" and not args.share_device else "gloo"  # ranks sharing one device cannot form an RCCL communicator
    if world > 1:
        dist.init_process_group(mixed, timeout=rdv)
    elif not args.no_dist:
        # N = 1: the count reductions still go through a (one-rank) RCCL communicator -- the code path of N > 1
        try:
            os.environ.setdefault("MASTER_PORT", str(free_port()))
            dist.init_process_group(mixed, rank=0, world_size=1, timeout=rdv)
        except Exception as e:
            dist_note = f"one-rank process group not created: {type(e).__name__}: {e}"

    from image_amd.device import DeviceDetector

    torch.cuda.set_stream(torch.cuda.Stream())   # everything below runs on one non-default stream (a default stream cannot be captured into the small-batch graph of imgfd_detect_dev)
    det = DeviceDetector(local)
    det.ctx.set_fir_mode(args.fir_mode)
    res = measure(args, det, rank, world, dist, want_cpu=not args.no_cpu)
    main_counts, main_reduced = args.local_counts, args.reduced_counts
    if rank == 0:
        default_line = args.config == 2 and args.batch == 0 and world == 1 and not args.no_overlap
        if default_line and not args.no_extra:
            res["configs"] = extra_configs(args, det, dist)
    # "RCCL only to gather feature counts": ONE all_gather of every rank's count vector on device tensors, after the last
    # timed region of the process -- the communicator is created here, by this call, and never lives beside a kernel that is
    # being timed (a live one cost the same passes 12 % at N = 1, profiles/r03)
    rccl = None
    if dist.is_initialized():
        try:
            rccl = stream.rccl_gather_counts(main_counts, dist)
        except Exception as e:
            dist_note = f"RCCL all_gather failed: {type(e).__name__}: {e}"
    if rank == 0:
        res["summary"] = line_summary(res)   # LAST key of the line: the tail of stdout the driver keeps always holds it
        res["config"]["summary"] = res["summary"]   # and inside a key the driver's parser keeps
        if rccl is not None:
            same = [int(v) for v in rccl.sum(0).tolist()] == main_reduced
            res["config"]["collectives"] += (f"; then ONE RCCL all_gather of the per-rank count vectors (nccl backend, device tensors), after every clock has stopped: "
                                             f"{rccl.shape[0]} ranks seen, sums {'equal' if same else 'DIFFER from'} the gloo reduction")
            res["config"]["rccl_ranks_seen"] = int(rccl.shape[0])
        elif dist.is_initialized():
            res["config"]["collectives"] += "; no RCCL communicator (the ranks share one device / gloo backend: functional check)"
            res["config"]["rccl_ranks_seen"] = 0
        if dist_note:
            res["config"]["collectives"] = res["config"].get("collectives", "") + " [" + dist_note + "]"
        emit(res)
    if dist.is_initialized():
        if world > 1:
            dist.all_reduce(torch.zeros(1))
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
