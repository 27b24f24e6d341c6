#!/usr/bin/env python3
"""Benchmark of the ORB front end on MI355X.

  python bench.py --gpus N --steps K --warmup W

One "step" is one pass of the extractor (pyramid, cell-grid FAST, selection, orientation, blur,
rotated-BRIEF descriptors, keypoint records) over one batch of synthetic frames that is already
resident in HBM.  The workload is the configuration the headline metric is quoted on: 640x480 frames,
1000 features per frame (8 levels, scale 1.2), 1024 frames per GPU and step (the rate grows with the
batch until the short launches of the upper pyramid levels are amortised; --max-batch shows it: with this
code 153 k / 209 k / 241 k / 261 k / 273 k frames/s in chunks of 32 / 64 / 128 / 256 / 1024 frames).  With
N > 1 (one process per GPU: the ranks torch.distributed.run starts, or, when the script is started plainly
with --gpus N, N copies of itself that it launches before anything touches the GPU) every rank processes its
own batch of 1024 frames: frames are independent, so the path is sharded by frame batch with no data-path
collective (weak scaling); the only collectives of the timed region are the timing barrier and a MAX over
ranks of the elapsed time.  --backend gloo rehearses the N > 1 legs with all ranks on the GPUs that are there.

Rank 0 prints ONE JSON line.  Besides the contract fields it carries
  roofline      the dominant kernel group against HBM bandwidth (durations from HIP events on the
                library's stream, collected in a separate profiled pass after the timed region)
  cpu_baseline  the CPU oracle (oracle/, a port of the reference's CPU path) timed on the host
                cores of this box on a bounded sample of the same frames: one thread (how the reference
                runs the path) with the all-core figure beside it
  extra         N = 1: the 1280x720 / 2000 and 1920x1080 / 4000 extraction legs with their stage
                rooflines, the Hamming kNN legs (4096 frame pairs; one 1.25 M-row map shard) with the
                popcount roofline, the rate with host buffers (PCIe inclusive), the latency of one frame
                per call, the windowed / projection / BoW matcher legs.
                N > 1: the rank-local legs of the multi-GPU configurations: Hamming kNN of 4096 pairs per
                rank, the 2048-frame 1920x1080 / 4000 batch split contiguously over the ranks, and 4000
                queries against a 10 M-row map split into one shard per rank with one RCCL all-gather.
"""
import argparse
import importlib
import json
import os
import subprocess
import sys
import tempfile
import threading
import time

import numpy as np

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

HBM_PEAK_GBS = 8000.0  # MI355X HBM3E, spec

KERNEL_OF_STAGE = {"level0": "k_level0", "resize": "k_resize", "fast_measure": "k_score", "select": "k_cell_select",
                   "orient": "k_orient_describe", "blur": "k_blur", "describe": "k_orient_describe"}


def pyramid_pixels(ext, nlevels):
    return sum(ext.level_geometry(l)["w"] * ext.level_geometry(l)["h"] for l in range(nlevels))


def make_frames(synth, w, h, n, first):
    return np.stack([synth.synth_image(w, h, first + i) for i in range(n)])


def oracle():
    """The CPU oracle: the checker and the CPU baseline, never part of the measured GPU path."""
    p = os.path.join(ROOT, "oracle")
    if p not in sys.path:
        sys.path.insert(0, p)
    import orbref
    orbref.lib()
    return orbref


class ExtractWorkload:
    """Batch of frames resident in HBM + preallocated device outputs."""

    def __init__(self, orb, torch, dev, w, h, nfeatures, batch, first_idx, frames=None, max_batch=None):
        self.torch = torch
        self.w, self.h, self.nf, self.batch = w, h, nfeatures, batch
        self.host = frames if frames is not None else make_frames(orb.synth, w, h, batch, first_idx)
        self.ext = orb.ORBextractor(nfeatures, 1.2, 8, 1, 20, device=dev, max_width=w, max_height=h,
                                    max_batch=max_batch or batch)
        self.d_img = torch.from_numpy(self.host).to("cuda:%d" % dev)
        self.d_kp = torch.zeros((batch, nfeatures, 7), dtype=torch.float32, device="cuda:%d" % dev)
        self.d_desc = torch.zeros((batch, nfeatures, 32), dtype=torch.uint8, device="cuda:%d" % dev)
        self.d_n = torch.zeros(batch, dtype=torch.int32, device="cuda:%d" % dev)
        torch.cuda.synchronize(dev)

    def step(self):
        self.ext.extract_batch_device(self.d_img.data_ptr(), self.batch, self.h, self.w, self.w, self.w * self.h,
                                      self.d_kp.data_ptr(), self.d_desc.data_ptr(), self.nf, self.d_n.data_ptr())

    def sync(self):
        self.ext.synchronize()

    def timed(self, steps, warmup):
        for _ in range(warmup):
            self.step()
        self.sync()
        t0 = time.perf_counter()
        for _ in range(steps):
            self.step()
        self.sync()
        return time.perf_counter() - t0

    def mismatching(self, ref_results):
        """Frames of this batch whose device output differs from the oracle's (keypoint records and
        descriptors compared bit for bit)."""
        kp = self.d_kp.cpu().numpy().view(np.uint32)
        desc = self.d_desc.cpu().numpy()
        ncount = self.d_n.cpu().numpy()
        mism = 0
        for i, (rkp, rdesc) in enumerate(ref_results):
            n = len(rkp)
            same = ncount[i] == n and np.array_equal(kp[i, :n].reshape(-1), rkp.view(np.uint32).reshape(-1)) \
                and np.array_equal(desc[i, :n], rdesc)
            mism += 0 if same else 1
        return mism


def cpu_extract(frames, nfeatures, threads, sample):
    """Oracle (CPU port of the reference path) on `threads` host threads over `sample` frames."""
    orbref = oracle()
    from concurrent.futures import ThreadPoolExecutor
    sample = min(sample, len(frames))
    exts = [orbref.RefExtractor(nfeatures, 1.2, 8, 1, 20) for _ in range(threads)]
    results = [None] * sample

    def work(t):
        for i in range(t, sample, threads):
            results[i] = exts[t](frames[i])

    exts[0](frames[0])  # touch code and pages once
    t0 = time.perf_counter()
    with ThreadPoolExecutor(threads) as pool:
        list(pool.map(work, range(threads)))
    dt = time.perf_counter() - t0
    return sample / dt, results


def _pmc_traffic(kernel, wl):
    """HBM bytes per launch of `kernel` from the committed PMC summary (profiles/*pmc_traffic.json, made by
    tools/pmc_summary.py from separate rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE passes of this same
    workload).  Counters cannot be collected from inside the timed run, so the figure is the one of the
    latest committed pass of the same workload; None if there is none."""
    import glob
    want = "%dx%dx%d/%d" % (wl.batch, wl.w, wl.h, wl.nf)
    for f in sorted(glob.glob(os.path.join(ROOT, "profiles", "*pmc_traffic.json")), reverse=True):
        try:
            j = json.load(open(f))
            if j.get("workload", "256x1280x720/2000") != want:
                continue
            k = j["kernels"].get(kernel)
            return k["traffic"] if k else None
        except (OSError, ValueError, KeyError):
            continue
    return None


FULL_RATE_CYCLES = 2.3  # SIMD cycles per wave64 instruction of the full-rate VALU class on this part (tools/valu_probe.hip)
SHADER_CLOCK_HZ = 2.4e9
SIMDS = 256 * 4


def _valu_roofline(kernel, wl, avg_launch_ms, pixels_per_launch):
    """Instruction-issue roofline of `kernel` beside the HBM one.  The wave-instruction count per launch comes from
    the latest committed SQ counter summary of this same workload (profiles/*_pmc_sq.json; counters cannot be read
    inside the timed run), the launch duration is the one measured live.  cycles_per_instr = SIMD cycles the launch
    had (1024 SIMDs x duration x 2.4 GHz) per VALU wave instruction it issued; a kernel made only of full-rate
    instructions that never waits would sit at 2.3."""
    import glob
    want = "%d x %dx%d frames, %d features" % (wl.batch, wl.w, wl.h, wl.nf)
    for f in sorted(glob.glob(os.path.join(ROOT, "profiles", "*_pmc_sq.json")), reverse=True):
        if "matcher" in os.path.basename(f):
            continue
        try:
            j = json.load(open(f))
            if want not in j.get("source", ""):
                continue
            k = next((v for n, v in j["kernels"].items() if n.split("<")[0] == kernel), None)
            if not k or avg_launch_ms <= 0:
                continue
            insts = float(k["SQ_INSTS_VALU"])
            avail = SIMDS * avg_launch_ms * 1e-3 * SHADER_CLOCK_HZ
            cpi = avail / insts
            lds_act, lds_conf = float(k.get("SQ_LDS_IDX_ACTIVE", 0)), float(k.get("SQ_LDS_BANK_CONFLICT", 0))
            return {"wave_instr_per_launch": insts, "lane_instr_per_pixel": round(insts * 64 / pixels_per_launch, 2),
                    "simd_cycles_available": float("%.4g" % avail), "cycles_per_instr": round(cpi, 3),
                    "full_rate_cycles_per_instr": FULL_RATE_CYCLES, "frac_of_full_rate": round(FULL_RATE_CYCLES / cpi, 4),
                    "lds_bank_conflict_share": round(lds_conf / lds_act, 4) if lds_act else None,
                    "counters_from": os.path.basename(f)}
        except (OSError, ValueError, KeyError):
            continue
    return None


def extract_roofline(wl, prof_steps):
    """Stage durations from HIP events on the library's stream (single-stream profiling pass of the
    library) against the algorithmic bytes of each kernel group: the 5 B/pixel + 60 B/keypoint of the
    whole path split by who must move them (pyramid: read + write; FAST: read; blur: read + write;
    outputs: 32 B descriptor + 28 B keypoint)."""
    nl = wl.ext.GetLevels()
    P = pyramid_pixels(wl.ext, nl)
    wl.ext.set_profiling(True)
    acc = {}
    for _ in range(prof_steps):
        wl.step()
        wl.sync()
        for s, (ms, n) in wl.ext.stage_times().items():
            a = acc.setdefault(s, [0.0, 0])
            a[0] += ms
            a[1] += n
    wl.ext.set_profiling(False)
    N_frame = int(wl.d_n.sum().item()) / wl.batch
    lvl0 = wl.w * wl.h
    # FAST reads every pyramid pixel once: in the fused per-cell kernel ("nms_compact" group, the
    # "fast_measure" group then has no launches) or in the separate measure kernel.
    fused = acc["fast_measure"][1] == 0
    alg = {"level0": 2 * lvl0, "resize": 2 * (P - lvl0), "fast_measure": 0 if fused else P,
           "nms_compact": P if fused else 0, "select": 0, "orient": 0, "blur": 2 * P, "describe": 60 * N_frame}
    names = dict(KERNEL_OF_STAGE, nms_compact="k_fast_cell" if fused else "k_nms")
    stage_ms = {s: a[0] / prof_steps for s, a in acc.items()}
    dominant = max(stage_ms, key=stage_ms.get)
    launches = acc[dominant][1] // prof_steps
    avg_launch_ms = stage_ms[dominant] / max(launches, 1)
    bytes_per_launch = alg[dominant] * wl.batch / max(launches, 1)
    achieved = bytes_per_launch / (avg_launch_ms * 1e-3) / 1e9 if avg_launch_ms > 0 else 0.0
    path_bytes = (5 * P + 60 * N_frame) * wl.batch
    path_ms = sum(stage_ms.values())
    stages = {}
    for s, ms in stage_ms.items():
        b = alg[s] * wl.batch
        stages[s] = {"kernel": names[s], "ms_per_step": round(ms, 4), "launches_per_step": acc[s][1] // prof_steps,
                     "algorithmic_bytes_per_step": int(b),
                     "achieved_gb_per_sec": round(b / (ms * 1e-3) / 1e9, 1) if ms > 0 and b else None,
                     "frac": round(b / (ms * 1e-3) / 1e9 / HBM_PEAK_GBS, 5) if ms > 0 and b else None}
    return {"bound": "hbm", "kernel": names[dominant], "stage": dominant, "achieved": round(achieved, 2),
            "peak": HBM_PEAK_GBS, "unit": "GB/s", "frac": round(achieved / HBM_PEAK_GBS, 5),
            "traffic": _pmc_traffic(names[dominant], wl),
            "valu": _valu_roofline(names[dominant], wl, avg_launch_ms, P * wl.batch / max(launches, 1)),
            "launches_per_step": launches, "avg_launch_ms": round(avg_launch_ms, 5),
            "algorithmic_bytes_per_launch": int(bytes_per_launch),
            "stages": stages,
            "whole_path": {"algorithmic_bytes_per_step": int(path_bytes), "kernel_ms_per_step": round(path_ms, 4),
                           "achieved": round(path_bytes / (path_ms * 1e-3) / 1e9, 2) if path_ms > 0 else 0.0,
                           "frac": round(path_bytes / (path_ms * 1e-3) / 1e9 / HBM_PEAK_GBS, 5) if path_ms > 0 else 0.0}}, N_frame


class StdoutGuard:
    """The contract is ONE JSON line on stdout.  RCCL prints a version banner to the process's stdout when its first
    communicator comes up, so while collectives may initialise, file descriptor 1 points at stderr; emit() puts the
    real stdout back and prints the line."""

    def __init__(self):
        self.saved = None

    def divert(self):
        sys.stdout.flush()
        self.saved = os.dup(1)
        os.dup2(2, 1)

    def emit(self, line):
        sys.stdout.flush()
        if self.saved is not None:
            os.dup2(self.saved, 1)
            os.close(self.saved)
            self.saved = None
        print(json.dumps(line), flush=True)


STDOUT = StdoutGuard()


class Watchdog:
    """The legs that run after the headline measurement must not be able to lose it: if they have not
    finished after `seconds` (a collective that never completes, a stuck host thread), rank 0 prints the
    line it has and every rank leaves."""

    def __init__(self, seconds, rank):
        self.line = None
        self.rank = rank
        self.timer = threading.Timer(seconds, self.fire)
        self.timer.daemon = True
        self.seconds = seconds

    def arm(self, line):
        self.line = line
        self.timer.start()

    def fire(self):
        if self.rank == 0 and self.line is not None:
            self.line["extra"] = {"error": "the legs after the headline measurement did not finish within %d s" % self.seconds}
            STDOUT.emit(self.line)
        os._exit(0)

    def cancel(self):
        self.timer.cancel()


def rank_environment(base_env, rank, world, port):
    """Environment of rank `rank` of `world` on this node: what torch.distributed.run would set."""
    env = dict(base_env)
    env.update({"RANK": str(rank), "LOCAL_RANK": str(rank), "WORLD_SIZE": str(world), "LOCAL_WORLD_SIZE": str(world),
                "MASTER_ADDR": "127.0.0.1", "MASTER_PORT": str(port)})
    env.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")  # dmabuf IPC: what RCCL needs between processes on this driver
    return env


def free_port():
    import socket
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    port = s.getsockname()[1]
    s.close()
    return port


def launch_ranks(world, argv, grace=30.0):
    """Starts one child process per rank (this file, same arguments), forwards rank 0's stdout (the ONE JSON line)
    to this process's stdout and everything else to stderr, and returns the worst exit code.  The parent never
    touches the GPU; no process that has touched it is ever replaced.  If a rank dies, the others get `grace`
    seconds to finish (their collectives cannot complete any more) and are then terminated by PID."""
    port = free_port()
    me = os.path.abspath(__file__)
    procs = []
    for r in range(world):
        procs.append(subprocess.Popen([sys.executable, "-u", me] + list(argv), env=rank_environment(os.environ, r, world, port),
                                      stdout=subprocess.PIPE if r == 0 else sys.stderr, stderr=sys.stderr))

    def forward():
        for line in procs[0].stdout:
            sys.stdout.write(line.decode(errors="replace"))
            sys.stdout.flush()

    fw = threading.Thread(target=forward, daemon=True)
    fw.start()
    first_failure = None
    while any(p.poll() is None for p in procs):
        time.sleep(0.2)
        failed = [p for p in procs if p.poll() not in (None, 0)]
        if failed and first_failure is None:
            first_failure = time.monotonic()
        if first_failure is not None and time.monotonic() - first_failure > grace:
            for p in procs:
                if p.poll() is None:
                    p.terminate()
            time.sleep(5.0)
            for p in procs:
                if p.poll() is None:
                    p.kill()
    fw.join(10.0)
    codes = [p.wait() for p in procs]
    worst = max((c for c in codes), key=lambda c: (c != 0, abs(c)))
    if worst != 0:
        sys.stderr.write("bench.py: exit codes of the %d ranks: %s\n" % (world, codes))
    return worst if 0 <= worst < 256 else 1


def run_dist_legs_isolated(args, dist, rank, local_rank, world):
    """The multi-rank legs (their own RCCL communicator inside the library, an all-gather that has only ever run with
    one rank where this was written) run in a child process per rank, a torch.distributed job of its own on a fresh
    port: whatever happens to it - an exception, an abort inside a collective, a hang - the rank that holds the
    headline measurement survives and reports it.  Returns the legs (rank 0) or an {"error": ...} record."""
    port = [free_port() if rank == 0 else None]
    dist.broadcast_object_list(port, src=0)
    env = rank_environment(os.environ, rank, world, port[0])
    env["LOCAL_RANK"] = str(local_rank)
    child = subprocess.Popen([sys.executable, "-u", os.path.abspath(__file__), "--legs-child", "--backend", args.backend],
                             env=env, stdout=subprocess.PIPE if rank == 0 else sys.stderr, stderr=sys.stderr)
    try:
        outb, _ = child.communicate(timeout=args.legs_timeout)
    except subprocess.TimeoutExpired:
        child.kill()
        child.wait()
        return {"error": "the multi-rank legs did not finish within %d s" % args.legs_timeout}
    if rank != 0:
        return None
    lines = [l for l in (outb or b"").decode(errors="replace").splitlines() if l.startswith("{")]
    if child.returncode != 0 or not lines:
        return {"error": "the multi-rank legs ended with exit code %s" % child.returncode}
    return json.loads(lines[-1])


def legs_child_main(args):
    """A rank of the job run_dist_legs_isolated starts: the multi-rank legs alone; rank 0 prints them as one JSON line."""
    rank, local_rank, world = (int(os.environ[k]) for k in ("RANK", "LOCAL_RANK", "WORLD_SIZE"))
    hook = os.environ.get("ORBHIP_BENCH_LEGS_CHILD_SELFTEST")
    if hook:  # tests/test_bench_launcher.py: how the parent copes with a child that answers, aborts or hangs
        if hook == "abort":
            os.abort()
        if hook == "hang":
            time.sleep(3600)
        print(json.dumps({"selftest": hook, "rank": rank, "world": world, "port": os.environ.get("MASTER_PORT")}), flush=True)
        return
    orb = importlib.import_module("orb-slam2_amd")
    orb._lib.lib()
    import torch
    import torch.distributed as dist
    dev = local_rank % torch.cuda.device_count()
    torch.cuda.set_device(dev)
    gloo = args.backend == "gloo"
    STDOUT.divert()
    if gloo:
        dist.init_process_group("gloo", rank=rank, world_size=world)
    else:
        dist.init_process_group("nccl", rank=rank, world_size=world, device_id=torch.device("cuda", dev))

    def max_over_ranks(x):
        t = torch.tensor([x], dtype=torch.float64, device="cpu" if gloo else "cuda:%d" % dev)
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        return float(t.item())

    legs = dist_legs(orb, torch, dist, dev, rank, world, dist.barrier, max_over_ranks, gloo)
    dist.barrier()
    if rank == 0:
        STDOUT.emit(legs)
    dist.destroy_process_group()


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=20)
    ap.add_argument("--warmup", type=int, default=3)
    ap.add_argument("--batch", type=int, default=1024, help="frames per GPU per step")
    ap.add_argument("--max-batch", type=int, default=0,
                    help="frames the extractor's arenas hold (default: the whole batch); smaller = the step runs as chunks of this size")
    ap.add_argument("--width", type=int, default=640)
    ap.add_argument("--height", type=int, default=480)
    ap.add_argument("--features", type=int, default=1000)
    ap.add_argument("--cpu-sample", type=int, default=256, help="frames of the CPU baseline sample")
    ap.add_argument("--no-extra", action="store_true", help="headline measurement only")
    ap.add_argument("--dist-legs", action="store_true",
                    help="run the multi-rank legs even with one rank (rehearsal of the N > 1 code on one GPU)")
    ap.add_argument("--legs-timeout", type=int, default=900, help="seconds the legs after the headline may take")
    ap.add_argument("--backend", choices=("nccl", "gloo"), default=os.environ.get("ORBHIP_BENCH_BACKEND", "nccl"),
                    help="torch.distributed backend of the N > 1 run.  nccl (RCCL) is the measured configuration; gloo is a "
                         "rehearsal of the multi-rank code on fewer GPUs than ranks (ranks share devices, records travel "
                         "through host memory, the library's own RCCL communicator is left out)")
    ap.add_argument("--legs-child", action="store_true", help=argparse.SUPPRESS)  # internal: see run_dist_legs_isolated
    args = ap.parse_args()

    if args.legs_child:
        return legs_child_main(args)
    if args.gpus > 1 and "WORLD_SIZE" not in os.environ:
        # `python bench.py --gpus N` without a launcher: this process starts the N ranks itself.  It has not
        # touched the GPU (no HIP call, no torch.cuda, liborbhip.so not loaded) and never does.
        sys.exit(launch_ranks(args.gpus, sys.argv[1:]))

    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    args.gpus = world
    if os.environ.get("ORBHIP_BENCH_LAUNCH_SELFTEST"):
        # tests/test_bench_launcher.py: what a rank was started with, before anything heavy is imported
        print(json.dumps({"rank": rank, "local_rank": local_rank, "world": world, "master": os.environ.get("MASTER_ADDR"),
                          "port": os.environ.get("MASTER_PORT"), "steps": args.steps}), flush=True)
        sys.exit(int(os.environ.get("ORBHIP_BENCH_LAUNCH_SELFTEST_RC_RANK%d" % rank, "0")))

    orb = importlib.import_module("orb-slam2_amd")
    orb._lib.lib()  # fails loudly if liborbhip.so is missing
    import torch
    import torch.distributed as dist

    if not torch.cuda.is_available():
        raise SystemExit("bench.py needs a GPU")
    dev = local_rank % torch.cuda.device_count()
    torch.cuda.set_device(dev)
    use_dist = world > 1 or args.dist_legs
    gloo = args.backend == "gloo"
    if world > torch.cuda.device_count() and not gloo:
        raise SystemExit("%d ranks on %d GPUs: RCCL needs one GPU per rank (--backend gloo rehearses the multi-rank code "
                         "with ranks sharing devices)" % (world, torch.cuda.device_count()))
    if use_dist:
        STDOUT.divert()
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        os.environ.setdefault("MASTER_PORT", "29533")
        if gloo:
            dist.init_process_group("gloo", rank=rank, world_size=world)
        else:
            dist.init_process_group("nccl", rank=rank, world_size=world, device_id=torch.device("cuda", dev))

    def barrier():
        if use_dist:
            dist.barrier()

    def max_over_ranks(x):
        if not use_dist:
            return x
        t = torch.tensor([x], dtype=torch.float64, device="cpu" if gloo else "cuda:%d" % dev)
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        return float(t.item())

    wl = ExtractWorkload(orb, torch, dev, args.width, args.height, args.features, args.batch, first_idx=rank * args.batch,
                         max_batch=args.max_batch or None)

    for _ in range(args.warmup):
        wl.step()
    wl.sync()
    barrier()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(args.steps):
        wl.step()
    wl.sync()
    torch.cuda.synchronize()
    barrier()
    elapsed = max_over_ranks(time.perf_counter() - t0)
    frames_total = args.batch * args.steps * world
    value = frames_total / elapsed

    dog = Watchdog(args.legs_timeout + 120, rank)  # the isolated multi-rank legs time out by themselves first
    out = None
    if rank == 0:
        roofline, N_frame = extract_roofline(wl, max(3, min(args.steps, 10)))
        out = {"metric": "orb_extract_frames_per_sec", "value": round(value, 2), "unit": "frames/s", "n_gpus": world,
               "steps": args.steps, "warmup": args.warmup, "ms_per_step": round(elapsed / args.steps * 1e3, 4),
               "higher_is_better": True, "scaling": "weak", "vs_baseline": None, "dtype": "u8", "data": "synthetic",
               "config": {"workload": "batch of %d synthetic %dx%d frames per GPU, %d features/frame, 8 levels, scale 1.2, "
                                      "full extract+describe" % (args.batch, args.width, args.height, args.features),
                          "frames_per_gpu_per_step": args.batch, "keypoints_per_frame": round(N_frame, 1),
                          "parallelism": "frame-batch sharding x%d, no data-path collective" % world,
                          "ranks": world, "backend": ("gloo (rehearsal: %d ranks on %d GPU(s))" % (world, torch.cuda.device_count()))
                          if gloo and use_dist else ("nccl (RCCL)" if use_dist else None)},
               "roofline": roofline, "cpu_baseline": None, "extra": {}}
        # the same algorithmic bytes over the timed step of THIS rank's GPU (sub-batches and the blur run side by
        # side there; the stage times above come from a pass with one stream and everything in line)
        wp = roofline["whole_path"]
        step_s = elapsed / args.steps
        wp["timed_step_ms"] = round(step_s * 1e3, 4)
        wp["achieved_over_timed_step"] = round(wp["algorithmic_bytes_per_step"] / step_s / 1e9, 2)
        wp["frac_over_timed_step"] = round(wp["algorithmic_bytes_per_step"] / step_s / 1e9 / HBM_PEAK_GBS, 5)
    dog.arm(out)

    if rank == 0 and world == 1 and args.cpu_sample > 0:
        # ---- CPU baseline on a bounded sample of the same frames + parity of that sample ------------
        threads = max(1, min(16, os.cpu_count() or 1))
        cpu1, ref_results = cpu_extract(wl.host, args.features, 1, args.cpu_sample)
        cpuN, _ = cpu_extract(wl.host, args.features, threads, args.cpu_sample)
        out["cpu_baseline"] = {
            "value": round(cpu1, 3), "unit": "frames/s", "cores": 1, "kind": "port",
            "sample": "%d of the %d frames of this workload (%dx%d, %d features), the oracle extractor on one thread "
                      "(the reference runs the path on one thread per frame)"
                      % (len(ref_results), args.batch, args.width, args.height, args.features),
            "all_cores": {"value": round(cpuN, 3), "cores": threads, "sample": "the same frames, one oracle extractor per thread"},
            "gpu_over_cpu_1_thread": round(value / cpu1, 1), "gpu_over_cpu_all_cores": round(value / cpuN, 1),
            "parity_frames_checked": len(ref_results), "parity_frames_mismatching": wl.mismatching(ref_results)}

    if not args.no_extra:
        try:
            if use_dist:
                legs = run_dist_legs_isolated(args, dist, rank, local_rank, world)
            else:
                legs = single_gpu_legs(orb, torch, dev, wl)
        except Exception as ex:  # a leg that raises must not lose the headline measurement either
            dog.cancel()
            if rank == 0:
                out["extra"] = {"error": "a leg after the headline measurement raised %s: %s" % (type(ex).__name__, ex)}
                STDOUT.emit(out)
            sys.stderr.write("rank %d: %r\n" % (rank, ex))
            sys.stderr.flush()
            os._exit(0)  # no collective teardown: the other ranks may be anywhere
        if rank == 0:
            out["extra"] = legs
    barrier()
    dog.cancel()
    if rank == 0:
        STDOUT.emit(out)
    if use_dist:
        dist.destroy_process_group()


# ------------------------------------------------------------------------------------------------
# Hamming kNN helpers
# ------------------------------------------------------------------------------------------------

MFMA_FP4_PEAK_TFLOPS = 10000.0  # dense 4-bit matrix rate of an MI355X (4 x the 2.5 PFLOP/s bf16 rate), nominal clock


def knn_roofline(rate, r_bcnt, kernel="k_knn2_mm"):
    """The scan runs on the matrix cores (k_knn2_mm): one comparison = 256 multiply-adds of 4-bit operands = 512
    flop, priced against the dense FP4 matrix peak.  `vs_popcount` keeps the previous formulation's yardstick: one
    comparison = 8 v_xor + 8 v_bcnt lane-ops against the popcount rate the device sustains (measured live by the
    library with every CU busy); above 1 means faster than ANY xor + popcount loop could be on this device."""
    tf = rate * 512 / 1e12
    return {"bound": "mfma", "kernel": kernel, "achieved": float("%.4g" % tf), "peak": MFMA_FP4_PEAK_TFLOPS,
            "unit": "TFLOP/s", "frac": round(tf / MFMA_FP4_PEAK_TFLOPS, 4), "traffic": None,
            "vs_popcount": {"achieved": float("%.4g" % (rate * 8)), "peak": float("%.4g" % r_bcnt),
                            "unit": "v_bcnt lane-ops/s", "frac": round(rate * 8 / r_bcnt, 4) if r_bcnt else None}}


def device_descriptors(torch, cuda, pairs, nq, nt, seed):
    """Query and train sets generated on the device with the make-up the configuration specifies (and
    synth.synth_descriptors has on the host): uniform random bytes; a twentieth of each train set overwritten with
    copies of other rows of the set; a quarter of the queries copies of a random train row of their pair with 0 - 40
    randomly chosen bits flipped (a bit drawn twice flips back, as in the host generator)."""
    g = torch.Generator(device=cuda)
    g.manual_seed(seed)
    q = torch.randint(0, 256, (pairs, nq, 32), dtype=torch.uint8, device=cuda, generator=g)
    t = torch.randint(0, 256, (pairs, nt, 32), dtype=torch.uint8, device=cuda, generator=g)
    pidx = torch.arange(pairs, device=cuda)[:, None]
    ndup = max(1, nt // 20)
    dst = torch.randint(0, nt, (pairs, ndup), device=cuda, generator=g)
    src = torch.randint(0, nt, (pairs, ndup), device=cuda, generator=g)
    t[pidx, dst] = t[pidx, src]
    ncopy = nq // 4
    qi = torch.rand((pairs, nq), device=cuda, generator=g).argsort(dim=1)[:, :ncopy]  # ncopy distinct queries per pair
    ti = torch.randint(0, nt, (pairs, ncopy), device=cuda, generator=g)
    planted = t[pidx, ti].clone()
    nflip = torch.randint(0, 41, (pairs, ncopy), device=cuda, generator=g)
    bits = torch.randint(0, 256, (pairs, ncopy, 40), device=cuda, generator=g)
    for k in range(40):
        on = (nflip > k).to(torch.uint8)
        byte = (bits[:, :, k] >> 3)[:, :, None]
        mask = (on * (1 << (bits[:, :, k] & 7)).to(torch.uint8))[:, :, None]
        planted.scatter_(2, byte, planted.gather(2, byte) ^ mask)
    q[pidx, qi] = planted
    return q, t


def falling_distance_descriptors(torch, cuda, pairs, nq, nt, seed, distinct=64):
    """A bad case for a streaming two-smallest scan: inside a pair every query is a near-copy of one base descriptor
    and train row j differs from that base in about 200 - 180 j / nt bits, so distances keep falling along the scan and
    the running second-best is beaten again and again to the very end.  `distinct` pairs are generated and repeated."""
    g = torch.Generator(device=cuda)
    g.manual_seed(seed)
    base = torch.randint(0, 256, (distinct, 1, 32), dtype=torch.uint8, device=cuda, generator=g)
    weights = (1 << torch.arange(8, device=cuda)).to(torch.uint8)

    def flips(rows, prob):  # [distinct, rows, 32] bytes whose bits are set with probability prob[row]
        r = torch.rand((distinct, rows, 32, 8), device=cuda, generator=g) < prob[None, :, None, None]
        return (r.to(torch.uint8) * weights).sum(dim=3).to(torch.uint8)

    p_t = (200.0 - 180.0 * torch.arange(nt, device=cuda) / nt) / 256.0
    t = base ^ flips(nt, p_t)
    q = base ^ flips(nq, torch.full((nq,), 4.0 / 256.0, device=cuda))
    rep = (pairs + distinct - 1) // distinct
    return q.repeat(rep, 1, 1)[:pairs].contiguous(), t.repeat(rep, 1, 1)[:pairs].contiguous()


def knn_pairs_leg(orb, torch, dev, m, pairs=4096, nq=2000, nt=2000, reps=5, data=None):
    """4096 independent pairs of 2000 x 2000 descriptors: returns (median ms, device tensors)."""
    cuda = "cuda:%d" % dev
    q, t = data if data is not None else device_descriptors(torch, cuda, pairs, nq, nt, seed=1)
    o = torch.zeros((3, pairs, nq), dtype=torch.int32, device=cuda)
    torch.cuda.synchronize(dev)

    def run():
        m.knn2_device(q.data_ptr(), nq, nq * 32, t.data_ptr(), nt, nt * 32, pairs, 0, o[0].data_ptr(), o[1].data_ptr(),
                      o[2].data_ptr())

    run()
    m.synchronize()
    ms = []
    for _ in range(reps):
        run()
        m.synchronize()
        ms.append(m.last_elapsed_ms())
    return float(np.median(ms)), run, q, t, o


def cpu_knn(orbref, q, t, threads):
    """Scalar scan of the oracle: `threads` pairs in parallel (ctypes releases the GIL)."""
    from concurrent.futures import ThreadPoolExecutor
    t0 = time.perf_counter()
    with ThreadPoolExecutor(threads) as pool:
        res = list(pool.map(lambda i: orbref.knn2(q[i], t[i]), range(len(q))))
    return len(q) * q.shape[1] * t.shape[1] / (time.perf_counter() - t0), res


# ------------------------------------------------------------------------------------------------
# N = 1: everything else the library does, beside the headline
# ------------------------------------------------------------------------------------------------

def single_gpu_legs(orb, torch, dev, wl0):
    orbref = oracle()
    threads = max(1, min(16, os.cpu_count() or 1))
    res = {}
    cuda = "cuda:%d" % dev

    # ---- host buffers in, host buffers out (PCIe inclusive), same 256 frames ---------------------
    import ctypes as C
    L = orb._lib.lib()
    b, cap = wl0.batch, wl0.nf
    kp = np.zeros((b, cap), dtype=orb.KP_DTYPE)
    desc = np.zeros((b, cap, 32), dtype=np.uint8)
    n = np.zeros(b, dtype=np.int32)

    def host_batch(src=None):
        orb._lib.check(L.orbhip_extract_batch(wl0.ext._h, src or wl0.host.ctypes.data_as(C.c_void_p), b, wl0.h, wl0.w, wl0.w,
                                              wl0.w * wl0.h, kp.ctypes.data_as(C.c_void_p), desc.ctypes.data_as(C.c_void_p),
                                              cap, n.ctypes.data_as(C.c_void_p)))

    # the same frames in page-locked memory: the library copies them to the device from where they lie
    pinned = torch.from_numpy(wl0.host).pin_memory()
    host_batch(C.c_void_p(pinned.data_ptr()))
    tp = []
    for _ in range(5):
        t0 = time.perf_counter()
        host_batch(C.c_void_p(pinned.data_ptr()))
        tp.append(time.perf_counter() - t0)
    n_pinned, desc0_pinned = n.copy(), desc[0].copy()
    host_batch()
    ts = []
    for _ in range(5):
        t0 = time.perf_counter()
        host_batch()
        ts.append(time.perf_counter() - t0)
    dn = wl0.d_n.cpu().numpy()
    same = np.array_equal(n, dn) and all(
        np.array_equal(desc[i, :n[i]], wl0.d_desc[i, :n[i]].cpu().numpy()) for i in range(0, b, 37))
    res["extract_host_buffers_pcie_inclusive"] = {
        "frames_per_sec": round(b / float(np.median(ts)), 1), "batch": b, "ms_per_batch": round(float(np.median(ts)) * 1e3, 3),
        "what": "orbhip_extract_batch: pageable host images in, keypoints and descriptors back in host memory, "
                "one blocking call per %d frames" % b,
        "same_output_as_device_path": bool(same),
        "page_locked_input": {"frames_per_sec": round(b / float(np.median(tp)), 1), "ms_per_batch": round(float(np.median(tp)) * 1e3, 3),
                              "what": "the same call with the frames in page-locked host memory (no staging copy on the host)",
                              "same_output": bool(np.array_equal(n_pinned, n) and np.array_equal(desc0_pinned, desc[0]))}}

    # ---- one frame per call: the call of src/Frame.cc:61 ------------------------------------------
    one = orb.ORBextractor(wl0.nf, 1.2, 8, 1, 20, device=dev, max_width=wl0.w, max_height=wl0.h, max_batch=1)
    img = wl0.host[0]
    kp1 = np.zeros(cap, dtype=orb.KP_DTYPE)
    d1 = np.zeros((cap, 32), dtype=np.uint8)
    n1 = C.c_int(0)

    def one_call():
        orb._lib.check(L.orbhip_extract(one._h, img.ctypes.data_as(C.c_void_p), wl0.h, wl0.w, wl0.w,
                                        kp1.ctypes.data_as(C.c_void_p), d1.ctypes.data_as(C.c_void_p), cap, C.byref(n1)))

    for _ in range(20):
        one_call()
    ts = []
    for _ in range(300):
        t0 = time.perf_counter()
        one_call()
        ts.append(time.perf_counter() - t0)
    rext = orbref.RefExtractor(wl0.nf, 1.2, 8, 1, 20)
    rext(img)
    tc = []
    for _ in range(20):
        t0 = time.perf_counter()
        rext(img)
        tc.append(time.perf_counter() - t0)
    lat = {"calls": 300, "c_abi_median_us": round(float(np.median(ts)) * 1e6, 1),
           "c_abi_p90_us": round(float(np.percentile(ts, 90)) * 1e6, 1),
           "cpu_oracle_1_thread_median_us": round(float(np.median(tc)) * 1e6, 1),
           "what": "orbhip_extract on one %dx%d frame in host memory, %d features, results back in host memory"
                   % (wl0.w, wl0.h, wl0.nf)}
    exe = os.path.join(ROOT, "orb-slam2_amd", "shim", "shim_test")
    if os.path.exists(exe):
        with tempfile.TemporaryDirectory() as td:
            raw = os.path.join(td, "in.raw")
            img.tofile(raw)
            try:
                r = subprocess.run([exe, "--latency", raw, str(wl0.h), str(wl0.w), str(wl0.nf), "300"], capture_output=True,
                                   text=True, timeout=120)
                j = json.loads(r.stdout.strip().splitlines()[-1])
                lat["cpp_class_median_us"] = j["median_us"]
                lat["cpp_class_p90_us"] = j["p90_us"]
                lat["cpp_class_what"] = "ORB_SLAM::ORBextractor::operator() of the forwarding C++ class, 300 calls (shim_test --latency)"
            except (OSError, ValueError, IndexError, KeyError, subprocess.SubprocessError) as e:
                lat["cpp_class_error"] = repr(e)[:200]
    res["single_frame_latency"] = lat
    del one

    # ---- the two larger extraction configurations, with their stage rooflines -----------------------
    for name, w, h, nf, first, distinct in (("extract_1280x720_2000", 1280, 720, 2000, 0, 256),
                                             ("extract_1920x1080_4000", 1920, 1080, 4000, 300000, 32)):
        # 1080p: 32 distinct frames, repeated (host generation of 256 such frames would take longer than the run);
        # it is the shard one GPU of eight owns in the 2048-frame configuration
        bb = 256
        fr = np.concatenate([make_frames(orb.synth, w, h, distinct, first)] * (bb // distinct))
        wl = ExtractWorkload(orb, torch, dev, w, h, nf, bb, first_idx=0, frames=fr)
        k = 10 if w < 1500 else 5
        dt = wl.timed(k, 2)
        roof, nkp = extract_roofline(wl, 3)
        nchk = 16  # of the 256 frames of the chunk (1080p: 16 of its 32 distinct ones), oracle on all host threads
        rr = cpu_extract(fr, nf, threads, nchk)[1]
        res[name] = {"frames_per_sec": round(bb * k / dt, 1), "batch": bb, "ms_per_step": round(dt / k * 1e3, 3),
                     "keypoints_per_frame": round(nkp, 1), "roofline": roof,
                     "parity_frames_checked": nchk, "parity_frames_mismatching": wl.mismatching(rr)}
        del wl, fr

    # ---- Hamming kNN k=2: 4096 pairs of 2000 x 2000 descriptors (1.6384e10 comparisons) ------------
    m = orb.ORBmatcher(0.6, True, device=dev)
    pairs, nq, nt = 4096, 2000, 2000
    best, _, q, t, o = knn_pairs_leg(orb, torch, dev, m, pairs, nq, nt, reps=8)
    rate = pairs * nq * nt / (best * 1e-3)
    # denominators measured live on this device (sustained, every CU busy)
    r_bcnt, r_mix, r_add = m.valu_rate(0), m.valu_rate(1), m.valu_rate(2)
    hq, ht = q[:threads].cpu().numpy(), t[:threads].cpu().numpy()
    cpu1, r1 = cpu_knn(orbref, hq[:1], ht[:1], 1)
    cpuN, rN = cpu_knn(orbref, hq, ht, threads)
    oo = o[:, :threads].cpu().numpy()
    ok = all(np.array_equal(oo[0, i], rN[i][0]) and np.array_equal(oo[1, i], rN[i][1]) and np.array_equal(oo[2, i], rN[i][2])
             for i in range(len(rN)))
    res["hamming_knn2"] = {"pairs": pairs, "nq": nq, "nt": nt, "comparisons_per_sec": float("%.4g" % rate),
                           "data": "per pair: uniform random descriptors, 5 % of the train rows copies of other rows, 25 % of "
                                   "the queries near-copies of a train row (0 - 40 bits flipped), generated on the device",
                           "ms": round(best, 4), "roofline": knn_roofline(rate, r_bcnt),
                           "sustained_lane_ops_per_sec": {"v_bcnt_u32_b32": float("%.4g" % r_bcnt),
                                                          "xor_bcnt_mix": float("%.4g" % r_mix),
                                                          "v_add_u32": float("%.4g" % r_add)},
                           "cpu_oracle_1_thread_comparisons_per_sec": float("%.4g" % cpu1),
                           "cpu_oracle_%d_threads_comparisons_per_sec" % threads: float("%.4g" % cpuN),
                           "parity_pairs_checked": len(rN), "parity": bool(ok)}
    # the same pairs through the xor + popcount kernel the matrix-core one replaced
    m.set_scan_kernel("popcount")
    best_pc = knn_pairs_leg(orb, torch, dev, m, pairs, nq, nt, reps=3)[0]
    m.set_scan_kernel("matrix")
    rate_pc = pairs * nq * nt / (best_pc * 1e-3)
    res["hamming_knn2"]["popcount_kernel_comparisons_per_sec"] = float("%.4g" % rate_pc)
    # the xor + popcount kernel against the popcount rate the device sustains (8 v_bcnt lane-ops per comparison); an
    # xor + popcount stream cannot exceed 0.64 of it (8 x 4.05 of every 8 x 4.05 + 8 x 2.3 issue cycles)
    res["hamming_knn2"]["popcount_kernel_roofline"] = {
        "bound": "valu", "kernel": "k_knn2", "achieved": float("%.4g" % (rate_pc * 8)), "peak": float("%.4g" % r_bcnt),
        "unit": "v_bcnt lane-ops/s", "frac": round(rate_pc * 8 / r_bcnt, 4) if r_bcnt else None, "ceiling_of_the_formulation": 0.64}
    del q, t, o
    # the same shape on two other inputs: featureless noise (nothing planted: the usual case of an unrelated frame
    # pair) and distances that fall along the whole scan (the bad case of a streaming two-smallest scan)
    g0 = torch.Generator(device=cuda)
    g0.manual_seed(1)
    noise = (torch.randint(0, 256, (pairs, nq, 32), dtype=torch.uint8, device=cuda, generator=g0),
             torch.randint(0, 256, (pairs, nt, 32), dtype=torch.uint8, device=cuda, generator=g0))
    for name, data in (("uniform_noise", noise), ("falling_distances", falling_distance_descriptors(torch, cuda, pairs, nq, nt, 3))):
        ms_x, _, qx, tx, ox = knn_pairs_leg(orb, torch, dev, m, pairs, nq, nt, reps=4, data=data)
        rate_x = pairs * nq * nt / (ms_x * 1e-3)
        rx = orbref.knn2(qx[0].cpu().numpy(), tx[0].cpu().numpy())
        ohx = ox[:, 0].cpu().numpy()
        m.set_scan_kernel("popcount")
        ms_xp = knn_pairs_leg(orb, torch, dev, m, pairs, nq, nt, reps=2, data=data)[0]
        m.set_scan_kernel("matrix")
        res["hamming_knn2"][name] = {
            "comparisons_per_sec": float("%.4g" % rate_x), "ms": round(ms_x, 4), "frac": knn_roofline(rate_x, r_bcnt)["frac"],
            "popcount_kernel_comparisons_per_sec": float("%.4g" % (pairs * nq * nt / (ms_xp * 1e-3))),
            "popcount_kernel_frac": round(pairs * nq * nt / (ms_xp * 1e-3) * 8 / r_bcnt, 4) if r_bcnt else None,
            "parity_pair0": bool(np.array_equal(ohx[0], rx[0]) and np.array_equal(ohx[1], rx[1]) and np.array_equal(ohx[2], rx[2]))}
        del qx, tx, ox, data
    del noise
    # ---- frame-vs-map shard: 4000 queries against the 1.25 M map rows one GPU of eight owns ---------
    nq2, nt2 = 4000, 1250000
    q2, t2 = device_descriptors(torch, cuda, 1, nq2, nt2, seed=2)
    q2, t2 = q2[0], t2[0]
    o2 = torch.zeros((3, nq2), dtype=torch.int32, device=cuda)
    torch.cuda.synchronize(dev)
    # the first calls after an idle or differently loaded period run slower (the clock follows the load): 16 calls,
    # median of the last 10, and the popcount rate of the device measured again right after them
    ms2 = []
    for _ in range(16):
        m.knn2_device(q2.data_ptr(), nq2, 0, t2.data_ptr(), nt2, 0, 1, 0, o2[0].data_ptr(), o2[1].data_ptr(), o2[2].data_ptr())
        m.synchronize()
        ms2.append(m.last_elapsed_ms())
    best2 = float(np.median(ms2[6:]))
    r_bcnt2 = 0.5 * (r_bcnt + m.valu_rate(0))
    hq2, ht2 = q2[:16].cpu().numpy(), t2.cpu().numpy()
    r_idx, r_d1, r_d2 = orbref.knn2(hq2, ht2)
    oo2 = o2[:, :16].cpu().numpy()
    rate2 = nq2 * nt2 / (best2 * 1e-3)
    res["hamming_knn2_map_shard"] = {"nq": nq2, "nt": nt2, "ms": round(best2, 4),
                                     "comparisons_per_sec": float("%.4g" % rate2), "roofline": knn_roofline(rate2, r_bcnt2),
                                     "calls": 16, "ms_first_call": round(ms2[0], 4),
                                     "parity_queries_checked": 16,
                                     "parity": bool(np.array_equal(oo2[0], r_idx) and np.array_equal(oo2[1], r_d1)
                                                    and np.array_equal(oo2[2], r_d2))}
    del q2, t2
    res["windowed_matchers"] = windowed_leg(orb, torch, dev, orbref)
    res["bow_transform"] = bow_leg(orb, torch, dev, orbref, m)
    return res


# ------------------------------------------------------------------------------------------------
# N > 1: the rank-local legs of the multi-GPU configurations
# ------------------------------------------------------------------------------------------------

def dist_legs(orb, torch, dist, dev, rank, world, barrier, max_over_ranks, gloo=False):
    """Every rank runs these; rank 0 returns the record.  Timing as for the headline: barrier and device
    synchronisation on both sides, MAX over ranks.  gloo: rehearsal on fewer GPUs than ranks — the per-shard records
    travel through host memory (gloo all-gather) into the device merge, and the library's own RCCL communicator,
    which needs one GPU per rank, is left out."""
    cuda = "cuda:%d" % dev
    shard_range = orb.sharded.shard_range
    res = {"ranks": world}

    def timed(fn, sync, steps, warmup=1):
        for _ in range(warmup):
            fn()
        sync()
        barrier()
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        for _ in range(steps):
            fn()
        sync()
        torch.cuda.synchronize()
        barrier()
        return max_over_ranks(time.perf_counter() - t0)

    # ---- frame-pair kNN: 4096 pairs of 2000 x 2000 per rank, no collective (weak scaling) -----------
    m = orb.ORBmatcher(0.6, True, device=dev)
    pairs, nq, nt = 4096, 2000, 2000
    _, run, q, t, o = knn_pairs_leg(orb, torch, dev, m, pairs, nq, nt, reps=1)
    steps = 3
    dt = timed(run, m.synchronize, steps)
    rate = world * pairs * nq * nt * steps / dt
    r_bcnt = m.valu_rate(0)
    if rank == 0:
        orbref = oracle()
        ridx, rd1, rd2 = orbref.knn2(q[0].cpu().numpy(), t[0].cpu().numpy())
        oo = o[:, 0].cpu().numpy()
        res["hamming_knn2_pairs"] = {
            "scaling": "weak", "pairs_per_rank": pairs, "nq": nq, "nt": nt, "steps": steps,
            "comparisons_per_sec": float("%.4g" % rate), "ms_per_step": round(dt / steps * 1e3, 3),
            "roofline": knn_roofline(rate / world, r_bcnt), "collective": None,
            "parity_pair0": bool(np.array_equal(oo[0], ridx) and np.array_equal(oo[1], rd1) and np.array_equal(oo[2], rd2))}
    del q, t, o, run

    # ---- 2048 frames of 1920x1080 / 4000 features, contiguous split (strong scaling) ----------------
    total, w, h, nf, distinct = 2048, 1920, 1080, 4000, 32
    f0, f1 = shard_range(total, rank, world)
    pool = make_frames(orb.synth, w, h, distinct, 300000)  # frame i of the batch is pool[i % 32] on every rank
    mine = np.stack([pool[i % distinct] for i in range(f0, f1)]) if f1 > f0 else np.zeros((0, h, w), np.uint8)
    wl = ExtractWorkload(orb, torch, dev, w, h, nf, f1 - f0, first_idx=0, frames=mine, max_batch=256)
    del mine
    steps = 3
    dt = timed(wl.step, wl.sync, steps)
    if rank == 0:
        nchk = min(8, f1 - f0)
        rr = cpu_extract(np.stack([pool[(f0 + i) % distinct] for i in range(nchk)]), nf, min(8, os.cpu_count() or 1), nchk)[1]
        P = pyramid_pixels(wl.ext, wl.ext.GetLevels())
        nkp = float(wl.d_n.float().mean().item())
        fps = total * steps / dt
        res["extract_2048x1920x1080_4000"] = {
            "scaling": "strong", "frames_total": total, "frames_this_rank": f1 - f0, "steps": steps,
            "frames_per_sec": round(fps, 1), "ms_per_step": round(dt / steps * 1e3, 3), "keypoints_per_frame": round(nkp, 1),
            "algorithmic_gb_per_sec": round(fps * (5 * P + 60 * nkp) / 1e9, 1),
            "frac_of_hbm_peak_all_ranks": round(fps * (5 * P + 60 * nkp) / 1e9 / (HBM_PEAK_GBS * world), 5),
            "collective": None, "parity_frames_checked": nchk, "parity_frames_mismatching": wl.mismatching(rr)}
    del wl, pool

    # ---- 4000 queries against a 10 M-row map, one index-contiguous shard per rank --------------------
    NT, nq = 10000000, 4000
    planted = 64

    def make_shard(r):
        """Shard r of the map, the same bytes on whichever rank builds it.  Query i < 64 is planted as an
        exact copy at row 1000 + 17 i of shard i % world and again further down in every later shard, so that
        the first of several equal rows must win across the shard boundary."""
        a, e = shard_range(NT, r, world)
        g = torch.Generator(device=cuda)
        g.manual_seed(5000 + r)
        s = torch.randint(0, 256, (e - a, 32), dtype=torch.uint8, device=cuda, generator=g)
        for i in range(planted):
            if i % world == r:
                s[1000 + 17 * i] = dq[i]
            elif i % world < r:
                s[(e - a) // 2 + 13 * i] = dq[i]
        return a, e, s

    g = torch.Generator(device=cuda)
    g.manual_seed(99)
    dq = torch.randint(0, 256, (nq, 32), dtype=torch.uint8, device=cuda, generator=g)  # replicated: same seed
    a, e, shard = make_shard(rank)
    out_c = torch.full((3, nq), -5, dtype=torch.int32, device=cuda)
    out_t = torch.full((3, nq), -6, dtype=torch.int32, device=cuda)
    parts_local = torch.zeros((3, nq), dtype=torch.int32, device=cuda)
    parts_all = torch.zeros((world, 3, nq), dtype=torch.int32, device=cuda)
    torch.cuda.synchronize()
    steps = 5
    dt_t = timed(lambda: orb.sharded.knn2_sharded(m, dq, nq, shard, e - a, a, parts_local, parts_all, out_t, dist=dist),
                 m.synchronize, steps, warmup=3)
    if gloo:
        out_c, dt_c = out_t, dt_t
    else:
        comm = orb.sharded.Communicator.from_torch_distributed(dist, dev)
        dt_c = timed(lambda: orb.sharded.knn2_sharded_rccl(m, comm, dq, nq, shard, e - a, a, out_c), m.synchronize, steps, warmup=6)
        comm.close()
    same = torch.tensor([int(torch.equal(out_c, out_t))], dtype=torch.int32, device="cpu" if gloo else cuda)
    dist.all_reduce(same, op=dist.ReduceOp.MIN)
    if rank == 0:
        orbref = oracle()
        sample = np.concatenate([np.arange(0, planted, 4), np.arange(planted, planted + 8)])
        hq = dq[torch.from_numpy(sample).to(cuda)].cpu().numpy()
        hmap = np.empty((NT, 32), np.uint8)
        for r in range(world):
            ra, re_, s = make_shard(r)
            hmap[ra:re_] = s.cpu().numpy()
            del s
        ridx, rd1, rd2 = orbref.knn2(hq, hmap)
        oc = out_c.cpu().numpy()[:, sample]
        ok = np.array_equal(oc[0], ridx) and np.array_equal(oc[1], rd1) and np.array_equal(oc[2], rd2)
        first_wins = all(int(oc[0, k]) == shard_range(NT, int(i) % world, world)[0] + 1000 + 17 * int(i)
                         for k, i in enumerate(sample) if i < planted)
        rate = nq * NT * steps / dt_c
        res["hamming_knn2_map_10M"] = {
            "scaling": "strong", "nq": nq, "map_rows": NT, "rows_this_rank": e - a, "steps": steps,
            "comparisons_per_sec": float("%.4g" % rate), "ms_per_step": round(dt_c / steps * 1e3, 3),
            "roofline": knn_roofline(rate / world, r_bcnt),
            "collective": ("REHEARSAL: one gloo all-gather of 12 B x %d queries per rank over %d ranks through host memory, "
                           "device merge (orbhip_knn2_merge_device)" % (nq, world)) if gloo else
                          ("one ncclAllGather of 12 B x %d queries per rank over %d ranks (orbhip_knn2_sharded, the "
                           "library's own RCCL communicator, queued on the matcher's stream)" % (nq, world)),
            "torch_distributed_form": {"ms_per_step": round(dt_t / steps * 1e3, 3),
                                       "comparisons_per_sec": float("%.4g" % (nq * NT * steps / dt_t)),
                                       "collective": "one all_gather_into_tensor on the nccl backend + device merge",
                                       "same_result_on_every_rank": bool(int(same.item()))},
            "parity_queries_checked": int(len(sample)), "parity": bool(ok), "first_of_equal_rows_wins_across_shards": bool(first_wins)}
    return res


def bow_leg(orb, torch, dev, orbref, m):
    """Vocabulary descent of 256 x 2000 descriptors through a synthetic tree of the ORB vocabulary's shape (k = 10,
    L = 6: 1,111,111 nodes, 35 MB of node descriptors), next to the oracle on one thread for 2000 descriptors."""
    k, L, n = 10, 6, 256 * 2000
    parent, desc, weight = orb.synth.synth_vocabulary_large(k, L, seed=1, tie_share=0.05)
    voc = orb.ORBVocabulary(parent, desc, weight, L, device=dev)
    cuda = "cuda:%d" % dev
    rng = np.random.default_rng(2)
    f = desc[rng.integers(1, len(desc), n)] ^ (rng.integers(0, 256, (n, 32), dtype=np.uint8)
                                                & rng.integers(0, 256, (n, 32), dtype=np.uint8)
                                                & rng.integers(0, 256, (n, 32), dtype=np.uint8))
    d_f = torch.from_numpy(f).to(cuda)
    word = torch.empty(n, dtype=torch.int32, device=cuda)
    node = torch.empty(n, dtype=torch.int32, device=cuda)
    wgt = torch.empty(n, dtype=torch.float64, device=cuda)
    torch.cuda.synchronize(dev)
    ts = []
    for _ in range(4):
        t0 = time.perf_counter()
        voc.transform_device(m, d_f.data_ptr(), n, 4, word.data_ptr(), node.data_ptr(), wgt.data_ptr())
        m.synchronize()
        ts.append(time.perf_counter() - t0)
    best = float(np.median(ts[1:]))
    ref = orbref.RefVocabulary(parent, desc, weight, L)
    t0 = time.perf_counter()
    rword, rnode, _ = ref.transform_features(f[:2000], 4)
    cdt = time.perf_counter() - t0
    ok = np.array_equal(word[:2000].cpu().numpy(), rword) and np.array_equal(node[:2000].cpu().numpy(), rnode)
    # SearchByBoW (KeyFrame/KeyFrame form) over 256 pairs of 2000 features: ComputeBoW of both batches, then the match
    b, nf = 256, 2000
    kp = np.zeros((b, nf, 7), np.float32)
    kp[:, :, 3] = rng.uniform(0, 360, (b, nf))
    d1 = f.reshape(b, nf, 32)
    flips = rng.integers(0, 256, (b, nf, 32), dtype=np.uint8)
    for _ in range(4):
        flips &= rng.integers(0, 256, (b, nf, 32), dtype=np.uint8)
    d2 = d1 ^ flips
    cnt = torch.full((b,), nf, dtype=torch.int32, device=cuda)
    mm = orb.ORBmatcher(0.8, True, device=dev)
    F1 = orb.FrameBatch(mm, torch.from_numpy(kp).to(cuda), torch.from_numpy(d1).to(cuda), cnt, [0, 0, 1280, 720])
    F2 = orb.FrameBatch(mm, torch.from_numpy(kp).to(cuda), torch.from_numpy(d2).to(cuda), cnt, [0, 0, 1280, 720])
    t0 = time.perf_counter()
    F1.ComputeBoW(voc, 4, node_cap=1024)
    F2.ComputeBoW(voc, 4, node_cap=1024)
    t_bow = (time.perf_counter() - t0) / 2
    ones = torch.ones((b, nf), dtype=torch.uint8, device=cuda)
    tsb = []
    for _ in range(4):
        torch.cuda.synchronize(dev)
        t0 = time.perf_counter()
        nm, m12, _ = mm.SearchByBoW(F1, F2, ones, ones)
        tsb.append(time.perf_counter() - t0)
    tb = float(np.median(tsb[1:]))
    fv1, fv2 = F1.feature_vector(0), F2.feature_vector(0)
    hk = kp[0].view(orbref.KP_DTYPE).reshape(nf)
    t0 = time.perf_counter()
    rn, r12 = orbref.search_by_bow_kf_kf(hk, d1[0], np.ones(nf, np.uint8), fv1, hk, d2[0], np.ones(nf, np.uint8), fv2,
                                         mm.mfNNratio, True)
    ct = time.perf_counter() - t0
    sbow = {"pairs": b, "features": nf, "ms": round(tb * 1e3, 3), "pairs_per_sec": round(b / tb, 1),
            "compute_bow_ms_per_%d_frames" % b: round(t_bow * 1e3, 3), "mean_matches_per_pair": round(float(nm.float().mean().item()), 1),
            "cpu_oracle_1_thread_pairs_per_sec": round(1 / ct, 1),
            "parity_pair0": bool(rn == int(nm[0].item()) and np.array_equal(r12, m12[0].cpu().numpy()))}
    # SearchForTriangulation on the same 256 pairs: every feature free, keypoints at the origin and F12 chosen so
    # that every candidate lies on its epipolar line (the sort-and-walk of the reference then depends on the
    # distances alone), next to the oracle on one thread for the first pair
    F12 = np.tile(np.array([0, 0, 0, 0, 0, -1, 0, 1, 0], np.float32), (b, 1))
    d_F12 = torch.from_numpy(F12).to(cuda)
    sig = torch.from_numpy((1.2 ** (2 * np.arange(8))).astype(np.float32)).to(cuda)
    tst = []
    for _ in range(4):
        torch.cuda.synchronize(dev)
        t0 = time.perf_counter()
        nmt, t12, _ = mm.SearchForTriangulation(F1, F2, ones, ones, d_F12, sig)
        mm.synchronize()
        tst.append(time.perf_counter() - t0)
    tt = float(np.median(tst[1:]))
    t0 = time.perf_counter()
    rnt, rt12 = orbref.search_for_triangulation(hk, d1[0], np.ones(nf, np.uint8), fv1, hk, d2[0], np.ones(nf, np.uint8), fv2,
                                                F12[0].reshape(3, 3), sig.cpu().numpy(), True)
    ctt = time.perf_counter() - t0
    stri = {"pairs": b, "features": nf, "ms": round(tt * 1e3, 3), "pairs_per_sec": round(b / tt, 1),
            "mean_matches_per_pair": round(float(nmt.float().mean().item()), 1),
            "cpu_oracle_1_thread_pairs_per_sec": round(1 / ctt, 1),
            "parity_pair0": bool(rnt == int(nmt[0].item()) and np.array_equal(rt12, t12[0].cpu().numpy()))}
    return {"search_by_bow": sbow, "search_for_triangulation": stri,
            "descriptors": n, "tree": "k=%d L=%d (%d nodes, synthetic)" % (k, L, len(parent)), "ms": round(best * 1e3, 3),
            "descriptors_per_sec": float("%.4g" % (n / best)),
            "cpu_oracle_1_thread_descriptors_per_sec": float("%.4g" % (2000 / cdt)), "parity_first_2000": bool(ok)}


def windowed_leg(orb, torch, dev, orbref):
    """Frame grid + SearchForInitialization(window 100) + WindowSearch(window 100) over 256 frame pairs
    of ~2000 keypoints (extractor output of 1280x720 frames; F2 = F1 jittered by up to 8 px with ~8
    descriptor bits flipped), next to the oracle on one thread for the first 4 pairs."""
    b, w, h, nf = 256, 1280, 720, 2000
    wl = ExtractWorkload(orb, torch, dev, w, h, nf, b, first_idx=200000)
    wl.step()
    wl.sync()
    g = torch.Generator(device="cuda:%d" % dev)
    g.manual_seed(7)
    cuda = "cuda:%d" % dev
    kp1 = wl.d_kp.view(torch.uint8).view(b, nf, 28).clone()
    kp2 = kp1.clone()
    xy = kp2.view(torch.float32).view(b, nf, 7)
    xy[:, :, 0:2] += torch.rand((b, nf, 2), device=cuda, generator=g) * 16 - 8
    d1 = wl.d_desc.view(b, nf, 32).clone()
    flips = torch.randint(0, 256, (b, nf, 32), dtype=torch.uint8, device=cuda, generator=g)
    for _ in range(4):  # each bit survives five ANDs with probability 1/32: ~8 of 256 bits flipped
        flips &= torch.randint(0, 256, (b, nf, 32), dtype=torch.uint8, device=cuda, generator=g)
    d2 = d1 ^ flips
    n = wl.d_n.clone()
    bounds = [0, 0, w, h]
    m_init, m_win = orb.ORBmatcher(0.9, True, device=dev), orb.ORBmatcher(0.6, True, device=dev)
    t0 = time.perf_counter()
    F1 = orb.FrameBatch(m_init, kp1, d1, n, bounds)
    F2 = orb.FrameBatch(m_init, kp2, d2, n, bounds)
    t_grid = (time.perf_counter() - t0) / 2
    prev0 = kp1.view(torch.float32).view(b, nf, 7)[:, :, 0:2].contiguous()
    valid = torch.ones((b, nf), dtype=torch.uint8, device=cuda)
    out = {}
    for name, fn in (("search_for_initialization_w100", lambda: m_init.SearchForInitialization(F1, F2, prev0.clone(), 100)),
                     ("window_search_w100", lambda: m_win.WindowSearch(F1, F2, 100, valid, 0))):
        fn()
        ts = []
        for _ in range(3):
            torch.cuda.synchronize(dev)
            t0 = time.perf_counter()
            nm, mm = fn()
            torch.cuda.synchronize(dev)
            ts.append(time.perf_counter() - t0)
        out[name] = {"pairs_per_sec": round(b / float(np.median(ts)), 1), "ms_per_%d_pairs" % b: round(float(np.median(ts)) * 1e3, 3),
                     "mean_matches_per_pair": round(float(nm.float().mean().item()), 1)}
    # oracle, one thread, first 4 pairs (+ parity of those)
    hk1, hk2 = kp1[:4].cpu().numpy().view(orbref.KP_DTYPE).reshape(4, nf), kp2[:4].cpu().numpy().view(orbref.KP_DTYPE).reshape(4, nf)
    hd1, hd2, hn = d1[:4].cpu().numpy(), d2[:4].cpu().numpy(), n[:4].cpu().numpy()
    nm_i, m12 = m_init.SearchForInitialization(F1, F2, prev0.clone(), 100)
    nm_w, m21 = m_win.WindowSearch(F1, F2, 100, valid, 0)
    m12, m21 = m12[:4].cpu().numpy(), m21[:4].cpu().numpy()
    ok = True
    t_i = t_w = 0.0
    for f in range(4):
        k = int(hn[f])
        pv = np.stack([hk1[f, :k]["x"], hk1[f, :k]["y"]], 1)
        t0 = time.perf_counter()
        rn, r12, _ = orbref.search_for_initialization(hk1[f, :k], hd1[f, :k], hk2[f, :k], hd2[f, :k], bounds, pv, 100,
                                                      m_init.mfNNratio, True)
        t_i += time.perf_counter() - t0
        t0 = time.perf_counter()
        rw, r21 = orbref.window_search(hk1[f, :k], hd1[f, :k], np.ones(k, np.uint8), hk2[f, :k], hd2[f, :k], bounds, 100, 0,
                                       2**31 - 1, m_win.mfNNratio, True)
        t_w += time.perf_counter() - t0
        ok = ok and np.array_equal(r12, m12[f, :k]) and np.array_equal(r21, m21[f, :k])
    # SearchByProjection(CurrentFrame, LastFrame, th = 15): one probe per keypoint of F1 at its own position
    # (identity motion), radius 15 * scale[octave], levels octave-1..octave+1, best distance <= TH_HIGH, rotation check
    sf = np.ones(8, np.float32)
    for l in range(1, 8):
        sf[l] = sf[l - 1] * np.float32(1.2)
    k1f = kp1.view(torch.float32).view(b, nf, 7)
    octv = kp1.view(torch.int32).view(b, nf, 7)[:, :, 5]
    pr = torch.zeros((b, nf, 8), dtype=torch.float32, device=cuda)
    pri = pr.view(torch.int32)
    pr[:, :, 0:2] = k1f[:, :, 0:2]
    pr[:, :, 2] = torch.from_numpy(sf * np.float32(15.0)).to(cuda)[octv.long()]
    pri[:, :, 3] = octv - 1
    pri[:, :, 4] = octv + 1
    pr[:, :, 5] = k1f[:, :, 3]
    pri[:, :, 6] = 1
    probes = pr.view(torch.uint8).view(b, nf, 32)
    m_proj = orb.ORBmatcher(0.9, True, device=dev)
    fn = lambda: m_proj.SearchByProjection(F2, probes, d1, n, orb.ORBmatcher.PROJ_BEST, th_dist=100)  # noqa: E731
    fn()
    ts = []
    for _ in range(3):
        torch.cuda.synchronize(dev)
        t0 = time.perf_counter()
        nm_p, mo = fn()
        torch.cuda.synchronize(dev)
        ts.append(time.perf_counter() - t0)
    out["search_by_projection_last_frame_th15"] = {
        "frames_per_sec": round(b / float(np.median(ts)), 1), "ms_per_%d_frames" % b: round(float(np.median(ts)) * 1e3, 3),
        "mean_matches_per_frame": round(float(nm_p.float().mean().item()), 1)}
    mo = mo[:4].cpu().numpy()
    t_p = 0.0
    ident = np.eye(4, dtype=np.float32)
    for f in range(4):
        k = int(hn[f])
        pos = np.stack([hk1[f, :k]["x"], hk1[f, :k]["y"], np.ones(k, np.float32)], 1)
        t0 = time.perf_counter()
        rn, rmo, _ = orbref.sbp_last_frame(hk2[f, :k], hd2[f, :k], bounds, None, (1.0, 1.0, 0.0, 0.0), ident, sf, hk1[f, :k],
                                           hd1[f, :k], np.zeros(k, np.uint8), pos, 15.0, True)
        t_p += time.perf_counter() - t0
        ok = ok and np.array_equal(rmo, mo[f, :k]) and rn == int(nm_p[f].item())
    # extraction -> grid -> SearchByProjection of the whole batch: every call blocking (the default) against the
    # same calls queued behind a candidate reservation with one synchronisation at the end
    mq = orb.ORBmatcher(0.9, True, device=dev)
    mq.reserve_candidates(64 << 20)

    # the probes of the chain are built on the device from a pose and map points (identity pose, unit intrinsics,
    # points at depth 1 over F1's keypoints: the same windows as `probes` above)
    cams = mq.projections([ident] * b, (1.0, 1.0, 0.0, 0.0))
    pos = torch.cat([k1f[:, :, 0:2], torch.ones((b, nf, 1), dtype=torch.float32, device=cuda)], 2).contiguous()
    d_sf = torch.from_numpy(sf).to(cuda)

    def chain(mm):
        Fc = orb.FrameBatch.from_images(wl.ext, mm, wl.d_img)
        pr_c = mm.BuildProbes(orb.ORBmatcher.PROBES_LAST_FRAME, cams, bounds, d_sf, pos, n, 15.0, src_kp=kp1)
        return mm.SearchByProjection(Fc, pr_c, d1, n, orb.ORBmatcher.PROJ_BEST, th_dist=100)

    res_chain = {}
    for name, mm in (("blocking", m_proj), ("queued", mq)):
        chain(mm)
        mm.synchronize()
        ts = []
        for _ in range(3):
            torch.cuda.synchronize(dev)
            t0 = time.perf_counter()
            nmc, moc = chain(mm)
            mm.synchronize()
            wl.ext.synchronize()
            torch.cuda.synchronize(dev)
            ts.append(time.perf_counter() - t0)
        res_chain[name] = (float(np.median(ts)), nmc, moc)
    out["extract_grid_project_chain"] = {
        "frames": b, "blocking_ms": round(res_chain["blocking"][0] * 1e3, 3), "queued_ms": round(res_chain["queued"][0] * 1e3, 3),
        "queued_frames_per_sec": round(b / res_chain["queued"][0], 1),
        "same_result": bool(torch.equal(res_chain["blocking"][2], res_chain["queued"][2])),
        "what": "images -> extraction -> grid -> probes from pose and map points -> SearchByProjection, all queued on the device",
        "device_built_probes_equal_host_built": bool(torch.equal(
            mq.BuildProbes(orb.ORBmatcher.PROBES_LAST_FRAME, cams, bounds, d_sf, pos, n, 15.0, src_kp=kp1)[:, :, :28]
            * (torch.arange(nf, device=cuda)[None, :, None] < n[:, None, None]),
            probes[:, :, :28] * (torch.arange(nf, device=cuda)[None, :, None] < n[:, None, None])))}
    # Fuse / SearchBySim3: the independent best-in-radius search with the same probes (radius 15 * scale, levels
    # octave-1..octave+1), next to the oracle's Fuse on one thread for the first 4 frames (identity pose, points at
    # depth 1 projecting onto F1's keypoints); parity on those frames with the probes the oracle built
    fn = lambda: m_proj.BestInRadius(F2, probes, d1, n)  # noqa: E731
    fn()
    ts = []
    for _ in range(3):
        torch.cuda.synchronize(dev)
        t0 = time.perf_counter()
        fn()
        m_proj.synchronize()
        ts.append(time.perf_counter() - t0)
    t_f, ok_f = 0.0, True
    fprobes = np.zeros((4, nf), orb.PROBE_DTYPE)
    refs = []
    for f in range(4):
        k = int(hn[f])
        pos = np.stack([hk1[f, :k]["x"], hk1[f, :k]["y"], np.ones(k, np.float32)], 1)
        dist = np.sqrt((pos.astype(np.float64) ** 2).sum(1))
        min_d = (dist / sf[hk1[f, :k]["octave"]]).astype(np.float32)
        nrm = (pos / dist[:, None]).astype(np.float32)
        t0 = time.perf_counter()
        r = orbref.fuse(hk2[f, :k], hd2[f, :k], bounds, (1.0, 1.0, 0.0, 0.0), sf, ident, np.zeros(k, np.uint8), pos, nrm, min_d,
                        min_d * np.float32(4.0), hd1[f, :k], 15.0)
        t_f += time.perf_counter() - t0
        fprobes[f, :k] = r[3]
        refs.append(r)
    fprobes[:, :]["reserved"] = 0
    F2s = orb.FrameBatch(m_proj, kp2[:4].contiguous(), d2[:4].contiguous(), n[:4].contiguous(), bounds)
    bi, bd = m_proj.BestInRadius(F2s, torch.from_numpy(fprobes.view(np.uint8).reshape(4, nf, 32).copy()).to(cuda), d1[:4].contiguous(),
                                 n[:4].contiguous())
    m_proj.synchronize()
    bi, bd = bi.cpu().numpy(), bd.cpu().numpy()
    for f in range(4):
        k = int(hn[f])
        ok_f = ok_f and np.array_equal(bi[f, :k], refs[f][1]) and np.array_equal(bd[f, :k], refs[f][2])
    out["best_in_radius_th15"] = {
        "what": "the per-map-point search of ORBmatcher::Fuse / SearchBySim3, one probe per keypoint of the other frame",
        "frames_per_sec": round(b / float(np.median(ts)), 1), "ms_per_%d_frames" % b: round(float(np.median(ts)) * 1e3, 3),
        "cpu_oracle_1_thread_frames_per_sec": round(4 / t_f, 1), "parity_first_4_frames": bool(ok_f),
        "mean_fused_per_frame_first_4": round(float(np.mean([r[0] for r in refs])), 1)}
    out["frame_grid_ms_per_%d_frames" % b] = round(t_grid * 1e3, 3)
    out["cpu_oracle_1_thread_pairs_per_sec"] = {"search_for_initialization_w100": round(4 / t_i, 1),
                                                "window_search_w100": round(4 / t_w, 1),
                                                "search_by_projection_last_frame_th15": round(4 / t_p, 1)}
    out["parity_first_4_pairs"] = bool(ok)
    return out



if __name__ == "__main__":
    main()
