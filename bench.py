#!/usr/bin/env python3
"""Benchmark of the ORB front end on MI355X.

  python bench.py --gpus N --steps K --warmup W

One "step" is one pass of the extractor (pyramid, cell-grid FAST, selection, orientation, blur,
rotated-BRIEF descriptors, keypoint records) over one batch of synthetic frames that is already
resident in HBM.  The workload is the 1-GPU configuration of the headline metric: a batch of 256
synthetic 1280x720 frames, 2000 features per frame.  With N > 1 (one process per GPU, launched by
torch.distributed.run) every rank processes its own batch of 256 frames: frames are independent,
so the path is sharded by frame batch with no data-path collective (weak scaling); the only
collectives are the timing barrier and a MAX over ranks of the elapsed time.

Rank 0 prints ONE JSON line.  Besides the contract fields it carries
  roofline      the dominant kernel group against HBM bandwidth (durations from HIP events on the
                library's stream, collected in a separate profiled pass after the timed region)
  cpu_baseline  the CPU oracle (oracle/, a port of the reference's CPU path) timed on the host
                cores of this box on a bounded sample of the same frames
  extra         the 640x480 / 1000-feature extraction rate and the Hamming kNN rate of the
                same library, measured after the main timed region.
"""
import argparse
import importlib
import json
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

HBM_PEAK_GBS = 8000.0  # MI355X HBM3E, spec


def pyramid_pixels(ext, nlevels):
    return sum(ext.level_geometry(l)["w"] * ext.level_geometry(l)["h"] for l in range(nlevels))


def make_frames(synth, w, h, n, first):
    return np.stack([synth.synth_image(w, h, first + i) for i in range(n)])


class ExtractWorkload:
    """Batch of frames resident in HBM + preallocated device outputs."""

    def __init__(self, orb, torch, dev, w, h, nfeatures, batch, first_idx, frames=None):
        self.torch = torch
        self.w, self.h, self.nf, self.batch = w, h, nfeatures, batch
        self.host = frames if frames is not None else make_frames(orb.synth, w, h, batch, first_idx)
        self.ext = orb.ORBextractor(nfeatures, 1.2, 8, 1, 20, device=dev, max_width=w, max_height=h, max_batch=batch)
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


def cpu_baseline(frames, nfeatures, threads, sample):
    """Oracle (CPU port of the reference path) on `threads` host threads over `sample` frames."""
    sys.path.insert(0, os.path.join(ROOT, "oracle"))
    import orbref
    from concurrent.futures import ThreadPoolExecutor
    orbref.lib()
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



def _pmc_traffic(kernel, args):
    """HBM bytes per launch of `kernel` from the committed PMC summary (profiles/*pmc_traffic.json,
    made by tools/pmc_summary.py from separate rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE passes of this
    same workload).  Counters cannot be collected from inside the timed run, so the figure is the one
    of the latest committed pass; None if there is none or the workload is not the default one."""
    if (args.batch, args.width, args.height, args.features) != (256, 1280, 720, 2000):
        return None
    import glob
    files = sorted(glob.glob(os.path.join(os.path.dirname(os.path.abspath(__file__)), "profiles", "*pmc_traffic.json")))
    if not files:
        return None
    try:
        k = json.load(open(files[-1]))["kernels"].get(kernel)
        return k["traffic"] if k else None
    except (OSError, ValueError, KeyError):
        return None


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=20)
    ap.add_argument("--warmup", type=int, default=3)
    ap.add_argument("--batch", type=int, default=256, help="frames per GPU per step")
    ap.add_argument("--width", type=int, default=1280)
    ap.add_argument("--height", type=int, default=720)
    ap.add_argument("--features", type=int, default=2000)
    ap.add_argument("--cpu-sample", type=int, default=96, help="frames of the CPU baseline sample")
    ap.add_argument("--no-extra", action="store_true", help="skip the 640x480 and Hamming kNN legs")
    args = ap.parse_args()

    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    if world != args.gpus:
        if args.gpus != 1 and world == 1:
            raise SystemExit("--gpus %d needs one process per GPU: launch with python -m torch.distributed.run "
                             "--nproc-per-node %d bench.py --gpus %d ..." % (args.gpus, args.gpus, args.gpus))
        args.gpus = world

    orb = importlib.import_module("orb-slam2_amd")
    orb._lib.lib()  # fails loudly if liborbhip.so is missing
    import torch
    import torch.distributed as dist

    if not torch.cuda.is_available():
        raise SystemExit("bench.py needs a GPU")
    dev = local_rank % torch.cuda.device_count()
    torch.cuda.set_device(dev)
    if world > 1:
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        dist.init_process_group("nccl", rank=rank, world_size=world, device_id=torch.device("cuda", dev))

    def barrier():
        if world > 1:
            dist.barrier()

    wl = ExtractWorkload(orb, torch, dev, args.width, args.height, args.features, args.batch, first_idx=rank * args.batch)

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
    elapsed = time.perf_counter() - t0
    if world > 1:
        t = torch.tensor([elapsed], dtype=torch.float64, device="cuda:%d" % dev)
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        elapsed = float(t.item())
    frames_total = args.batch * args.steps * world
    value = frames_total / elapsed

    out = None
    if rank == 0:
        # ---- roofline of the dominant kernel group: separate profiled pass ----------------------
        nl = wl.ext.GetLevels()
        P = pyramid_pixels(wl.ext, nl)
        wl.ext.set_profiling(True)
        prof_steps = max(3, min(args.steps, 10))
        acc = {}
        for _ in range(prof_steps):
            wl.step()
            wl.sync()
            for s, (ms, n) in wl.ext.stage_times().items():
                a = acc.setdefault(s, [0.0, 0])
                a[0] += ms
                a[1] += n
        wl.ext.set_profiling(False)
        n_kp = int(wl.d_n.sum().item())
        # algorithmic bytes per frame of each kernel group: the 5 B/pixel + 60 B/keypoint of the
        # whole path split by who must move them (pyramid: read + write; FAST: read; blur: read +
        # write; outputs: 32 B descriptor + 28 B keypoint).
        N_frame = n_kp / args.batch
        lvl0 = args.width * args.height
        # FAST reads every pyramid pixel once: in the fused per-cell kernel ("nms_compact" group, the
        # "fast_measure" group then has no launches) or in the separate measure kernel.
        fused = acc["fast_measure"][1] == 0
        alg = {"level0": 2 * lvl0, "resize": 2 * (P - lvl0), "fast_measure": 0 if fused else P,
               "nms_compact": P if fused else 0, "select": 0, "orient": 0, "blur": 2 * P, "describe": 60 * N_frame}
        stage_ms = {s: a[0] / prof_steps for s, a in acc.items()}
        dominant = max(stage_ms, key=stage_ms.get)
        launches = acc[dominant][1] // prof_steps
        avg_launch_ms = stage_ms[dominant] / max(launches, 1)
        bytes_per_launch = alg[dominant] * args.batch / max(launches, 1)
        achieved = bytes_per_launch / (avg_launch_ms * 1e-3) / 1e9 if avg_launch_ms > 0 else 0.0
        path_bytes = (5 * P + 60 * N_frame) * args.batch
        path_ms = sum(stage_ms.values())
        kernel_names = {"level0": "k_level0", "resize": "k_resize", "fast_measure": "k_score",
                        "nms_compact": "k_fast_cell" if fused else "k_nms", "select": "k_cell_select", "orient": "k_orient_describe",
                        "blur": "k_blur", "describe": "k_orient_describe"}
        roofline = {"bound": "hbm", "kernel": kernel_names[dominant], "stage": dominant, "achieved": round(achieved, 2), "peak": HBM_PEAK_GBS,
                    "unit": "GB/s", "frac": round(achieved / HBM_PEAK_GBS, 5),
                    "traffic": _pmc_traffic(kernel_names[dominant], args),
                    "launches_per_step": launches, "avg_launch_ms": round(avg_launch_ms, 5),
                    "algorithmic_bytes_per_launch": int(bytes_per_launch),
                    "stage_ms_per_step": {s: round(v, 4) for s, v in stage_ms.items()},
                    "whole_path": {"algorithmic_bytes_per_step": int(path_bytes), "kernel_ms_per_step": round(path_ms, 4),
                                   "achieved": round(path_bytes / (path_ms * 1e-3) / 1e9, 2) if path_ms > 0 else 0.0,
                                   "frac": round(path_bytes / (path_ms * 1e-3) / 1e9 / HBM_PEAK_GBS, 5) if path_ms > 0 else 0.0}}

        # ---- CPU baseline on a bounded sample of the same frames + parity of that sample --------
        threads = max(1, min(16, os.cpu_count() or 1))
        if world > 1:  # the CPU baseline and the extra legs belong to the N = 1 run; the other ranks are waiting
            args.cpu_sample, args.no_extra = 0, True
        cpu_fps, ref_results = cpu_baseline(wl.host, args.features, threads, args.cpu_sample) if args.cpu_sample > 0 else (0.0, [])
        kp = wl.d_kp.cpu().numpy().view(np.uint32)
        desc = wl.d_desc.cpu().numpy()
        ncount = wl.d_n.cpu().numpy()
        mism = 0
        for i, (rkp, rdesc) in enumerate(ref_results):
            n = len(rkp)
            same = ncount[i] == n and np.array_equal(kp[i, :n].reshape(-1), rkp.view(np.uint32).reshape(-1)) \
                and np.array_equal(desc[i, :n], rdesc)
            mism += 0 if same else 1
        cpu = None if not ref_results else {"value": round(cpu_fps, 3), "unit": "frames/s", "cores": threads, "kind": "port",
               "sample": "%d of the %d frames of this workload (%dx%d, %d features), one oracle extractor per thread"
                         % (len(ref_results), args.batch, args.width, args.height, args.features),
               "parity_frames_checked": len(ref_results), "parity_frames_mismatching": mism}

        extra = {}
        if not args.no_extra:
            extra = extra_legs(orb, torch, dev, threads)

        out = {"metric": "orb_extract_frames_per_sec", "value": round(value, 2), "unit": "frames/s", "n_gpus": world,
               "steps": args.steps, "warmup": args.warmup, "ms_per_step": round(elapsed / args.steps * 1e3, 4),
               "higher_is_better": True, "scaling": "weak", "vs_baseline": None, "dtype": "u8", "data": "synthetic",
               "config": {"workload": "batch of %d synthetic %dx%d frames per GPU, %d features/frame, 8 levels, scale 1.2, "
                                      "full extract+describe" % (args.batch, args.width, args.height, args.features),
                          "frames_per_gpu_per_step": args.batch, "keypoints_per_frame": round(N_frame, 1),
                          "parallelism": "frame-batch sharding x%d, no data-path collective" % world},
               "roofline": roofline, "cpu_baseline": cpu, "extra": extra}
    barrier()
    if rank == 0:
        print(json.dumps(out), flush=True)
    if world > 1:
        dist.destroy_process_group()


def extra_legs(orb, torch, dev, threads):
    """Not the headline value: the 640x480 / 1000-feature and 1920x1080 / 4000-feature extraction rates (the
    reference's own CPU-runnable case and the per-GPU shard of the 2048-frame configuration), the Hamming kNN
    rates (4096 frame pairs; one map shard), the windowed / projection matchers and the BoW legs."""
    res = {}
    # extraction, 640x480, 1000 features, batch 256
    b = 256
    wl = ExtractWorkload(orb, torch, dev, 640, 480, 1000, b, first_idx=100000)
    for _ in range(2):
        wl.step()
    wl.sync()
    t0 = time.perf_counter()
    k = 10
    for _ in range(k):
        wl.step()
    wl.sync()
    dt = time.perf_counter() - t0
    fps = b * k / dt
    cpu1, _ = cpu_baseline(wl.host, 1000, 1, 24)
    cpuN, _ = cpu_baseline(wl.host, 1000, threads, 16 * threads)
    res["extract_640x480_1000"] = {"frames_per_sec": round(fps, 1), "batch": b,
                                   "cpu_oracle_1_thread_frames_per_sec": round(cpu1, 2),
                                   "cpu_oracle_%d_threads_frames_per_sec" % threads: round(cpuN, 2),
                                   "speedup_vs_1_thread": round(fps / cpu1, 1)}
    del wl
    # extraction, 1920x1080, 4000 features, batch 256: the shard one GPU of eight owns in the 2048-frame
    # configuration (32 distinct frames, repeated: host generation of 256 such frames would take longer than the run)
    b3 = 256
    f3 = np.concatenate([make_frames(orb.synth, 1920, 1080, 32, 300000)] * (b3 // 32))
    wl = ExtractWorkload(orb, torch, dev, 1920, 1080, 4000, b3, first_idx=0, frames=f3)
    for _ in range(2):
        wl.step()
    wl.sync()
    t0 = time.perf_counter()
    for _ in range(5):
        wl.step()
    wl.sync()
    dt3 = time.perf_counter() - t0
    nl3 = wl.ext.GetLevels()
    bytes3 = 5 * pyramid_pixels(wl.ext, nl3) + 60 * float(wl.d_n.float().mean().item())
    res["extract_1920x1080_4000"] = {"frames_per_sec": round(b3 * 5 / dt3, 1), "batch": b3, "ms_per_step": round(dt3 / 5 * 1e3, 3),
                                     "algorithmic_gb_per_sec": round(b3 * 5 / dt3 * bytes3 / 1e9, 1),
                                     "keypoints_per_frame": round(float(wl.d_n.float().mean().item()), 1)}
    del wl, f3
    # Hamming kNN k=2: 4096 pairs of 2000 x 2000 descriptors (1.6384e10 comparisons)
    m = orb.ORBmatcher(0.6, True, device=dev)
    pairs, nq, nt = 4096, 2000, 2000
    g = torch.Generator(device="cuda:%d" % dev)
    g.manual_seed(1)
    q = torch.randint(0, 256, (pairs, nq, 32), dtype=torch.uint8, device="cuda:%d" % dev, generator=g)
    t = torch.randint(0, 256, (pairs, nt, 32), dtype=torch.uint8, device="cuda:%d" % dev, generator=g)
    o = torch.zeros((3, pairs, nq), dtype=torch.int32, device="cuda:%d" % dev)
    torch.cuda.synchronize(dev)

    def run():
        m.knn2_device(q.data_ptr(), nq, nq * 32, t.data_ptr(), nt, nt * 32, pairs, 0, o[0].data_ptr(), o[1].data_ptr(),
                      o[2].data_ptr())

    run()
    m.synchronize()
    ms = []
    for _ in range(5):
        run()
        m.synchronize()
        ms.append(m.last_elapsed_ms())
    best = float(np.median(ms))
    comps = pairs * nq * nt
    rate = comps / (best * 1e-3)
    # denominators measured live on this device (sustained, every CU busy): v_bcnt_u32_b32 issues at
    # half the rate of v_add_u32 / v_xor_b32 on gfx950, so one comparison costs 8 xor slots + 16 bcnt
    # slots + 3 slots of top-2 bookkeeping = 27 full-rate issue slots
    r_bcnt, r_mix, r_add = m.valu_rate(0), m.valu_rate(1), m.valu_rate(2)
    nominal = 256 * 4 * 32 * 2.4e9  # CUs x SIMDs x lanes/clk x clock
    # CPU: scalar scan of the oracle on one pair, one thread
    sys.path.insert(0, os.path.join(ROOT, "oracle"))
    import orbref
    hq, ht = q[0].cpu().numpy(), t[0].cpu().numpy()
    t0 = time.perf_counter()
    ridx, rd1, rd2 = orbref.knn2(hq, ht)
    cdt = time.perf_counter() - t0
    oo = o[:, 0].cpu().numpy()
    res["hamming_knn2"] = {"pairs": pairs, "nq": nq, "nt": nt, "comparisons_per_sec": float("%.4g" % rate),
                           "ms": round(best, 4),
                           "sustained_lane_ops_per_sec": {"v_bcnt_u32_b32": float("%.4g" % r_bcnt),
                                                          "xor_bcnt_mix": float("%.4g" % r_mix),
                                                          "v_add_u32": float("%.4g" % r_add),
                                                          "nominal_full_rate": float("%.4g" % nominal)},
                           "popcount_frac_of_sustained_bcnt_peak": round(rate * 8 / r_bcnt, 4) if r_bcnt else None,
                           "issue_slot_frac_of_sustained_valu": round(rate * 27 / r_add, 4) if r_add else None,
                           "cpu_oracle_1_thread_comparisons_per_sec": float("%.4g" % (nq * nt / cdt)),
                           "parity_pair0": bool(np.array_equal(oo[0], ridx) and np.array_equal(oo[1], rd1)
                                                and np.array_equal(oo[2], rd2))}
    # frame-vs-map shard: 4000 queries against the 1.25 M map descriptors one GPU of eight would own
    nq2, nt2 = 4000, 1250000
    q2 = torch.randint(0, 256, (nq2, 32), dtype=torch.uint8, device="cuda:%d" % dev, generator=g)
    t2 = torch.randint(0, 256, (nt2, 32), dtype=torch.uint8, device="cuda:%d" % dev, generator=g)
    o2 = torch.zeros((3, nq2), dtype=torch.int32, device="cuda:%d" % dev)
    torch.cuda.synchronize(dev)
    ms2 = []
    for _ in range(4):
        m.knn2_device(q2.data_ptr(), nq2, 0, t2.data_ptr(), nt2, 0, 1, 0, o2[0].data_ptr(), o2[1].data_ptr(), o2[2].data_ptr())
        m.synchronize()
        ms2.append(m.last_elapsed_ms())
    best2 = float(np.median(ms2[1:]))
    # spot check of 8 queries against the scalar scan
    hq2, ht2 = q2[:8].cpu().numpy(), t2.cpu().numpy()
    r_idx, r_d1, r_d2 = orbref.knn2(hq2, ht2)
    oo2 = o2[:, :8].cpu().numpy()
    res["hamming_knn2_map_shard"] = {"nq": nq2, "nt": nt2, "ms": round(best2, 4),
                                     "comparisons_per_sec": float("%.4g" % (nq2 * nt2 / (best2 * 1e-3))),
                                     "parity_first_8_queries": bool(np.array_equal(oo2[0], r_idx) and np.array_equal(oo2[1], r_d1)
                                                                    and np.array_equal(oo2[2], r_d2))}
    del q2, t2
    res["windowed_matchers"] = windowed_leg(orb, torch, dev, orbref)
    res["bow_transform"] = bow_leg(orb, torch, dev, orbref, m)
    return res


def bow_leg(orb, torch, dev, orbref, m):
    """Vocabulary descent of 256 x 2000 descriptors through a synthetic k = 10, L = 5 tree (111 k nodes),
    next to the oracle on one thread for 2000 descriptors."""
    k, L, n = 10, 5, 256 * 2000
    parent, desc, weight = orb.synth.synth_vocabulary(k, L, seed=1, tie_share=0.05)
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
    F1.ComputeBoW(voc, 3, node_cap=1024)
    F2.ComputeBoW(voc, 3, node_cap=1024)
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
    return {"search_by_bow": sbow,
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
    out["frame_grid_ms_per_%d_frames" % b] = round(t_grid * 1e3, 3)
    out["cpu_oracle_1_thread_pairs_per_sec"] = {"search_for_initialization_w100": round(4 / t_i, 1),
                                                "window_search_w100": round(4 / t_w, 1),
                                                "search_by_projection_last_frame_th15": round(4 / t_p, 1)}
    out["parity_first_4_pairs"] = bool(ok)
    return out


if __name__ == "__main__":
    main()
