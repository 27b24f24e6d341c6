"""Multi-process partitioning logic on CPU (gloo, world_size 2): contiguous shard ranges and the
all-gather + ordered merge of frame-vs-map matching.  The per-shard scan and the merge are done
here by the oracle (no GPU in this test); what is covered is the sharding arithmetic, the
collective's layout and that merging index-contiguous shards in rank order equals the serial scan."""
import os
import socket
import sys

import numpy as np
import pytest

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(HERE)


def test_shard_range_covers_exactly(orb):
    sr = orb.sharded.shard_range
    for n in (0, 1, 7, 8, 2048, 10_000_000):
        for world in (1, 2, 3, 8):
            spans = [sr(n, r, world) for r in range(world)]
            assert spans[0][0] == 0 and spans[-1][1] == n
            for a, b in zip(spans, spans[1:]):
                assert a[1] == b[0]
            sizes = [e - b for b, e in spans]
            assert max(sizes) - min(sizes) <= 1


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


def _worker(rank, world, port, q_out):
    import importlib
    sys.path.insert(0, ROOT)
    sys.path.insert(0, os.path.join(ROOT, "oracle"))
    import torch
    import torch.distributed as dist
    import orbref
    orb = importlib.import_module("orb-slam2_amd")
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    nq, nt = 300, 5001
    q, t = orb.synth.synth_descriptors(nq, nt, seed=9)
    t[4000:4010] = t[17]
    q[5] = t[17]

    class OracleMatcher:  # stands in for ORBmatcher on CPU tensors
        def knn2_device(self, dq, nq_, qs, dt, nt_, ts, npairs, idx_base, pi, p1, p2):
            idx, d1, d2 = orbref.knn2(self.q, self.t)
            self.parts[0] = torch.from_numpy(np.where(idx >= 0, idx + idx_base, idx))
            self.parts[1] = torch.from_numpy(d1)
            self.parts[2] = torch.from_numpy(d2)

        def merge_device(self, dparts, nshards, nq_, oi, o1, o2):
            p = self.all.numpy()
            best = np.full(nq_, 2147483647, np.int64)
            best2 = best.copy()
            idx = np.full(nq_, -1, np.int64)
            for s in range(nshards):
                i, d1, d2 = p[s]
                ok = i >= 0
                lt = ok & (d1 < best)
                mid = ok & ~lt & (d1 < best2)
                best2 = np.where(lt, best, np.where(mid, d1, best2))
                idx = np.where(lt, i, idx)
                best = np.where(lt, d1, best)
                best2 = np.where(ok & (d2 < best2), d2, best2)
            self.out[0] = torch.from_numpy(idx.astype(np.int32))
            self.out[1] = torch.from_numpy(best.astype(np.int32))
            self.out[2] = torch.from_numpy(best2.astype(np.int32))

        def synchronize(self):
            pass

    b, e = orb.sharded.shard_range(nt, rank, world)
    m = OracleMatcher()
    m.q, m.t = q, t[b:e]
    m.parts = torch.zeros((3, nq), dtype=torch.int32)
    m.all = torch.zeros((world, 3, nq), dtype=torch.int32)
    m.out = torch.zeros((3, nq), dtype=torch.int32)
    orb.sharded.knn2_sharded(m, torch.from_numpy(q), nq, torch.from_numpy(t[b:e].copy()), e - b, b, m.parts, m.all, m.out,
                             dist=dist)
    ridx, rd1, rd2 = orbref.knn2(q, t)
    ok = bool(np.array_equal(m.out[0].numpy(), ridx) and np.array_equal(m.out[1].numpy(), rd1)
              and np.array_equal(m.out[2].numpy(), rd2) and ridx[5] == 17)
    q_out.put((rank, ok))
    dist.destroy_process_group()


def test_sharded_knn_merge_two_ranks_gloo():
    import torch.multiprocessing as mp
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_worker, args=(r, 2, port, q)) for r in range(2)]
    for p in procs:
        p.start()
    res = [q.get(timeout=180) for _ in procs]
    for p in procs:
        p.join(60)
    assert sorted(res) == [(0, True), (1, True)]
