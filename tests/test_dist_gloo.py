"""Multi-process partitioning logic on CPU (gloo, world_size 2 and 3): contiguous shard ranges and the
all-gather + ordered merge of frame-vs-map matching.  There is no GPU in this test, so the per-shard scan is
done by the oracle; the records then travel through a real gloo all-gather in the layout the C entry point
uses (orbhip_knn2_gather_layout) and are merged by the library's own merge function — the host twin
orbhip_knn2_merge_host, compiled from the same source as the device kernel k_knn2_merge.  Covered: the sharding
arithmetic, the collective's layout, and that merging index-contiguous shards in rank order equals the serial
scan, including equal rows on both sides of a shard boundary."""
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
            # the library's merge (host twin of k_knn2_merge, same source) on the gathered records
            merged = orb.sharded.merge_host(self.all.numpy())
            self.out.copy_(torch.from_numpy(merged))

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
    # the gathered buffer holds rank r's record at the offset the C entry point uses for its in-place all-gather
    rec, off, total = orb.sharded.gather_layout(nq, world, rank)
    flat = m.all.numpy().reshape(-1)
    ok = ok and rec == 3 * nq and total == 4 * rec * world and np.array_equal(flat[off:off + rec], m.parts.numpy().reshape(-1))
    q_out.put((rank, ok))
    dist.destroy_process_group()


@pytest.mark.parametrize("world", [2, 3])
def test_sharded_knn_merge_gloo(world):
    import torch.multiprocessing as mp
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_worker, args=(r, world, port, q)) for r in range(world)]
    for p in procs:
        p.start()
    res = [q.get(timeout=180) for _ in procs]
    for p in procs:
        p.join(60)
    assert sorted(res) == [(r, True) for r in range(world)]


def test_merge_host_rule(orb):
    """Known answers of the merge by hand: the earlier shard wins a tie on the best, a shard's second-best and a
    later shard's best both compete for the second place, empty shards (idx -1) are skipped."""
    big = 2147483647
    parts = np.array([
        # shard 0: idx, d1, d2          queries: 0      1      2      3     4
        [[3, 3, -1, 7, 2], [10, 10, big, 30, 5], [20, 12, big, 31, 5]],
        [[103, 104, 100, 105, -1], [10, 9, 40, 29, big], [11, 50, 41, 29, big]],
    ], dtype=np.int32)
    out = orb.sharded.merge_host(parts)
    assert out[0].tolist() == [3, 104, 100, 105, 2]
    assert out[1].tolist() == [10, 9, 40, 29, 5]
    assert out[2].tolist() == [10, 10, 41, 29, 5]
    for world in (1, 2, 8):
        for rank in range(world):
            assert orb.sharded.gather_layout(4000, world, rank) == (12000, 12000 * rank, 48000 * world)
