"""Multi-process path on CPU (gloo, world_size 2): the index built by rank 0 and broadcast as flat arrays is identical on
every rank, the read shards partition the batch, and the bench's max-over-ranks / sum-over-ranks reductions work.
(On the GPU box the same code runs with backend "nccl" = RCCL; mapping itself needs no collective.)"""
import os
import socket
import sys
import tempfile
import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


def _worker(rank, world, port, tmp, q):
    sys.path.insert(0, ROOT)
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    import torch
    import torch.distributed as dist
    from winnowmap_amd import gpu, synth
    from winnowmap_amd import dist as wmdist
    dist.init_process_group("gloo", rank=rank, world_size=world)
    dev = torch.device("cpu")
    idx = None
    if rank == 0:
        ref = synth.make_reference(2, 150000, 11, repeat_frac=0.1)
        fa = os.path.join(tmp, "ref.fa")
        synth.write_fasta(fa, ref)
        km, cnt = synth.repetitive_kmers(ref, 15)
        kf = os.path.join(tmp, "rep.txt")
        synth.write_kmer_list(kf, km, cnt, 15)
        idx = gpu.Index(fa, kf, k=15, w=50, n_threads=2)
    idx = wmdist.broadcast_index(idx, rank, dist, dev)
    sizes, arrs = idx.export_arrays()
    digest = [int(x) for x in sizes] + [int(np.frombuffer(a.tobytes(), np.uint8).astype(np.uint64).sum()) for a in arrs] + [idx.n_minimizers]
    mine = wmdist.shard(37, rank, world)
    t = wmdist.max_over_ranks(1.0 + rank, dist, dev)
    tot = wmdist.sum_over_ranks([len(mine), 100.0 * (rank + 1)], dist, dev)
    q.put((rank, digest, mine, t, tot, idx.names()))
    dist.barrier()
    dist.destroy_process_group()


def test_index_broadcast_and_read_sharding_world2():
    import torch.multiprocessing as mp
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    tmp = tempfile.mkdtemp()
    port = _free_port()
    ps = [ctx.Process(target=_worker, args=(r, 2, port, tmp, q)) for r in range(2)]
    for p in ps:
        p.start()
    res = sorted([q.get(timeout=300) for _ in ps])
    for p in ps:
        p.join(60)
        assert p.exitcode == 0
    (r0, d0, m0, t0, s0, n0), (r1, d1, m1, t1, s1, n1) = res
    assert d0 == d1 and n0 == n1 and d0[-1] > 1000          # identical index content on both ranks
    assert sorted(m0 + m1) == list(range(37)) and not set(m0) & set(m1)
    assert t0 == t1 == 2.0                                   # max over ranks
    assert s0 == s1 == [37.0, 300.0]                         # sum over ranks
