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
    idx, on_dev = wmdist.broadcast_index(idx, rank, dist, dev)
    assert not on_dev               # (gloo on CPU: no context to receive into; the RCCL form is tests/test_e2e_gpu.py::test_index_handed_on_device_to_device)
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


def _map_worker(rank, world, port, tmp, q):
    sys.path.insert(0, ROOT)
    sys.path.insert(0, os.path.join(ROOT, "tests"))
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    import ctypes as C
    import torch
    import torch.distributed as dist
    import e2e_common as E
    import wmtest as W
    from winnowmap_amd import gpu, build
    from winnowmap_amd import dist as wmdist
    dist.init_process_group("gloo", rank=rank, world_size=world)
    preset, fa, kf, k, reads = E.make_golden.inputs("ont_short", os.path.join(tmp, "r%d" % rank))   # (seeded: every rank regenerates the same reads)
    idx = gpu.Index(fa, kf, k=k, w=50, n_threads=2) if rank == 0 else None
    idx, _ = wmdist.broadcast_index(idx, rank, dist, torch.device("cpu"))
    # the received index drives the HOST mapper of this rank (oracle-backed device ops: no GPU here) on this rank's shard of the reads
    mmi = os.path.join(tmp, "rank%d.mmi" % rank)
    idx.save(mmi)
    H = C.CDLL(build.build_harness())
    H.h_index_load_mmi.restype = C.c_void_p
    H.h_index_load_mmi.argtypes = [C.c_char_p, C.c_char_p]
    H.h_map.argtypes = [C.c_void_p, C.c_char_p, C.c_int64, C.c_char_p, C.c_int, C.c_char_p, W.i32p, C.c_int, W.u32p, C.c_int64, C.POINTER(C.c_int64), W.u64p]
    h = H.h_index_load_mmi(mmi.encode(), b"")
    out = {}
    for i in wmdist.shard(len(reads), rank, world):
        s = reads[i]
        ho = np.zeros(16 * 256, np.int32); co = np.zeros(2000000, np.uint32); nc = C.c_int64(); st = np.zeros(4, np.uint64)
        n = H.h_map(h, preset.encode(), 0x4 | 0x20, s, len(s), ("read%d" % i).encode(), ho, 256, co, len(co), C.byref(nc), st)
        out[i] = (ho[:16 * n].reshape(-1, 16).copy(), co[:nc.value].copy())
    q.put((rank, out))
    dist.barrier()
    dist.destroy_process_group()


def test_sharded_mapping_with_broadcast_index_world2():
    """world_size 2: rank 0 builds the index, gloo broadcasts it, each rank maps its shard with the received index; the union
    of the shards equals the reference's golden result for the whole batch."""
    import torch.multiprocessing as mp
    sys.path.insert(0, os.path.join(ROOT, "tests"))
    import e2e_common as E
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    tmp = tempfile.mkdtemp()
    for r in range(2):
        os.makedirs(os.path.join(tmp, "r%d" % r))
    port = _free_port()
    ps = [ctx.Process(target=_map_worker, args=(r, 2, port, tmp, q)) for r in range(2)]
    for p in ps:
        p.start()
    res = dict()
    for _ in ps:
        rank, out = q.get(timeout=600)
        res.update(out)
    for p in ps:
        p.join(60)
        assert p.exitcode == 0
    gh, gc, gf = E.golden("ont_short")
    reads = E.make_golden.inputs("ont_short", tmp)[4]
    assert sorted(res) == list(range(len(gf) - 1))
    co = 0
    for i in range(len(gf) - 1):
        hits, cig = res[i]
        want = gh[gf[i]:gf[i + 1]]
        E.mask_mapq(len(reads[i]), hits)
        assert np.array_equal(hits, want), i
        nc = int(want[:, 7].sum())
        assert np.array_equal(cig, gc[co:co + nc]), i
        co += nc


def test_launcher_starts_n_ranks():
    """bench.py's own launcher (winnowmap_amd.dist.launch_ranks, used for a plain `python bench.py --gpus N`): N ranks come up
    with WORLD_SIZE == N and distinct LOCAL_RANKs; the host threads are divided between them."""
    sys.path.insert(0, ROOT)
    from winnowmap_amd import dist as wmdist
    tmp = tempfile.mkdtemp()
    rc = wmdist.launch_ranks(2, [os.path.join(ROOT, "tests", "dist_probe.py"), tmp])
    assert rc == 0
    seen = sorted(open(os.path.join(tmp, f)).read().split() for f in os.listdir(tmp))
    assert seen == [["0", "2", "0"], ["1", "2", "1"]]
    assert wmdist.host_threads_per_rank(256, 1) == 32 and wmdist.host_threads_per_rank(128, 8) == 16 and wmdist.host_threads_per_rank(16, 1) == 16 and wmdist.host_threads_per_rank(4, 8) == 1
    assert 1 <= wmdist.available_cores() <= (os.cpu_count() or 1)
