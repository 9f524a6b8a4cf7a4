"""Multi-GPU plumbing (one process per GPU, torch.distributed; backend "nccl" = RCCL on ROCm, "gloo" in the CPU tests).

The mapping path shards by READS and needs no data-path collective; the only exchange is the one-off
broadcast of the flat index arrays from the rank that built it."""
import numpy as np

_EXPORT_DTYPES = (np.uint32, np.uint64, np.uint64, np.uint64, np.uint8, np.uint64, np.uint8)


def export_lengths(sizes):
    """element counts of the 7 exported arrays (S, hkey, hval, P, bloom, seq_meta, names) from the 9 header sizes"""
    s = [int(x) for x in sizes]
    return [s[0], s[1], s[1], s[2], s[3], 2 * s[4], s[5]]


def broadcast_index(idx, rank, dist, device, ctx=None):
    """rank 0 passes its gpu.Index, the others None; every rank returns an Index with identical content.
    One broadcast for the 9-entry size header, then one per flat array (bytes). With `ctx` (a gpu.Context on `device`, RCCL): the five arrays the
    kernels use go from the broadcast's receive buffers straight into the context (wm_index_upload_dev: device to device, no second trip through
    host memory) and the function returns (index, True); the host copy of the index (contig table, packed bases and occurrence counts for the
    host-side glue) is filled from one device-to-host copy per array. Without `ctx` (gloo on CPU): (index, False), the caller uploads."""
    import torch
    from . import gpu
    if rank == 0:
        sizes, arrs = idx.export_arrays()
        st = torch.from_numpy(sizes.astype(np.int64)).to(device)
    else:
        st = torch.zeros(9, dtype=torch.int64, device=device)
    dist.broadcast(st, 0)
    sizes_b = st.cpu().numpy().astype(np.uint64)
    on_device = ctx is not None and getattr(device, "type", "cpu") == "cuda"
    recv, bufs = [], []
    for i, (m, dt) in enumerate(zip(export_lengths(sizes_b), _EXPORT_DTYPES)):
        nbytes = max(m, 1) * np.dtype(dt).itemsize
        if rank == 0:
            t = torch.from_numpy(arrs[i].view(np.uint8)).to(device)
        else:
            t = torch.empty(nbytes, dtype=torch.uint8, device=device)
        dist.broadcast(t, 0)
        bufs.append(t)
        if rank != 0:
            recv.append(t.cpu().numpy().view(dt))
    out = idx if rank == 0 else gpu.Index.from_arrays(sizes_b, recv)
    if on_device:
        torch.cuda.synchronize(device)
        out.upload_dev(ctx, [b.data_ptr() for b in bufs[:5]], device.index if device.index is not None else torch.cuda.current_device())
    return out, on_device


def shard(n, rank, world):
    """indices of the reads rank `rank` maps (round-robin: balances long reads after the reference's length sort, src/map.c:1124-1143)"""
    return list(range(rank, n, world))


def max_over_ranks(value, dist, device):
    import torch
    t = torch.tensor([value], dtype=torch.float64, device=device)
    dist.all_reduce(t, op=dist.ReduceOp.MAX)
    return float(t.item())


def sum_over_ranks(values, dist, device):
    import torch
    t = torch.tensor(list(values), dtype=torch.float64, device=device)
    dist.all_reduce(t, op=dist.ReduceOp.SUM)
    return [float(x) for x in t.tolist()]


def available_cores():
    """Host cores this process can really use: the smallest of the hardware threads, the affinity mask and the container's
    CPU quota (cgroup v2 cpu.max / v1 cfs_quota). The GPU boxes report 256 hardware threads under a 16-CPU quota: threads
    beyond the quota only add throttling (profiles/r02_cpu_scaling.txt)."""
    import os
    n = os.cpu_count() or 1
    try:
        n = min(n, len(os.sched_getaffinity(0)))
    except (AttributeError, OSError):
        pass
    try:
        with open("/sys/fs/cgroup/cpu.max") as f:
            q, p = f.read().split()[:2]
        if q != "max":
            n = min(n, max(1, int(int(q) / int(p))))
    except (OSError, ValueError):
        try:
            with open("/sys/fs/cgroup/cpu/cpu.cfs_quota_us") as f:
                q = int(f.read())
            with open("/sys/fs/cgroup/cpu/cpu.cfs_period_us") as f:
                p = int(f.read())
            if q > 0:
                n = min(n, max(1, q // p))
        except (OSError, ValueError):
            pass
    return n


def host_threads_per_rank(n_cores, world):
    """Host threads one rank's mapper uses: the ranks of a node share its (usable) cores, so they are divided explicitly;
    at most 32 per rank."""
    return max(1, min(32, n_cores // max(1, world)))


def free_port():
    import socket
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


def launch_ranks(n, script_argv, port=None, env=None):
    """Run `script_argv` as n ranks of one node under torch.distributed.run (rendezvous on 127.0.0.1); returns its exit code.
    bench.py uses this when it is started plainly with --gpus N; the gloo tests drive the same launcher on CPU."""
    import subprocess
    import sys
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", str(n),
           "--master-addr", "127.0.0.1", "--master-port", str(port or free_port())] + list(script_argv)
    return subprocess.call(cmd, env=env)
