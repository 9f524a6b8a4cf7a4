"""Device vs host index build on the bench reference (250 Mb): times and equality of the flat arrays."""
import os, sys, time, tempfile
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
from winnowmap_amd import gpu, synth
import importlib.util
spec = importlib.util.spec_from_file_location("bench", os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "bench.py"))
B = importlib.util.module_from_spec(spec); spec.loader.exec_module(B)
mb = float(sys.argv[1]) if len(sys.argv) > 1 else 250
tmp = tempfile.mkdtemp()
ref, fa, kf = B.make_workload(mb, tmp)
c = gpu.Context(0, 24 << 30)
t0 = time.time(); dev, st = gpu.Index.build_on_device(c, fa, kf, k=15, w=50, n_threads=16); t_dev = time.time() - t0
t0 = time.time(); host = gpu.Index(fa, kf, k=15, w=50, n_threads=16); t_host = time.time() - t0
hs, ha = host.export_arrays(); ds, da = dev.export_arrays()
same = bool(np.array_equal(hs, ds) and all(np.array_equal(a, b) for a, b in zip(ha, da)))
print("reference %.0f Mb: host build %.2f s | device build %.2f s (read+pack %.2f, device sketch %.2f, table %.2f) | %d minimizers | identical arrays: %s" %
      (mb, t_host, t_dev, st["read_pack_s"], st["device_sketch_s"], st["table_s"], st["minimizers"], same))
t0 = time.time(); n_dev, kst = gpu.write_repetitive_kmers_gpu(c, fa, 15, tmp + "/rep_dev.txt"); t_k = time.time() - t0
t0 = time.time(); n_host = gpu.write_repetitive_kmers(fa, 15, tmp + "/rep_host.txt"); t_kh = time.time() - t0
print("-W list: host %.2f s | device %.2f s (read+encode %.2f, device %.2f, write %.2f) | %d k-mers of %d distinct | identical files: %s" %
      (t_kh, t_k, kst["read_encode_s"], kst["device_s"], kst["write_s"], n_dev, kst["distinct_kmers"], open(tmp + "/rep_dev.txt", "rb").read() == open(tmp + "/rep_host.txt", "rb").read() and n_dev == n_host))
