"""winnowmap_amd — MI355X-native seed→chain→align hot path of Winnowmap behind a C-ABI (include/wm_gpu.h)."""
import os

# The mapper drives the GPU from many host threads, one HIP stream each; ROCm maps streams onto 4 hardware queues by
# default, which serialises them. 16 is what an MI355X runs cleanly at once (profiles/r02f_stream_conc.txt: 15.5 kernels in
# flight with 16 or 20 queues; 24 and more oversubscribe the queue slots and fall back to 4..8). Must be set before the HIP
# runtime initialises.
os.environ.setdefault("GPU_MAX_HW_QUEUES", "16")
