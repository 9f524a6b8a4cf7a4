"""winnowmap_amd — MI355X-native seed→chain→align hot path of Winnowmap behind a C-ABI (include/wm_gpu.h)."""
import os

# The mapper drives the GPU from many host threads, one HIP stream each; ROCm maps streams onto 4 hardware queues by
# default, which serialises them. 16 or 20 queues run that many kernels at once (profiles/r02f_stream_conc.txt); 24 and more
# oversubscribe the queue slots. 20 = 6 device contexts + 14 side streams split by the weight of a call (round 4: +14 % over 16,
# profiles/r04g_sched_sweep.txt). Must be set before the HIP runtime initialises (the library sets the same default itself).
os.environ.setdefault("GPU_MAX_HW_QUEUES", "20")
