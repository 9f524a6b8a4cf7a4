"""winnowmap_amd — MI355X-native seed→chain→align hot path of Winnowmap behind a C-ABI (include/wm_gpu.h)."""
import os

# The mapper drives the GPU from many host threads, one HIP stream each; ROCm maps streams onto 4 hardware queues by
# default, which serialises them. Must be set before the HIP runtime initialises.
os.environ.setdefault("GPU_MAX_HW_QUEUES", "32")
