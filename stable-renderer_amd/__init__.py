"""MI355X-native render-then-diffuse hot path behind Stable-Renderer's node / script API.

Layout: ``csrc/`` hand-written HIP kernels + the C-ABI (``include/sr_hip.h``); the Python modules are the
host-side mirror of the reference's operator interface (same names / argument meaning / error behaviour)
and only do plumbing (device memory via torch, ctypes calls into ``libsr_hip.so``)."""
__version__ = "0.1.0"
