"""gpuar_amd -- MI355X-native (gfx950) drop-in for the GPU encode/decode path of jiahansu/GPUAR.

The product is the C-ABI shared library ``gpuar_amd/lib/libgpuar_hip.so``
(include/gpuar_hip.h): hand-written HIP kernels for the per-packet adaptive
arithmetic codec plus the reference-named executors.  This package is the thin
Python side used by the tests and the benchmark: ctypes bindings
(:mod:`gpuar_amd.hip`) and the synthetic input streams (:mod:`gpuar_amd.synth`).
PyTorch appears only as plumbing for device memory, streams and process groups.
"""
__all__ = ["hip", "synth"]
