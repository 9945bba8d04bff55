"""rmqtt_amd — MI355X-native publish-time topic matching for the rmqtt broker.

Only the hot path (TopicTree / Router::matches and the RetainTree twin) lives here:
``csrc/`` holds the HIP kernels (gfx950), the host table compiler and the C ABI
declared in ``include/rmqtt_gpu_router.h``; ``host/`` the C++ mirror of the
reference's Router trait; ``capi.py`` a ctypes view of the C ABI for tests/bench.
"""
__all__ = ["build", "workload", "capi"]
