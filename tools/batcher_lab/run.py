#!/usr/bin/env python3
"""Batcher lab: the C++ twin's host machinery (Batcher::submit, drivers, worker pool, GpuRouter::expand_chunk) against a STUB of the
C ABI whose device pass is a sleep — what the boundary costs per publish on the host, measurable without a GPU.
  python tools/batcher_lab/run.py [submitters outstanding workers passes seconds]"""
import ctypes as C
import os
import subprocess
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
HOST = os.path.join(ROOT, "rmqtt_amd", "host")
OUT = "/tmp/libhost_stub.so"
srcs = [os.path.join(HOST, f) for f in ("gpu_router.cpp", "gpu_retain.cpp", "raft_snapshot.cpp", "router_capi.cpp")] + [os.path.join(ROOT, "tools", "batcher_lab", "stub_abi.cpp")]
subprocess.check_call(["g++", "-O2", "-g", "-std=c++17", "-fPIC", "-shared", "-pthread", "-I", os.path.join(ROOT, "include"), "-I", HOST] + srcs + ["-o", OUT, "-lz", "-ldl"])
L = C.CDLL(OUT)
vp = C.c_void_p
L.hr_new.restype = vp; L.hr_new.argtypes = [C.c_uint64, C.c_int]
L.hr_set_match_mode.argtypes = [vp, C.c_int]; L.hr_set_match_mode.restype = None
L.hr_e2e_run_async.argtypes = [vp, vp, vp, C.c_uint32, C.c_uint32, C.c_uint32, C.c_uint32, C.c_uint32, C.c_uint32, C.c_uint32, C.c_double, vp, vp, vp, C.c_uint32, vp]
a = [int(x) for x in sys.argv[1:5]] if len(sys.argv) >= 5 else [4, 8192, 4, 2]
secs = float(sys.argv[5]) if len(sys.argv) > 5 else 3.0
topics = [f"l0x{i % 16}/l1x{i % 64}/l2x{i % 256}/l3x{i % 1024}/l4x{i}".encode() for i in range(20000)]
blob = np.frombuffer(b"".join(topics), dtype=np.uint8).copy()
offs = np.zeros(len(topics) + 1, dtype=np.uint64); offs[1:] = np.cumsum([len(t) for t in topics])
g = L.hr_new(1, 0)
L.hr_set_match_mode(g, 1)
res = (C.c_uint64 * 10)(); wall = C.c_double(0); lat = np.zeros(400000, dtype=np.float32); nl = C.c_uint32(0)
L.hr_e2e_run_async(g, blob.ctypes.data, offs.ctypes.data, len(topics), a[0], a[1], a[2], a[3], 4096, 200, secs, res, C.byref(wall), lat.ctypes.data, len(lat), C.byref(nl))
l = np.sort(lat[:nl.value])
print(f"submitters {a[0]} outstanding {a[1]} workers {a[2]} passes {a[3]}: {res[0] / wall.value / 1e6:.3f} M publishes/s, {res[2]} passes ({res[0] / max(1, res[2]):.0f} per pass), "
      f"per pass ms: collect {res[4] / max(1, res[2]) / 1e6:.3f} device {res[5] / max(1, res[2]) / 1e6:.3f} dispatch {res[6] / max(1, res[2]) / 1e6:.3f}; "
      f"worker task {res[7] / max(1, res[8]) / 1e3:.1f} us ({res[7] / max(1, res[0]):.0f} ns per publish), max task queue {res[9]}; "
      f"latency p50 {l[len(l) // 2]:.0f} us p99 {l[int(len(l) * 0.99)]:.0f} us")
