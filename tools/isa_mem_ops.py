"""Count the memory instructions (ds_ / flat_ / global_ / buffer_ / scratch_) of each kernel in the gfx950 ISA of kernels.hip.
Used to check that LDS-resident data is read with ds_read (not flat_load) and that nothing spills to scratch.
  python tools/isa_mem_ops.py [kernel-name-substring ...]"""
import collections
import glob
import os
import re
import subprocess
import sys
import tempfile

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
with tempfile.TemporaryDirectory() as td:
    subprocess.run(["hipcc", "--offload-arch=gfx950", "-O3", "-std=c++17", "-fPIC", "-c", "-I", os.path.join(ROOT, "include"), "-I",
                    os.path.join(ROOT, "rmqtt_amd", "csrc"), "-x", "hip", *os.environ.get("RGR_EXTRA_FLAGS", "").split(),
                    os.path.join(ROOT, "rmqtt_amd", "csrc", "kernels.hip"), "-o", os.path.join(td, "k.o"), "--save-temps=obj"],
                   check=True, capture_output=True)
    txt = open(glob.glob(os.path.join(td, "*gfx950.s"))[0]).read()
want = sys.argv[1:]
for m in re.finditer(r"^(_ZN3rgr\S+):[^\n]*\n(.*?)\n\s*s_endpgm", txt, re.S | re.M):
    name = subprocess.run(["c++filt", m.group(1)], capture_output=True, text=True).stdout.strip()
    name = re.sub(r"\(anonymous namespace\)::|^void |\(.*", "", name)
    if want and not any(w in name for w in want):
        continue
    c = collections.Counter(re.findall(r"^\s+((?:ds|flat|global|buffer|scratch)_[a-z0-9_]+)", m.group(2), re.M))
    print(f"{name:44s}", " ".join(f"{k}:{v}" for k, v in sorted(c.items())))
