"""ISA-level record of every kernel of the product library: VGPR / AGPR / SGPR, scratch (spills), occupancy, LDS.

Compiles rmqtt_amd/csrc/kernels.hip for gfx950 with -Rpass-analysis=kernel-resource-usage (no GPU needed: hipcc
cross-compiles) and prints one line per kernel.  `python tools/kernel_resources.py > profiles/rNN_kernel_resource_usage.txt`
"""
import os
import re
import subprocess
import sys
import tempfile

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def main():
    extra = os.environ.get("RGR_EXTRA_FLAGS", "").split()
    with tempfile.TemporaryDirectory() as td:
        cmd = ["hipcc", "--offload-arch=gfx950", "-O3", "-std=c++17", "-fPIC", "-c", "-I", os.path.join(ROOT, "include"),
               "-I", os.path.join(ROOT, "rmqtt_amd", "csrc"), "-x", "hip", *extra, os.path.join(ROOT, "rmqtt_amd", "csrc", "kernels.hip"),
               "-o", os.path.join(td, "k.o"), "-Rpass-analysis=kernel-resource-usage"]
        txt = subprocess.run(cmd, capture_output=True, text=True, check=True).stderr
    rows = []
    for b in re.split(r"(?=remark: [^\n]*Function Name)", txt):
        m = re.search(r"Function Name: (\S+)", b)
        if not m:
            continue

        def g(k):
            mm = re.search(k + r": (\d+)", b)
            return mm.group(1) if mm else "?"
        name = subprocess.run(["c++filt", m.group(1)], capture_output=True, text=True).stdout.strip()
        name = re.sub(r"^void ", "", name)
        name = re.sub(r"\(anonymous namespace\)::", "", name)
        name = re.sub(r"\(.*", "", name)
        rows.append((name, g("VGPRs"), g("AGPRs"), g("SGPRs"), g(r"ScratchSize \[bytes/lane\]"), g(r"Occupancy \[waves/SIMD\]"),
                     g(r"LDS Size \[bytes/block\]")))
    print("# hipcc -O3 --offload-arch=gfx950 -Rpass-analysis=kernel-resource-usage rmqtt_amd/csrc/kernels.hip" + (" " + " ".join(extra) if extra else ""))
    print(f"{'kernel':64s} {'VGPR':>5} {'AGPR':>5} {'SGPR':>5} {'scratch B/lane':>15} {'waves/SIMD':>11} {'LDS B/block':>12}")
    for r in rows:
        print(f"{r[0][:64]:64s} {r[1]:>5} {r[2]:>5} {r[3]:>5} {r[4]:>15} {r[5]:>11} {r[6]:>12}")
    return 0


if __name__ == "__main__":
    sys.exit(main())
