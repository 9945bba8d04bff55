#!/usr/bin/env python3
"""Per basic block instruction counts (VALU / SALU / LDS / VMEM / SMEM) of one kernel in a `hipcc -S` listing, with each block's
branches — for tracing the dynamic instruction count of a path by hand (profiles/r05g_*: the delivery expansions are bound by
instruction issue, not by memory).
  hipcc --offload-arch=gfx950 -O3 -std=c++17 -I include -I rmqtt_amd/csrc -x hip -S --cuda-device-only -o k.s rmqtt_amd/csrc/kernels.hip
  python tools/isa_blocks.py k.s expand_deliver_lean_kernel [--path L0,L164,L223,...]
"""
import re
import sys


def blocks(path, name):
    on = False
    cur, out = "entry", []
    cnt = dict(v=0, s=0, lds=0, vmem=0, smem=0, br=[])
    for line in open(path):
        if not on:
            if re.match(r"^_Z\w*%s\w*:" % re.escape(name), line):
                on = True
            continue
        t = line.strip()
        m = re.match(r"^\.LBB\d+_(\d+):", t)
        if m:
            out.append((cur, cnt))
            cur, cnt = "L" + m.group(1), dict(v=0, s=0, lds=0, vmem=0, smem=0, br=[])
            continue
        m = re.match(r"^; %bb\.(\d+):", t)
        if m:
            out.append((cur, cnt))
            cur, cnt = "b" + m.group(1), dict(v=0, s=0, lds=0, vmem=0, smem=0, br=[])
            continue
        if not t or t.startswith(";") or t.startswith("."):
            continue
        op = t.split()[0]
        if op.startswith("v_"): cnt["v"] += 1
        elif op.startswith("ds_"): cnt["lds"] += 1
        elif op.startswith("global_") or op.startswith("buffer_") or op.startswith("flat_") or op.startswith("scratch_"): cnt["vmem"] += 1
        elif op.startswith("s_load") or op.startswith("s_buffer_load"): cnt["smem"] += 1
        elif op.startswith("s_"):
            cnt["s"] += 1
            if op.startswith("s_cbranch") or op == "s_branch":
                cnt["br"].append(op.replace("s_cbranch_", "").replace("s_branch", "jmp") + "->L" + t.split("_")[-1])
        if op == "s_endpgm":
            break
    out.append((cur, cnt))
    return out


def main():
    path, name = sys.argv[1], sys.argv[2]
    bl = blocks(path, name)
    if len(sys.argv) > 4 and sys.argv[3] == "--path":
        want = sys.argv[4].split(",")
        tot = dict(v=0, s=0, lds=0, vmem=0, smem=0)
        names = [n for n, _ in bl]
        for w in want:
            rep = 1
            if "*" in w: w, rep = w.split("*")[0], int(w.split("*")[1])
            # a path element "A-B" takes every block from A to B in listing order
            if "-" in w:
                a, b = w.split("-")
                ia, ib = names.index(a), names.index(b)
                sel = bl[ia:ib + 1]
            else:
                sel = [bl[names.index(w)]]
            for _, c in sel:
                for k in tot: tot[k] += rep * c[k]
        print(tot)
        return
    for n, c in bl:
        print(f"{n:>6}  v {c['v']:3d}  s {c['s']:3d}  lds {c['lds']:2d}  vmem {c['vmem']:2d}  smem {c['smem']:2d}  {' '.join(c['br'])}")


if __name__ == "__main__":
    main()
