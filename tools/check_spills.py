#!/usr/bin/env python3
"""Scratch (spill) instructions in the built column-pass kernels, per function.

The hull path of k_column_pass_wave needs 127 of the 128 VGPRs that four waves per SIMD leave it; a change anywhere
in the kernel body can tip its hot loops into scratch with byte-identical hull code (measured: cfg2 0.69 -> 1.05 ms).
Run this after touching csrc/edt_colwave_kernel.h / edt_colwave_lane.h:  python tools/check_spills.py [cw ...]
Kernel bodies must show 0 (a handful at most).  (With -DEDT_BRUTE_INLINE='__attribute__((noinline))' the windowed
path is a function of its own and shows ~30-40: callee-saved registers saved once per call, not spills in loops.)"""
import os, re, subprocess, sys, tempfile

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
CSRC = os.path.join(ROOT, "euclidean-distance-transform-3d_amd", "csrc")
FLAGS = ["--offload-arch=gfx950", "-O3", "-std=c++17", "-fPIC", "-ffp-contract=off", "-fno-fast-math",
         "-fhip-fp32-correctly-rounded-divide-sqrt", "-I" + os.path.join(ROOT, "include"), "-I" + CSRC]
FLAGS += os.environ.get("EDT_SPILL_EXTRA", "").split()  # e.g. EDT_SPILL_EXTRA=-DEDT_CONTIG: a variant build


def scan(cw, workdir):
    src = os.path.join(CSRC, f"edt_colwave_cw{cw}.hip")
    out_s = os.path.join(workdir, f"cw{cw}.s")
    subprocess.run(["/opt/rocm/bin/hipcc", *FLAGS, "-S", "--cuda-device-only", "-o", out_s, src],
                   check=True, capture_output=True, cwd=workdir)
    asm = open(out_s).read()
    out = []
    for m in re.finditer(r"^(_ZN\w+):\s*; @", asm, re.M):
        end = asm.index(".Lfunc_end", m.start())
        body = asm[m.start():end]
        tail = asm[end:end + 4000]
        vg = re.search(r"; NumVgprs: (\d+)", tail)
        occ = re.search(r"; Occupancy: (\d+)", tail)
        out.append({"name": m.group(1), "kernel": "k_column_pass_wave" in m.group(1),
                    "scratch_ops": len(re.findall(r"\bscratch_(?:load|store)", body)),
                    "vgprs": int(vg.group(1)) if vg else None, "occupancy": int(occ.group(1)) if occ else None})
    return out


if __name__ == "__main__":
    cws = [int(a) for a in sys.argv[1:]] or [32, 16, 8, 4, 2, 1]
    worst = 0
    with tempfile.TemporaryDirectory() as d:
        for cw in cws:
            for f in scan(cw, d):
                print(f"cw{cw:<3d} {'kernel  ' if f['kernel'] else 'function'} scratch ops {f['scratch_ops']:4d}  vgprs {f['vgprs']}  "
                      f"occupancy {f['occupancy']}  {f['name'][:70]}")
                if f["kernel"]:
                    worst = max(worst, f["scratch_ops"])
    sys.exit(1 if worst > 8 else 0)
