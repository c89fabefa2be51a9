#!/usr/bin/env python3
"""VGPRs, scratch instructions and code size of the built integer column kernels (edt_colq16.hip), per instantiation -- and,
since round 6, how many of the fill's row loads are IN FLIGHT TOGETHER: the longest run of 8-byte row loads (global_load_dwordx2)
that no `s_waitcnt vmcnt` interrupts.  Sixteen = the whole tile of a thread in one trip to memory; the build before round 6's last
session had 2 (a wait behind the second load: a register copy the allocator had put there), which cost every tile a second trip."""
import os, re, subprocess, sys, tempfile
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
CSRC = os.path.join(ROOT, "euclidean-distance-transform-3d_amd", "csrc")
FLAGS = ["--offload-arch=gfx950", "-O3", "-std=c++17", "-fPIC", "-ffp-contract=off", "-fno-fast-math",
         "-fhip-fp32-correctly-rounded-divide-sqrt", "-I" + os.path.join(ROOT, "include"), "-I" + CSRC]
FLAGS += os.environ.get("EDT_SPILL_EXTRA", "").split()


def scan(src="edt_colq16.hip"):
    with tempfile.TemporaryDirectory() as d:
        out_s = os.path.join(d, "k.s")
        subprocess.run(["/opt/rocm/bin/hipcc", *FLAGS, "-S", "--cuda-device-only", "-o", out_s, os.path.join(CSRC, src)],
                       check=True, capture_output=True, cwd=d)
        asm = open(out_s).read()
    out = []
    for m in re.finditer(r"^(_ZN\w+):\s*; @", asm, re.M):
        end = asm.index(".Lfunc_end", m.start())
        body = asm[m.start():end]
        tail = asm[end:end + 4000]
        vg = re.search(r"; NumVgprs: (\d+)", tail)
        occ = re.search(r"; Occupancy: (\d+)", tail)
        cl = re.search(r"; codeLenInByte = (\d+)", tail)
        dem = subprocess.run(["c++filt", m.group(1)], capture_output=True, text=True).stdout.strip()
        run = best = 0
        for ins in re.findall(r"^\s+(global_load_dwordx2|s_waitcnt vmcnt)", body, re.M):
            run = run + 1 if ins == "global_load_dwordx2" else 0
            best = max(best, run)
        out.append({"name": dem, "scratch_ops": len(re.findall(r"\bscratch_(?:load|store)", body)), "loads_in_flight": best,
                    "vgprs": int(vg.group(1)) if vg else None, "occupancy": int(occ.group(1)) if occ else None,
                    "bytes": int(cl.group(1)) if cl else None})
    return out


if __name__ == "__main__":
    worst = 0
    for f in scan(*(sys.argv[1:2])):
        short = re.sub(r"\(.*", "", f["name"]).replace("void edt_amd::", "")
        print(f"scratch {f['scratch_ops']:4d}  vgprs {f['vgprs']:4d}  occupancy {f['occupancy']}  code {f['bytes']:7d} B  row loads in flight {f['loads_in_flight']:2d}  {short}")
        worst = max(worst, f["scratch_ops"])
    sys.exit(1 if worst > 0 else 0)
