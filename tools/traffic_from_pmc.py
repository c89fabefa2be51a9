#!/usr/bin/env python3
"""profiles/<tag>_traffic.json from the raw PMC passes of tools/gpu_session.sh pmc: HBM bytes per STEP and pass.

FETCH_SIZE / WRITE_SIZE are collected in separate rocprofv3 --pmc passes (pass 1 and pass 2 of the session step); unit
KiB; corrected with the factors the calibration found on copies of known size (profiles/r02_counter_calibration.txt):
FETCH_SIZE x 2.0 for 4 B/lane and 16 B/lane reads, WRITE_SIZE x 1.0.

Since round 4 a pass is more than one kernel (the 16-bit integer column kernel, then the fp32 kernel over the tiles it
refused; slab-wise X / Y passes of volumes beyond 2^27 voxels), so the table is built from EVERY dispatch of the run: the
dispatches are sorted into passes by kernel name, summed, and divided by the number of steps the run made (= dispatches of
the bit-plane transposer, one per 3-D transform; voxel graph: of k_vg_rows).  usage: traffic_from_pmc.py <tag> [cfg ...]"""
import collections, csv, glob, json, os, re, sys
tag = sys.argv[1] if len(sys.argv) > 1 else "r04"
cfgs = sys.argv[2:] or ["cfg2", "cfg3", "cfg3m", "cfg3L", "cfg3La", "cfg3M", "cfg3Ma", "cfg4", "cfg5"]
FETCH_FACTOR, WRITE_FACTOR, KIB = 2.0, 1.0, 1024


def pass_of(name):
    # only this library's kernels (and its memset of the hand-over counters) belong to a step: the copies of the bench's
    # verification, torch's fills and RCCL's kernels do not
    if "edt_amd::" not in name:
        return "other" if "fillBufferAligned" in name else None
    n = name.replace("void ", "").replace("edt_amd::", "").replace("(anonymous namespace)::", "")
    if "k_row_pass" in n or "k_line_" in n or "k_rows_" in n:
        return "x_pass"
    if "k_bits_transpose_yz" in n:
        return "z_bits"
    if "k_negate_background" in n:
        return "sign"
    m = re.match(r"k_column_pass_q16<(true|false), (\d)", n)
    if m:
        return "y_pass" if m.group(2) == "1" else "z_pass"
    m = re.match(r"k_column_pass_wave<\d+, (true|false), (true|false)", n)
    if m:
        return "y_pass" if m.group(2) == "true" else "z_pass"
    if "k_column_pass" in n:
        return "z_pass"
    if n.startswith("k_vg_rows"):
        return "x_pass"
    if n.startswith("k_vg_"):
        return "vg_bits"
    if "k_pack_record_bits" in n:
        return "pack_bits"
    return "other"


def read_pass(cfg, i, counter):
    per = collections.defaultdict(float)
    names = collections.defaultdict(set)
    steps = {"t": 0, "vg": 0}
    files = glob.glob(f"gpurun_out/pmc_{tag}{cfg}_{i}/**/*counter_collection.csv", recursive=True)
    if not files:
        return None
    with open(files[0]) as f:
        for row in csv.DictReader(f):
            if row["Counter_Name"] != counter:
                continue
            k = row["Kernel_Name"]
            p = pass_of(k)
            if p is None:
                continue
            per[p] += float(row["Counter_Value"])
            names[p].add(k.replace("(anonymous namespace)::", "").split("(")[0].replace("void ", "").replace("edt_amd::", ""))
            if "k_bits_transpose_yz" in k:
                steps["t"] += 1
            if "k_vg_rows" in k:
                steps["vg"] += 1
    n = steps["vg"] or steps["t"]
    return per, names, n


out = {}
for cfg in cfgs:
    rd = read_pass(cfg, 1, "FETCH_SIZE")
    wr = read_pass(cfg, 2, "WRITE_SIZE")
    if not rd or not wr or not rd[2] or not wr[2]:
        continue
    res = {}
    for p in sorted(set(rd[0]) | set(wr[0])):
        r = int(rd[0].get(p, 0.0) * FETCH_FACTOR * KIB / rd[2])
        w = int(wr[0].get(p, 0.0) * WRITE_FACTOR * KIB / wr[2])
        res[p] = {"unit": "bytes/step", "read": r, "write": w, "total": r + w, "kernels": sorted(rd[1].get(p, set()) | wr[1].get(p, set()))}
    res["_steps_seen"] = [rd[2], wr[2]]
    out[cfg] = res
out["_note"] = ("rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE in separate passes (tools/gpu_session.sh pmc), every dispatch of the run "
                "sorted into passes by kernel name and divided by the steps of the run; counter unit KiB; FETCH_SIZE x 2.0, "
                "WRITE_SIZE x 1.0 -- profiles/r02_counter_calibration.txt; tools/traffic_from_pmc.py")
os.makedirs("gpurun_out", exist_ok=True)
json.dump(out, open(f"gpurun_out/{tag}_traffic.json", "w"), indent=1)
print(json.dumps({c: {p: v["total"] for p, v in r.items() if isinstance(v, dict)} for c, r in out.items() if c != "_note"}))
