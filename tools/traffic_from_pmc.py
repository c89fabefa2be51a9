#!/usr/bin/env python3
"""profiles/<tag>_traffic.json from the PMC summaries of tools/gpu_session.sh pmc (tools/profile_r03.sh): HBM bytes per launch and pass.

FETCH_SIZE / WRITE_SIZE are collected in separate rocprofv3 --pmc passes; unit KiB; corrected with the factors the
calibration of this round found on copies of known size (profiles/r02_counter_calibration.txt): FETCH_SIZE x 2.0 for
4 B/lane and 16 B/lane reads, WRITE_SIZE x 1.0.  With the index form of pass 1 the first column pass is the XF
instantiation of the column kernel (third template argument true), the second the plain one, so the two passes have
counters of their own; otherwise both carry the mean of the kernel's launches."""
import json, re, sys
tag = sys.argv[1] if len(sys.argv) > 1 else "r02"
FETCH_FACTOR, WRITE_FACTOR, KIB = 2.0, 1.0, 1024
out = {}
for cfg in ("cfg2", "cfg3", "cfg3m", "cfg3L", "cfg3La", "cfg3M", "cfg3Ma"):
    try:
        text = open(f"gpurun_out/pmc_{tag}{cfg}_summary.txt").read()
    except OSError:
        continue
    kernels, cur = {}, None
    for line in text.splitlines():
        if not line.startswith(" "):
            cur = kernels.setdefault(line.strip(), {})
        else:
            m = re.match(r"\s+(\S+)\s+([0-9.]+)", line)
            if m and cur is not None:
                cur[m.group(1)] = float(m.group(2))
    def entry(k):
        rd = int(kernels[k].get("FETCH_SIZE", 0) * FETCH_FACTOR * KIB)
        wr = int(kernels[k].get("WRITE_SIZE", 0) * WRITE_FACTOR * KIB)
        return {"unit": "bytes/launch", "read": rd, "write": wr, "total": rd + wr, "kernel": k}
    res = {}
    cols = [k for k in kernels if k.startswith("k_column_pass_wave")]
    for k in kernels:
        if k.startswith("k_row_pass_wave"):
            res["x_pass"] = entry(k)
        elif k.startswith("edt_amd::k_bits_transpose_yz") or "k_bits_transpose_yz" in k:
            res["z_bits"] = entry(k)
    xf = [k for k in cols if re.match(r"k_column_pass_wave<\d+, (true|false), true", k)]
    plain = [k for k in cols if k not in xf]
    if xf and plain:
        res["y_pass"] = entry(xf[0])
        res["z_pass"] = entry(plain[0])
    elif cols:
        res["y_pass"] = res["z_pass"] = entry(cols[0])
    out[cfg] = res
out["_note"] = ("rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE in separate passes (tools/gpu_session.sh pmc), 512^3 uint32, counter unit KiB; "
                "FETCH_SIZE x 2.0 (4 B/lane and 16 B/lane reads), WRITE_SIZE x 1.0 -- profiles/r02_counter_calibration.txt; "
                "tools/traffic_from_pmc.py")
json.dump(out, open(f"gpurun_out/{tag}_traffic.json", "w"), indent=1)
print(json.dumps({c: {p: v["total"] for p, v in r.items()} for c, r in out.items() if c != "_note"}))
