#!/usr/bin/env python3
"""The measurement table of DESIGN.md 5.1 from a bench line: python tools/bench_table.py profiles/r06_bench_line.json"""
import json, sys
d = json.load(open(sys.argv[1] if len(sys.argv) > 1 else "profiles/r06_bench_line.json"))
k = d["roofline"]["kernel_ms"]
print(f"| headline {d['config']['workload'].split(':')[0]} | {d['ms_per_step']:.4f} | {d['value']:.0f} | {d['roofline']['frac']:.3f} | "
      f"{k.get('x_pass', 0):.3f} | {k.get('y_pass', 0):.3f} | {k.get('z_pass', 0):.3f} | bits {k.get('z_bits', 0):.3f} |")
for s in d.get("secondary", []):
    km = s.get("kernel_ms") or {}
    xy = km.get("xy_pass")
    print(f"| {s['config']} | {s.get('ms_per_step', s.get('gpu_seconds_total'))} | {s.get('mvox_per_s', '')} | {s.get('whole_job_frac', s.get('model_frac', ''))} | "
          f"{('X + Y %.2f' % xy) if xy else '%.3f' % km.get('x_pass', 0)} | {km.get('y_pass', 0):.3f} | {km.get('z_pass', 0):.3f} | "
          f"verified {s.get('output_verified')} {('two-transform %.3f' % s['two_transform_ms']) if 'two_transform_ms' in s else ''} |")
r = d["roofline"]
print("real:", {x: r.get(x) for x in ("whole_job_real_bytes", "whole_job_real_GBs", "whole_job_real_frac_of_spec", "whole_job_real_frac_of_achievable")})
print("cpu:", d["cpu_baseline"]["sample"], "| end_to_end:", d["end_to_end"]["numpy_to_numpy_ms"])
