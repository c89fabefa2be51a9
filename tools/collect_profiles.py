#!/usr/bin/env python3
"""Copy the summaries of a profiling session from gpurun_out/ (scratch) into profiles/ (tracked):
   python tools/collect_profiles.py r03   ->  profiles/r03_kernel_stats_<cfg>.csv, r03_pmc_summary_<cfg>.txt, r03_traffic.json"""
import glob, os, shutil, sys
tag = sys.argv[1] if len(sys.argv) > 1 else "r03"
root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
go, pr = os.path.join(root, "gpurun_out"), os.path.join(root, "profiles")
os.makedirs(pr, exist_ok=True)
for d in sorted(glob.glob(os.path.join(go, f"prof_{tag}_*"))):
    if not os.path.isdir(d):
        continue
    cfg = os.path.basename(d)[len(f"prof_{tag}_"):]
    stats = glob.glob(os.path.join(d, "**", "*kernel_stats.csv"), recursive=True)
    if stats:
        shutil.copy(stats[0], os.path.join(pr, f"{tag}_kernel_stats_{cfg}.csv"))
        print("stats", cfg)
for f in sorted(glob.glob(os.path.join(go, f"pmc_{tag}*_summary.txt"))):
    cfg = os.path.basename(f)[len(f"pmc_{tag}"):-len("_summary.txt")]
    shutil.copy(f, os.path.join(pr, f"{tag}_pmc_summary_{cfg}.txt"))
    print("pmc", cfg)
t = os.path.join(go, f"{tag}_traffic.json")
if os.path.exists(t):
    shutil.copy(t, os.path.join(pr, f"{tag}_traffic.json"))
    print("traffic")
