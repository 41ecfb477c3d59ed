"""Condense a rocprofv3 *_kernel_stats.csv into a short, committed summary (kernel names truncated)."""
import csv, sys
src, dst = sys.argv[1], sys.argv[2]
rows = list(csv.DictReader(open(src)))
with open(dst, "w") as f:
    f.write("# source: rocprofv3 --kernel-trace --stats -- python bench.py --steps 3 --warmup 1 --no-cpu-baseline (4 forwards incl. warm-up)\n")
    f.write("name,calls,total_ms,avg_us,percent,min_us,max_us\n")
    for r in rows:
        if float(r["Percentage"]) < 0.004:
            continue
        name = r["Name"].replace("(anonymous namespace)::", "").replace("void ", "")
        name = name[:110].replace(",", ";")
        f.write(f'{name},{r["Calls"]},{float(r["TotalDurationNs"])/1e6:.3f},{float(r["AverageNs"])/1e3:.2f},{float(r["Percentage"]):.3f},'
                f'{float(r["MinNs"])/1e3:.2f},{float(r["MaxNs"])/1e3:.2f}\n')
