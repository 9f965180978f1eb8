"""Per-frame timeline from a rocprofv3 --kernel-trace CSV: kernel durations and the gaps between them (us)."""
import csv, sys, collections
rows = list(csv.DictReader(open(sys.argv[1])))
rows.sort(key=lambda r: int(r["Start_Timestamp"]))
last = int(sys.argv[2]) if len(sys.argv) > 2 else 40
t_prev_end = None
out = []
for r in rows:
    s, e = int(r["Start_Timestamp"]), int(r["End_Timestamp"])
    name = r["Kernel_Name"].split("(")[0][:60]
    if "k_fused" in r["Kernel_Name"]:
        name = "k_fused<" + r["Kernel_Name"].split("<")[1][:40]
    out.append((name, (e - s) / 1e3, (s - t_prev_end) / 1e3 if t_prev_end else 0.0, r.get("Stream_Id", "?")))
    t_prev_end = max(e, t_prev_end or 0)
for o in out[-last:]:
    print(f"{o[0]:60s} dur {o[1]:9.1f} us   gap-before {o[2]:8.1f} us  stream {o[3]}")
