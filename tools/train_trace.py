"""Summarise a rocprofv3 --kernel-trace of `python -m be_hip.train_local`: the launch sequence of ONE training step (the
last complete one: from one k_pack_jobs to the next) with each launch's duration, and per-kernel totals of that step."""
import csv
import glob
import sys

root = sys.argv[1]
files = glob.glob(root + "/**/*kernel_trace.csv", recursive=True)
if not files:
    sys.exit("no kernel_trace.csv under " + root)
rows = list(csv.DictReader(open(files[0])))
rows.sort(key=lambda r: int(r["Start_Timestamp"]))
names = [r["Kernel_Name"] for r in rows]
marks = [i for i, n in enumerate(names) if "k_pack_jobs" in n]
if len(marks) < 3:
    sys.exit("fewer than three steps in the trace")
lo, hi = marks[-2], marks[-1]
step = rows[lo:hi]
t0 = int(step[0]["Start_Timestamp"])
tot = {}
busy = 0
for i, r in enumerate(step):
    s, e = int(r["Start_Timestamp"]), int(r["End_Timestamp"])
    d = (e - s) / 1e3
    busy += d
    nm = r["Kernel_Name"].replace("(anonymous namespace)::", "").replace("void ", "").split("(")[0][:70]
    g = (r.get("Grid_Size_X") or r.get("Grid_Size") or "?", r.get("Grid_Size_Y", ""), r.get("Grid_Size_Z", ""))
    print(f"{i:4d} {(s - t0) / 1e3:9.1f} us  {d:8.2f} us  grid {g[0]:>8}x{g[1]}x{g[2]}  {nm}")
    k = tot.setdefault(nm, [0, 0.0])
    k[0] += 1
    k[1] += d
span = (int(step[-1]["End_Timestamp"]) - t0) / 1e3
print(f"\nlaunches {len(step)}  kernel time {busy:.1f} us  span {span:.1f} us (eager)")
for nm, (n, d) in sorted(tot.items(), key=lambda kv: -kv[1][1]):
    print(f"{d:9.1f} us {n:4d} x  {nm}")
