"""Summarise one GlobalStage training step (between two k_pack_jobs launches) of a rocprofv3 --kernel-trace of be_hip.train_global."""
import csv, glob, sys
f = glob.glob(sys.argv[1] + "/**/*kernel_trace.csv", recursive=True)[0]
rows = list(csv.DictReader(open(f)))
rows.sort(key=lambda r: int(r["Start_Timestamp"]))
names = [r["Kernel_Name"] for r in rows]
marks = [i for i, n in enumerate(names) if "k_pack_jobs" in n]
lo, hi = marks[-2], marks[-1]
step = rows[lo:hi]
tot = {}
for r in step:
    nm = r["Kernel_Name"].replace("(anonymous namespace)::", "").replace("void ", "").split("(")[0][:60]
    d = (int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e3
    k = tot.setdefault(nm, [0, 0.0]); k[0] += 1; k[1] += d
print(len(step), "launches", round(sum(v[1] for v in tot.values()) / 1e3, 3), "ms kernel time; span",
      (int(step[-1]["End_Timestamp"]) - int(step[0]["Start_Timestamp"])) / 1e6, "ms")
for nm, (n, d) in sorted(tot.items(), key=lambda kv: -kv[1][1])[:int(sys.argv[2]) if len(sys.argv) > 2 else 30]:
    print(f"{d / 1e3:8.3f} ms {n:4d} x {d / n:8.1f} us  {nm}")
