"""Prints the kernel timeline of one steady-state training step from a rocprofv3 --kernel-trace CSV
(start/end relative to the step's first kernel, queue, kernel name) and the step span."""
import csv
import re
import sys

rows = list(csv.DictReader(open(sys.argv[1])))
rows.sort(key=lambda r: int(r["Start_Timestamp"]))
idx = [i for i, r in enumerate(rows) if "k_adam" in r["Kernel_Name"]]
a, b = idx[-4], idx[-3]
t0 = int(rows[a + 1]["Start_Timestamp"])
for r in rows[a + 1:b + 1]:
    s, e = int(r["Start_Timestamp"]) - t0, int(r["End_Timestamp"]) - t0
    name = re.sub(r"\(anonymous namespace\)::", "", r["Kernel_Name"])
    name = re.sub(r"^void ", "", name)[:44]
    print(f"{s / 1000:8.1f} {e / 1000:8.1f} {(e - s) / 1000:7.1f}  q{r['Queue_Id']} {name}")
print("step span us", (int(rows[b]["End_Timestamp"]) - int(rows[a]["End_Timestamp"])) / 1000)
