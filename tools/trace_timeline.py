"""Timeline excerpt of a rocprofv3 --kernel-trace CSV (diagnostic): kernels of the LAST training step between two anchors.
python tools/trace_timeline.py <k_kernel_trace.csv> <anchor kernel substring> [before] [after]"""
import csv, re, sys
f, anchor = sys.argv[1], sys.argv[2]
before, after = (int(sys.argv[3]) if len(sys.argv) > 3 else 10), (int(sys.argv[4]) if len(sys.argv) > 4 else 40)
which = sys.argv[5] if len(sys.argv) > 5 else "last"
rows = list(csv.DictReader(open(f)))
rows.sort(key=lambda r: int(r["Start_Timestamp"]))
def short(n):
    n = re.sub(r"\(anonymous namespace\)::", "", n); n = re.sub(r"^void ", "", n)
    return n[:60]
idx = [i for i, r in enumerate(rows) if anchor in r["Kernel_Name"]]
a = idx[-1] if which == "last" else idx[0]
t0 = int(rows[a]["Start_Timestamp"])
for r in rows[max(0, a - before):a + after]:
    s, e = int(r["Start_Timestamp"]), int(r["End_Timestamp"])
    print(f"{(s - t0) / 1e3:10.1f} {(e - s) / 1e3:8.1f} q{r['Queue_Id']} {short(r['Kernel_Name'])} wg {int(r['Grid_Size_X']) // max(int(r['Workgroup_Size_X']), 1)}x{r['Grid_Size_Y']}")
