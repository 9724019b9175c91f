import csv, re, sys
rows = list(csv.DictReader(open(sys.argv[1])))
rows.sort(key=lambda r: int(r["Start_Timestamp"]))
def short(n):
    n = re.sub(r"\(anonymous namespace\)::", "", n); n = re.sub(r"^void ", "", n)
    return n.split("(")[0][:48]
idx = [i for i, r in enumerate(rows) if "gemm_v4_kernel<1," in r["Kernel_Name"]]          # video FFN-up: once per layer
a, b = idx[-10], idx[-9]
t0 = int(rows[a]["Start_Timestamp"])
qs = {}
for r in rows[a:b + 1]:
    q = qs.setdefault(r.get("Queue_Id", "?"), len(qs))
    s, e = int(r["Start_Timestamp"]), int(r["End_Timestamp"])
    print(f"{(s - t0) / 1e3:8.1f} {(e - s) / 1e3:7.1f} q{q} {short(r['Kernel_Name'])}")
