import csv, sys, collections
rows = list(csv.DictReader(open(sys.argv[1])))
rows.sort(key=lambda r: int(r["Start_Timestamp"]))
# take the last 4000 kernels; group k_gal_down / k_gal_up by grid size
seg = rows[-6000:]
acc = collections.defaultdict(list)
for r in seg:
    n = r["Kernel_Name"]
    if "k_gal_" in n or "k_mg_down<float, double" in n or "k_mg_up<float, double" in n:
        key = (n.split("(")[0].replace("void ", "").replace("fl::", "")[:40], r.get("Grid_Size", r.get("Grid_Size_X", "?")), r.get("Workgroup_Size", r.get("Workgroup_Size_X", "?")))
        acc[key].append((int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e3)
for k, v in sorted(acc.items()):
    print(k, len(v), "avg us %.2f" % (sum(v) / len(v)))
