import csv, sys
rows = list(csv.DictReader(open(sys.argv[1])))
ev = sorted((int(r["Start_Timestamp"]), int(r["End_Timestamp"]), r["Kernel_Name"], r.get("Queue_Id", "")) for r in rows)
i0 = int(len(ev) * float(sys.argv[2])); n = int(sys.argv[3])
t0 = ev[i0][0]
short = lambda s: (s.replace("void ", "").split("(")[0])[:34]
for s, e, nme, q in ev[i0:i0 + n]:
    print(f"{(s - t0)/1e3:9.1f} -> {(e - t0)/1e3:9.1f} us  ({(e - s)/1e3:7.1f})  q{q}  {short(nme)}")
