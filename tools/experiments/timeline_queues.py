import csv, sys, collections
rows = list(csv.DictReader(open(sys.argv[1])))
print(rows[0].keys())
q = collections.Counter((r.get("Queue_Id"), r.get("Stream_Id")) for r in rows if "conv" in r["Kernel_Name"])
print("conv kernels by (queue, stream):", q.most_common(12))
q2 = collections.Counter(r.get("Queue_Id") for r in rows)
print("all kernels by queue:", q2.most_common(12))
