import csv, glob, collections
seen=collections.defaultdict(list)
for f in glob.glob("/tmp/pmc_C/**/*counter_collection.csv", recursive=True):
    for r in csv.DictReader(open(f)):
        if r["Counter_Name"]=="FETCH_SIZE" and ("conv_wino" in r["Kernel_Name"] or "halo<128" in r["Kernel_Name"] or "dma<128, 1, 0, 0, 0" in r["Kernel_Name"]):
            seen[(r["Kernel_Name"][12:45], r.get("Grid_Size"))].append((int(r["Dispatch_Id"]), float(r["Counter_Value"])*1024*2/1e9))
for k,v in sorted(seen.items()):
    v.sort()
    print(k, len(v), "read GB of the first dispatches in order:", [round(a,2) for _,a in v[:9]])
