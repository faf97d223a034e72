import csv, glob, sys, collections, re
for d in sys.argv[1:]:
    print("==", d)
    dur = collections.defaultdict(lambda: [0, 0.0])
    for f in glob.glob(d + "/**/*kernel_trace.csv", recursive=True):
        for r in csv.DictReader(open(f)):
            n = re.sub(r"\(.*", "", r["Kernel_Name"].replace("void (anonymous namespace)::", "").replace("(anonymous namespace)::", ""))
            dur[n][0] += 1; dur[n][1] += (int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e6
    for n, (c, ms) in sorted(dur.items(), key=lambda kv: -kv[1][1])[:6]:
        print(f"  {n:40s} calls {c:5d} total {ms:9.3f} ms")
    cnt = collections.defaultdict(lambda: collections.defaultdict(float))
    for f in glob.glob(d + "/**/*counter_collection.csv", recursive=True):
        for r in csv.DictReader(open(f)):
            n = re.sub(r"\(.*", "", r["Kernel_Name"].replace("void (anonymous namespace)::", "").replace("(anonymous namespace)::", ""))
            cnt[n][r["Counter_Name"]] += float(r["Counter_Value"])
    for n in sorted(cnt, key=lambda k: -sum(cnt[k].values()))[:3]:
        print("  ", n, {k: f"{v:.4g}" for k, v in cnt[n].items()})
