import csv, glob, sys, collections
d = sys.argv[1]
files = glob.glob(d + '/**/*counter_collection.csv', recursive=True)
agg = collections.defaultdict(lambda: collections.defaultdict(float)); cnt = collections.Counter()
for f in files:
    for r in csv.DictReader(open(f)):
        name = r['Kernel_Name']
        if 'felics' not in name: continue
        name = name.split('felics::')[1].split('<')[0].split('(')[0]
        agg[name][r['Counter_Name']] += float(r['Counter_Value'])
        cnt[(name, r['Counter_Name'])] += 1
steps = int(sys.argv[2]) if len(sys.argv) > 2 else 7
for name in sorted(agg):
    print(name, {k: round(v / steps / 1e6, 3) for k, v in sorted(agg[name].items())}, "(M per step)")
