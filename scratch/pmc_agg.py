"""Per-kernel averages of rocprofv3 counter_collection.csv files (one directory per --pmc pass)."""
import csv, glob, json, os, sys, collections
out = collections.defaultdict(dict)
for d in sys.argv[1:]:
    if not os.path.isdir(d):
        continue
    for f in glob.glob(os.path.join(d, '**', '*counter_collection.csv'), recursive=True):
        acc = collections.defaultdict(lambda: collections.defaultdict(list))
        for row in csv.DictReader(open(f)):
            name = row['Kernel_Name'].split('(')[0]
            acc[name][row['Counter_Name']].append((row['Dispatch_Id'], float(row['Counter_Value'])))
        for k, cs in acc.items():
            for c, vals in cs.items():
                per = collections.defaultdict(float)
                for did, v in vals:
                    per[did] += v
                out[k][c] = round(sum(per.values()) / len(per), 1)
                out[k]['launches'] = len(per)
keep = {k: v for k, v in out.items() if any(s in k for s in ('forward_move', 'diffuse_rows', 'k_reduce', 'k_pic_'))}
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), '..'))
try:
    import bench
    keep['kernel_source_sha'] = bench.kernel_source_sha()       # which build of die_amd/csrc these counters belong to
except Exception as e:
    keep['kernel_source_sha'] = None
print(json.dumps(keep, indent=1))
