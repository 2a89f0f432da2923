import csv, glob, collections, sys
pat = sys.argv[1] if len(sys.argv) > 1 else 'gpurun_out/pmc_*'
for d in sorted(glob.glob(pat)):
    for f in glob.glob(d + '/*/*counter_collection.csv'):
        rows = list(csv.DictReader(open(f)))
        agg = collections.defaultdict(float)
        for r in rows:
            if 'point_kernel' in r['Kernel_Name']:
                agg[r['Counter_Name']] += float(r['Counter_Value'])
        print(d.split('/')[-1], {k: f'{v:.4g}' for k, v in agg.items()})
