import csv, sys, collections, glob
tag = sys.argv[1]
for f in sorted(glob.glob('gpurun_out/pmc_%s_*/p_counter_collection.csv' % tag)):
    rows = list(csv.DictReader(open(f)))
    agg = collections.defaultdict(lambda: collections.defaultdict(list))
    for r in rows:
        agg[r['Kernel_Name'][:40]][r['Counter_Name']].append(float(r['Counter_Value']))
    for k, d in agg.items():
        if 'gemm' not in k and 'mha' not in k and 'ln_' not in k: continue
        print(f.split('/')[1], k, ' '.join('%s=%.4g' % (c, sum(v)/len(v)) for c, v in d.items()))
