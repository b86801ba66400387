"""Sum rocprofv3 --pmc counters per kernel from a counter_collection.csv."""
import collections, csv, sys
agg = collections.defaultdict(lambda: collections.defaultdict(float))
disp = collections.defaultdict(set)
for path in sys.argv[1:]:
    for r in csv.DictReader(open(path)):
        k = r['Kernel_Name'].replace('(anonymous namespace)::', '')[:70]
        agg[k][r['Counter_Name']] += float(r['Counter_Value'])
        disp[k].add(r['Dispatch_Id'])
for k, d in agg.items():
    print(k, 'dispatches', len(disp[k]))
    for c, v in sorted(d.items()):
        print('   %-28s %.6g' % (c, v))
