"""Summarise a rocprofv3 --pmc counter_collection.csv per conv dispatch (debug aid)."""
import csv, collections, sys, glob
d = sys.argv[1]
cc = glob.glob(d + '/*/*counter_collection.csv')[0]
kt = glob.glob(d + '/*/*kernel_trace.csv')[0]
tr = {r['Dispatch_Id']: r for r in csv.DictReader(open(kt))}
by = collections.OrderedDict()
for r in csv.DictReader(open(cc)):
    if 'conv_igemm' not in r['Kernel_Name']:
        continue
    e = by.setdefault(r['Dispatch_Id'], {'name': r['Kernel_Name'][46:72], 'grid': r['Grid_Size']})
    e[r['Counter_Name']] = e.get(r['Counter_Name'], 0) + float(r['Counter_Value'])
seen = set()
for k, e in by.items():
    key = (e['name'], e['grid'])
    if key in seen:
        continue
    seen.add(key)
    t = tr[k]
    ns = int(t['End_Timestamp']) - int(t['Start_Timestamp'])
    print(e['name'], e['grid'], 'ms %.3f' % (ns / 1e6), ' '.join('%s=%.4g' % (c, v) for c, v in e.items() if c not in ('name', 'grid')))
