"""Aggregate a rocprofv3 --pmc counter_collection CSV per kernel: launches, mean and total of each counter.
Usage: python tools/pmc_summary.py <dir-or-csv> [out.json]   (run on the GPU box right after the rocprofv3 pass)."""
import csv, glob, json, os, sys
from collections import defaultdict


def main():
    src = sys.argv[1]
    files = [src] if src.endswith('.csv') else glob.glob(os.path.join(src, '**', '*counter_collection.csv'), recursive=True)
    agg = defaultdict(lambda: defaultdict(lambda: [0, 0.0]))
    for f in files:
        for r in csv.DictReader(open(f)):
            name = r.get('Kernel_Name') or r.get('Kernel Name') or r.get('Name')
            cn, cv = r.get('Counter_Name'), r.get('Counter_Value')
            if name is None or cn is None:
                continue
            a = agg[name.split('(')[0][:120]][cn]
            a[0] += 1; a[1] += float(cv)
    out = {k: {c: {'launches': n, 'mean': t / n, 'total': t} for c, (n, t) in v.items()} for k, v in agg.items()}
    # kernel durations, when the pass also ran with --kernel-trace (effective clock = GRBM_GUI_ACTIVE per XCD / duration)
    tfiles = [] if src.endswith('.csv') else glob.glob(os.path.join(src, '**', '*kernel_trace.csv'), recursive=True)
    dur = defaultdict(lambda: [0, 0.0])
    for f in tfiles:
        for r in csv.DictReader(open(f)):
            name = (r.get('Kernel_Name') or '').split('(')[0][:120]
            d = dur[name]; d[0] += 1; d[1] += float(r['End_Timestamp']) - float(r['Start_Timestamp'])
    for k, (n, t) in dur.items():
        if k in out:
            out[k]['duration_ns'] = {'launches': n, 'mean': t / n, 'total': t}
    top = dict(sorted(out.items(), key=lambda kv: -max(x['total'] for x in kv[1].values()))[:40])
    js = json.dumps(top, indent=1)
    if len(sys.argv) > 2:
        open(sys.argv[2], 'w').write(js)
    print(js[:3000])


if __name__ == '__main__':
    main()
