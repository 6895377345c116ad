"""Turn rocprofv3 outputs (CSV) into the summaries committed under profiles/.

  python scripts/make_profiles.py stats <dir with *kernel_trace.csv> <out.csv>
  python scripts/make_profiles.py steady <dir with *kernel_trace.csv> <out.csv>   (first batch dropped)
  python scripts/make_profiles.py pmc <fetch dir> <write dir> <out.md> [<commit>]
      (also rewrites profiles/hbm_traffic_latest.json, which bench.py reads for roofline.traffic)
  python scripts/make_profiles.py pmc3 <prof dir> <out.md> <commit> <leg> [<leg> ...]
      (round 3: <prof dir>/<leg>_fetch, _write, _sq summaries of scripts/profile_r03.sh; one traffic
      and one SQ table per leg; hbm_traffic_latest.json = union over the legs)

PMC units follow MI355X_MICROARCH.md (HBM section): FETCH_SIZE / WRITE_SIZE count KiB; on gfx950
FETCH_SIZE tallies 128-byte read requests as 64 bytes, so reads are doubled; WRITE_SIZE is exact."""
import collections
import csv
import glob
import os
import re
import sys


def short(name):
    m = re.search(r'(\w+_kernel)(I[^E]*E)?', name)
    if not m:
        return name[:60]
    tmpl = m.group(2) or ''
    args = re.findall(r'L[ib](\d+)E', tmpl)
    return m.group(1) + ('<' + ','.join(args) + '>' if args else '')


def find(d, pat):
    f = glob.glob(d + '/**/*' + pat, recursive=True)
    if not f:
        raise SystemExit('no %s under %s' % (pat, d))
    # (gpurun merges every call's output into the same scratch tree: take the latest run's file)
    return max(f, key=os.path.getmtime)


def stats(d, out, steady=False, nconf=3):
    """steady=True drops the warm-up batch of the run (every dispatch up to the `nconf`-th
    confusion_kernel = the end of the first batch; a bench step calls val_fn three times since round
    2, twice in round 1) together with the load-time border folding and the weight packing in front
    of it, so what is left is what bench.py's timed steps (and its roofline pass) run."""
    rows = sorted(csv.DictReader(open(find(d, 'kernel_trace.csv'))),
                  key=lambda r: int(r['Start_Timestamp']))
    if steady:
        seen = 0
        for i, r in enumerate(rows):
            if 'confusion_kernel' in r['Kernel_Name']:
                seen += 1
                if seen == nconf:
                    rows = rows[i + 1:]
                    break
    per = collections.OrderedDict()
    for r in rows:
        if 'spin_kernel' in r['Kernel_Name']:      # torch.cuda._sleep: the roofline pass's stream blocker
            continue
        e = per.setdefault(short(r['Kernel_Name']), [0, 0.0])
        e[0] += 1
        e[1] += (int(r['End_Timestamp']) - int(r['Start_Timestamp'])) / 1e6
    tot = sum(v[1] for v in per.values())
    with open(out, 'w') as f:
        f.write('kernel,calls,total_ms,avg_ms,percent\n')
        for k, (n, ms) in sorted(per.items(), key=lambda kv: -kv[1][1]):
            f.write('%s,%d,%.3f,%.4f,%.2f\n' % (k, n, ms, ms / n, 100 * ms / tot))
        f.write('TOTAL,%d,%.3f,,100\n' % (sum(v[0] for v in per.values()), tot))
    print(open(out).read())


def pmc(fd, wd, out, commit=None):
    def load(d, counter):
        per = collections.OrderedDict()
        seen = collections.defaultdict(set)
        for r in csv.DictReader(open(find(d, 'counter_collection.csv'))):
            if r['Counter_Name'] != counter:
                continue
            k = short(r['Kernel_Name'])
            e = per.setdefault(k, [0, 0.0])
            if r['Dispatch_Id'] not in seen[k]:
                seen[k].add(r['Dispatch_Id'])
                e[0] += 1
            e[1] += float(r['Counter_Value'])
        return per
    def load_summary(d, counter):
        """scripts/profile_r02.sh reduces the per-dispatch CSV on the GPU box to one row per
        (kernel, counter): Kernel_Name, Counter_Name, dispatches, Counter_Sum."""
        per = collections.OrderedDict()
        for r in csv.DictReader(open(d + '/summary.csv')):
            if r['Counter_Name'] != counter:
                continue
            e = per.setdefault(short(r['Kernel_Name']), [0, 0.0])
            e[0] += int(r['dispatches'])
            e[1] += float(r['Counter_Sum'])
        return per
    import os
    if os.path.exists(fd + '/summary.csv'):
        load = load_summary
    fe, wr = load(fd, 'FETCH_SIZE'), load(wd, 'WRITE_SIZE')
    rows = []
    for k in fe:
        n = fe[k][0]
        rd = 2 * fe[k][1] * 1024 / n / 1e9
        w = wr.get(k, [n, 0.0])
        rows.append((k, n, fe[k][1], rd, w[1], w[1] * 1024 / max(w[0], 1) / 1e9))
    rows.sort(key=lambda r: -(r[3] + r[5]) * r[1])
    with open(out, 'w') as f:
        f.write('| kernel | launches | FETCH_SIZE sum (KiB) | read GB/launch (x2) | WRITE_SIZE sum (KiB) '
                '| write GB/launch | total GB/launch |\n|---|---:|---:|---:|---:|---:|---:|\n')
        for k, n, fs, rd, ws, wg in rows:
            f.write('| %s | %d | %.4g | %.3f | %.4g | %.3f | %.3f |\n' % (k, n, fs, rd, ws, wg, rd + wg))
    print(open(out).read())
    import json
    import os
    latest = os.path.join(os.path.dirname(os.path.abspath(out)), 'hbm_traffic_latest.json')
    json.dump({'commit': commit, 'source': os.path.basename(out),
               'method': 'rocprofv3 --pmc FETCH_SIZE and --pmc WRITE_SIZE in separate passes; '
                         'reads doubled (gfx950 tallies 128-B requests at 64 B), KiB units',
               'gb_per_launch': {re.sub(r'<.*', '', k): round(rd + wg, 4)
                                 for k, n, fs, rd, ws, wg in rows}},
              open(latest, 'w'), indent=1)


def pmc3(prof, out, commit, legs):
    import json
    def load(d, counter):
        per = collections.OrderedDict()
        for r in csv.DictReader(open(d + '/summary.csv')):
            if r['Counter_Name'] != counter:
                continue
            e = per.setdefault(short(r['Kernel_Name']), [0, 0.0])
            e[0] += int(r['dispatches'])
            e[1] += float(r['Counter_Sum'])
        return per
    union = {}
    with open(out, 'w') as f:
        f.write('# ' + os.path.basename(out)[:3] + ' -- rocprofv3 PMC passes (scripts/profile_' + os.path.basename(out)[:3] + '.sh; one bench batch + load-time work '
                'per leg)\n\nEvery counter set in its own pass with --kernel-trace only, within the gfx950 '
                'slots (SQ 8, TCC 4: FETCH_SIZE 3, WRITE_SIZE 2; GRBM 2).  FETCH_SIZE / WRITE_SIZE in KiB; '
                'reads doubled (gfx950 tallies 128-byte requests at 64 bytes -- calibrated for 16-byte-per-lane '
                'streaming reads, which is what the LDS-DMA staging of the conv kernels issues; the 8-byte '
                'stores of conv_c8_kernel are outside the calibrated widths: ratios only).\n')
        for leg in legs:
            fe, wr = load('%s/%s_fetch' % (prof, leg), 'FETCH_SIZE'), load('%s/%s_write' % (prof, leg), 'WRITE_SIZE')
            rows = []
            for k in fe:
                n = fe[k][0]
                rd = 2 * fe[k][1] * 1024 / n / 1e9
                w = wr.get(k, [n, 0.0])
                rows.append((k, n, rd, w[1] * 1024 / max(w[0], 1) / 1e9))
            rows.sort(key=lambda r: -(r[2] + r[3]) * r[1])
            f.write('\n## leg %s: HBM-side traffic per launch\n\n| kernel | launches | read GB/launch (x2) | '
                    'write GB/launch | total GB/launch |\n|---|---:|---:|---:|---:|\n' % leg)
            for k, n, rd, wg in rows[:14]:
                f.write('| %s | %d | %.3f | %.3f | %.3f |\n' % (k, n, rd, wg, rd + wg))
                # (the X3 instantiations of conv_c8_kernel run only in the bf16x3 leg: own key)
                key = re.sub(r'<.*', '', k)
                union.setdefault(key + '<x3>' if leg == 'bf16x3' and key == 'conv_c8_kernel' else key,
                                 round(rd + wg, 4))
            sq = {c: load('%s/%s_sq' % (prof, leg), c) for c in
                  ('SQ_VALU_MFMA_BUSY_CYCLES', 'SQ_WAVE_CYCLES', 'SQ_WAIT_ANY', 'SQ_WAIT_INST_ANY',
                   'SQ_ACTIVE_INST_ANY', 'GRBM_GUI_ACTIVE')}
            f.write('\n## leg %s: matrix pipe and wave states\n\nmatrix pipe busy = SQ_VALU_MFMA_BUSY_CYCLES / '
                    '(GRBM_GUI_ACTIVE / 8 XCDs x 1024 SIMDs); wave states as fractions of SQ_WAVE_CYCLES '
                    '(parked on s_waitcnt / barrier, issue stall, issuing).\n\n| kernel | launches | matrix pipe '
                    'busy | parked | issue stall | issuing |\n|---|---:|---:|---:|---:|---:|\n' % leg)
            ks = sorted(sq['GRBM_GUI_ACTIVE'], key=lambda k: -sq['GRBM_GUI_ACTIVE'][k][1])
            for k in ks[:10]:
                gui = sq['GRBM_GUI_ACTIVE'][k][1] / 8.0
                wc = max(sq['SQ_WAVE_CYCLES'].get(k, [0, 0.0])[1], 1.0)
                g = lambda c: sq[c].get(k, [0, 0.0])[1]
                f.write('| %s | %d | %.3f | %.3f | %.3f | %.3f |\n'
                        % (k, sq['GRBM_GUI_ACTIVE'][k][0], g('SQ_VALU_MFMA_BUSY_CYCLES') / (gui * 1024),
                           g('SQ_WAIT_ANY') / wc, g('SQ_WAIT_INST_ANY') / wc, g('SQ_ACTIVE_INST_ANY') / wc))
    print(open(out).read())
    latest = os.path.join(os.path.dirname(os.path.abspath(out)), 'hbm_traffic_latest.json')
    json.dump({'commit': commit, 'source': os.path.basename(out),
               'method': 'rocprofv3 --pmc FETCH_SIZE and --pmc WRITE_SIZE in separate passes per leg '
                         '(%s); reads doubled (gfx950 tallies 128-B requests at 64 B), KiB units' % ', '.join(legs),
               'gb_per_launch': union}, open(latest, 'w'), indent=1)


if __name__ == '__main__':
    if sys.argv[1] == 'pmc3':
        pmc3(sys.argv[2], sys.argv[3], sys.argv[4], sys.argv[5:])
    elif sys.argv[1] == 'stats':
        stats(sys.argv[2], sys.argv[3])
    elif sys.argv[1] == 'steady':
        stats(sys.argv[2], sys.argv[3], steady=True, nconf=int(sys.argv[4]) if len(sys.argv) > 4 else 3)
    else:
        pmc(sys.argv[2], sys.argv[3], sys.argv[4], sys.argv[5] if len(sys.argv) > 5 else None)
