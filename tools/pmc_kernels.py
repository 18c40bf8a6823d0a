"""Per-kernel HBM traffic and matrix-pipe utilisation of a command from separate rocprofv3 PMC passes (csv): for every kernel whose name contains one of the
given patterns: dispatches, average duration (kernel trace of the FETCH pass), FETCH_SIZE x 2 + WRITE_SIZE bytes per dispatch (gfx950 correction of
MI355X_MICROARCH.md: FETCH_SIZE counts wide coalesced reads at half their bytes) -> achieved HBM GB/s and fraction of 8 TB/s, and — when the MFMA pass is
given — SQ_VALU_MFMA_BUSY_CYCLES / (1024 SIMDs x SQ_BUSY_CYCLES / 32 shader engines): the fraction of SIMD cycles with the matrix pipe busy.
usage: pmc_kernels.py <out.json> <tag> <commit> <command description> <fetch dir> <write dir> <mfma dir or -> <pattern> [pattern ...]"""
import csv, glob, json, sys
out, tag, commit, what, fdir, wdir, mdir = sys.argv[1:8]
patterns = sys.argv[8:]


def load(d, suffix):
    f = glob.glob(f'{d}/**/*{suffix}', recursive=True)
    return list(csv.DictReader(open(f[0]))) if f else []


def per_kernel(rows, counter):
    acc = {}
    for r in rows:
        if r.get('Counter_Name') != counter:
            continue
        for p in patterns:
            if p in r['Kernel_Name']:
                acc.setdefault(p, []).append(float(r['Counter_Value']))
    return acc


fetch, write = load(fdir, 'counter_collection.csv'), load(wdir, 'counter_collection.csv')
trace = load(fdir, 'kernel_trace.csv')
dur = {}
for r in trace:
    for p in patterns:
        if p in r['Kernel_Name']:
            dur.setdefault(p, []).append((int(r['End_Timestamp']) - int(r['Start_Timestamp'])) / 1e3)
f, w = per_kernel(fetch, 'FETCH_SIZE'), per_kernel(write, 'WRITE_SIZE')
rec = {'round': tag, 'commit': commit, 'command': what, 'gfx950_fetch_correction': 2.0, 'hbm_peak_GBps': 8000.0, 'kernels': {}}
mf = {}
if mdir != '-':
    m = load(mdir, 'counter_collection.csv')
    busy, sq = per_kernel(m, 'SQ_VALU_MFMA_BUSY_CYCLES'), per_kernel(m, 'SQ_BUSY_CYCLES')
    for p in patterns:
        if p in busy and p in sq and sq[p] and sum(sq[p]) > 0:
            # SQ_VALU_MFMA_BUSY_CYCLES sums over the chip's 1024 SIMDs, SQ_BUSY_CYCLES over its 32 shader engines (profiles/r02_pmc_mfma_utilisation.json)
            mf[p] = (sum(busy[p]) / len(busy[p])) / (1024.0 * (sum(sq[p]) / len(sq[p])) / 32.0)
for p in patterns:
    if p not in f or p not in w or p not in dur:
        continue
    fb, wb, us = sum(f[p]) / len(f[p]) * 1024 * 2, sum(w[p]) / len(w[p]) * 1024, sum(dur[p]) / len(dur[p])
    rec['kernels'][p] = {'dispatches': len(dur[p]), 'avg_duration_us': round(us, 2), 'read_bytes_per_dispatch': int(fb), 'write_bytes_per_dispatch': int(wb),
                         'hbm_GBps': round((fb + wb) / us / 1e3, 1), 'hbm_frac_of_8TBps': round((fb + wb) / us / 1e3 / 8000.0, 4)}
    if p in mf:
        rec['kernels'][p]['mfma_busy_frac_of_simd_cycles'] = round(mf[p], 4)
json.dump(rec, open(out, 'w'), indent=1)
print(json.dumps(rec, indent=1))
