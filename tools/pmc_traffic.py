"""HBM traffic of the two persistent launches from two separate rocprofv3 PMC passes (FETCH_SIZE, WRITE_SIZE; `--output-format csv`), corrected as
MI355X_MICROARCH.md prescribes for gfx950 (FETCH_SIZE counts the 128-B requests of wide coalesced reads at 64 B: doubled; WRITE_SIZE is exact for
16-B-per-lane stores).  Writes profiles/rNN_pmc_resident_traffic.json, which bench.py reports as `roofline.traffic`.
usage: pmc_traffic.py <fetch counter_collection.csv> <write counter_collection.csv> <out.json> <round tag> [commit]"""
import csv, json, sys
fetch_csv, write_csv, out, tag = sys.argv[1:5]
commit = sys.argv[5] if len(sys.argv) > 5 else '?'


def mean_kb(path, kernel, counter):
    vals = [float(r['Counter_Value']) for r in csv.DictReader(open(path)) if kernel in r['Kernel_Name'] and r['Counter_Name'] == counter and int(r['Grid_Size']) == 256 * 1024]
    vals = vals[-150:]          # the re-launches of bench.py's dominant_kernel_roofline (512 samples = 256 blocks of 1024 threads)
    return (sum(vals) / len(vals), len(vals)) if vals else (0.0, 0)


rec = {'round': tag, 'commit': commit,
       'command': 'rocprofv3 --kernel-trace --pmc FETCH_SIZE --output-format csv -- python3 bench.py --steps 5 --warmup 2 --no-cpu-baseline --no-secondary --no-parity ; same with --pmc WRITE_SIZE (separate passes)',
       'gfx950_fetch_correction': 2.0}
for kernel in ('resident_fwd_kernel', 'resident_bwd_kernel'):
    f, nf = mean_kb(fetch_csv, kernel, 'FETCH_SIZE')
    w, nw = mean_kb(write_csv, kernel, 'WRITE_SIZE')
    rec[kernel] = {'dispatches_averaged': [nf, nw], 'FETCH_SIZE_KB_per_dispatch': round(f, 2), 'WRITE_SIZE_KB_per_dispatch': round(w, 2),
                   'read_bytes_per_launch': int(f * 1024 * 2), 'write_bytes_per_launch': int(w * 1024), 'traffic_bytes_per_launch': int(f * 1024 * 2 + w * 1024)}
json.dump(rec, open(out, 'w'), indent=1)
print(json.dumps(rec))
