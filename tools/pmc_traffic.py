"""HBM traffic of the dominant kernel from two separate rocprofv3 PMC passes (FETCH_SIZE, WRITE_SIZE; `--output-format csv`), corrected as
MI355X_MICROARCH.md prescribes for gfx950 (FETCH_SIZE counts the 128-B requests of wide coalesced reads at 64 B: doubled; WRITE_SIZE is exact for
16-B-per-lane stores).  Writes profiles/rNN_pmc_conv_tile_traffic.json, which bench.py reports as `roofline.traffic`.
usage: pmc_traffic.py <fetch counter_collection.csv> <write counter_collection.csv> <out.json> <round tag> [commit]"""
import csv, json, sys
fetch_csv, write_csv, out, tag = sys.argv[1:5]
commit = sys.argv[5] if len(sys.argv) > 5 else '?'
KERNEL = 'conv_tile_kernel<4, 5, 2>'


def mean_kb(path, counter):
    vals = [float(r['Counter_Value']) for r in csv.DictReader(open(path)) if KERNEL in r['Kernel_Name'] and r['Counter_Name'] == counter and int(r['Grid_Size']) == 262144]
    vals = vals[-200:]          # the micro-benchmark launches of bench.py's dominant_kernel_roofline (N = 512)
    return sum(vals) / len(vals), len(vals)


f, nf = mean_kb(fetch_csv, 'FETCH_SIZE')
w, nw = mean_kb(write_csv, 'WRITE_SIZE')
algo = 4 * (512 * 32 * 32 * 4 + 4 * 4 * 5 * 5 + 512 * 32 * 32 * 4)
rec = {'round': tag, 'commit': commit,
       'kernel': 'conv_tile_kernel<4,5,2> (512x32x32x4 -> 4, 5x5, pad 2, bias + ReLU + statistics epilogue), launched by bench.py dominant_kernel_roofline through dcv_conv2d_fwd_prepared',
       'command': 'rocprofv3 --kernel-trace --pmc FETCH_SIZE --output-format csv -- python3 bench.py --steps 5 --warmup 2 --no-cpu-baseline --no-secondary --no-parity ; same with --pmc WRITE_SIZE (separate passes)',
       'dispatches_averaged': [nf, nw], 'FETCH_SIZE_KB_per_dispatch': round(f, 2), 'WRITE_SIZE_KB_per_dispatch': round(w, 2), 'gfx950_fetch_correction': 2.0,
       'read_bytes_per_launch': int(f * 1024 * 2), 'write_bytes_per_launch': int(w * 1024), 'traffic_bytes_per_launch': int(f * 1024 * 2 + w * 1024),
       'algorithmic_bytes_per_launch': algo, 'traffic_over_algorithmic': round((f * 1024 * 2 + w * 1024) / algo, 3)}
json.dump(rec, open(out, 'w'), indent=1)
print(json.dumps(rec))
