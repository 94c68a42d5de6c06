#!/usr/bin/env python3
"""Condense rocprofv3 output into the small, committed summaries under profiles/.

    python tools/pmc_summary.py --kt gpurun_out/prof/kt --fetch gpurun_out/prof/pmc_fetch \
        --write gpurun_out/prof/pmc_write --kernel logprob_kernel --out profiles/r1_logprob

Writes <out>_kernel_stats.csv (the --stats table as is) and <out>_traffic.json with per-launch HBM bytes:
FETCH_SIZE and WRITE_SIZE are reported by rocprofv3 in KiB-like units of 1024 B (MI355X_MICROARCH.md
§HBM / cdna_hip_programming.md §7: hbm_bytes = (FETCH_SIZE + WRITE_SIZE) * 1024) and on gfx950 FETCH_SIZE
counts 64 B per 128-B request for wide coalesced reads, so the read side is doubled.
"""
import argparse
import csv
import glob
import json
import os
import shutil


def find(d, suffix):
    hits = glob.glob(os.path.join(d, '**', '*' + suffix), recursive=True)
    return hits[0] if hits else None


def counter_mean(d, kernel, counter):
    f = find(d, '_counter_collection.csv')
    vals = []
    with open(f) as fh:
        for row in csv.DictReader(fh):
            if kernel in row.get('Kernel_Name', '') and row.get('Counter_Name') == counter:
                vals.append(float(row['Counter_Value']))
    return (sum(vals) / len(vals), len(vals)) if vals else (None, 0)


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument('--kt')
    ap.add_argument('--fetch')
    ap.add_argument('--write')
    ap.add_argument('--kernel', default='logprob_kernel')
    ap.add_argument('--out', required=True)
    ap.add_argument('--note', default='')
    ap.add_argument('--walkers', type=int, default=256)
    ap.add_argument('--npix', type=int, default=4096)
    ap.add_argument('--phot', type=int, default=0)
    a = ap.parse_args()
    os.makedirs(os.path.dirname(a.out), exist_ok=True)
    summary = {'kernel': a.kernel, 'note': a.note,
               'config': {'walkers': a.walkers, 'npix': a.npix, 'phot': bool(a.phot)}}  # bench.py matches on this
    if a.kt:
        ks = find(a.kt, '_kernel_stats.csv')
        shutil.copy(ks, a.out + '_kernel_stats.csv')
        with open(ks) as fh:
            for row in csv.DictReader(fh):
                if a.kernel in row['Name']:
                    summary['calls'] = int(row['Calls'])
                    summary['avg_ns'] = float(row['AverageNs'])
                    summary['min_ns'] = float(row['MinNs'])
                    summary['max_ns'] = float(row['MaxNs'])
    if a.fetch:
        v, n = counter_mean(a.fetch, a.kernel, 'FETCH_SIZE')
        summary['FETCH_SIZE_raw_mean'] = v
        summary['fetch_dispatches'] = n
    if a.write:
        v, n = counter_mean(a.write, a.kernel, 'WRITE_SIZE')
        summary['WRITE_SIZE_raw_mean'] = v
    if summary.get('FETCH_SIZE_raw_mean') is not None:
        rd = summary['FETCH_SIZE_raw_mean'] * 1024 * 2  # gfx950: wide coalesced reads are counted at half
        wr = (summary.get('WRITE_SIZE_raw_mean') or 0.0) * 1024
        summary['hbm_bytes_per_launch'] = rd + wr
        summary['hbm_read_bytes_per_launch_x2_corrected'] = rd
        summary['hbm_write_bytes_per_launch'] = wr
    with open(a.out + '_traffic.json', 'w') as fh:
        json.dump(summary, fh, indent=1)
    print(json.dumps(summary))


if __name__ == '__main__':
    main()
