#!/usr/bin/env python3
"""Kernel durations and the gaps between consecutive kernels of a rocprofv3 --kernel-trace run of tools/chain_bench.py.

    cd /tmp && rocprofv3 --kernel-trace --output-format csv -d $R/gpurun_out/chain_kt -o kt -- python3 $R/tools/chain_bench.py --no-host --walkers 256 --device-steps 1000
    python3 tools/chain_gaps.py gpurun_out/chain_kt
"""
import csv
import glob
import sys

import numpy as np


def main():
    f = glob.glob(sys.argv[1] + '/**/*kernel_trace.csv', recursive=True)[0]
    rows = [r for r in csv.DictReader(open(f)) if 'logprob_kernel' in r['Kernel_Name']]
    rows.sort(key=lambda r: int(r['Start_Timestamp']))
    st = np.array([int(r['Start_Timestamp']) for r in rows], dtype=np.int64)
    en = np.array([int(r['End_Timestamp']) for r in rows], dtype=np.int64)
    dur = en - st
    gap = st[1:] - en[:-1]
    ok = gap < 20000   # (chunk boundaries, the warm-up's end: not the steady state)
    tail = slice(len(dur) // 2, None)
    big = gap[~ok]
    print('span {:.1f} us for {} kernels = {:.2f} us per kernel; {} gaps >= 20 us: total {:.1f} us, median {:.1f} us'.format(
        (en[-1] - st[0]) / 1e3, len(rows), (en[-1] - st[0]) / 1e3 / len(rows), len(big), big.sum() / 1e3, np.median(big) / 1e3 if len(big) else 0.0))
    print('kernels {}: duration median {:.2f} us (p10 {:.2f}, p90 {:.2f}); gap end -> next start median {:.2f} us (p10 {:.2f}, p90 {:.2f}); '
          'start -> start median {:.2f} us'.format(len(rows), np.median(dur[tail]) / 1e3, np.percentile(dur[tail], 10) / 1e3,
                                                   np.percentile(dur[tail], 90) / 1e3, np.median(gap[ok]) / 1e3, np.percentile(gap[ok], 10) / 1e3,
                                                   np.percentile(gap[ok], 90) / 1e3, np.median(np.diff(st)[ok]) / 1e3))


if __name__ == '__main__':
    main()
