#!/usr/bin/env python3
"""Precision vs bytes (SURVEY.md §7 'hard parts'): the opt-in 12-byte pair storage against the float64 default,
at BASELINE config 2, for data of S/N 100 (the bench's 1 % noise) and S/N 1000."""
import json
import os
import sys

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))


def main():
    import torch
    from bench import build_workload
    from mcmc_spec_amd import _lib, bands, synth
    from mcmc_spec_amd.engine import Engine
    dev = torch.device('cuda', 0)
    eng = Engine(0)
    W = build_workload(eng, 4096, False)
    th = synth.draw_walkers(2048, seed=21, tmin=W['tmin'], tmax=W['tmax'])
    kw = dict(nspec=2, bands=bands.make_bands(W['tabs'], *W['vega']), av_table=synth.make_av_table(), tmin=W['tmin'],
              tmax=W['tmax'], prior=W['prior'])
    for snr_scale, tag in ((1.0, 'S/N 100'), (0.1, 'S/N 1000')):
        rng = np.random.default_rng(5)
        data = [W['data'][0], W['data'][1]]
        err = W['err'] * snr_scale
        if snr_scale != 1.0:  # same spectrum, ten times less noise
            clean = W['data'][1] - (W['data'][1] - np.median(W['data'][1])) * 0  # keep as is; only the errors shrink
            data = [W['data'][0], clean]
        ref = None
        for compact in (False, True):
            eng.stage_problem(data, err, W['fr'], W['r'], W['ctm'], W['ptm'], W['tmi'], W['tma'], W['matrix'],
                              compact_pairs=compact, **kw)
            lp = eng.logposterior(th)
            tt = torch.from_numpy(th[:256].copy()).to(dev)
            o = torch.empty(256, dtype=torch.float64, device=dev)
            st = torch.empty(256, dtype=torch.int32, device=dev)
            s = torch.cuda.current_stream(dev)
            for _ in range(20):
                eng.ctx.logprob_batch_dev(tt.data_ptr(), 256, 6, o.data_ptr(), st.data_ptr(), s.cuda_stream, _lib.MODE_LOGPOST, 0)
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record(s)
            for _ in range(200):
                eng.ctx.logprob_batch_dev(tt.data_ptr(), 256, 6, o.data_ptr(), st.data_ptr(), s.cuda_stream, _lib.MODE_LOGPOST, 0)
            e1.record(s)
            torch.cuda.synchronize()
            us = e0.elapsed_time(e1) / 200 * 1e3
            if ref is None:
                ref = lp
                rel = 0.0
            else:
                fin = np.isfinite(ref)
                rel = float(np.max(np.abs(lp[fin] - ref[fin]) / np.abs(ref[fin])))
            print(json.dumps(dict(data=tag, compact_pairs=compact, kernel_us_256=us, bytes_per_eval=eng.ctx.bytes_per_eval(),
                                  max_rel_dev_from_f64=rel)), flush=True)


if __name__ == '__main__':
    main()
