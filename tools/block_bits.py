import sys, numpy as np, torch
sys.path.insert(0, '.')
from bench import build_workload
from mcmc_spec_amd import _lib, synth
from mcmc_spec_amd.engine import Engine
eng = Engine(0); W = build_workload(eng, 4096, False)
dev = torch.device('cuda', 0)
th = torch.from_numpy(synth.draw_walkers(256, seed=3, tmin=W['tmin'], tmax=W['tmax'])).to(dev)
out = {}
for B in (0, 256, 512, 1024):
    lp = torch.empty(256, dtype=torch.float64, device=dev); st = torch.empty(256, dtype=torch.int32, device=dev)
    eng.ctx.logprob_batch_dev(th.data_ptr(), 256, 6, lp.data_ptr(), st.data_ptr(), torch.cuda.current_stream(dev).cuda_stream, _lib.MODE_LOGPOST, B)
    torch.cuda.synchronize(); out[B] = lp.cpu().numpy()
for B in (256, 512, 1024):
    d = out[B] != out[0]
    rel = np.abs(out[B] - out[0]) / np.abs(out[0])
    print(B, 'differing', int(d.sum()), 'max rel', float(np.nanmax(rel)))
print('finite', int(np.isfinite(out[0]).sum()), out[0][:4], out[1024][:4])
