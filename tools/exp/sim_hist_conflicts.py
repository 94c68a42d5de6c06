"""Simulate the LDS bank-conflict cycles of the median histogram's atomics for one model vector under the lane -> pixel
mappings tried in round 4 (tools/exp: analysis, never shipped).  Conflict model from tools/exp/lds_atomic_probe.hip on gfx950:
an LDS atomic instruction costs 2 x (max lanes per bank - 1) extra cycles, same address or not."""
import sys, time
import numpy as np
sys.path.insert(0, '/root/repo')
from mcmc_spec_amd import synth
from oracle import mft6_oracle as orc
from scipy.interpolate import interp1d
t0 = time.time()
wl = np.arange(3000, 30000, 0.2)
teffs = np.array([3000, 3100, 3800, 3900]); loggs = np.array([4.5, 5.0, 5.5])
flux = synth.make_grid(teffs, loggs, wl)
specs = synth.grid_to_specs(teffs, loggs, wl, flux)
wl_um = synth.data_wavelengths_um(4096)
win = [np.floor(wl_um.min() * 1e4), np.ceil(wl_um.max() * 1e4)]
specs = orc.broaden_specs_window(specs, win, 1700)
print('grid', time.time() - t0)
# a model vector: blend of nodes near (3850, logg 4.9) + (3025, 5.1), reddened
def node(t, g): return specs['{}, {}'.format(t, g)]
mA = 0.5 * (0.6 * node(3800, 5.0) + 0.4 * node(3800, 4.5)) + 0.5 * (0.6 * node(3900, 5.0) + 0.4 * node(3900, 4.5))
mB = 0.75 * (0.8 * node(3000, 5.0) + 0.2 * node(3000, 5.5)) + 0.25 * (0.8 * node(3100, 5.0) + 0.2 * node(3100, 5.5))
comp = mA + 0.09 * mB
comp = orc.extinct(specs['wl'], comp, 0.106)
m = interp1d(specs['wl'], comp)(wl_um * 1e4)
fx = (m.view(np.uint64) >> 44).astype(np.int64)   # hi32 >> 12
bins = fx & 2047
print('distinct bins', len(np.unique(bins)), 'range', bins.min(), bins.max())
def cost(groups):
    """groups: list of arrays of 64 bin indices (one per wave instruction); conflict cycles = 2 * (max lanes per BANK - 1),
    same-address lanes included (probe: p1..p4, p6)"""
    tot = 0
    for g in groups:
        banks = np.bincount(g % 32, minlength=32)
        tot += 2 * (banks.max() - 1)
    return tot
px = np.arange(4096)
# baseline (512 threads, quad trips): instruction = 64 consecutive pixels
base = [bins[i:i + 64] for i in range(0, 4096, 64)]
print('baseline conflict cycles per walker', cost(base))
# parity rotation within an element {p, p+256}: instruction 1: even lanes p, odd lanes p+256
rot = []
for blk in range(0, 4096, 512):
    for w in range(0, 256, 64):
        a = bins[blk + w: blk + w + 64]; b = bins[blk + 256 + w: blk + 256 + w + 64]
        odd = (np.arange(64) & 1).astype(bool)
        rot.append(np.where(odd, b, a)); rot.append(np.where(odd, a, b))
print('parity rotation', cost(rot))
# 4-way rotation over the quad {p, p+256, p+1024, p+1280} by lane & 3
rot4 = []
for half in (0, 2048):
    for blk in (0, 512):
        for w in range(0, 256, 64):
            vals = [bins[half + blk + off + w: half + blk + off + w + 64] for off in (0, 256, 1024, 1280)]
            r = np.arange(64) & 3
            for k in range(4):
                idx = (k + r) & 3
                rot4.append(np.choose(idx, vals))
print('4-way rotation', cost(rot4))
# sum over banks instead of max (if banks were serial)
def cost_sum(groups):
    return sum(2 * (np.bincount(g % 32, minlength=32) - 1).clip(0).sum() for g in groups)
print('sum model: base', cost_sum(base), 'parity', cost_sum(rot), '4way', cost_sum(rot4))
# same-address only model (different addresses in one bank free)
def cost_addr(groups):
    return sum(2 * (np.bincount(g, minlength=2048).max() - 1) for g in groups)
print('same-address model: base', cost_addr(base), 'parity', cost_addr(rot), '4way', cost_addr(rot4))
print('avg run length of equal bins', np.mean([len(list(g)) for g in np.split(bins, np.nonzero(np.diff(bins))[0] + 1)]))
