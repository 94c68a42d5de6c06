"""The oracle's restatement of the in-path broadening placement (oracle/mft6_oracle.py,
loglikelihood(inpath=...)) against its restatement of the reference's live placement (per node at staging,
mft6.py:366-378): broadening is linear, so the two agree to the order of the sums."""
import numpy as np

import common
from common import golden_case, rel_err


def test_oracle_inpath_placement_equals_the_staging_placement():
    orc = common.orc
    c = golden_case('B')
    raw = c.specs                                               # the golden grid, unbroadened
    w = [float(min(c.data[0])) * 1e4 - 3.0, float(max(c.data[0])) * 1e4 + 3.0]   # the data window, Angstrom
    res = 1700.0
    staged = orc.broaden_specs_window(raw, w, res)
    th = c.theta[:6]
    a = np.array([orc.loglikelihood(list(t), c.fr, 2, c.data, c.err, c.r, staged, c.ctm, c.ptm, c.tmi, c.tma, c.matrix,
                                    bandlib=c.bandlib) for t in th])
    b = np.array([orc.loglikelihood(list(t), c.fr, 2, c.data, c.err, c.r, staged, c.ctm, c.ptm, c.tmi, c.tma, c.matrix,
                                    bandlib=c.bandlib, inpath=dict(specs_raw=raw, w=w, resolution=res)) for t in th])
    plain = np.array([orc.loglikelihood(list(t), c.fr, 2, c.data, c.err, c.r, raw, c.ctm, c.ptm, c.tmi, c.tma, c.matrix,
                                        bandlib=c.bandlib) for t in th])
    assert np.all(np.isfinite(a)) and rel_err(b, a).max() < 1e-10
    assert rel_err(plain, a).max() > 1e-6                        # (the broadening does something on this grid)
