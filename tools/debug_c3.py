import os, sys
import numpy as np, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from mused_amd import synth
from mused_amd.swfd import SeqBasedSWFD
from oracle.swfd_oracle import SeqBasedSWFD as Ora
N, ell, d = [int(v) for v in os.environ.get('CFG','700,256,320').split(',')]
X, _ = synth.make_stream("blob", N + 300, d, 11)
R = float((X.astype(np.float64) ** 2).sum(1).max())
dev, ora = SeqBasedSWFD(N=N, R=R, d=d, sketch_dim=ell, sweeps=int(os.environ.get('SW','0'))), Ora(N=N, R=R, d=d, sketch_dim=ell)
for lo, hi in [(0, 300), (300, 700), (700, 1000)]:
    dev.fit(X[lo:hi]); ora.fit(X[lo:hi])
    Bd, sd, ld, dd = dev.get(); Bo, so, lo_, do = ora.get()
    big = so > 1e-3 * so[0]
    print(hi, "level", ld, lo_, "max abs/s0 %.2e" % (np.abs(sd - so).max() / so[0]), "max rel(big) %.2e" % (np.abs(sd[big] - so[big]) / so[big]).max(), "nbig", big.sum(), "min big/s0 %.1e" % (so[big].min() / so[0]))
