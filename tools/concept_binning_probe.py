"""What can be recovered of CONCEPT's power-spectrum binning from the reference's tables (VERDICT r1 "missing" 5)?  The reference's parameter
files leave `powerspec_options['bins per decade']` at its default (TheFeedingPipeline/NN.param:28-45), which CONCEPT documents as a k-dependent
bin density, 4 bins per decade at 4 k_min rising to 40 at 100 k_min. From the 16 mode counts of PPPMP3M/PP/powerspec/powerspec_a=1.00 (64^3 grid, 500 Mpc):
  * every half-space mode with 0 < |k| <= k_Nyquist is binned (sum of the counts = 68 529 = the modes inside the Nyquist sphere);
  * the bin edges are bracketed by consecutive |n| shells; the bracket midpoints give the local bin density 1 / dlog10(k);
  * bins CENTRED on k_min with 4 bins per decade below 4 k_min reproduce the first three counts (3, 25, 124) exactly, and a density linear in log k through
    (4 k_min, 4) and (100 k_min, 40) reproduces the widths of bins 7-16 (12.4 ... 26.6 per decade) to 3 %, but bins 4-6 (418, 730, 1012 modes) are
    20-40 % wider than any smooth interpolation gives — the exact rule (mode-count or shell-based adjustment) is not recoverable without CONCEPT's source.
Consequence: tables are reproduced through the reference's own mode counts (`cosmo.rebin_like_concept`), which is exact for any table the reference ships."""
import os, sys
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
t = np.loadtxt(os.path.join(ROOT, "tests", "golden", "concept", "PP_powerspec_a=1.00"))
modes = t[:, 1].astype(int)
N, L = 64, 500.0
kf = 2 * np.pi / L
n = np.fft.fftfreq(N, d=1.0 / N)
k = np.sqrt(n[:, None, None] ** 2 + n[None, :, None] ** 2 + n[None, None, :] ** 2).ravel()
ks = np.sort(k[(k > 0) & (k <= N / 2)])
print("modes in the table:", modes.sum(), " half-space modes inside the Nyquist sphere:", ks.size // 2)
cum = np.cumsum(2 * modes)
lo, hi = ks[np.minimum(cum - 1, ks.size - 1)], ks[np.minimum(cum, ks.size - 1)]
mid = np.sqrt(lo * hi)
print("upper bin edges bracketed by |n| shells:", np.round(np.column_stack([lo, hi]), 3).tolist())
w = np.diff(np.log10(np.concatenate([[mid[0] / 10 ** 0.25], mid])))
model = np.interp(np.log10(np.sqrt(np.concatenate([[mid[0] / 10 ** 0.25], mid[:-1]]) * mid)), np.log10([4.0, 100.0]), [4.0, 40.0])
print("bins per decade implied by the table :", np.round(1 / w, 1).tolist())
print("documented default, linear in log k  :", np.round(model, 1).tolist())
