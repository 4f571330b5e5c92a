# How often would a wave-uniform "rare block" for the Germ_Max 'better than the best later record' test (EE:1266) fire?
# For every 64-position tile and every sample row of a wave's chunk: is there a lane (x any nucleotide) whose record strictly beats
# the best later record so far?  config-3-like synthetic normals (tests.helpers.synth_recs), chunks of 64 samples per wave.
import sys, numpy as np
sys.path.insert(0, "/root/repo")
from tests.helpers import synth_recs
P, S, CH = 6400, 256, 64
r = synth_recs(P, S).astype(np.int64)              # [S][P][8]
absent = r[:, :, 0] == np.iinfo(np.int32).min
fw, bw = r[:, :, :4].copy(), r[:, :, 4:]
fw[absent] = 0
FW, BW = fw.sum(-1), bw.sum(-1)
RD = FW + BW
cov = (~absent) & (FW >= 100) & (BW >= 100)
x = fw + bw                                         # [S][P][4]
lim = (RD * 26843545) >> 29
cand = cov[:, :, None] & (x <= lim[:, :, None])
rows_any = rows = 0
events = 0
first_rows = 0
for c0 in range(0, S, CH):
    have_first = np.zeros((P, 4), bool)
    bx = np.zeros((P, 4), np.int64); bd = np.ones((P, 4), np.int64)
    for s in range(c0, c0 + CH):
        c = cand[s]
        is_first = c & ~have_first
        later = c & have_first
        better = later & (x[s] * bd > bx * RD[s][:, None])
        bx = np.where(better, x[s], bx); bd = np.where(better, RD[s][:, None], bd)
        have_first |= c
        t = better.reshape(P // 64, 64 * 4).any(1)
        rows_any += t.sum(); rows += t.size; events += better.sum()
        first_rows += is_first.reshape(P // 64, 64 * 4).any(1).sum()
print(f"rows (tile x sample): {rows}; rows where some lane's best later record improves: {rows_any} = {rows_any / rows:.3f}; "
      f"improvements per row: {events / rows:.2f}; rows where some lane meets its FIRST record: {first_rows / rows:.3f}")
