import sys, numpy as np, torch
sys.path.insert(0, '.')
from amplisolve_amd import Context
from oracle import pyoracle as orc
from tests.helpers import synth_recs
ctx = Context(0)
for (P,S,splits) in [(64,4,1),(63,4,1),(64,3,1),(64,8,1),(64,8,2),(128,4,1),(64,5,1)]:
    recs = synth_recs(P,S)
    ctx.set_tuning(splits)
    acc = ctx.error_reduce(torch.from_numpy(recs).cuda(), P)
    got = acc.snt.cpu().numpy(); nrec = acc.nrec.cpu().numpy()
    ref = orc.error_reduce(recs,P)
    ok = np.array_equal(got, ref["snt"])
    print(P,S,splits,"ok" if ok else "MISMATCH", "nrec ok", np.array_equal(nrec, ref["nrec"]))
    if not ok:
        per = [orc.error_reduce(recs[s:s+1],P)["snt"] for s in range(S)]
        bad = np.argwhere(got != ref["snt"])
        print(" n bad", len(bad), "first", bad[:5].tolist())
        st,nt,p = bad[0]
        print(" got", got[st,nt,p], "exp", ref["snt"][st,nt,p], "per-sample", [x[st,nt,p] for x in per])
        print(" got row", got[0,0,:8], "\n exp row", ref["snt"][0,0,:8])
        print(" nrec got", nrec[:8], "exp", ref["nrec"][:8])
