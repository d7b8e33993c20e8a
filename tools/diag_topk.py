import sys, torch
sys.path.insert(0, ".")
from tests.helpers import Golden, GOLDEN
from tests.test_hip_parity import run_hip
g = Golden(GOLDEN + "/worst_topk_regclsvar_s101.npz")
hp, det = run_hip(g)
n = int(hp.n_total.item()); counts = hp.sel_count.cpu().tolist()
idx = hp.cand_anchor_idx[:n].cpu().long(); sc = hp.cand_score[:n].cpu()
ref_sc = g.t("aw0_prob")
off = 0
for lvl, cnt in enumerate(counts):
    ref = g.t("topk_%d" % lvl)[:cnt]
    bad = (idx[off:off+cnt] != ref).nonzero().squeeze(1)
    print("level", lvl, "cnt", cnt, "mismatch positions", bad.tolist()[:20])
    for b in bad.tolist()[:6]:
        print("   pos", b, "hip idx", int(idx[off+b]), "ref idx", int(ref[b]), "hip score %.9g ref score %.9g" % (float(sc[off+b]), float(ref_sc[off+b])))
    off += cnt
d = (sc - ref_sc).abs() / ref_sc
print("score rel diff max %.3g, frac nonzero %.3f" % (float(d.max()), float((d > 0).float().mean())))
