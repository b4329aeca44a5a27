"""hyphy_amd/fel.py on the device next to the reference's own FEL.bf table (tests/golden/ref_fel_12x60.npz, written by
`python -m oracle.make_golden fel`): prints the per-site comparison the tolerances of
tests/test_gpu_parity.py::test_fel_driver_matches_the_reference_fel were set from.  Usage (GPU box): python tests/fel_reference_check.py"""
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np

from hyphy_amd import fel, hip, models
from tests import common


def templates(rev, pf):
    T = np.zeros((2, 61, 61))
    rv = dict(AC=rev[0], AT=rev[1], CG=rev[2], CT=rev[3], GT=rev[4], AG=1.0)
    for (i, j, name, ns, f) in models.mg94rev_template(pf):
        T[1 if ns else 0, i, j] = rv[name] * f
    return T


def run(max_iter=400, name="ref_fel_12x60"):
    fx = common.load(name)
    S = fx["leaf_codes"].shape[1]
    with hip.HipPartition(61, fx["flat_parents"], int(fx["L"]), fx["leaf_codes"], None, np.ones(S, dtype=np.int64)) as part:
        part.set_q_templates(templates(fx["rev"], fx["pos_freqs"]))
        ts = fx["syn_rate"]
        B = len(ts)
        nodes = np.arange(B, dtype=np.int64)
        om = np.where(fx["tested"], float(fx["omega_test"]), float(fx["omega_background"]))
        glob = part.prepare_built_step(nodes, nodes, fx["root_freqs"], np.ascontiguousarray(np.stack([ts, ts * om], axis=1)))()
        glob = float(glob)   # (patterns = sites with frequency 1: the same sum as the reference's compressed one)
        res = fel.fel(part, fx["tested"], ts, ts, fx["root_freqs"], max_iter=max_iter)
    return fx, glob, res


def run_meme(max_iter=400):
    fx = common.load("ref_meme_12x60")
    S = fx["leaf_codes"].shape[1]
    with hip.HipPartition(61, fx["flat_parents"], int(fx["L"]), fx["leaf_codes"], None, np.ones(S, dtype=np.int64)) as part:
        part.set_q_templates(templates(fx["rev"], fx["pos_freqs"]))
        bl = fx["branch_length"]   # (MEME.bf scales every site rate by the branch's total length MLE, MEME.bf:837-845)
        res = fel.meme(part, fx["tested"], bl, bl, fx["root_freqs"], max_iter=max_iter)
    return fx, res


if __name__ == "__main__" and len(sys.argv) > 1 and sys.argv[1] == "meme":
    fx, res = run_meme()
    ref = fx["fel_table"]   # alpha, beta-, p-, beta+, p+, LRT, p-value, MEME LogL, FEL LogL
    print("launches", res.launches)
    print(" site | alpha ref/dev | beta- ref/dev | beta+ ref/dev | q ref/dev | LRT ref/dev | p ref/dev | site logL ref/dev")
    for s in range(ref.shape[0]):
        print(f"{s:4d} | {ref[s,0]:7.3f} {res.alpha[s]:7.3f} | {ref[s,1]:6.3f} {res.beta_minus[s]:6.3f} | {ref[s,3]:7.3f} {res.beta_plus[s]:7.3f} | {ref[s,2]:5.3f} {res.weight_minus[s]:5.3f} |"
              f" {ref[s,5]:7.3f} {res.lrt[s]:7.3f} | {ref[s,6]:6.4f} {res.p_value[s]:6.4f} | {ref[s,7]:9.4f} {res.logl_alt[s]:9.4f}")
    fitted = ref[:, 7] != 0
    print("device alt logL - reference MEME LogL: min", (res.logl_alt - ref[:, 7])[fitted].min(), "max", (res.logl_alt - ref[:, 7])[fitted].max())
    print("max |LRT diff|", np.abs(ref[:, 5] - res.lrt).max(), " max |p diff|", np.abs(ref[:, 6] - res.p_value).max())
    sys.exit(0)

if __name__ == "__main__":
    fx, glob, res = run(name=sys.argv[1] if len(sys.argv) > 1 else "ref_fel_12x60")
    ref = fx["fel_table"]
    print("global logL: device", glob, "reference", float(fx["global_logl"]))
    print("launches", res.launches)
    print(" site |  alpha ref / dev   |   beta ref / dev   |  LRT ref / dev   |   p ref / dev")
    for s in range(ref.shape[0]):
        print(f"{s:5d} | {ref[s,0]:8.4f} {res.alpha[s]:8.4f} | {ref[s,1]:8.4f} {res.beta[s]:8.4f} | {ref[s,3]:8.4f} {res.lrt[s]:8.4f} | {ref[s,4]:7.4f} {res.p_value[s]:7.4f}")
    lrt_ref = np.maximum(ref[:, 3], 0.0)
    print("max |LRT diff|", np.abs(lrt_ref - res.lrt).max(), " max |p diff|", np.abs(ref[:, 4] - res.p_value).max())
    print("sites where device LRT < reference LRT - 1e-3:", np.where(res.lrt < lrt_ref - 1e-3)[0])
    print("sites where device LRT > reference LRT + 1e-3:", np.where(res.lrt > lrt_ref + 1e-3)[0])
