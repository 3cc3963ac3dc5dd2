"""ITM loss error of the bf16 engine format under candidate treatments: fp32 pooler + ITM-head weights, fp32 last
sub-layers, error-diffused bf16 rounding of the weights along K.  CPU study on the oracle (tests/study_itm_noise.py); results in
profiles/r04_itm_variants.txt.  Not collected by pytest.  Usage: python tests/study_itm_variants.py"""
import sys, os, json, time, torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import study_itm_noise as S
from oracle import volta_ref as R
torch.set_num_threads(6)
rcfg = R.RefConfig(json.load(open(os.path.join(ROOT, "config", "ctrl_vilbert_base.json"))))
for seed in (3, 4):
    sd = R.make_weights(rcfg, seed=seed, std=0.03)
    b = R.synthetic_batch(rcfg, B=256, T=20, R=36, seed=7)
    with torch.no_grad():
        t0 = time.time()
        ref, rt, rv, ritm = S.forward(sd, rcfg, b, ())
        print("seed", seed, "fp32 ITM loss %.6f (%.1f s)" % (ref, time.time() - t0), flush=True)
        P = S.POINTS
        nopool = tuple(p for p in P if p != "pool")
        variants = [
            ("all bf16 (engine format)", P, None),
            ("pool+itm weights fp32, pooled bf16", P, r"^(?!.*(pooler|bi_seq)).*$"),
            ("pool+itm weights fp32, pooled fp32", nopool, r"^(?!.*(pooler|bi_seq)).*$"),
            ("  + last 2 sublayers (34,35) weights fp32", nopool, r"^(?!.*(pooler|bi_seq|layer\.3[45]\.)).*$"),
            ("  + last 6 sublayers (30-35) weights fp32", nopool, r"^(?!.*(pooler|bi_seq|layer\.3[0-5]\.)).*$"),
            ("weights only bf16", ("w",), None),
            ("weights bf16 except pool+itm", ("w",), r"^(?!.*(pooler|bi_seq)).*$"),
        ]
        for name, on, pat in variants:
            l, t, v, itm = S.forward(sd, rcfg, b, on, pat)
            print("  %-46s ITM rel %.2e" % (name, abs(l - ref) / ref), flush=True)
