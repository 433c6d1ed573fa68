"""Golden vectors for caption_metrics.py, generated from the reference's own Python scorers
(/root/reference/coco_caption/pycocoevalcap/{bleu,rouge,cider}: pure Python, importable in the build container).
TEST INFRASTRUCTURE: writes tests/golden/metrics.json = the token strings fed to the scorers and the numbers they
returned.  usage: PYTHONDONTWRITEBYTECODE=1 python oracle/make_metrics_golden.py"""
import contextlib
import io
import json
import os
import random
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
REF = "/root/reference/coco_caption"


def corpus(seed=7, n_ids=48):
    rng = random.Random(seed)
    vocab = ("a man woman dog cat is are playing riding cutting the guitar horse onion with on in water street two people "
             "running slowly kitchen ball small someone something and piano").split()
    gts, res = {}, {}
    for i in range(n_ids):
        vid = "video%d" % (1200 + i)
        refs = [" ".join(rng.choice(vocab) for _ in range(rng.randint(3, 12))) for _ in range(rng.randint(1, 6))]
        words = rng.choice(refs).split()
        kind = i % 6
        if kind == 0:                                  # a reference word for word
            hyp = words
        elif kind == 1:                                # truncated: brevity penalty
            hyp = words[:max(1, len(words) // 2)]
        elif kind == 2:                                # no overlap at all
            hyp = ["zebra", "quietly", "juggles"]
        elif kind == 3:                                # repeated words: clipping
            hyp = [words[0]] * 5 + words[:2]
        else:                                          # random edits
            hyp = [w if rng.random() < 0.7 else rng.choice(vocab) for w in words] + \
                  [rng.choice(vocab) for _ in range(rng.randint(0, 3))]
        gts[vid] = refs
        res[vid] = [" ".join(hyp)]
    return gts, res


def main():
    sys.path.insert(0, REF)
    from pycocoevalcap.bleu.bleu import Bleu
    from pycocoevalcap.cider.cider import Cider
    from pycocoevalcap.rouge.rouge import Rouge
    cases = {}
    for name, (seed, n_ids) in {"corpus48": (7, 48), "corpus5": (11, 5), "single": (3, 1)}.items():
        gts, res = corpus(seed, n_ids)
        with contextlib.redirect_stdout(io.StringIO()):
            b, b_each = Bleu(4).compute_score(gts, res)
        r, r_each = Rouge().compute_score(gts, res)
        c, c_each = Cider().compute_score(gts, res)
        cases[name] = {"gts": gts, "res": res, "bleu": [float(x) for x in b],
                       "bleu_each": [[float(x) for x in row] for row in b_each], "rouge": float(r),
                       "rouge_each": [float(x) for x in r_each], "cider": float(c), "cider_each": [float(x) for x in c_each]}
        print(name, "Bleu", ["%.4f" % x for x in b], "ROUGE_L %.4f CIDEr %.4f" % (r, c))
    with open(os.path.join(ROOT, "tests", "golden", "metrics.json"), "w") as f:
        json.dump(cases, f, indent=0, sort_keys=True)


if __name__ == "__main__":
    main()
