"""Caption metrics for eval.py: BLEU-1..4, ROUGE-L and CIDEr over {video_id: [captions]} dictionaries.

Replaces what the reference reaches through `COCOScorer.score` (eval.py:154-204 -> the vendored coco-caption scorers
coco_caption/pycocoevalcap/{bleu,rouge,cider}); same definitions, constants and corpus conventions, so the numbers
agree with those scorers on the same token strings (tests/test_caption_metrics.py pins them on fixtures generated from
the reference's own Python scorers by oracle/make_metrics_golden.py).  Two pieces of the reference's scorer stack are
Java programs whose jars the reference does not ship (stanford-corenlp PTBTokenizer, meteor-1.5):
  * tokenisation is done here by `ptb_tokenize`, a regular-expression restatement of the PTB conventions that matter for
    captions (lower-casing, clitics split off, punctuation tokens dropped) - unpinned against the Java tokenizer;
  * METEOR is not computed (its aligner needs the jar's paraphrase / synonym tables); `CaptionScorer.score` leaves it out.
CPU string processing, outside the GPU path."""
import math
import re
from collections import Counter

import numpy as np

# tokens the reference's tokenizer wrapper removes after PTB tokenisation (ptbtokenizer.py:20-21)
_DROPPED = {"''", "'", "``", "`", "-LRB-", "-RRB-", "-LCB-", "-RCB-", ".", "?", "!", ",", ":", "-", "--", "...", ";",
            "-lrb-", "-rrb-", "-lcb-", "-rcb-"}
_CLITIC = re.compile(r"^(.*\w)(n't|'s|'re|'ve|'m|'ll|'d)$")
_SPLIT = re.compile(r"\.\.\.|--|``|''|[\"`.,;:?!()\[\]{}]")


def ptb_tokenize(sentence):
    """One caption -> its space-joined token string, lower-cased, punctuation tokens removed."""
    s = _SPLIT.sub(lambda m: " %s " % m.group(0), sentence.replace("\n", " ").lower())   # punctuation = its own token
    out = []
    for w in s.split():
        w = w.strip("'")                               # quote marks around a word are tokens of their own (dropped)
        m = _CLITIC.match(w)
        if m:                                          # "isn't" -> "is n't", "man's" -> "man 's"
            out += [m.group(1), m.group(2)]
        elif w and w not in _DROPPED and w not in '"()[]{}':
            out.append(w)
    return " ".join(out)


def tokenize(captions):
    """{id: [{'caption': str, ...}, ...]} -> {id: [token string, ...]} (the shape the scorers take)."""
    return {k: [ptb_tokenize(c["caption"]) for c in v] for k, v in captions.items()}


def _ngrams(words, n):
    """Counter over all 1..n-grams of a token list."""
    c = Counter()
    for k in range(1, n + 1):
        c.update(tuple(words[i:i + k]) for i in range(len(words) - k + 1))
    return c


def _check(gts, res):
    ids = sorted(gts.keys())
    if ids != sorted(res.keys()):
        raise ValueError("references and candidates must cover the same ids")
    for i in ids:
        if len(res[i]) != 1 or len(gts[i]) < 1:
            raise ValueError("id %r: need exactly one candidate and at least one reference" % (i,))
    return ids


def bleu(gts, res, n=4):
    """Corpus BLEU-1..n with the 'closest' reference length (ties -> the shorter reference) and the scorer's epsilon
    conventions (1e-15 on matches, 1e-9 on totals).  Returns ([BLEU_1..BLEU_n], [per-id lists, one per order])."""
    tiny, small = 1e-15, 1e-9
    ids = _check(gts, res)
    tot_match, tot_guess = [0] * n, [0] * n
    tot_test = tot_ref = 0
    per_id = [[] for _ in range(n)]
    for i in ids:
        hyp = res[i][0].split()
        hyp_counts = _ngrams(hyp, n)
        ref_max, ref_lens = Counter(), []
        for r in gts[i]:
            rw = r.split()
            ref_lens.append(len(rw))
            for g, c in _ngrams(rw, n).items():
                if c > ref_max[g]:
                    ref_max[g] = c
        match = [0] * n
        for g, c in hyp_counts.items():
            match[len(g) - 1] += min(c, ref_max.get(g, 0))
        guess = [max(0, len(hyp) - k) for k in range(n)]
        ref_len = min((abs(l - len(hyp)), l) for l in ref_lens)[1]
        tot_test += len(hyp)
        tot_ref += ref_len
        ratio = (len(hyp) + tiny) / (ref_len + small)
        bp = math.exp(1.0 - 1.0 / ratio) if ratio < 1.0 else 1.0
        prod = 1.0
        for k in range(n):
            tot_match[k] += match[k]
            tot_guess[k] += guess[k]
            prod *= (match[k] + tiny) / (guess[k] + small)
            per_id[k].append(prod ** (1.0 / (k + 1)) * bp)
    ratio = (tot_test + tiny) / (tot_ref + small)
    bp = math.exp(1.0 - 1.0 / ratio) if ratio < 1.0 else 1.0
    out, prod = [], 1.0
    for k in range(n):
        prod *= (tot_match[k] + tiny) / (tot_guess[k] + small)
        out.append(prod ** (1.0 / (k + 1)) * bp)
    return out, per_id


def _lcs(a, b):
    """Length of the longest common subsequence of two token lists (one rolling row)."""
    if len(a) < len(b):
        a, b = b, a
    row = [0] * (len(b) + 1)
    for x in a:
        diag = 0
        for j, y in enumerate(b, 1):
            keep = row[j]
            row[j] = diag + 1 if x == y else max(row[j], row[j - 1])
            diag = keep
    return row[len(b)]


def rouge_l(gts, res, beta=1.2):
    """ROUGE-L F-measure (beta 1.2) from the best LCS precision and the best LCS recall over the references.
    Returns (mean, per-id array)."""
    ids = _check(gts, res)
    scores = []
    for i in ids:
        hyp = res[i][0].split(" ")
        p = r = 0.0
        for ref in gts[i]:
            rw = ref.split(" ")
            l = _lcs(rw, hyp)
            p = max(p, l / float(len(hyp)))
            r = max(r, l / float(len(rw)))
        scores.append((1 + beta ** 2) * p * r / (r + beta ** 2 * p) if p != 0 and r != 0 else 0.0)
    scores = np.array(scores)
    return float(np.mean(scores)), scores


def cider(gts, res, n=4, sigma=6.0):
    """CIDEr: TF-IDF weighted n-gram cosine similarity (document frequency over the reference sets of the corpus,
    idf = log(#ids) - log(max(1, df))), clipped on the candidate side, Gaussian length penalty, averaged over orders and
    references, x10.  The length that enters the penalty is the scorer's (the number of bigrams).  Returns
    (mean, per-id array)."""
    ids = _check(gts, res)
    refs = {i: [_ngrams(r.split(), n) for r in gts[i]] for i in ids}
    df = Counter()
    for i in ids:
        df.update(set(g for c in refs[i] for g in c))
    log_n = np.log(float(len(ids)))

    def vec(counts):
        v = [dict() for _ in range(n)]
        length = 0
        for g, tf in counts.items():
            k = len(g) - 1
            v[k][g] = float(tf) * (log_n - np.log(max(1.0, df.get(g, 0.0))))
            if k == 1:
                length += tf
        norm = [np.sqrt(sum(w * w for w in d.values())) for d in v]
        return v, norm, length

    scores = []
    for i in ids:
        hv, hn, hl = vec(_ngrams(res[i][0].split(), n))
        total = np.zeros(n)
        for rc in refs[i]:
            rv, rn, rl = vec(rc)
            pen = np.e ** (-(float(hl - rl) ** 2) / (2 * sigma ** 2))
            for k in range(n):
                dot = sum(min(w, rv[k].get(g, 0.0)) * rv[k].get(g, 0.0) for g, w in hv[k].items())
                if hn[k] != 0 and rn[k] != 0:
                    dot /= hn[k] * rn[k]
                total[k] += dot * pen
        scores.append(np.mean(total) / len(refs[i]) * 10.0)
    scores = np.array(scores)
    return float(np.mean(scores)), scores


class CaptionScorer(object):
    """Same call as the reference's COCOScorer (eval.py:154-204): score(GT, RES, IDs) with GT / RES in the formats of
    eval.py:103-151 ({id: [{'caption': ...}, ...]}); returns {'Bleu_1'..'Bleu_4', 'ROUGE_L', 'CIDEr'} and keeps the
    per-id values in `imgToEval`.  METEOR is absent (see the module docstring)."""

    def __init__(self):
        self.eval = {}
        self.imgToEval = {}

    def score(self, GT, RES, IDs):
        self.eval, self.imgToEval = {}, {}
        gts = tokenize({i: GT[i] for i in IDs})
        res = tokenize({i: RES[i] for i in IDs})
        order = sorted(gts.keys())
        b, b_each = bleu(gts, res, 4)
        for k in range(4):
            self._set("Bleu_%d" % (k + 1), b[k], b_each[k], order)
        r, r_each = rouge_l(gts, res)
        self._set("ROUGE_L", r, r_each, order)
        c, c_each = cider(gts, res)
        self._set("CIDEr", c, c_each, order)
        return self.eval

    def _set(self, name, value, each, order):
        self.eval[name] = float(value)
        for i, v in zip(order, each):
            self.imgToEval.setdefault(i, {"image_id": i})[name] = float(v)
