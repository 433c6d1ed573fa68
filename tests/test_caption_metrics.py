"""caption_metrics.py against numbers produced by the reference's own Python scorers (tests/golden/metrics.json, written
by oracle/make_metrics_golden.py), plus the tokeniser conventions and the COCOScorer-shaped wrapper."""
import json
import os

import numpy as np
import pytest

import caption_metrics as cm

GOLD = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden", "metrics.json")


@pytest.fixture(scope="module")
def cases():
    with open(GOLD) as f:
        return json.load(f)


@pytest.mark.parametrize("name", ["corpus48", "corpus5", "single"])
def test_bleu_rouge_cider_match_the_reference_scorers(cases, name):
    c = cases[name]
    b, b_each = cm.bleu(c["gts"], c["res"], 4)
    np.testing.assert_allclose(b, c["bleu"], rtol=1e-12, atol=1e-15)
    np.testing.assert_allclose(b_each, c["bleu_each"], rtol=1e-12, atol=1e-15)
    r, r_each = cm.rouge_l(c["gts"], c["res"])
    np.testing.assert_allclose(r, c["rouge"], rtol=1e-12)
    np.testing.assert_allclose(r_each, c["rouge_each"], rtol=1e-12, atol=0)
    s, s_each = cm.cider(c["gts"], c["res"])
    np.testing.assert_allclose(s, c["cider"], rtol=1e-10, atol=1e-12)
    np.testing.assert_allclose(s_each, c["cider_each"], rtol=1e-10, atol=1e-12)


def test_known_answers():
    gts = {"v": ["a man is playing a guitar"]}
    assert cm.bleu(gts, {"v": ["a man is playing a guitar"]})[0] == pytest.approx([1.0] * 4)
    # 3 of 4 unigrams match, 2 of 3 bigrams; candidate shorter than the reference: brevity penalty exp(1 - 6/4)
    b = cm.bleu(gts, {"v": ["a man is singing"]})[0]
    assert b[0] == pytest.approx(0.75 * np.exp(-0.5), rel=1e-6)
    assert b[1] == pytest.approx(np.sqrt(0.75 * 2.0 / 3.0) * np.exp(-0.5), rel=1e-6)
    # LCS("a man is singing", ref) = 3: P = 3/4, R = 3/6
    p, r, beta = 0.75, 0.5, 1.2
    assert cm.rouge_l(gts, {"v": ["a man is singing"]})[0] == pytest.approx((1 + beta ** 2) * p * r / (r + beta ** 2 * p))
    assert cm.rouge_l(gts, {"v": ["zebra"]})[0] == 0.0
    # one id: log(1) - log(1) = 0 -> every tf-idf weight is 0 -> CIDEr 0 (the reference's behaviour on a 1-video corpus)
    assert cm.cider(gts, {"v": ["a man is playing a guitar"]})[0] == 0.0


def test_mismatched_ids_and_multiple_candidates_are_rejected():
    with pytest.raises(ValueError):
        cm.bleu({"a": ["x"]}, {"b": ["x"]})
    with pytest.raises(ValueError):
        cm.rouge_l({"a": ["x"]}, {"a": ["x", "y"]})


def test_tokeniser_conventions():
    assert cm.ptb_tokenize("A man isn't playing the Guitar.") == "a man is n't playing the guitar"
    assert cm.ptb_tokenize("The dog's ball, (red) -- rolls...") == "the dog 's ball red rolls"
    assert cm.ptb_tokenize('"Hello", she said!') == "hello she said"
    assert cm.ptb_tokenize("two\nlines") == "two lines"


def test_scorer_wrapper_has_the_reference_call_shape():
    GT = {"video1": [{"image_id": "video1", "cap_id": 0, "caption": "A man is playing a guitar."},
                     {"image_id": "video1", "cap_id": 1, "caption": "Someone plays the guitar"}],
          "video2": [{"image_id": "video2", "cap_id": 0, "caption": "A dog runs in the water"}]}
    RES = {"video1": [{"image_id": "video1", "caption": "a man is playing a guitar"}],
           "video2": [{"image_id": "video2", "caption": "a cat runs in the street"}]}
    sc = cm.CaptionScorer()
    out = sc.score(GT, RES, ["video1", "video2"])
    assert set(out) == {"Bleu_1", "Bleu_2", "Bleu_3", "Bleu_4", "ROUGE_L", "CIDEr"}
    assert 0.0 < out["Bleu_4"] < out["Bleu_1"] <= 1.0
    assert sc.imgToEval["video1"]["Bleu_4"] == pytest.approx(1.0)
    assert sc.imgToEval["video2"]["ROUGE_L"] < 1.0


def test_eval_score_reads_a_gts_file_and_skips_videos_without_ground_truth(tmp_path):
    import importlib.util
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    spec = importlib.util.spec_from_file_location("s2vt_eval", os.path.join(root, "eval.py"))
    s2vt_eval = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(s2vt_eval)
    gts = {"gts": {"video1": [{"image_id": "video1", "cap_id": 0, "caption": "a man is playing a guitar"}],
                   "video2": [{"image_id": "video2", "cap_id": 0, "caption": "a dog runs"},
                              {"image_id": "video2", "cap_id": 1, "caption": "the dog is running in water"}]}}
    with open(tmp_path / "gts.json", "w") as f:
        json.dump(gts, f)
    preds = {"video1": "a man is playing a guitar", "video2": "a dog is running", "video9": "no ground truth for this one"}
    sc = s2vt_eval.score(preds, str(tmp_path / "gts.json"))
    assert sorted(sc.imgToEval) == ["video1", "video2"]
    assert sc.imgToEval["video1"]["Bleu_4"] == pytest.approx(1.0)
    assert 0.0 < sc.eval["ROUGE_L"] <= 1.0 and sc.eval["CIDEr"] >= 0.0
