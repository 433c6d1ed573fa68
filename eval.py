"""Caption generation for the MI355X S2VT path: the decode loops of the reference's eval.py
(`eval()` eval.py:30-60, `beam_eval()` :63-99) on top of the drop-in model — load a full-module checkpoint, decode the
test split greedily or by beam search, map ids to words and cut at `<eos>` (eval.py:54-58, :90-96).

`main()` writes the predictions as JSON ({video_id: caption}) and, with --gts, scores them as the reference's
`__main__` does (eval.py:222-236: gts.json -> pred_to_coco_samples_IDs -> COCOScorer.score) with caption_metrics.py:
BLEU-1..4, ROUGE-L, CIDEr.  METEOR and the Stanford tokenizer are Java jars the reference does not ship
(`.MISSING_LARGE_BLOBS`); see caption_metrics.py for what stands in for them.
"""
import argparse
import json

import torch


def ids_to_caption(ids, ix2word, drop_sos=False):
    """[token ids] -> 'w1 w2 ...' cut at the first <eos> (eval.py:54-58); beam results start with <sos> (eval.py:92-95)."""
    words = [ix2word[str(int(i))] for i in ids]
    if '<eos>' in words:
        words = words[:words.index('<eos>')]
    if drop_sos and '<sos>' in words:
        words.remove('<sos>')
    return ' '.join(words)


def generate(model_path, caption_file, feats_path, batch_size=10, mode='test', beam_width=5, max_beam_depth=30,
             split='test', device=None):
    import dataloader
    dev = device or torch.device('cuda', 0)
    dataset = dataloader.VideoDataset(caption_file, feats_path, mode=split)
    loader = torch.utils.data.DataLoader(dataset, batch_size=batch_size, shuffle=False)
    model = torch.load(model_path, weights_only=False).to(dev)          # full-module pickle (eval.py:41)
    model.eval()
    if mode == 'beam_search':                                           # old pickles lack these attrs (eval.py:84-86)
        model.rnn_type, model.sos_ix, model.eos_ix = 'lstm', dataset.word2ix.get('<sos>', 3), dataset.word2ix.get('<eos>', 4)
    preds = {}
    with torch.no_grad():
        for feats, targets, ids, masks in dataloader.feed_batches(loader, dev):
            if mode == 'beam_search':
                out = model(feats, mode='beam_search', beam_width=beam_width, max_beam_depth=max_beam_depth)
                for vid, seq in zip(ids, out):
                    preds[vid] = ids_to_caption([int(t.item()) for t in seq], dataset.ix2word, drop_sos=True)
            else:
                out = model(feats, mode='test').cpu()
                for vid, seq in zip(ids, out):
                    preds[vid] = ids_to_caption(seq.tolist(), dataset.ix2word)
    return preds


def to_samples(prediction_dict, gts):
    """{video_id: caption} -> ({video_id: [{'image_id', 'caption'}]}, ids) for the ids that have ground truth
    (pred_to_coco_samples_IDs, eval.py:138-151)."""
    samples = {k: [{'image_id': k, 'caption': v}] for k, v in prediction_dict.items() if k in gts}
    return samples, list(samples.keys())


def score(prediction_dict, gts_file):
    """Metrics of a prediction dictionary against a gts.json ({'gts': {video_id: [{'caption': ...}, ...]}})."""
    import caption_metrics
    with open(gts_file, encoding='utf-8') as f:
        gts = json.load(f)['gts']
    samples, ids = to_samples(prediction_dict, gts)
    scorer = caption_metrics.CaptionScorer()
    scorer.score(gts, samples, ids)
    return scorer


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--model-path", required=True)
    ap.add_argument("--caption-file", default="./data/captions_server.json")
    ap.add_argument("--feats-path", default="./data/feats/vgg16_bn")
    ap.add_argument("--batch-size", type=int, default=10)
    ap.add_argument("--beam", type=int, default=0, help="beam width (0: greedy)")
    ap.add_argument("--out", default="predictions.json")
    ap.add_argument("--gts", default=None, help="gts.json: also print BLEU / ROUGE-L / CIDEr of the predictions")
    a = ap.parse_args()
    preds = generate(a.model_path, a.caption_file, a.feats_path, a.batch_size,
                     'beam_search' if a.beam else 'test', beam_width=a.beam or 5)
    with open(a.out, 'w', encoding='utf-8') as f:
        json.dump(preds, f, ensure_ascii=False, indent=1)
    print("wrote {} captions to {}".format(len(preds), a.out))
    if a.gts:
        scorer = score(preds, a.gts)
        print("***********************")
        print(scorer.eval)
        print("***********************")


if __name__ == '__main__':
    main()
