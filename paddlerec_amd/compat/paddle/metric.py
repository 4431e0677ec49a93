"""paddle.metric.Auc [EXT] (deepfm/dygraph_model.py:69-73,83-84): the reference feeds it HOST arrays
(`predict_2d.numpy()`, `label.numpy()`), so the bucket update is host arithmetic here too; the device-resident
variant of the engine's own trainer is rec_auc_histogram."""
import numpy as _np


class Auc:
    def __init__(self, curve="ROC", num_thresholds=4095, name="auc"):
        self._n = num_thresholds
        self.reset()

    def update(self, preds, labels):
        p = _np.asarray(preds)[:, 1].reshape(-1)
        t = _np.asarray(labels).reshape(-1)
        b = _np.minimum((p * self._n).astype(_np.int64), self._n)      # bin = int(p * num_thresholds)
        _np.add.at(self._pos, b[t != 0], 1)
        _np.add.at(self._neg, b[t == 0], 1)

    def accumulate(self):
        tot_pos = tot_neg = 0.0
        auc = 0.0
        for i in range(self._n, -1, -1):      # trapezoid sweep from the top bucket
            p0, n0 = tot_pos, tot_neg
            tot_pos += float(self._pos[i])
            tot_neg += float(self._neg[i])
            auc += abs(tot_neg - n0) * (tot_pos + p0) / 2.0
        return auc / tot_pos / tot_neg if tot_pos > 0 and tot_neg > 0 else 0.0

    def reset(self):
        self._pos = _np.zeros(self._n + 1, _np.int64)
        self._neg = _np.zeros(self._n + 1, _np.int64)
