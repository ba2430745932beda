"""AUC(num_thresholds=500) restated in numpy fp32 (test infrastructure).

Follows utils/auc.py:118-126 (thresholds), utils/metrics_utils.py:297-354
(confusion counts by tiling predictions against thresholds, `pred > thr`),
utils/auc.py:248-281 (`result()`: ROC, 'interpolation' summation).
Pinned by the reference's only known-answer vector, utils/auc.py:46-55.
"""
import numpy as np

F32 = np.float32
K_EPSILON = 1e-7   # K.epsilon(), utils/auc.py:126


def thresholds(num_thresholds=500):
    """utils/auc.py:118-126: python floats, turned into an fp32 constant at
    utils/metrics_utils.py:301-303 (array_ops.constant(thresholds))."""
    inner = [(i + 1) * 1.0 / (num_thresholds - 1) for i in range(num_thresholds - 2)]
    return np.array([0.0 - K_EPSILON] + inner + [1.0 + K_EPSILON], dtype=F32)


def confusion_counts(y_true, y_pred, thr):
    """utils/metrics_utils.py:297-354: returns float32 tp, fp, tn, fn of shape [T]."""
    pred = np.asarray(y_pred, F32).reshape(1, -1)
    lab = np.asarray(y_true).reshape(1, -1).astype(bool)
    pred_pos = pred > thr.reshape(-1, 1)
    tp = np.sum(lab & pred_pos, axis=1)
    fn = np.sum(lab & ~pred_pos, axis=1)
    fp = np.sum(~lab & pred_pos, axis=1)
    tn = np.sum(~lab & ~pred_pos, axis=1)
    return tp.astype(F32), fp.astype(F32), tn.astype(F32), fn.astype(F32)


def div_no_nan(a, b):
    out = np.zeros_like(a, dtype=F32)
    nz = b != 0
    out[nz] = (a[nz] / b[nz]).astype(F32)
    return out


def auc_from_counts(tp, fp, tn, fn):
    """utils/auc.py:248-281, ROC + interpolation, all in fp32."""
    tp, fp, tn, fn = (np.asarray(a, F32) for a in (tp, fp, tn, fn))
    recall = div_no_nan(tp, (tp + fn).astype(F32))
    fpr = div_no_nan(fp, (fp + tn).astype(F32))
    n = tp.shape[0]
    heights = ((recall[:n - 1] + recall[1:]) / F32(2.0)).astype(F32)
    terms = ((fpr[:n - 1] - fpr[1:]) * heights).astype(F32)
    return F32(np.sum(terms, dtype=F32))


class AUC(object):
    """stateful metric: update_state over batches, result at the end."""

    def __init__(self, num_thresholds=500):
        self.thr = thresholds(num_thresholds)
        self.reset_states()

    def reset_states(self):
        T = self.thr.shape[0]
        self.tp = np.zeros(T, F32)
        self.fp = np.zeros(T, F32)
        self.tn = np.zeros(T, F32)
        self.fn = np.zeros(T, F32)

    def update_state(self, y_true, y_pred):
        tp, fp, tn, fn = confusion_counts(y_true, y_pred, self.thr)
        self.tp += tp
        self.fp += fp
        self.tn += tn
        self.fn += fn

    def result(self):
        return auc_from_counts(self.tp, self.fp, self.tn, self.fn)


def auc500(y_true, y_pred, batch_size=None):
    m = AUC(500)
    n = len(y_pred)
    bs = batch_size or n
    for s in range(0, n, bs):
        m.update_state(y_true[s:s + bs], y_pred[s:s + bs])
    return m.result()
