"""metrics.py -- the evaluation metrics of the reference's model_fns, as streaming accumulators on the device.

Mirrors `_get_metric_op` (models/DeepCrossNetwork/DeepCrossNetwork.py:293-319; the canned head of DeepFM and
ESMM.py:178-215 report the same family): average_loss = tf.metrics.mean(unweighted_loss, weights), accuracy / precision /
recall of class_ids against the labels (weighted), accuracy_baseline = max(label mean, 1 - label mean), and auc =
[TF-upstream] tf.metrics.auc with its defaults -- 200 thresholds (0, 1 and 198 interior points, each side moved out by
1e-7), ROC curve, trapezoidal summation.  Host-side bookkeeping of a few counters; nothing here is on the hot path.
"""
import torch


class BinaryMetrics:
    def __init__(self, num_thresholds=200, device="cuda"):
        eps = 1e-7
        t = [(i + 1) * 1.0 / (num_thresholds - 1) for i in range(num_thresholds - 2)]
        self.thresholds = torch.tensor([0.0 - eps] + t + [1.0 + eps], dtype=torch.float64, device=device)
        z = lambda *s: torch.zeros(*s, dtype=torch.float64, device=device)  # noqa: E731
        self.tp, self.fp, self.tn, self.fn = z(num_thresholds), z(num_thresholds), z(num_thresholds), z(num_thresholds)
        self.sums = z(8)   # w, w*loss, w*label, w*correct, w*tp@.5, w*fp@.5, w*fn@.5, w*logistic

    @torch.no_grad()
    def update(self, labels, logistic, unweighted_loss=None, weights=None):
        """labels, logistic (and loss, weights): [B] or [B,1] on the metrics' device."""
        y = labels.reshape(-1).to(torch.float64)
        p = logistic.reshape(-1).to(torch.float64)
        w = torch.ones_like(y) if weights is None else weights.reshape(-1).to(torch.float64)
        loss = torch.zeros_like(y) if unweighted_loss is None else unweighted_loss.reshape(-1).to(torch.float64)
        pred = (p > 0.5).to(torch.float64)                  # class_ids = argmax([1-p, p])
        pos = y > 0.5
        above = p.unsqueeze(0) > self.thresholds.unsqueeze(1)              # [T, B]
        wp, wn = (w * pos).unsqueeze(0), (w * (~pos)).unsqueeze(0)
        self.tp += (above * wp).sum(1)
        self.fn += ((~above) * wp).sum(1)
        self.fp += (above * wn).sum(1)
        self.tn += ((~above) * wn).sum(1)
        self.sums += torch.stack([w.sum(), (w * loss).sum(), (w * y).sum(), (w * (pred == y)).sum(), (w * pred * y).sum(),
                                  (w * pred * (1 - y)).sum(), (w * (1 - pred) * y).sum(), (w * p).sum()])
        return self

    def result(self):
        s = self.sums.tolist()
        wsum = s[0]
        div = lambda a, b: a / b if b > 0 else 0.0  # noqa: E731
        eps = 1e-6                                           # tf.metrics.auc's epsilon in the rates
        tpr = (self.tp + eps) / (self.tp + self.fn + eps)
        fpr = self.fp / (self.fp + self.tn + eps)
        auc = float(((fpr[:-1] - fpr[1:]) * (tpr[:-1] + tpr[1:]) / 2.0).sum())
        label_mean = div(s[2], wsum)
        return {"average_loss": div(s[1], wsum), "accuracy": div(s[3], wsum), "precision": div(s[4], s[4] + s[5]),
                "recall": div(s[4], s[4] + s[6]), "accuracy_baseline": max(label_mean, 1.0 - label_mean), "auc": auc,
                "label/mean": label_mean, "prediction/mean": div(s[7], wsum)}
