"""TEST INFRASTRUCTURE ONLY (see oracle/__init__.py): CPU restatement of the epoch metrics of the CA / single-stream drivers.

Follows the reference's evaluation code, MAIN_CA = main_vit_covid_..._crossvit_2vits_2additionaloutputs_trainval_sum.py:
  * `_, preds = torch.max(output, 1)`                                   MAIN_CA:870   (first maximum wins)
  * `all_gt_one_hot = label_binarize(all_gt, classes=[0,1,2])`          MAIN_CA:901
  * per class `metrics.roc_curve(onehot[:, i], all_val[:, i])` + `metrics.auc(fpr, tpr)`, mean over the 3 classes   MAIN_CA:905-909
  * `epoch_acc = sum(all_pred == all_gt) / num_imgs`, `epoch_loss = running_loss / num_imgs`                         MAIN_CA:910-911
The ROC arithmetic itself lives in scikit-learn (reference dependency, unpinned); its published algorithm - thresholds at the
distinct scores in decreasing order, cumulative TP / FP counts, trapezoid area - is restated here with integers: the area equals
(2 * #{(pos, neg): s_pos > s_neg} + #{(pos, neg): s_pos == s_neg}) / (2 * n_pos * n_neg).  tests/test_oracle_golden.py pins this
restatement against the installed sklearn.metrics.roc_curve / auc on seeded scores with ties.
"""
import numpy as np


def argmax_first(scores):
    """torch.max(output, 1)[1] / np.argmax: index of the first maximum of each row."""
    return np.argmax(scores, axis=1)


def confusion_matrix(preds, labels, num_classes):
    """conf[t][p] = number of samples with label t predicted as p."""
    conf = np.zeros((num_classes, num_classes), dtype=np.int64)
    for t, p in zip(labels.tolist(), preds.tolist()):
        conf[t][p] += 1
    return conf


def auc_pair_counts(scores, labels, num_classes):
    """Per class c: (u2, n_pos, n_neg) with u2 = 2 * #{pos > neg} + #{pos == neg} over all (positive, negative) pairs."""
    out = []
    for c in range(num_classes):
        s = scores[:, c]
        pos, neg = s[labels == c], s[labels != c]
        neg_sorted = np.sort(neg)
        lt = np.searchsorted(neg_sorted, pos, side="left")      # negatives strictly below each positive
        le = np.searchsorted(neg_sorted, pos, side="right")     # negatives below or equal
        u2 = int(2 * lt.sum() + (le - lt).sum())
        out.append((u2, int(pos.size), int(neg.size)))
    return out


def roc_auc_ovr(scores, labels, num_classes):
    """Per-class one-vs-rest ROC AUC (MAIN_CA:905-907) and their mean (MAIN_CA:909).  NaN for a class without both kinds."""
    aucs = []
    for u2, npos, nneg in auc_pair_counts(scores, labels, num_classes):
        aucs.append(u2 / (2.0 * npos * nneg) if npos and nneg else float("nan"))
    return np.array(aucs), float(np.mean(aucs))


def epoch_metrics(all_val, all_gt, loss_sum, num_imgs, num_classes=3):
    """(epoch_loss, epoch_auc, epoch_acc) exactly as MAIN_CA:909-911 combines them."""
    preds = argmax_first(all_val)
    _, auc = roc_auc_ovr(all_val, all_gt, num_classes)
    return loss_sum / num_imgs, auc, float(np.sum(preds == all_gt)) / num_imgs
