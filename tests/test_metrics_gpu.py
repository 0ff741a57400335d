"""GPU: epoch metrics kernels through the C ABI, bit-exact against oracle/ref_metrics.py."""
import numpy as np
import pytest
import torch

from oracle import ref_metrics

pytestmark = pytest.mark.gpu


def _case(n, quant, seed, C=3):
    rng = np.random.Generator(np.random.PCG64(seed))
    scores = rng.standard_normal((n, C)).astype(np.float32)
    if quant:
        scores = (np.round(scores * quant) / quant).astype(np.float32)      # exact ties, incl. tied row maxima
    labels = rng.choice(C, size=n, p=[0.6, 0.3, 0.1] if C == 3 else None).astype(np.int64)
    return scores, labels


@pytest.mark.parametrize("n,quant", [(1, None), (5, 1), (257, None), (1000, 4), (4099, 8)])
def test_eval_counts_bit_exact(n, quant):
    from mfvit import metrics
    scores, labels = _case(n, quant, seed=100 + n)
    dev = torch.device("cuda:0")
    conf, preds, u2, npos = metrics.eval_counts(torch.from_numpy(scores).to(dev), torch.from_numpy(labels).to(dev))
    want_preds = ref_metrics.argmax_first(scores)
    np.testing.assert_array_equal(preds.cpu().numpy(), want_preds)                      # first maximum wins (torch.max, MAIN_CA:870)
    np.testing.assert_array_equal(conf.cpu().numpy(), ref_metrics.confusion_matrix(want_preds, labels, 3))
    pairs = ref_metrics.auc_pair_counts(scores, labels, 3)
    assert u2.cpu().tolist() == [p[0] for p in pairs]                                     # integers: bit-exact
    assert npos.cpu().tolist() == [p[1] for p in pairs]


def test_epoch_meter_matches_reference_epoch_arithmetic():
    """EpochMeter over several ragged batches == the oracle's restatement of MAIN_CA:884-911 on the concatenated epoch."""
    from mfvit import metrics
    dev = torch.device("cuda:0")
    scores, labels = _case(777, 16, seed=5)
    meter = metrics.EpochMeter(3)
    loss_sum, i = 0.0, 0
    for bs in (128, 128, 128, 128, 128, 128, 9):
        s, t = scores[i:i + bs], labels[i:i + bs]
        loss = torch.nn.functional.cross_entropy(torch.from_numpy(s), torch.from_numpy(t))
        loss_sum += float(loss) * len(t)                                                  # running_loss += loss.item() * n
        meter.update(torch.from_numpy(s).to(dev), torch.from_numpy(t).to(dev).float(), loss.to(dev))   # labels arrive as float, MAIN_CA:856
        i += bs
    got = meter.compute()
    want = ref_metrics.epoch_metrics(scores, labels, loss_sum, 777)
    assert abs(got[0] - want[0]) < 1e-6 * abs(want[0])
    assert abs(got[1] - want[1]) < 1e-12 and abs(got[2] - want[2]) < 1e-12
    per, _ = ref_metrics.roc_auc_ovr(scores, labels, 3)
    np.testing.assert_allclose(meter.auc_per_class.numpy(), per, rtol=0, atol=1e-12)
