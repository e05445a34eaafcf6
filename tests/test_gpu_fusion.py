"""Multi-stream logit fusion + top-k on the GPU vs the numpy oracle: bit-exact fusion."""
import numpy as np
import pytest
import torch

import _bootstrap
from oracle import stgcn_oracle as o

pytestmark = pytest.mark.gpu
pkg = _bootstrap.load()
DEV = "cuda:0"


@pytest.mark.parametrize("n_streams", [1, 2, 4])
@pytest.mark.parametrize("method", ["add", "maximum"])
@pytest.mark.parametrize("steps", [0, 3])
def test_fusion_and_topk(n_streams, method, steps):
    rng = np.random.default_rng(10 * n_streams + steps)
    n, classes = 517, 60
    shape = (n, classes, steps) if steps else (n, classes)
    preds = [rng.standard_normal(shape).astype(np.float32) for _ in range(n_streams)]
    targets = rng.integers(0, classes, n)
    want = o.fuse_preds(preds, np.add if method == "add" else np.maximum)
    dev = [torch.from_numpy(p).to(DEV) for p in preds]
    got = pkg.fusion.aggregate_preds(dev, method).cpu().numpy()
    assert got.shape == want.shape and np.array_equal(got, want)
    accs = pkg.fusion.topk_accuracies(dev, targets, (1, 3, 5), method)
    assert np.allclose(accs, o.topk_accuracies_np(want, targets), atol=1e-7)


@pytest.mark.parametrize("dims", ["2d", "3d"])
def test_fusion_equals_the_reference_fixture_g10(dims):
    """csk_fuse_rank_f32 against G10 = outputs of the reference's own aggregate_preds (scripts/multi_stream_eval.py:33-42,
    executed by tests/golden/make_golden.py): bit-exact for add and maximum over 1-4 streams, 2-D and 3-D predictions."""
    from tests.helpers import load_golden
    a, _ = load_golden("g10_fusion")
    dev = [torch.from_numpy(a[f"{dims}/pred{i}"]).to(DEV) for i in range(4)]
    for n in (1, 2, 3, 4):
        for name in ("add", "maximum"):
            want = a[f"{dims}/{name}{n}"]
            want = want[:, :, 0] if want.ndim == 3 else want
            got = pkg.fusion.aggregate_preds(dev[:n], name).cpu().numpy()
            assert got.shape == want.shape and np.array_equal(got, want), (dims, name, n)


def test_fusion_errors():
    a = torch.rand(4, 60, device=DEV)
    with pytest.raises(AssertionError):
        pkg.fusion.aggregate_preds([a, torch.rand(4, 61, device=DEV)])
    with pytest.raises(ValueError):
        pkg.fusion.aggregate_preds([a] * 5)
    with pytest.raises(RuntimeError, match="no CPU fallback"):
        pkg.fusion.aggregate_preds([a.cpu()])
