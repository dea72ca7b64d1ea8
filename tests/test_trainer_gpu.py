"""FlatTrainer on the GPU: the fused clip + AdamW kernels against torch.nn.utils.clip_grad_norm_ +
torch.optim.AdamW (the reference's recipe, train.py:140, train_untils.py:35-42), eager and hipGraph replay."""
import copy
import pytest
import torch

from adnm_hip import recipe
from adnm_hip.trainer import FlatTrainer
from util import assert_close

pytestmark = pytest.mark.gpu
DEV = "cuda"


def small_model():
    from models.ADNMUNet import create_block
    torch.manual_seed(0)
    m = create_block(32, 16, headdim=4, norm_epsilon=1e-6)
    recipe.fill_parameters(m)
    return m.to(DEV)


@pytest.mark.parametrize("use_graph", [False, True])
@pytest.mark.parametrize("max_norm", [0.0, 0.05])
def test_fused_step_matches_torch(use_graph, max_norm):
    ref = small_model()
    mine = copy.deepcopy(ref)
    x, tgt = recipe.tensor("tr.x", (2, 64, 32)).to(DEV), recipe.tensor("tr.t", (2, 64, 16)).to(DEV)
    loss_fn = lambda o, t: ((o - t) ** 2).mean()
    opt = torch.optim.AdamW(ref.parameters(), lr=1e-3, betas=(0.9, 0.999), eps=1e-9, weight_decay=1e-2)
    tr = FlatTrainer(mine, loss_fn, lr=1e-3, betas=(0.9, 0.999), eps=1e-9, weight_decay=1e-2, max_norm=max_norm, use_graph=use_graph)
    for step in range(4):
        loss_ref = loss_fn(ref(x), tgt)
        loss_ref.backward()
        if max_norm > 0:
            norm_ref = torch.nn.utils.clip_grad_norm_(ref.parameters(), max_norm)
        opt.step()
        opt.zero_grad(set_to_none=True)
        loss = tr.step(x, tgt)
        assert abs(float(loss) - float(loss_ref)) <= 1e-5 * abs(float(loss_ref)) + 1e-7
        if max_norm > 0:
            assert abs(float(tr.grad_norm()) - float(norm_ref)) <= 1e-4 * float(norm_ref)
    for (k, a), (_, b) in zip(mine.named_parameters(), ref.named_parameters()):
        assert_close(a, b, 2e-5, k, atol=1e-6)
    # parameters that never receive a gradient are untouched (no decay), as torch.optim.AdamW leaves them
    unused = [k for k, p in mine.named_parameters() if all(p is not q for q in tr.used)]
    assert "alpha1" in unused and "act.beta" in unused
