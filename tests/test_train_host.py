"""Host-side logic of the training path that needs no GPU: the EMA schedule against the oracle's ema_pytorch restatement."""
import pytest
import torch

from ddmi_amd.train import ema_pytorch_plan
from oracle import ddmi_oracle as O


@pytest.mark.parametrize("beta,every", [(0.9999, 10), (0.995, 1), (0.9999, 7)])
def test_ema_plan_reproduces_the_ema_pytorch_restatement(beta, every):
    """1 300 `update()` calls on a drifting parameter: the (action, decay) plan LDMTrainStep.ema_update executes gives the tensor the
    oracle's EmaPytorchOracle (ema_pytorch.EMA restated: copy while step <= 100, warm-up decay, cap at beta) gives."""
    g = torch.Generator().manual_seed(5)
    online = torch.randn(64, generator=g)
    orc = O.EmaPytorchOracle([online], beta=beta, update_every=every)
    ema, initted = online.clone(), False
    kinds = set()
    for step in range(1300):
        online += 0.01 * torch.randn(64, generator=g)     # "an optimizer step"
        action, decay = ema_pytorch_plan(step, beta, every)
        kinds.add(action)
        if action == "copy":
            ema.copy_(online)
        elif action == "lerp":
            if not initted:
                ema.copy_(online)
                initted = True
            ema = decay * ema + (1.0 - decay) * online
        orc.update()
        assert torch.allclose(ema, orc.ema[0], rtol=1e-5, atol=1e-6), step
    assert int(orc.step) == 1300 and bool(orc.initted) == initted
    assert kinds == ({"copy", "lerp"} if every == 1 else {"skip", "copy", "lerp"})
    # after 1 300 calls the EMA is a recent average, not the initial weights (what a constant 0.9999 from step 0 would leave)
    assert float((ema - online).abs().max()) < 0.5
