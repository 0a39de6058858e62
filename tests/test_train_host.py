"""Host-side logic of the training path that needs no GPU: the EMA schedule against the oracle's ema_pytorch restatement."""
import pytest
import torch

from ddmi_amd.train import ema_pytorch_plan, module_parameters_order
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


def test_adamw_state_is_indexed_in_module_parameters_order():
    """`dae_opt` is `AdamW(diffusion_process.parameters()).state_dict()` (tools/ldm/image.py:62, 113): parameter index i is the i-th
    parameter of `nn.Module.parameters()`, which yields the root module's own parameters BEFORE its submodules' - so the reference's
    `mixing_logit` (ddpm.py:156) is index 0 although `model` (ddpm.py:124) was registered first.  Checked against a real nn.Module."""
    class Diffusion(torch.nn.Module):
        def __init__(self):
            super().__init__()
            self.model = torch.nn.Sequential(torch.nn.Linear(3, 4), torch.nn.Linear(4, 2))     # registered first, as in ddpm.py:124
            self.register_buffer("betas", torch.zeros(5))
            self.mixing_logit = torch.nn.Parameter(torch.zeros(1, 2, 1, 1))                    # ddpm.py:156
    m = Diffusion()
    torch_order = [k for k, _ in m.named_parameters()]
    assert torch_order[0] == "mixing_logit"
    state_dict_order = [k for k in torch_order if k.startswith("model.")] + ["mixing_logit"]   # LDMTrainStep's slab order: model.* first
    assert module_parameters_order(state_dict_order) == torch_order
    # and that is the indexing of the optimizer's state dict
    opt = torch.optim.AdamW(m.parameters(), lr=1e-3)
    m.model(torch.ones(1, 3)).sum().backward()
    m.mixing_logit.grad = torch.ones_like(m.mixing_logit)
    opt.step()
    sd = opt.state_dict()
    shapes = [tuple(sd["state"][i]["exp_avg"].shape) for i in sd["param_groups"][0]["params"]]
    assert shapes == [tuple(dict(m.named_parameters())[k].shape) for k in module_parameters_order(state_dict_order)]
    assert module_parameters_order(["model.a", "model.b"]) == ["model.a", "model.b"]          # no mixed prediction: unchanged
