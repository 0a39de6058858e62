import sys, os, json
import numpy as np, torch
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
from ddmi_amd import initweights as iw
from ddmi_amd import train as T
from helpers import named_state, rel_err
cfg = json.load(open(os.path.join(ROOT, "tests/golden/configs.json")))["TOY_UNET"]
t1 = dict(np.load(os.path.join(ROOT, "tests/golden/T1_train_step.npz")))
tr = T.LDMTrainStep(cfg, dict(timesteps=1000, channels=64), numerics="fp32", device="cuda:0")
st = {"model." + k: v for k, v in named_state("T1.unet", [(k, tuple(p.shape)) for k, p in tr.model.state_dict().items()]).items()}
st["mixing_logit"] = iw.normal("T1", "mixing_logit", (1, 64, 1, 1), 0.0, 0.5)
tr.load_state_dict(st).finalize()
cap = {}
rb = tr.model.input_blocks[1][0]
orig = rb.emb_layers[1].run
def run(tc, x, silu_in):
    y = orig(tc, x, silu_in)
    cap["emb"] = x.detach().clone()
    y.register_hook(lambda g: cap.__setitem__("demb", g.detach().clone()))
    return y
rb.emb_layers[1].run = run
g = lambda k, s: torch.from_numpy(iw.normal("T1", k, s))
loss = tr.loss_and_backward(g("x_start", (2, 64, 8, 8)), torch.from_numpy(t1["t"]), g("noise", (2, 64, 8, 8)))
print("loss", float(loss), float(t1["loss"]))
W = rb.emb_layers[1].weight
exp = cap["demb"].t() @ torch.nn.functional.silu(cap["emb"])
ref = torch.from_numpy(t1["grad.model.input_blocks.1.0.emb_layers.1.weight"]).cuda()
print("grad vs manual", rel_err(W.grad, exp), "manual vs fixture", rel_err(exp, ref), "grad vs fixture", rel_err(W.grad, ref))
print("demb", cap["demb"].shape, cap["demb"].abs().max().item(), "emb", cap["emb"].shape)
bad = {}
for k, s_ref, l2_ref in zip(t1["all_keys"], t1["all_sum"], t1["all_l2"]):
    gk = dict(tr.named_parameters())[str(k)].grad.double()
    e = abs(float(gk.pow(2).sum().sqrt()) - l2_ref) / (l2_ref + 1e-30)
    if e > 1e-3: bad[str(k)] = (e, float(gk.pow(2).sum().sqrt()), l2_ref)
print("bad l2:", bad)
