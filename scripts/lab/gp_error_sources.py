"""Lab: where does the half-precision error of the reported gradient penalty come from?  After 27 updates in f16 mode, the
penalty of 4096 demo rows is evaluated in f64 on the GPU with ONE ingredient of the f16 path at a time rounded to half:
weights / logit weights / activations (-> ReLU masks) / chain values / input.  (plain torch; scripts/lab only)"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import ase_amd
ase_amd.configure()
import torch
import bench

dev = 'cuda:0'
agent, cfg, _ = bench.make_agent(dev, sys.argv[1] if len(sys.argv) > 1 else 'f16', 'program', 1, 0)
bench.fill_rollout(agent, dev)
agent._init_amp_demo_buf()
for _ in range(int(os.environ.get('UPDATES', 27))):
    agent.update(agent._play_steps_tail())
bench.fill_rollout(agent, dev)
agent._play_steps_tail()
torch.cuda.synchronize()
net = agent.model.a2c_network
lin = [m for m in net._disc_mlp.modules() if isinstance(m, torch.nn.Linear)]
Ws = [m.weight.detach().double() for m in lin]
bs = [m.bias.detach().double() for m in lin]
w = net._disc_logits.weight.detach().double().view(-1)
st = bench._rms_dict(agent.engine.amp_state)
mean, var = st['mean'].double().to(dev), st['var'].double().to(dev)
g = torch.Generator().manual_seed(0)
demo = agent._amp_obs_demo_buffer.data
x = demo[torch.randint(0, demo.shape[0], (4096,), generator=g).to(dev)].double()
x = ((x - mean) / torch.sqrt(var + 1e-5)).clamp(-5, 5)
h16 = lambda t: t.half().double()


def gp(Ws, w, x, r_act=False, r_chain=False, bf=False):
    rnd = (lambda t: t.bfloat16().double()) if bf else h16
    h, masks = x, []
    for W, b in zip(Ws, bs):
        z = h @ W.T + b
        masks.append(z > 0)
        h = torch.relu(z)
        if r_act:
            h = rnd(h)
            masks[-1] = h > 0
    gch = masks[-1] * w
    if r_chain:
        gch = rnd(gch)
    for l in range(len(Ws) - 1, 0, -1):
        gch = (gch @ Ws[l]) * masks[l - 1]
        if r_chain:
            gch = rnd(gch)
    gin = gch @ Ws[0]
    if r_chain:
        gin = rnd(gin)
    return float((gin * gin).sum(-1).mean())


ref = gp(Ws, w, x)
print('penalty (f64): %.6g   logit-weight max %.3g  rows 4096' % (ref, float(w.abs().max())))
rel = lambda v: (v - ref) / ref
for tag, rnd in (('f16', h16), ('bf16', lambda t: t.bfloat16().double())):
    bf = tag == 'bf16'
    W16, w16 = [rnd(W) for W in Ws], rnd(w)
    print(f'[{tag}] all weights rounded          %+.2e' % rel(gp(W16, w16, x)))
    print(f'[{tag}] logit weights only           %+.2e' % rel(gp(Ws, w16, x)))
    print(f'[{tag}] trunk weights only           %+.2e' % rel(gp(W16, w, x)))
    for l in range(len(Ws)):
        Wl = [W16[i] if i == l else Ws[i] for i in range(len(Ws))]
        print(f'[{tag}]   trunk layer {l} only         %+.2e' % rel(gp(Wl, w, x)))
    print(f'[{tag}] activations rounded (masks)  %+.2e' % rel(gp(Ws, w, x, r_act=True, bf=bf)))
    print(f'[{tag}] chain values rounded         %+.2e' % rel(gp(Ws, w, x, r_chain=True, bf=bf)))
    print(f'[{tag}] input rounded                %+.2e' % rel(gp(Ws, w, rnd(x))))
    print(f'[{tag}] everything                   %+.2e' % rel(gp(W16, w16, rnd(x), r_act=True, r_chain=True, bf=bf)))
