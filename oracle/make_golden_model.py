"""Golden vectors of the MODEL WRAPPERS' forward: the reference's OWN ``ModelASEContinuous / ModelAMPContinuous /
ModelHRLContinuous .Network.forward`` (learning/ase_models.py:19-29, learning/amp_models.py:20-38, learning/hrl_models.py under
/root/reference/ase, imported unmodified through oracle/ref_runner.py) over the reference's own networks, with the weights,
inputs and network tables of the tiny update goldens (tests/golden/{ase,amp,ppo}_tiny.pt).  TEST INFRASTRUCTURE ONLY.

    python oracle/make_golden_model.py        # writes tests/golden/model_forward.pt

Train mode (``is_train`` True): prev_neglogp, values, entropy, mus, sigmas, the three discriminator logits and enc_pred.
Play mode: mus / sigmas / values and - under a fixed torch seed - the sampled actions with their neglogpacs.  The base class
``ModelA2CContinuousLogStd.Network.forward`` is rl_games' (restated in oracle/rl_games_shim: parity unpinned at that boundary,
see its README); the discriminator / encoder extensions and every network called underneath are the reference's."""
import os
import sys

import torch

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, os.path.dirname(HERE))

from oracle import ref_runner  # noqa: E402

GOLDEN = os.path.join(os.path.dirname(HERE), 'tests', 'golden')
OUT = os.environ.get('ASE_GOLDEN_OUT', GOLDEN)


def one(name):
    G = torch.load(os.path.join(GOLDEN, name + '.pt'), weights_only=False)
    kind, spec, cfg = G['kind'], G['spec'], G['cfg']
    A = ref_runner.build_ref_agent(kind, G['net'], dict(cfg), num_envs=spec['num_envs'], obs_size=spec['obs_size'],
                                   act_size=spec['act_size'], amp_obs_size=spec['amp_obs_size'] if kind != 'ppo' else None)
    A.model.load_state_dict({'a2c_network.' + k: v for k, v in G['init_sd'].items()})
    A.model.eval()                       # (no dropout / batch norm in these nets: eval() only silences module state)
    mb = G['epochs'][0]['first_minibatch']
    g = torch.Generator().manual_seed(77)
    n = mb['obs'].shape[0]
    # the model sees NORMALISED observations (the agents normalise first: learning/ase_agent.py:200-214): any tensors of the
    # right shapes do - the minibatch's raw observations scaled into the normaliser's clip range
    inp = {'is_train': True, 'obs': (mb['obs'] * 0.5).clamp(-5, 5), 'prev_actions': mb['actions'].clone()}
    if kind == 'ase':
        inp['ase_latents'] = mb['ase_latents'].clone()
    if kind != 'ppo':
        for k in ('amp_obs', 'amp_obs_replay', 'amp_obs_demo'):
            inp[k] = (mb[k] * 0.5).clamp(-5, 5)
    with torch.no_grad():
        train = {k: (v.clone() if torch.is_tensor(v) else v) for k, v in A.model(dict(inp)).items()}
        play_in = dict(inp, is_train=False)
        torch.manual_seed(4321)
        play = {k: (v.clone() if torch.is_tensor(v) else v) for k, v in A.model(play_in).items()}
    inp = {k: v for k, v in inp.items() if k != 'is_train'}
    return {'golden': name, 'kind': kind, 'inputs': inp, 'train': train, 'play': play, 'play_seed': 4321, 'rows': n}


def main():
    out = {name: one(name) for name in ('ase_tiny', 'amp_tiny', 'ppo_tiny')}
    os.makedirs(OUT, exist_ok=True)
    torch.save(out, os.path.join(OUT, 'model_forward.pt'))
    for k, v in out.items():
        print(k, sorted(v['train']), sorted(v['play']))


if __name__ == '__main__':
    main()
