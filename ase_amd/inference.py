"""Forward-only use of the HIP engine behind the network-level API (eval_actor / eval_critic /
eval_disc / eval_enc / sample_latents; learning/ase_network_builder.py:115-144,214-225 and
learning/amp_network_builder.py:51-84 under /root/reference/ase).  Inputs are already-normalised
observations, as in the reference."""
import torch

from .engine import UpdateEngine

_INFER_CFG = {'learning_rate': 0.0, 'normalize_input': True, 'normalize_value': True, 'normalize_amp_input': True,
              'amp_diversity_bonus': 0.0, 'disc_coef': 0.0, 'disc_weight_decay': 0.0, 'disc_logit_reg': 0.0,
              'enc_weight_decay': 0.0, 'enc_coef': 0.0}


class InferenceEngine:
    def __init__(self, net, engine=None, dtype=torch.bfloat16):
        if engine is None:
            from .backend import HipBackend
            engine = UpdateEngine(net.kind, net, dict(_INFER_CFG), HipBackend(net.flat_params.device), minibatch=0,
                                  amp_minibatch=0, dtype=dtype)
        self.eng = engine

    def refresh(self):
        self.eng.refresh_shadows()

    def actor(self, obs, z=None):
        return self.eng.policy_forward(obs, z, normalize=False, want=('mu',))['mu']

    def critic(self, obs, z=None):
        return self.eng.policy_forward(obs, z, normalize=False, unnorm_value=False, want=('value',))['value']

    def disc(self, amp_obs):
        HD, _ = self.eng.amp_heads(amp_obs, normalize=False)
        return HD[:, 0:1].clone()

    def enc(self, amp_obs):
        _, e = self.eng.amp_heads(amp_obs, normalize=False)
        e = e[:, :self.eng.z]
        return e / e.norm(dim=-1, keepdim=True).clamp_min(1e-12)

    def sample_latents(self, n):
        z = torch.empty(n, self.eng.z, dtype=torch.float32, device=self.eng.dev)
        self.eng.be.sample_latents(z, n, self.eng.z, self.eng.rng_state)
        return z
