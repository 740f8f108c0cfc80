"""rl_games.algos_torch.a2c_continuous.A2CAgent (1.1.4): only train_actor_critic is reached
(CommonAgent.__init__ bypasses A2CAgent.__init__, common_agent.py:27)."""
from rl_games.common import a2c_common


class A2CAgent(a2c_common.ContinuousA2CBase):
    def train_actor_critic(self, input_dict):
        self.calc_gradients(input_dict)
        return self.train_result
