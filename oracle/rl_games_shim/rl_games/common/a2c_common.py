"""rl_games.common.a2c_common (1.1.4): swap_and_flatten01 + the A2CBase methods the reference's
update path calls. The oracle builds agents by attribute injection (oracle/ref_runner.py), so
__init__ here is intentionally empty: no env, no writer, no config parsing side effects."""


def swap_and_flatten01(arr):
    if arr is None:
        return arr
    s = arr.size()
    return arr.transpose(0, 1).reshape(s[0] * s[1], *s[2:])


class A2CBase:
    def __init__(self, base_name, config):
        pass

    def set_eval(self):
        self.model.eval()
        if self.normalize_input:
            self.running_mean_std.eval()
        if self.normalize_value:
            self.value_mean_std.eval()

    def set_train(self):
        self.model.train()
        if self.normalize_input:
            self.running_mean_std.train()
        if self.normalize_value:
            self.value_mean_std.train()

    def _preproc_obs(self, obs_batch):
        if obs_batch.dtype == __import__('torch').uint8:
            obs_batch = obs_batch.float() / 255.0
        if self.normalize_input:
            obs_batch = self.running_mean_std(obs_batch)
        return obs_batch

    def update_lr(self, lr):
        for param_group in self.optimizer.param_groups:
            param_group['lr'] = lr

    def update_epoch(self):
        self.epoch_num += 1
        return self.epoch_num

    def get_stats_weights(self):
        state = {}
        if self.normalize_input:
            state['running_mean_std'] = self.running_mean_std.state_dict()
        if self.normalize_value:
            state['reward_mean_std'] = self.value_mean_std.state_dict()
        return state

    def set_stats_weights(self, weights):
        if self.normalize_input:
            self.running_mean_std.load_state_dict(weights['running_mean_std'])
        if self.normalize_value:
            self.value_mean_std.load_state_dict(weights['reward_mean_std'])

    def get_weights(self):
        state = self.get_stats_weights()
        state['model'] = self.model.state_dict()
        return state

    def set_weights(self, weights):
        self.model.load_state_dict(weights['model'])
        self.set_stats_weights(weights)

    def get_full_state_weights(self):
        state = self.get_weights()
        state['epoch'] = self.epoch_num
        state['optimizer'] = self.optimizer.state_dict()
        state['frame'] = getattr(self, 'frame', 0)
        state['last_mean_rewards'] = getattr(self, 'last_mean_rewards', -100500)
        state['env_state'] = None
        return state


def rescale_actions(low, high, action):
    """rl_games.algos_torch.players / a2c_common.rescale_actions (1.1.4)"""
    d = (high - low) / 2.0
    m = (high + low) / 2.0
    return action * d + m


class ContinuousA2CBase(A2CBase):
    def preprocess_actions(self, actions):
        """rl_games ContinuousA2CBase.preprocess_actions (1.1.4): clamp to [-1, 1] and scale to the action space."""
        import torch
        if self.clip_actions:
            rescaled_actions = rescale_actions(self.actions_low, self.actions_high, torch.clamp(actions, -1.0, 1.0))
        else:
            rescaled_actions = actions
        if not self.is_tensor_obses:
            rescaled_actions = rescaled_actions.cpu().numpy()
        return rescaled_actions

    def obs_to_tensors(self, obs):
        """rl_games A2CBase.obs_to_tensors for tensor observations: wrap a bare tensor as {'obs': tensor}."""
        if isinstance(obs, dict):
            return {k: v for k, v in obs.items()}
        return {'obs': obs}
