"""rl_games.common.schedulers: lr_schedule 'constant' -> IdentityScheduler (every reference yaml)."""


class RLScheduler:
    def update(self, current_lr, entropy_coef, epoch, frames, kl_dist, **kwargs):
        pass


class IdentityScheduler(RLScheduler):
    def update(self, current_lr, entropy_coef, epoch, frames, kl_dist, **kwargs):
        return current_lr, entropy_coef
