"""rl_games.algos_torch.torch_ext (1.1.4) — the functions the reference calls, restated."""
import torch


def policy_kl(p0_mu, p0_sigma, p1_mu, p1_sigma, reduce=True):
    c1 = torch.log(p1_sigma / p0_sigma + 1e-5)
    c2 = (p0_sigma ** 2 + (p1_mu - p0_mu) ** 2) / (2.0 * (p1_sigma ** 2 + 1e-5))
    c3 = -1.0 / 2.0
    kl = c1 + c2 + c3
    kl = kl.sum(dim=-1)
    if reduce:
        return kl.mean()
    return kl


def mean_list(val):
    return torch.mean(torch.stack(val))


def shape_whc_to_cwh(shape):
    if len(shape) == 3:
        return (shape[2], shape[0], shape[1])
    return shape


def get_mean_std_with_masks(values, masks):
    sum_mask = masks.sum()
    values_mask = values * masks
    values_mean = values_mask.sum() / sum_mask
    min_sqr = ((values_mask ** 2) / sum_mask).sum() - ((values_mask / sum_mask).sum()) ** 2
    values_std = torch.sqrt(min_sqr * sum_mask / (sum_mask - 1))
    return values_mean, values_std


def normalization_with_masks(values, masks):
    values_mean, values_std = get_mean_std_with_masks(values, masks)
    return (values - values_mean) / (values_std + 1e-8)


def save_checkpoint(filename, state):
    torch.save(state, filename + '.pth')


def load_checkpoint(filename):
    return torch.load(filename, weights_only=False)
