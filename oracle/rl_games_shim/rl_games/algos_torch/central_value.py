"""rl_games.algos_torch.central_value — imported by common_agent.py:11; unused (no central value
in any reference config)."""


class CentralValueTrain:  # pragma: no cover
    def __init__(self, *a, **k):
        raise NotImplementedError("central value nets are not part of the hot path")
