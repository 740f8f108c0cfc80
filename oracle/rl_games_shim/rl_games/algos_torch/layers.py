"""rl_games.algos_torch.layers — imported by the reference builders, nothing used."""
