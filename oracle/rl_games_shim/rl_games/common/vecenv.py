"""rl_games.common.vecenv — imported by the agents, unused on the update path."""
