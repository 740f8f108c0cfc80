"""Restated subset of rl-games==1.1.4 (see ../README.md). Test infrastructure only."""
