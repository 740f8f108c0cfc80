"""Host-side mirror of the reference's rl_games plugin interface for the update path
(``ase/learning/*`` under /root/reference): same class names, method names, config keys and
checkpoint layout; the arithmetic runs in libase_hip.so."""
