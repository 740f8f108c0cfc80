"""The boundary tests of tests/test_boundary_emu.py (training loop, rollout action head, latents, players, HRL env_step over the
frozen low-level controller) re-run through libase_hip.so on the GPU: same functions, HIP backend, device tensors."""
import pytest

import tests.test_boundary_emu as T

pytestmark = pytest.mark.gpu


@pytest.fixture(autouse=True)
def _on_gpu(monkeypatch):
    from ase_amd.backend import HipBackend
    monkeypatch.setattr(T, '_DEV', 'cuda:0')
    monkeypatch.setattr(T, '_BE', lambda: HipBackend('cuda:0'))


@pytest.mark.parametrize('kind', ['ase', 'amp', 'ppo'])
def test_train_loop_gpu(kind, golden_dir, tmp_path):
    T.test_train_loop_runs_like_runner(kind, golden_dir, tmp_path)


def test_rollout_action_head_gpu(golden_dir):
    T.test_rollout_action_head_semantics(golden_dir)


def test_latents_follow_progress_gpu(golden_dir):
    T.test_latents_follow_progress(golden_dir)


@pytest.mark.parametrize('kind', ['ase', 'amp', 'ppo'])
def test_player_restores_agent_checkpoint_gpu(kind, golden_dir, tmp_path):
    T.test_player_restores_agent_checkpoint(kind, golden_dir, tmp_path)


def test_hrl_env_step_gpu(golden_dir, tmp_path):
    T.test_hrl_env_step_with_frozen_llc(golden_dir, tmp_path)


def test_hrl_train_and_player_gpu(golden_dir, tmp_path):
    T.test_hrl_train_and_player(golden_dir, tmp_path)


def test_hrl_env_step_matches_reference_golden_gpu(golden_dir):
    T.test_hrl_env_step_matches_reference_golden(golden_dir)


@pytest.mark.parametrize('local_root,root_h', [(True, True), (False, False)])
def test_fetch_amp_obs_demo_gpu(local_root, root_h, golden_dir):
    T.test_fetch_amp_obs_demo_matches_reference(local_root, root_h, golden_dir)
