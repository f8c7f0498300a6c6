"""Device-resident rollout of a vectorised environment: VectorAgentManager against the
numpy restatement of the reference's trajectory assembly (oracle/host.py::lockstep_rollout), and the whole Learner loop
running on it."""
import json
import os
import sys

import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu
sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))

import synthetic_env  # noqa: E402
from oracle import host, nets  # noqa: E402


@pytest.mark.parametrize("standardize", [True, False, "per_feature"])
def test_vector_rollout_matches_trajectory_assembly(standardize):
    per_feature = standardize == "per_feature"   # every feature with its own statistics (not the reference)
    standardize = bool(standardize)
    from rlgym_ppo_amd.batched_agents import VectorAgentManager
    from rlgym_ppo_amd.ppo import DiscreteFF
    torch.manual_seed(11)
    pol = DiscreteFF(107, 90, (32, 32), "cuda:0")
    params = [(l.weight.detach().cpu(), l.bias.detach().cpu()) for l in pol.arena.linears]
    mgr = VectorAgentManager(pol, seed=5, standardize_obs=standardize)
    mgr.per_feature_obs_standardization = per_feature
    d, n_act, code = mgr.init_processes(0, lambda: synthetic_env.SyntheticVectorEnv(seed=3))
    assert (d, n_act, code) == (107, 90, 0)

    env = synthetic_env.SyntheticVectorEnv(seed=3)           # the oracle's own copy of the environment

    def act_fn(obs):                                         # the reference's sampling chain on the CPU
        probs = nets.discrete_probs(params, obs)
        a, lp = nets.discrete_sample(probs, nets.draw_exp_noise(obs.shape[0], 90))
        return a.numpy().astype(np.float32).reshape(-1, 1), lp.numpy()

    step_fn = lambda a: env.step(a)[:4]
    state = None
    reset_obs = env.reset()
    for n_req in (16 * 9, 16 * 5 - 3):                       # second request is rounded up to whole steps: 5 per agent
        torch.manual_seed(100 + n_req)
        exp, _, n_col, _ = mgr.collect_timesteps(n_req)
        torch.manual_seed(100 + n_req)
        T = -(-n_req // 16)
        ref, state = host.lockstep_rollout(reset_obs, step_fn, act_fn, T, standardize=standardize, state=state, per_feature=per_feature)
        assert n_col == 16 * T and mgr.value_input_rows.shape[0] == n_col + 1
        states, actions, logp, rews, nxt, dones, trunc = [x.cpu().numpy() for x in exp]
        assert np.array_equal(actions, ref[1])                                  # action indices: exact
        np.testing.assert_allclose(states[:, :107], ref[0], rtol=1e-6, atol=1e-7)
        assert (states[:, 107:] == 0).all()
        np.testing.assert_allclose(logp, ref[2], rtol=1e-5, atol=1e-6)
        assert np.array_equal(rews, ref[3]) and np.array_equal(dones, ref[5]) and np.array_equal(trunc, ref[6])
        np.testing.assert_allclose(nxt[:, :107], ref[4], rtol=1e-6, atol=1e-7)
        np.testing.assert_allclose(mgr.value_input_rows[n_col, :107].cpu().numpy(), ref[4][-1], rtol=1e-6, atol=1e-7)
        assert trunc.reshape(16, T)[:, -1].tolist() == [1.0 - x for x in dones.reshape(16, T)[:, -1].tolist()]
    assert mgr.cumulative_timesteps == 16 * 9 + 16 * 5
    mgr.cleanup()


def test_collect_learn_collect_consumes_the_reference_noise_stream_with_prefetch():
    """[r4] The next collect's noise is drawn AHEAD, during add_new_experience + learn (engine.HostExponential.prefetch, called at the
    end of a collect) and parked in HBM in one copy when the burst is complete; both are speculative and must be transparent.  A
    collect -> add_new_experience -> learn -> collect sequence of the product (768 agents x 90 actions: above the look-ahead
    threshold) against the oracle's lock-step rollout that draws torch.empty(n, 90).exponential_(1) step by step: the second
    collect's action indices are the oracle's (only a learn() that left the CPU generator alone, and a prefetch that predicted
    its states exactly, give that), the generator ends in the reference's state, and every draw of the second collect was served
    from the chain (hits), the prefetched ones with their noise already resident in HBM (resident_hits)."""
    from rlgym_ppo_amd import Learner, engine
    na, T = 768, 6
    mk = lambda: synthetic_env.SyntheticVectorEnv(n_agents=na, seed=9)
    import contextlib, io
    with contextlib.redirect_stdout(io.StringIO()):
        learner = Learner(mk, vector_env=True, n_proc=1, timestep_limit=10**9, exp_buffer_size=na * T, ts_per_iteration=na * T,
                          ppo_epochs=2, ppo_batch_size=na * T, ppo_minibatch_size=na * T // 2, policy_layer_sizes=(64, 64),
                          critic_layer_sizes=(64, 64), checkpoints_save_folder=None, checkpoint_load_folder=None, save_every_ts=10**12,
                          log_to_wandb=False, random_seed=5, standardize_obs=False)
    try:
        pol = learner.ppo_learner.policy
        assert pol.noise_mode == "host"
        engine._HOST_EXP = None
        torch.manual_seed(4321)
        exp1, _, n1, _ = learner.agent.collect_timesteps(na * T)
        h = engine._HOST_EXP
        assert len(h._chain) == T                     # the next collect's T draws are in the chain before learn() starts
        a1 = exp1[1].cpu().numpy().copy()
        params1 = [(l.weight.detach().cpu().clone(), l.bias.detach().cpu().clone()) for l in pol.arena.linears]
        learner.add_new_experience(exp1)
        with contextlib.redirect_stdout(io.StringIO()):
            learner.ppo_learner.learn(learner.experience_buffer)
        params2 = [(l.weight.detach().cpu().clone(), l.bias.detach().cpu().clone()) for l in pol.arena.linears]
        hits0, pre0 = h.hits, h.resident_hits
        exp2, _, n2, _ = learner.agent.collect_timesteps(na * T)
        a2 = exp2[1].cpu().numpy().copy()
        s_got = torch.get_rng_state()
        assert h.hits - hits0 == T and h.misses == 1 and h.resident_hits - pre0 == T - h.depth   # (the first `depth` draws: the ordinary look-ahead)
    finally:
        learner.agent.cleanup()
    # the oracle: the same interaction, noise drawn on the spot from the same seed
    env = mk()
    torch.manual_seed(4321)
    state, acts = None, []
    reset_obs = env.reset()
    for params in (params1, params2):
        def act_fn(obs, params=params):
            probs = nets.discrete_probs(params, obs)
            a, lp = nets.discrete_sample(probs, nets.draw_exp_noise(obs.shape[0], 90))
            return a.numpy().astype(np.float32).reshape(-1, 1), lp.numpy()
        ref, state = host.lockstep_rollout(reset_obs, lambda a: env.step(a)[:4], act_fn, T, standardize=False, state=state)
        acts.append(ref[1])
    assert torch.equal(s_got, torch.get_rng_state())                 # 2 T draws of [768, 90], nothing else, in the reference's order
    for got, want in zip((a1, a2), acts):
        diff = (got.reshape(-1) != want.reshape(-1)).sum()
        assert diff <= 2, diff                                       # (a near-tie of p/q may flip under an ulp of the probabilities)


def test_learner_loop_on_vector_env(tmp_path, capsys):
    from rlgym_ppo_amd import Learner
    learner = Learner(synthetic_env.make_vector_env, vector_env=True, n_proc=1, timestep_limit=1500, exp_buffer_size=1024,
                      ts_per_iteration=512, ppo_epochs=2, ppo_batch_size=512, ppo_minibatch_size=256,
                      policy_layer_sizes=(64, 64), critic_layer_sizes=(64, 64), checkpoints_save_folder=str(tmp_path / "ck"),
                      add_unix_timestamp=False, save_every_ts=1000, checkpoint_load_folder=None, random_seed=3)
    try:
        learner._learn()
    finally:
        learner.agent.cleanup()
    out = capsys.readouterr().out
    assert out.count("BEGIN ITERATION REPORT") == 3 and "Policy Entropy" in out
    assert learner.agent.cumulative_timesteps == 3 * 512 and learner.epoch == 3
    assert len(learner.experience_buffer) == 1024
    assert learner.ppo_learner.cumulative_model_updates == 2 * (1 + 2 + 2)
    assert learner.agent.average_reward is not None and np.isfinite(learner.agent.average_reward)
    b = learner.experience_buffer
    assert b.states.shape == (1024, 107) and b.next_states.shape == (1024, 107) and b.actions.shape[0] == 1024
    assert torch.isfinite(b.advantages).all() and torch.isfinite(b.values).all()


def test_welford_increment_bit_exact():
    """rlppo_welford_increment == WelfordRunningStat.increment (sample-by-sample float32), bit for bit, from a non-trivial
    state, for the observation statistics (d = 107, 4096 samples) and the return statistics (d = 1)."""
    import ctypes
    from rlgym_ppo_amd import _native as N
    L = N.lib()
    P = lambda t: ctypes.c_void_p(t.data_ptr())
    st = lambda: ctypes.c_void_p(torch.cuda.current_stream().cuda_stream)
    rs = np.random.RandomState(0)
    for d, n, ld in ((107, 4096, 107), (1, 150, 1), (231, 700, 256)):
        w = host.Welford(d)
        warm = (rs.randn(37, d) * 3 + 1).astype(np.float32)
        w.increment(warm, 37)
        x = (rs.randn(n, d) * 2 + 0.5).astype(np.float32)
        mean, m2 = torch.from_numpy(w.mean_.copy()).cuda(), torch.from_numpy(w.m2.copy()).cuda()
        xp = np.zeros((n, ld), np.float32)
        xp[:, :d] = x
        xd = torch.from_numpy(xp).cuda()
        N.check(L.rlppo_welford_increment(st(), P(xd), ld, n, d, P(mean), P(m2), w.count, 0))
        w.increment(x, n)
        assert np.array_equal(mean.cpu().numpy(), w.mean_) and np.array_equal(m2.cpu().numpy(), w.m2)


def test_welford_device_forms_follow_the_state_dtype():
    """After WelfordRunningStat.from_json the state is float64 (np.asarray of Python floats, running_stats.py:121-125) and the
    reference keeps updating in float64.  device_stats.increment / merge must do the same -- bit for bit what the host class
    computes -- instead of reinterpreting the doubles as floats (the resume corruption the round-1 advisor found)."""
    from rlgym_ppo_amd.util import WelfordRunningStat, device_stats
    rs = np.random.RandomState(1)
    for d in (107, 1):
        src = WelfordRunningStat(d)
        src.increment((rs.randn(41, d) * 2 + 1).astype(np.float32), 41)
        x = (rs.randn(300, d) * 3 - 0.5).astype(np.float32)
        other = WelfordRunningStat(d)
        other.increment((rs.randn(29, d) + 4).astype(np.float32), 29)
        for f64 in (False, True):
            a, b = WelfordRunningStat(d), WelfordRunningStat(d)
            for w in (a, b):
                if f64:
                    w.from_json(json.loads(json.dumps(src.to_json())))
                    assert w.running_mean.dtype == np.float64
                else:
                    w.deserialize(src.serialize())
                    w.running_mean = np.asarray(w.running_mean, np.float32)
                    w.running_variance = np.asarray(w.running_variance, np.float32)
            a.increment(x, 300)                                   # host class (the reference's arithmetic)
            device_stats.increment(b, torch.from_numpy(x).cuda())
            assert b.count == a.count and b.running_mean.dtype == a.running_mean.dtype
            assert np.array_equal(a.running_mean, b.running_mean) and np.array_equal(a.running_variance, b.running_variance)
            a.increment_from_serialized_other(other.serialize())
            device_stats.merge(b, other.serialize(), "cuda:0")
            assert b.count == a.count
            assert np.array_equal(np.asarray(a.running_mean), b.running_mean)
            assert np.array_equal(np.asarray(a.running_variance), b.running_variance)


def test_vector_env_run_resumes_with_sane_observation_statistics(tmp_path, capsys):
    """save -> load("latest") -> collect on a vector_env run: the reloaded (float64) statistics keep evolving exactly as the
    host class would evolve them, and the standardised rows stay finite."""
    from rlgym_ppo_amd import Learner
    from rlgym_ppo_amd.util import WelfordRunningStat
    kw = dict(vector_env=True, n_proc=1, exp_buffer_size=1024, ts_per_iteration=512, ppo_epochs=1, ppo_batch_size=512,
              ppo_minibatch_size=256, policy_layer_sizes=(32, 32), critic_layer_sizes=(32, 32),
              checkpoints_save_folder=str(tmp_path / "ck"), add_unix_timestamp=False, save_every_ts=500, random_seed=3)
    first = Learner(synthetic_env.make_vector_env, timestep_limit=600, checkpoint_load_folder=None, **kw)
    try:
        first._learn()
    finally:
        first.agent.cleanup()
    second = Learner(synthetic_env.make_vector_env, timestep_limit=10_000, checkpoint_load_folder="latest", **kw)
    try:
        st = second.agent.obs_stats
        assert st.running_mean.dtype == np.float64 and st.count == first.agent.obs_stats.count
        np.testing.assert_allclose(st.running_mean, first.agent.obs_stats.running_mean, rtol=1e-6)
        shadow = WelfordRunningStat(1)
        shadow.from_json(st.to_json())
        seen = []
        orig = second.agent._increment_obs_stats
        second.agent._increment_obs_stats = lambda obs: (seen.append(obs.cpu().numpy()), orig(obs))[1]
        exp, _, n, _ = second.agent.collect_timesteps(512)
        assert len(seen) >= 2
        for obs in seen:
            shadow.increment(obs, obs.shape[0])
        assert st.count == shadow.count and st.running_mean.dtype == np.float64
        assert np.array_equal(st.running_mean, shadow.running_mean) and np.array_equal(st.running_variance, shadow.running_variance)
        rows = exp[0]
        assert torch.isfinite(rows).all() and rows.abs().max().item() < 20.0   # first rows are the raw reset observations
        assert 0.05 < float(st.std[0]) < 50
    finally:
        second.agent.cleanup()


def test_vector_rollout_gaussian_policy():
    """Same comparison for the continuous head (actions [n, k] fp32, summed log-probs)."""
    from rlgym_ppo_amd.batched_agents import VectorAgentManager
    from rlgym_ppo_amd.ppo import ContinuousPolicy
    torch.manual_seed(21)
    pol = ContinuousPolicy(107, 8, (32, 32), "cuda:0")
    params = [(l.weight.detach().cpu(), l.bias.detach().cpu()) for l in pol.arena.linears]
    mgr = VectorAgentManager(pol, seed=5, standardize_obs=True)
    make = lambda: synthetic_env.SyntheticVectorEnv(n_actions=4, n_agents=12, seed=8, kind="continuous")
    d, n_act, code = mgr.init_processes(0, make)
    assert (d, n_act, code) == (107, 4, 2)
    env = make()

    def act_fn(obs):
        mean, std = nets.gauss_out(params, obs)
        a, lp = nets.gauss_sample(mean, std, torch.empty(obs.shape[0], 4).normal_(0, 1))
        return a.numpy(), lp.numpy()

    state, reset_obs = None, env.reset()
    for n_req in (12 * 7, 12 * 3):
        torch.manual_seed(7 + n_req)
        exp, _, n_col, _ = mgr.collect_timesteps(n_req)
        torch.manual_seed(7 + n_req)
        ref, state = host.lockstep_rollout(reset_obs, lambda a: env.step(a)[:4], act_fn, n_req // 12, state=state)
        states, actions, logp, rews, nxt, dones, trunc = [x.cpu().numpy() for x in exp]
        np.testing.assert_allclose(actions, ref[1], rtol=1e-5, atol=2e-6)
        np.testing.assert_allclose(states[:, :107], ref[0], rtol=1e-6, atol=1e-7)
        np.testing.assert_allclose(logp, ref[2], rtol=1e-4, atol=1e-4)
        np.testing.assert_allclose(rews, ref[3], rtol=1e-5, atol=1e-6)   # rewards depend on the (fp32-close) actions
        assert np.array_equal(dones, ref[5]) and np.array_equal(trunc, ref[6])
    mgr.cleanup()
