"""[r6] The learner-side collection loop in C++ (csrc/collector.cpp, rlppo_collector_*) against the Python loop it replaces
(BatchedAgentManager with native_collect = False: the statement of rlgym_ppo/batched_agents/batched_agent_manager.py:126-350), on real
worker processes speaking the reference's wire format: with ONE worker the message order is deterministic, so the two loops must agree
value for value -- trajectories, forced truncation at the flush, observation statistics (bit for bit), the running average reward, the
cadence counter, the metrics records -- over several collect_timesteps calls (an action is in flight across each boundary).  CPU only."""
import numpy as np
import pytest
import torch

import synthetic_env


class _DiscretePolicy:
    def get_action(self, obs, standardize=None):
        obs = np.asarray(obs, np.float32)
        a = (np.abs(obs[:, :5]).sum(1) * 7).astype(np.int64) % 7
        return torch.as_tensor(a), torch.as_tensor(-np.abs(obs[:, 0]).astype(np.float32))


class _ContinuousPolicy:
    def get_action(self, obs, standardize=None):
        obs = np.asarray(obs, np.float32)
        return torch.as_tensor(np.tanh(obs[:, :3] * 0.3).astype(np.float32)), torch.as_tensor(-np.abs(obs[:, 1]).astype(np.float32))


def _run(native, env_fn, policy, calls, n_proc=1, metrics_fn=None, standardize=True, per_feature=False, min_inference_size=1, restored=None):
    from rlgym_ppo_amd.batched_agents import BatchedAgentManager
    from rlgym_ppo_amd.util import WelfordRunningStat
    mgr = BatchedAgentManager(policy, min_inference_size=min_inference_size, seed=5, standardize_obs=standardize)
    mgr.native_collect = native
    mgr.per_feature_obs_standardization = per_feature
    try:
        shapes = mgr.init_processes(n_proc, env_fn, collect_metrics_fn=metrics_fn, shm_buffer_size=4096)
        if restored is not None:   # what Learner.load does with a checkpoint's statistics: float64 arrays from JSON (learner.py:308-310)
            mgr.obs_stats = WelfordRunningStat(1)
            mgr.obs_stats.from_json(restored)
        out = [mgr.collect_timesteps(k) for k in calls]
        assert (mgr._native is not None) == native, "the loop that ran is not the one the test asked for"
        st = mgr.obs_stats
        state = dict(avg=mgr.average_reward, total=mgr.cumulative_timesteps, since=mgr.steps_since_obs_stats_update,
                     stats=None if st is None else (st.running_mean.copy(), st.running_variance.copy(), st.count))
        return shapes, out, state
    finally:
        mgr.cleanup()


def _same(a, b):
    (sa, oa, ta), (sb, ob, tb) = a, b
    assert sa == sb
    assert ta["total"] == tb["total"] and ta["since"] == tb["since"]
    assert (ta["avg"] is None) == (tb["avg"] is None) and (ta["avg"] is None or ta["avg"] == tb["avg"])   # doubles, same order of operations
    if ta["stats"] is not None:
        for x, y in zip(ta["stats"][:2], tb["stats"][:2]):
            assert x.dtype == y.dtype == np.float32 and np.array_equal(x, y)                               # float32 Welford: bit for bit
        assert ta["stats"][2] == tb["stats"][2]
    for (ea, ma, na, _), (eb, mb, nb, _) in zip(oa, ob):
        assert na == nb and len(ma) == len(mb)
        for x, y in zip(ma, mb):
            assert np.asarray(x).shape == np.asarray(y).shape and np.array_equal(x, y)
        for x, y, name in zip(ea, eb, ("states", "actions", "log_probs", "rewards", "next_states", "dones", "truncated")):
            # (`truncated` is a list of python floats AND the ints the flush writes: numpy makes it float64 -- quirk Q2 -- unless every
            # sequence of the call is one step long, then int64; the C++ loop always hands out float64)
            assert x.shape == y.shape and (x.dtype == y.dtype or name == "truncated"), (name, x.dtype, y.dtype, x.shape, y.shape)
            assert np.array_equal(x, y), name


@pytest.mark.parametrize("case", ["discrete", "metrics", "continuous", "per_feature", "raw_obs", "varying_team", "single_agent_rank1"])
def test_native_loop_equals_the_python_loop_with_one_worker(case):
    kw = dict(env_fn=synthetic_env.make_wire_env, policy=_DiscretePolicy(), calls=(40, 17, 1, 33))
    if case == "metrics":
        kw.update(env_fn=synthetic_env.make_single_env, metrics_fn=synthetic_env.step_count_metrics, calls=(25, 9))
    elif case == "single_agent_rank1":
        kw.update(env_fn=synthetic_env.make_single_env, calls=(30, 7))
    elif case == "continuous":
        kw.update(env_fn=synthetic_env.make_continuous_wire_env, policy=_ContinuousPolicy())
    elif case == "per_feature":
        kw.update(per_feature=True)
    elif case == "raw_obs":
        kw.update(standardize=False)
    elif case == "varying_team":
        kw.update(env_fn=synthetic_env.make_varying_env, metrics_fn=synthetic_env.step_count_metrics, calls=(30, 11, 26))
    py, nat = _run(False, **kw), _run(True, **kw)
    _same(py, nat)
    states = nat[1][0][0][0]
    assert len(states) > 0
    if case == "varying_team":   # the flush at a team-size change really happened: some next-state rows are the zero padding
        nxt = np.concatenate([o[0][4] for o in nat[1]])
        assert (np.abs(nxt).sum(1) == 0).any()


@pytest.mark.parametrize("per_feature,count", [(False, 40), (True, 40), (True, 1)])
def test_native_loop_with_statistics_restored_from_a_checkpoint(per_feature, count, capsys):
    """A resumed run: WelfordRunningStat.from_json leaves float64 arrays (running_stats.py:120-125), numpy then forms the standardised
    observation in float64 and it is rounded to float32 where it meets the policy / the experience buffer.  The C++ loop does the same
    -- float64 statistics advanced in place bit for bit, (x - mean) / std in double, one rounding -- so a resumed run keeps the fast
    loop: equal to the Python loop's float64 arrays once those are rounded to float32.  (count 1: the constant mean / std branch.)"""
    rs = np.random.RandomState(3)
    book = {"mean": (rs.randn(13) * 0.3).tolist(), "var": (np.abs(rs.randn(13)) * 30 + 5).tolist(), "shape": [13], "count": count}
    kw = dict(env_fn=synthetic_env.make_wire_env, policy=_DiscretePolicy(), calls=(40, 17), per_feature=per_feature, restored=book)
    (s0, o0, t0), (s1, o1, t1) = _run(False, **kw), _run(True, **kw)
    assert s0 == s1 and t0["total"] == t1["total"] and t0["since"] == t1["since"] and t0["avg"] == t1["avg"]
    for x, y in zip(t0["stats"][:2], t1["stats"][:2]):
        assert x.dtype == y.dtype == np.float64 and np.array_equal(x, y)
    assert t0["stats"][2] == t1["stats"][2] > count
    for (ea, _, na, _), (eb, _, nb, _) in zip(o0, o1):
        assert na == nb
        for x, y, name in zip(ea, eb, ("states", "actions", "log_probs", "rewards", "next_states", "dones", "truncated")):
            assert x.shape == y.shape and np.array_equal(np.asarray(x, y.dtype), y), name
        assert ea[0].dtype == np.float64 and eb[0].dtype == np.float32


def test_native_loop_with_three_workers_is_structurally_right():
    """Three processes: arrival order is timing dependent, so (as for the Python loop) the check is structural -- every stored action
    is the policy's action for the stored state, flags are 0 / 1, the flush marks the last step, the statistics advanced."""
    pol = _DiscretePolicy()
    shapes, out, state = _run(True, synthetic_env.make_wire_env, pol, (90, 40), n_proc=3, min_inference_size=2)
    assert shapes == (13, 7, 0)
    total = 0
    for (states, actions, log_probs, rewards, next_states, dones, truncated), metrics, n, _ in out:
        total += n
        assert n >= 40 and len(states) > 0 and states.shape[1] == 13 and next_states.shape == states.shape
        a, lp = pol.get_action(states)
        assert np.array_equal(actions, a.numpy().astype(np.float32)) and np.allclose(log_probs, lp.numpy())
        assert rewards.dtype == dones.dtype == truncated.dtype == np.float64 and states.dtype == np.float32
        assert set(np.unique(dones)) <= {0.0, 1.0} and set(np.unique(truncated)) <= {0.0, 1.0}
        assert (dones + truncated).max() <= 1.0 and (dones + truncated)[-1] == 1.0
        assert np.abs(next_states).max() <= 5.0 and len(metrics) > 0
    assert state["total"] == total and state["stats"][2] > 6 and state["avg"] is not None


def test_signals_interrupt_the_native_wait_without_changing_the_results():
    """A signal that arrives while the C++ loop waits for workers makes it hand control back (RLPPO_ERR_INTERRUPTED) so that python's
    handlers run -- Ctrl-C works -- and the wait is resumed: with a 1 kHz timer firing throughout, the rollout is still the Python
    loop's, value for value; a handler that raises ends the collection with its exception."""
    import signal
    kw = dict(env_fn=synthetic_env.make_wire_env, policy=_DiscretePolicy(), calls=(60, 25))
    py = _run(False, **kw)
    ticks = []
    old = signal.signal(signal.SIGALRM, lambda *a: ticks.append(1))
    signal.setitimer(signal.ITIMER_REAL, 0.001, 0.001)
    try:
        nat = _run(True, **kw)
    finally:
        signal.setitimer(signal.ITIMER_REAL, 0, 0)
        signal.signal(signal.SIGALRM, old)
    assert len(ticks) > 10
    _same(py, nat)

    class Stop(Exception):
        pass

    def raising(*a):
        raise Stop()
    from rlgym_ppo_amd.batched_agents import BatchedAgentManager
    mgr = BatchedAgentManager(_DiscretePolicy(), min_inference_size=1, seed=5, standardize_obs=True)
    old = signal.signal(signal.SIGALRM, raising)
    try:
        mgr.init_processes(1, synthetic_env.make_wire_env, shm_buffer_size=4096)
        signal.setitimer(signal.ITIMER_REAL, 0.05, 0)
        with pytest.raises(Stop):
            mgr.collect_timesteps(10_000_000)
    finally:
        signal.setitimer(signal.ITIMER_REAL, 0, 0)
        signal.signal(signal.SIGALRM, old)
        mgr.cleanup()


def test_collector_entry_points_report_errors_as_codes_and_text():
    import ctypes
    from rlgym_ppo_amd import _native as N
    L = N.lib()
    h = ctypes.c_void_p()
    assert L.rlppo_collector_create(0, None, None, None, 0, 0, ctypes.byref(h)) == 1001 and b"collector_create" in L.rlppo_last_error()
    rows = ctypes.c_int64(0)
    assert L.rlppo_collector_ready(None, None, 0, ctypes.byref(rows)) == 1001 and b"collector_ready" in L.rlppo_last_error()
    assert L.rlppo_collector_send(None, None, 0, None) == 1001 and b"collector_send" in L.rlppo_last_error()
    with pytest.raises(RuntimeError, match="collector_send"):
        N.check(L.rlppo_collector_send(None, None, 0, None))
