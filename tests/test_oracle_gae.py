"""Pins the GAE oracle (oracle/gae_oracle.c) to the reference (fixture G3, produced by importing the reference:
tests/golden/make_golden.py).  np2 mode must be bit-exact; f64 mode (what the HIP kernel implements) must sit
inside the stated tolerance atol 1e-5 + rtol 1e-5."""
import numpy as np
import pytest

from oracle import gae as ogae


def cases(golden):
    g = golden("g3_gae")
    for c in range(int(g["n_cases"])):
        p = f"c{c}."
        std = g[p + "ret_std"]
        yield c, dict(rews=g[p + "rews"], dones=g[p + "dones"], trunc=g[p + "trunc"], values=g[p + "values"],
                      std=None if np.isnan(std) else std,
                      gamma=float(g[p + "gamma"]) if p + "gamma" in g else 0.99,
                      lmbda=float(g[p + "lmbda"]) if p + "lmbda" in g else 0.95,
                      vt=g[p + "value_targets"], adv=g[p + "advantages"], ret=g[p + "returns"])


def test_np2_mode_is_bit_exact_against_reference(golden):
    n = 0
    for c, k in cases(golden):
        vt, adv, ret = ogae.gae(k["rews"], k["dones"], k["trunc"], k["values"], k["gamma"], k["lmbda"], k["std"], "np2")
        assert np.array_equal(vt, k["vt"]), c
        assert np.array_equal(adv, k["adv"]), c
        assert np.array_equal(ret, k["ret"]), c
        n += 1
    assert n == 19


def test_f64_mode_within_tolerance_of_reference(golden):
    for c, k in cases(golden):
        vt, adv, ret = ogae.gae(k["rews"], k["dones"], k["trunc"], k["values"], k["gamma"], k["lmbda"], k["std"], "f64")
        np.testing.assert_allclose(vt, k["vt"], rtol=1e-5, atol=1e-5)
        np.testing.assert_allclose(adv, k["adv"], rtol=1e-5, atol=1e-5)
        np.testing.assert_allclose(ret, k["ret"], rtol=1e-5, atol=1e-5)


def test_c_and_python_affine_scan_forms_agree(golden):
    for c, k in cases(golden):
        a = ogae.gae(k["rews"], k["dones"], k["trunc"], k["values"], k["gamma"], k["lmbda"], k["std"], "f64")
        b = ogae.gae_python(k["rews"], k["dones"], k["trunc"], k["values"], k["gamma"], k["lmbda"], k["std"])
        for x, y in zip(a, b):
            np.testing.assert_allclose(x, y, rtol=1e-12, atol=1e-12)


def test_empty_and_single():
    vt, adv, ret = ogae.gae(np.zeros(0, np.float32), np.zeros(0, np.float32), np.zeros(0), np.zeros(1, np.float32))
    assert vt.shape == (0,) and adv.shape == (0,) and ret.shape == (0,)
    vt, adv, ret = ogae.gae([2.0], [0.0], [1.0], [0.5, 3.0], 0.99, 0.95, None)
    assert adv[0] == np.float32(2.0 + 0.99 * 3.0 - 0.5) and ret[0] == 2.0
    assert vt[0] == np.float32(0.5 + (2.0 + 0.99 * 3.0 - 0.5))


def test_truncation_bootstraps_from_next_slot_quirk_q3():
    # truncated (not done) step bootstraps with values[t+1] even though the trajectory changes there
    vt, adv, ret = ogae.gae([1.0, 1.0], [0, 0], [1.0, 1.0], [0.0, 10.0, 20.0], 0.5, 1.0, None)
    assert adv[0] == np.float32(1.0 + 0.5 * 10.0) and adv[1] == np.float32(1.0 + 0.5 * 20.0 - 10.0)
    assert list(ret) == [1.0, 1.0]
