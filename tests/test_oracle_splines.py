"""Pins the oracle's EndEffectorSplines restatement against the reference's own known-answer / finite-difference
tests: /root/reference/test/splines_tests.cpp (line ranges cited per test).  CPU only."""
import math

import numpy as np
import pytest

from oracle_py import OracleSpline

F, P = OracleSpline.FORCE, OracleSpline.POSITION
FORCE_MULT = 100.0
MARGIN = 1e-3
TIMES = [0.2 * i for i in range(5)]            # splines_tests.cpp:19-27


def make_splines():
    # splines.emplace_back(num_contacts, times, !start_constant, 3); (..., start_constant, 3)   :29-31
    return [OracleSpline(TIMES, False, 3), OracleSpline(TIMES, True, 3)]


def test_setting_vars():                          # :34-56
    for sp in make_splines():
        times, _ = sp.times()
        for coord in range(3):
            for it in sp.mutable_nodes(P, coord):
                sp.set_vars(P, coord, it, it, 2.0)
                assert sp.value_at(P, coord, times[it]) == it
            for it in sp.mutable_nodes(F, coord):
                sp.set_vars(F, coord, it, it, 2.0 / FORCE_MULT)
                assert sp.value_at(F, coord, times[it]) == it


def test_known_values():                          # :58-107
    s0, s1 = make_splines()
    for coord in range(3):
        for it in s0.mutable_nodes(P, coord):
            s0.set_vars(P, coord, it, it, it - 1)
        assert s0.value_at(P, coord, 0) == 0
        if coord != 2:
            assert abs(s0.value_at(P, coord, 0.103448) - 1.0517) < MARGIN
            assert abs(s0.value_at(P, coord, 0.503448) - 4.62926) < MARGIN
        else:
            assert abs(s0.value_at(P, coord, 0.162069) - 1.67752) < MARGIN
            assert abs(s0.value_at(P, coord, 0.5) - 6) < MARGIN
        for it in s1.mutable_nodes(P, coord):
            s1.set_vars(P, coord, it, it, it - 1)
        assert s1.value_at(P, coord, 0) == 0
        if coord != 2:
            assert abs(s1.value_at(P, coord, 0.103448) - 0.0) < MARGIN
            assert abs(s1.value_at(P, coord, 0.25517) - 0.93156) < MARGIN
        else:
            assert abs(s1.value_at(P, coord, 0.162069) - 0) < MARGIN
            assert abs(s1.value_at(P, coord, 0.25517) - 2.2683) < MARGIN
        for it in s0.mutable_nodes(F, coord):
            s0.set_vars(F, coord, it, it, (it - 1) / FORCE_MULT)
        assert s0.value_at(F, coord, 0) == 0
        assert abs(s0.value_at(F, coord, 0.103448) - 0.0) < MARGIN
        assert abs(s0.value_at(F, coord, 0.26666 + 0.0229885) - 3.27887) < MARGIN


def _check_force_lin(sp, coord, start, total):
    vec = sp.qp_vec(F, coord)
    N = 100.0
    for i in range(100):
        t = i * ((total - start) / N) + start
        if sp.is_force_mutable(t):
            idx, k = sp.vars_idx(F, coord, t)
            lin = sp.lin(F, coord, t)
            assert len(lin) == k
            assert abs(sp.value_at(F, coord, t) - float(vec[idx:idx + k] @ lin)) < MARGIN
        else:
            assert sp.value_at(F, coord, t) == 0


def test_linearisation_identity():                # :109-158
    for sp in make_splines():
        total = sp.end_time()
        for coord in range(3):
            for it in sp.mutable_nodes(P, coord):
                sp.set_vars(P, coord, it, it, 3.1)
            vec = sp.qp_vec(P, coord)
            for i in range(100):
                t = i * (total / 100.0)
                idx, k = sp.vars_idx(P, coord, t)
                lin = sp.lin(P, coord, t)
                assert len(lin) == k
                assert abs(sp.value_at(P, coord, t) - float(vec[idx:idx + k] @ lin)) < MARGIN
            for it in sp.mutable_nodes(F, coord):
                sp.set_vars(F, coord, it, it, 1.4 / FORCE_MULT)
            _check_force_lin(sp, coord, 0.0, total)


def test_add_remove_polys():                      # :160-237
    splines = make_splines()
    for sp in splines:
        for i in range(3):
            sp.add_poly(0.2)
            assert abs(sp.end_time() - (TIMES[-1] + (i + 1) * 0.2)) < MARGIN
        for coord in range(3):
            for it in sp.mutable_nodes(F, coord):
                sp.set_vars(F, coord, it, it - 1, .75 / FORCE_MULT)
            _check_force_lin(sp, coord, 0.0, sp.end_time())
    for sp in splines:
        assert sp.remove_poly(0.5) == 0
        assert abs(sp.end_time() - (TIMES[-1] + 3 * 0.2)) < MARGIN
        for coord in range(3):
            for it in sp.mutable_nodes(F, coord):
                sp.set_vars(F, coord, it, 2 * it - 1, .5 / FORCE_MULT)
            _check_force_lin(sp, coord, sp.start_time(), sp.end_time())
            for i in range(20):
                sp.remove_poly(0.5 + i * 0.1)


def test_knot_pattern_and_counts():
    # contact times [0,.3,.6,.9,1.2] -> 11 knots, 4 mutable force nodes, 3 position vars per coord
    for start in (False, True):
        sp = OracleSpline([0, 0.3, 0.6, 0.9, 1.2], start, 3)
        t, ty = sp.times()
        assert len(t) == 11
        assert len(sp.mutable_nodes(F, 0)) == 4
        assert len(sp.mutable_nodes(P, 0)) == 3 and len(sp.mutable_nodes(P, 1)) == 3
    sw = OracleSpline([0, 0.3, 0.6, 0.9, 1.2], False, 3)
    _, ty = sw.times()
    # swing-first: LO, mid, TD, F, F, LO, mid, TD, F, F, LO  (end_effector_splines.cpp:45-100)
    assert list(ty) == [0, 2, 1, 2, 2, 0, 2, 1, 2, 2, 0]
    st = OracleSpline([0, 0.3, 0.6, 0.9, 1.2], True, 3)
    _, ty = st.times()
    assert list(ty) == [1, 2, 2, 0, 2, 1, 2, 2, 0, 2, 1]
    # the FP-sensitive lookup of SURVEY.md section 7: 6*0.05 = 0.30000000000000004 >= 0.3 selects the node AT 0.3
    assert 6 * 0.05 > 0.3
    assert st.vars_idx(P, 0, 6 * 0.05) == st.vars_idx(P, 0, 0.3)


@pytest.mark.parametrize('which', [0, 1])
def test_value_derivatives_fd(which):             # :239-325
    sp = make_splines()[which]
    dt = math.sqrt(1e-16)
    TOL = 1e-4
    for coord in range(3):
        for it in sp.mutable_nodes(F, coord):
            sp.set_vars(F, coord, it, 2 * it - 1, .5 / FORCE_MULT)
    ct = sp.contact_times()
    sp2 = sp.clone()
    t = 0.0
    while t < sp.end_time():
        for c in range(len(ct)):
            for coord in range(3):
                v1 = sp.value_at(F, coord, t)
                ct2 = ct.copy(); ct2[c] += dt
                sp2.set_contact_times(ct2)
                v2 = sp2.value_at(F, coord, t)
                assert abs(sp.partial_wrt_time(F, coord, t, c) - (v2 - v1) / dt) < TOL, (t, c, coord)
                sp2.set_contact_times(ct)
        t += 0.01
    for coord in range(2):
        for it in sp.mutable_nodes(P, coord):
            sp.set_vars(P, coord, it, 2 * it - 1, .5)
    sp2 = sp.clone()
    t = 0.0
    while t < sp.end_time():
        for c in range(len(ct)):
            for coord in range(2):
                v1 = sp.value_at(P, coord, t)
                ct2 = ct.copy(); ct2[c] += dt
                sp2.set_contact_times(ct2)
                v2 = sp2.value_at(P, coord, t)
                assert abs(sp.partial_wrt_time(P, coord, t, c) - (v2 - v1) / dt) < TOL, (t, c, coord)
                sp2.set_contact_times(ct)
        t += 0.01


@pytest.mark.parametrize('which', [0, 1])
def test_coefficient_derivatives_fd(which):       # :327-443
    sp = make_splines()[which]
    dt = math.sqrt(1e-16)
    TOL = 1e-4
    for coord in range(3):
        for it in sp.mutable_nodes(F, coord):
            sp.set_vars(F, coord, it, 2 * it - 1, .5 / FORCE_MULT)
    ct = sp.contact_times()
    sp2 = sp.clone()
    t = 0.0
    while t < sp.end_time():
        if sp.is_force_mutable(t):
            for c in range(len(ct)):
                for coord in range(3):
                    coefs = sp.lin(F, coord, t)
                    ct2 = ct.copy(); ct2[c] += dt
                    sp2.set_contact_times(ct2)
                    if sp2.is_force_mutable(t):
                        coefs2 = sp2.lin(F, coord, t)
                        assert len(coefs) == len(coefs2)
                        part = sp.coef_partial_wrt_time(F, coord, t, c)
                        assert len(part) == len(coefs)
                        assert np.all(np.abs(part - (coefs2 - coefs) / dt) < TOL), (t, c, coord)
                    sp2.set_contact_times(ct)
        t += 0.01
    for coord in range(3):
        for it in sp.mutable_nodes(P, coord):
            sp.set_vars(P, coord, it, 2 * it - 1, .5 / FORCE_MULT)
    sp2 = sp.clone()
    t = 0.0
    while t < sp.end_time():
        for c in range(len(ct)):
            for coord in range(2):
                coefs = sp.lin(P, coord, t)
                ct2 = ct.copy(); ct2[c] += dt
                sp2.set_contact_times(ct2)
                coefs2 = sp2.lin(P, coord, t)
                if len(coefs) == len(coefs2):
                    part = sp.coef_partial_wrt_time(P, coord, t, c)
                    assert len(part) == len(coefs)
                    assert np.all(np.abs(part - (coefs2 - coefs) / dt) < TOL), (t, c, coord)
                sp2.set_contact_times(ct)
        t += 0.01
