"""Comparison of two Levenberg-Marquardt traces (GPU vs oracle) that sum in different orders.

Accept / reject and the function-tolerance stop are threshold tests on sums: the GPU accumulates with atomics, the oracle in
observation order, so a step that lands within rounding of a threshold may be decided differently -- which only happens once
the cost has settled (the step changes it by less than `settle` relative).  Up to there the traces must agree decision for
decision and cost for cost; from there on only the final cost is compared."""
import numpy as np


def assert_same_trace(sg, so, rtol=1e-7, settle=1e-5):
    """sg / so: summaries with initial_cost, final_cost, trace_len, trace_cost, trace_accepted.  Returns True when the two
    traces are identical in length and decisions (the caller may then compare states tightly)."""
    co, cg = np.array(so.trace_cost[:so.trace_len]), np.array(sg.trace_cost[:sg.trace_len])
    ao, ag = list(so.trace_accepted[:so.trace_len]), list(sg.trace_accepted[:sg.trace_len])
    assert np.isclose(sg.initial_cost, so.initial_cost, rtol=1e-12)
    cur, stop = so.initial_cost, len(co)
    for i in range(len(co)):
        if abs(cur - co[i]) <= settle * abs(cur):
            stop = i
            break
        if ao[i]:
            cur = co[i]
    n = min(stop, len(cg))
    assert n == stop or len(cg) == len(co), (len(cg), len(co), stop)  # (the GPU may only stop early in the settled tail)
    assert ag[:n] == ao[:n], (ag, ao)
    assert np.allclose(cg[:n], co[:n], rtol=rtol, atol=1e-15), (cg, co)
    identical = len(cg) == len(co) and ag == ao
    if identical:
        assert np.allclose(cg, co, rtol=max(rtol, 1e-6), atol=1e-15)
    assert np.isclose(sg.final_cost, so.final_cost, rtol=rtol if identical else max(rtol, 10 * settle), atol=1e-15)
    return identical
