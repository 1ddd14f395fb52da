"""Comparison of two Levenberg-Marquardt traces (GPU vs oracle) that sum in different orders.

assert_identical_trace: the bar of SURVEY 8(c) -- same length, same accept / reject decisions, every cost within rtol (1e-9).
The GPU's sums are reproducible since round 5 (gh_ba_options.deterministic: pre-rounded accumulation, posegraph.hip), so a
test either always meets it or never does.

assert_same_trace (rounds 3-4, kept for A/B runs of the atomics mode, deterministic = 0): accept / reject and the
function-tolerance stop are threshold tests on sums, and with plain atomics a step that lands within rounding of a threshold
may be decided differently from run to run once the cost has settled (the step changes it by less than `settle` relative).
Up to there the traces must agree decision for decision and cost for cost; from there on only the final cost is compared."""
import numpy as np


def assert_identical_trace(sg, so, rtol=1e-9):
    co, cg = np.array(so.trace_cost[:so.trace_len]), np.array(sg.trace_cost[:sg.trace_len])
    ao, ag = list(so.trace_accepted[:so.trace_len]), list(sg.trace_accepted[:sg.trace_len])
    assert np.isclose(sg.initial_cost, so.initial_cost, rtol=1e-12)
    assert len(cg) == len(co) and ag == ao, (ag, ao, cg, co)
    assert sg.iterations == so.iterations
    assert np.allclose(cg, co, rtol=rtol, atol=1e-15), (np.abs(cg - co) / np.abs(co)).max()
    assert np.isclose(sg.final_cost, so.final_cost, rtol=rtol, atol=1e-15)
    return True


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
