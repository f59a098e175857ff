"""Back-end part of __graft_entry__.smoke(): one small window through the HIP solver + marginalisation, checked against the oracle."""
import numpy as np


def run():
    import gfamd
    import oracle_py
    import synth_window as SW
    w0 = SW.make_window(21, oracle_py, max_features=60, n_landmarks=90)
    wo, wg = w0.copy(), w0.copy()
    so = oracle_py.ba_solve(wo, 4)
    est = gfamd.Estimator(max_features=60, max_visual=600)
    sg = est.solve([wg], 4)[0]
    assert sg["iterations"] == so["iterations"] and sg["successful_steps"] == so["successful_steps"]
    pa, pb = wo["para_Pose"].reshape(-1, 7), wg["para_Pose"].reshape(-1, 7)
    assert np.abs(pa[:, :3] - pb[:, :3]).max() < 1e-6, "positions differ from the oracle"
    assert min(np.abs(pa[:, 3:] - pb[:, 3:]).max(), np.abs(pa[:, 3:] + pb[:, 3:]).max()) < 5e-7, "rotations differ from the oracle"
    pg = est.marginalize([wg], 0)[0]
    po = oracle_py.ba_marginalize(wo, 0)
    assert pg is not None and np.array_equal(pg["block_id"], po["block_id"]) and pg["n"] == po["n"]
    est.close()
    print("backend smoke ok: %d iterations, cost %.3f -> %.3f, prior n=%d" % (sg["iterations"], sg["initial_cost"], sg["final_cost"], pg["n"]))
