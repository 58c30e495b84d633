"""GPU tests of ifd_opt_params.knn_reference_form (validation only): the reference's neighbour choice bug for bug - float32 expanded
form |a|^2 + |b|^2 - 2 a.b in torch's accumulation order, top-6, column 0 dropped whatever it is (ConvONet/defense/pn_utils.py:72-83).

With the switch on, what tests/test_gpu_parity.py has to ATTRIBUTE to the neighbour choice (category B, and the "self column" quirk
of converged point pairs) is GONE: the teacher-forced steps on the trained-like field at Adam t = 100 / 300 / 500 agree with the
reference at every coordinate, no exclusions.  The product path keeps the exact 5-NN; the cost of the difference is reported as the
Chamfer distance between the two runs after 501 steps."""
import os

import numpy as np
import pytest
import torch

from test_gpu_parity import PL, _chamfer, _oracle_restore_from_hip_draws, _ulp_floor

pytestmark = pytest.mark.gpu

HERE = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden")


def _restated_dist(x):
    """The accumulation order the kernel restates (csrc/knn_device.h knn_scan_ref2): dot = fma(z, z', fma(y, y', x x')),
    xx = (x^2 + y^2) + z^2, dist = (xx_j - 2 dot) + xx_i - in numpy, fma through float64 (exact for float32 operands)."""
    def fma(a, b, c):
        return (a.astype(np.float64) * b.astype(np.float64) + c.astype(np.float64)).astype(np.float32)
    xi, xj = x[:, :, None, :], x[:, None, :, :]
    dot = fma(xi[..., 2], xj[..., 2], fma(xi[..., 1], xj[..., 1], xi[..., 0] * xj[..., 0]))
    sq = [x[..., k] * x[..., k] for k in range(3)]
    xxn = (sq[0] + sq[1]) + sq[2]
    return (xxn[:, None, :] + np.float32(-2.0) * dot) + xxn[:, :, None]


def fixture_host_expanded_form_matches_restatement():
    """PINNED (round-5 verdict, weak 3): the restatement against the distance matrix the reference's own knn_point computed on the
    host that made the trajectory fixtures (tests/golden/knn_ref_dist.npz, captured by make_knn_ref_dist.py inside
    ConvONet/defense/pn_utils.py:64-83) - bit for bit, independent of the BLAS of the host this test runs on."""
    z = np.load(os.path.join(HERE, "knn_ref_dist.npz"))
    return bool(np.array_equal(_restated_dist(z["pc"]), z["dist"]))


def host_expanded_form_matches_torch():
    """The same restatement against THIS host's torch CPU kernels (only matters where the oracle is run live: the config #5 test)."""
    g = torch.Generator().manual_seed(5)
    pc = (torch.rand(2, 512, 3, generator=g) - 0.5) * 0.9
    inner = -2.0 * torch.matmul(pc, pc.transpose(2, 1))
    xx = torch.sum(pc.transpose(2, 1) ** 2, dim=1, keepdim=True)
    dist = (xx + inner + xx.transpose(2, 1)).numpy()
    return bool(np.array_equal(_restated_dist(pc.numpy()), dist))


def _trained():
    import ifdefense_amd as I
    z = np.load(os.path.join(HERE, "trained_like_f16.npz"))
    w = {k: z[k].astype(np.float32) for k in z.files}
    f = dict(np.load(os.path.join(HERE, "convonet_golden_trained.npz")))
    long = np.load(os.path.join(HERE, "convonet_golden_trained_long.npz"))
    f.update({k: long[k] for k in long.files if k != "init_points"})
    planes = {pl: torch.from_numpy(f["planes_f16"][:, i].astype(np.float32)) for i, pl in enumerate(PL)}
    return I.Restorer(I.weights.pack_state_dict(w), device="cuda:0"), f, planes, w


def test_reference_form_removes_the_neighbour_choice_differences_on_the_trained_like_field():
    from oracle import convonet_oracle as O
    # the trajectory fixtures come from the fixture host; the strict branch is pinned to ITS arithmetic (a committed tensor), not to
    # the BLAS of the host that runs this test
    assert fixture_host_expanded_form_matches_restatement(), "knn_ref_dist.npz no longer matches the restatement: regenerate the fixtures"
    print("BRANCH: strict (pinned to tests/golden/knn_ref_dist.npz); this host's torch.matmul sums a 3-term dot like the restatement: %s"
          % host_expanded_form_matches_torch())
    r, f, planes, w_np = _trained()
    try:
        for t in (0, 9, 99, 299, 499):
            x = torch.from_numpy(f[f"traj{t}_x"])
            m0, v0 = f[f"traj{t}_m"], f[f"traj{t}_v"]
            res = {}
            for ref_form in (False, True):
                out, (m1, _, _) = r.optimize_points(x, planes, rep_weight=500.0, steps=1, normalize=False, return_state=True,
                                                    state=(torch.from_numpy(m0), torch.from_numpy(v0), t), knn_reference_form=ref_form)
                g = m0 + (m1.cpu().numpy().astype(np.float64) - m0) / 0.1
                g_ref = f[f"traj{t}_g"].astype(np.float64)
                res[ref_form] = (int((np.abs(out.cpu().numpy() - f[f"traj{t}_x_next"]) > 1e-6).sum()),
                                 int((np.abs(g - g_ref).max(-1) > 5e-6 * np.abs(g_ref).max()).sum()),
                                 float(np.abs(g - g_ref).max() / np.abs(g_ref).max()))
            print("trained-like t=%d: coordinates of x_next off by > 1e-6 / points whose gradient differs by > 5e-6 of max / largest gradient "
                  "difference: exact 5-NN %s, reference-form kNN %s" % (t + 1, res[False], res[True]))
            assert res[True][0] == 0 and res[True][1] == 0 and res[True][2] < 5e-6, (t, res[True])      # no exclusions, always
        # what the difference costs after a whole run: exact 5-NN vs the reference's choice, 501 steps + normalisation, 8 clouds
        init = torch.from_numpy(f["init_points"])
        a = r.optimize_points(init, planes, rep_weight=500.0, iterations=500).cpu().numpy()
        b = r.optimize_points(init, planes, rep_weight=500.0, iterations=500, knn_reference_form=True).cpu().numpy()
        ref = f.get("out500_normalised")
        ch = [_chamfer(a[i], b[i]) for i in range(len(a))]
        d = np.linalg.norm(a - b, axis=-1)
        nn = np.mean([np.sort(np.linalg.norm(a[i][:, None] - a[i][None], axis=-1), axis=1)[:, 1].mean() for i in range(len(a))])
        print("501 steps + normalisation, exact 5-NN vs reference-form kNN: Chamfer %.2e (mean over 8 clouds; mean nearest-neighbour spacing of a "
              "restored cloud %.2e), per-point distance median %.2e, beyond 1e-3: %.1f %%%s" %
              (np.mean(ch), nn, np.median(d), 100 * (d > 1e-3).mean(),
               "" if ref is None else "; vs the reference's own 500-iteration output: exact %.2e, reference-form %.2e (Chamfer)" %
               (np.mean([_chamfer(a[i], ref[i]) for i in range(len(a))]), np.mean([_chamfer(b[i], ref[i]) for i in range(len(a))]))))
        assert np.mean(ch) < 0.25 * nn                     # the same surface sampling: far inside the point spacing
    finally:
        r.close()


def test_reference_form_on_config5_sparse_inputs(np_weights, oracle_weights, golden):
    """BASELINE config #5, K = 256 + SOR (micro-clusters from drawing 1024 points out of <= 256: the case with the most neighbour
    near-ties): the optimiser alone, 10 steps on the build's planes, against the oracle on the same planes - with the reference's
    neighbour choice the count of points beyond 1e-3 is inside the oracle's own 1-ulp floor, WITHOUT the +5 the exact-kNN
    comparison is allowed (tests/test_gpu_parity.py::_attribute_config)."""
    import bench
    import ifdefense_amd as I
    from oracle import convonet_oracle as O
    r = I.Restorer(I.weights.pack_state_dict(np_weights), device="cuda:0")
    try:
        clouds = bench.subsample_like(golden["raw"], 256)
        hip, ref, prep, keep = _oracle_restore_from_hip_draws(r, oracle_weights, clouds, iterations=9, sor=True)
        B = len(clouds)
        pd = I.planes_from_channel_last(prep["_planes_hip"].cpu())
        ox = O.optimize_points(oracle_weights, prep["init"].cpu(), pd, rep_weight=500.0, iterations=9, loss_batch=B, normalize=False).numpy()
        n = {}
        for ref_form in (False, True):
            hx = r.optimize_points(prep["init"], prep["_planes_hip"], rep_weight=500.0, steps=10, loss_batch=B, normalize=False,
                                   knn_reference_form=ref_form).cpu().numpy()
            d = np.linalg.norm(hx - ox, axis=-1)
            n[ref_form] = (int((d > 1e-3).sum()), float(d.max()), float(np.median(d)))
        floor = _ulp_floor(O, oracle_weights, prep["init"].cpu(),
                           lambda q: O.optimize_points(oracle_weights, q, pd, rep_weight=500.0, iterations=9, loss_batch=B,
                                                       normalize=False).numpy(), 9, B, ox)
        print("config #5 K=256 + SOR, optimiser alone, 10 steps: points beyond 1e-3 / max / median vs the oracle: exact 5-NN %s, "
              "reference-form kNN %s; the oracle vs its own 1-ulp-perturbed runs: %s (oracle run LIVE on this host; its matmul sums a "
              "3-term dot like the kernel's restatement: %s)" % (n[False], n[True], floor, host_expanded_form_matches_torch()))
        assert n[True][0] <= max(floor), (n, floor)
        assert n[True][0] <= n[False][0]
    finally:
        r.close()
