"""Pin the CPU oracle (oracle/) against outputs of the reference's own modules.

The fixtures in tests/golden/convonet_golden.npz were produced by
tests/golden/make_golden.py, which imports /root/reference/ConvONet on CPU.
Tolerances are float32 round-off: both sides run torch CPU kernels, but op
ordering (e.g. torch.where vs masked assignment, scatter order) may differ.
"""
import numpy as np
import torch

from oracle import closed_form as CF
from oracle import convonet_oracle as O

PL = ("xz", "xy", "yz")


def _planes(golden, n=2):
    return {pl: torch.from_numpy(golden["planes01"][:n, i]) for i, pl in enumerate(PL)}


def _proc(golden):
    return [golden["proc_pad"][b, :golden["proc_len"][b]] for b in range(4)]


def test_sor_mask_and_value(golden):
    keep, value = O.sor_keep_mask(torch.from_numpy(golden["raw"]))
    assert np.array_equal(keep.numpy(), golden["sor_keep"])            # bit-exact mask
    np.testing.assert_allclose(value.numpy(), golden["sor_value"], rtol=1e-12, atol=0)
    assert golden["sor_keep"].sum(1).tolist() == [909, 999, 885, 906]


def test_preprocess(golden):
    for b in range(4):
        got = O.preprocess_pc(golden["raw"][b][golden["sor_keep"][b]])
        np.testing.assert_array_equal(got, _proc(golden)[b])           # same numpy f32 ops


def test_init_points(golden):
    got = O.init_points(_proc(golden), golden["init_idx"], golden["noise"])
    np.testing.assert_array_equal(got.numpy(), golden["init_points"])


def test_encoder_pointnet(golden, oracle_weights):
    proc = _proc(golden)
    sel = torch.from_numpy(np.stack([proc[b][golden["sel_idx"][b]] for b in range(4)]))
    c, index, stages = O.pointnet_features(oracle_weights, sel, return_stages=True)
    got_index = np.stack([index[pl].numpy() for pl in PL], 1)
    assert np.array_equal(got_index, golden["enc_index"])
    np.testing.assert_allclose(stages[0][:2].numpy(), golden["enc_stage0"], rtol=1e-5, atol=1e-6)
    np.testing.assert_allclose(c.numpy(), golden["enc_c"], rtol=1e-4, atol=2e-6)
    pre = O.scatter_mean_plane(c, index["xz"])
    np.testing.assert_allclose(pre[0].numpy(), golden["enc_pre_xz0"], rtol=1e-4, atol=2e-6)
    # G1: the pooled-stage tensors - pool_local's output (scatter_max over the three planes' cells, gathered back,
    # summed; pointnet.py:104-122) at each of the four pooling stages, as the reference module produced them
    pooled = O.pointnet_features(oracle_weights, sel, return_stages="pooled")[3]
    for k in range(4):
        np.testing.assert_allclose(pooled[k][:2].numpy(), golden["enc_pooled"][:, k], rtol=1e-4, atol=2e-6)


def test_encoder_planes(golden, oracle_weights):
    proc = _proc(golden)
    sel = torch.from_numpy(np.stack([proc[b][golden["sel_idx"][b]] for b in range(4)]))
    planes = O.encode_inputs(oracle_weights, sel)
    for i, pl in enumerate(PL):
        ref = golden["planes01"][:, i]
        np.testing.assert_allclose(planes[pl][:2].numpy(), ref, rtol=1e-3, atol=2e-5 * np.abs(ref).max())
        for b in range(4):
            assert abs(float(planes[pl][b].abs().mean()) - golden["planes_stats"][b, i, 1]) < 1e-5


def test_decoder_logits_and_grad(golden, oracle_weights):
    p = torch.from_numpy(golden["init_points"][:2]).clone().requires_grad_()
    logits = O.decode_logits(oracle_weights, p, _planes(golden))
    logits.sum().backward()
    np.testing.assert_allclose(logits.detach().numpy(), golden["dec_logits"], rtol=1e-5, atol=1e-6)
    ref = golden["dec_dlogit_dp"]
    np.testing.assert_allclose(p.grad.numpy(), ref, rtol=1e-4, atol=1e-5 * np.abs(ref).max())


def test_knn_and_repulsion(golden):
    p = torch.from_numpy(golden["init_points"][:2])
    idx = O.knn_point(5, p)
    assert np.array_equal(idx.numpy(), golden["knn_idx"])
    np.testing.assert_allclose(O.repulsion_loss(p).numpy(), golden["rep_loss_b"], rtol=1e-5)
    # the exact float64 direct-form neighbour SETS agree with the reference's f32 expanded form
    for b in range(2):
        ex = CF.knn_exact(golden["init_points"][b])
        same = [set(ex[i]) == set(golden["knn_idx"][b, i]) for i in range(1024)]
        assert np.mean(same) > 0.995


def test_total_gradient_and_losses(golden, oracle_weights):
    for step in (0, 1, 9, 49):
        x = torch.from_numpy(golden[f"traj{step}_x"]).clone().requires_grad_()
        total, occ, rep, _ = O.losses(oracle_weights, x, _planes(golden), 500.0)
        total.backward()
        ref = golden[f"traj{step}_g"]
        np.testing.assert_allclose(x.grad.numpy(), ref, rtol=2e-3, atol=2e-5 * np.abs(ref).max())
        np.testing.assert_allclose([float(occ.detach()), float(rep.detach())], golden[f"traj{step}_loss"], rtol=1e-5)


def test_closed_form_matches_reference_gradient(golden, np_weights):
    """float64 hand-derived backward == the reference's autograd gradient."""
    for step in (0, 9):
        x = golden[f"traj{step}_x"]
        ref = golden[f"traj{step}_g"]
        for b in range(2):
            planes = {pl: golden["planes01"][b, i] for i, pl in enumerate(PL)}
            d = CF.decoder_forward_backward(np_weights, x[b], planes, loss_batch=2.0)
            idx = O.knn_point(5, torch.from_numpy(x[b:b + 1]))[0].numpy()
            r = CF.repulsion_forward_backward(x[b], idx=idx, loss_batch=2.0)
            g = d["grad"] + r["grad"]
            err = np.abs(g - ref[b]).max() / np.abs(ref[b]).max()
            assert err < 2e-5, err


def test_adam_teacher_forced(golden):
    for step in (0, 1, 9, 49):
        x, g = (torch.from_numpy(golden[f"traj{step}_{k}"]) for k in ("x", "g"))
        m, v = (torch.from_numpy(golden[f"traj{step}_{k}"]) for k in ("m", "v"))
        x1, _, _ = O.adam_step(x, g, m, v, t=step + 1)
        np.testing.assert_allclose(x1.numpy(), golden[f"traj{step}_x_next"], rtol=0, atol=2e-7)


def test_trajectory_short_horizon(golden, oracle_weights):
    init = torch.from_numpy(golden["init_points"][:2])
    _, snaps = O.optimize_points(oracle_weights, init, _planes(golden), iterations=9, normalize=False,
                                 record=(1, 2, 10))
    for n, key in ((1, "traj0_x_next"), (2, "traj1_x_next"), (10, "traj9_x_next")):
        d = (snaps[n] - torch.from_numpy(golden[key])).norm(dim=-1)
        assert float(d.max()) < 1e-3, (n, float(d.max()))
    assert float((snaps[1] - torch.from_numpy(golden["traj0_x_next"])).abs().max()) < 1e-6


def test_end_to_end_20(golden, oracle_weights):
    init = torch.from_numpy(golden["init_points"])
    planes4 = O.encode_inputs(oracle_weights, torch.from_numpy(
        np.stack([_proc(golden)[b][golden["sel_idx"][b]] for b in range(4)])))
    out = O.optimize_points(oracle_weights, init, planes4, iterations=20)
    d = (out - torch.from_numpy(golden["e2e20_out"])).norm(dim=-1)
    # 21 chaotic steps: bulk must agree tightly, a few points may have diverged
    assert float(d.median()) < 1e-4 and float((d > 1e-2).float().mean()) < 0.01, (float(d.median()), float(d.max()))


def test_oracle_late_teacher_forced_steps(oracle_weights, golden):
    """G4 at Adam t = 100 and 500: the oracle's gradient and update from the reference's recorded state."""
    import os
    from oracle import convonet_oracle as O
    gl = np.load(os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden", "convonet_golden_long.npz"))
    planes2 = {pl: torch.from_numpy(golden["planes01"][:, i]) for i, pl in enumerate(("xz", "xy", "yz"))}
    for t in (99, 499):
        x = torch.from_numpy(gl[f"traj{t}_x"]).clone().requires_grad_()
        total, occ, rep, _ = O.losses(oracle_weights, x, planes2, 500.0)
        total.backward()
        dg = np.abs(x.grad.numpy() - gl[f"traj{t}_g"]) / np.abs(gl[f"traj{t}_g"]).max()
        assert (dg > 2e-5).mean() < 1e-3 and dg.max() < 2e-2, dg.max()
        np.testing.assert_allclose([float(occ.detach()), float(rep.detach())], gl[f"traj{t}_loss"], rtol=1e-4)
        xn, _, _ = O.adam_step(x.detach(), torch.from_numpy(gl[f"traj{t}_g"]), torch.from_numpy(gl[f"traj{t}_m"]),
                               torch.from_numpy(gl[f"traj{t}_v"]), t + 1)
        assert np.abs(xn.numpy() - gl[f"traj{t}_x_next"]).max() < 1e-6


# ---- the trained-like fixtures (converged-surface regime; tests/golden/make_golden_trained.py) ---------------------------
def _trained():
    import os
    here = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden")
    z = np.load(os.path.join(here, "trained_like_f16.npz"))
    w = O.to_torch({k: z[k].astype(np.float32) for k in z.files})
    f = np.load(os.path.join(here, "convonet_golden_trained.npz"))
    planes = {pl: torch.from_numpy(f["planes_f16"][:, i].astype(np.float32)) for i, pl in enumerate(PL)}
    return w, f, planes


def test_trained_like_oracle_driver_and_decoder():
    """The oracle's driver restatements against what the reference's OWN functions (executed from opt_defense.py's source)
    produced on the trained-like checkpoint: SOR sizes, preprocess + recorded subset, init points - bit for bit - and the
    decoder's logits / input gradient."""
    w, f, planes = _trained()
    keep, _ = O.sor_keep_mask(torch.from_numpy(f["raw"]))
    assert (keep.sum(1).numpy() == f["sor_len"]).all()
    proc = [O.preprocess_pc(f["raw"][b][keep[b].numpy()]) for b in range(8)]
    sel = np.stack([proc[b][f["sel_idx"][b]] for b in range(8)])
    np.testing.assert_array_equal(sel, f["sel"])
    init = O.init_points(proc, f["init_idx"], f["noise"])
    np.testing.assert_array_equal(init.numpy(), f["init_points"])
    p = init[:4].clone().requires_grad_()
    logits = O.decode_logits(w, p, {k: v[:4] for k, v in planes.items()})
    logits.sum().backward()
    np.testing.assert_allclose(logits.detach().numpy(), f["dec_logits"][:4], rtol=1e-4, atol=2e-5)
    np.testing.assert_allclose(p.grad.numpy(), f["dec_dlogit_dp"][:4], rtol=1e-3, atol=1e-3 * np.abs(f["dec_dlogit_dp"]).max())


def test_trained_like_oracle_trajectory():
    """Teacher-forced Adam steps of the oracle from the reference's states at t = 1, 10, 100 (B = 8) and 10 free steps."""
    w, f, planes = _trained()
    for t in (0, 9, 99):
        x = torch.from_numpy(f[f"traj{t}_x"]).clone().requires_grad_()
        O.losses(w, x, planes, 500.0)[0].backward()
        g_ref = f[f"traj{t}_g"]
        assert np.abs(x.grad.numpy() - g_ref).max() / np.abs(g_ref).max() < 2e-5
        x1, _, _ = O.adam_step(x.detach(), x.grad, torch.from_numpy(f[f"traj{t}_m"]), torch.from_numpy(f[f"traj{t}_v"]), t + 1)
        assert np.abs(x1.numpy() - f[f"traj{t}_x_next"]).max() < 2e-6
    out = O.optimize_points(w, torch.from_numpy(f["init_points"]), planes, rep_weight=500.0, iterations=9, normalize=False)
    assert np.linalg.norm(out.numpy() - f["traj9_x_next"], axis=-1).max() < 1e-3


# ------------------------------------------------------------------------------------------------
# The reference's kNN, restated to the bit (what csrc/knn_device.h knn_ref_point runs under ifd_opt_params.knn_reference_form)
# ------------------------------------------------------------------------------------------------
def _ref_topk6_restated(neg_dist_row):
    """std::partial_sort of libstdc++ (bits/stl_heap.h) over (value, index) with comp(a, b) = a.value > b.value: what torch.topk(k = 6)
    runs on a CPU row of 1024 (ATen/native/cpu/TopKImpl.h, k * 64 <= n) - including what it does with EQUAL values."""
    v = [float(x) for x in neg_dist_row[:6]]
    ix = list(range(6))

    def adjust(hole, length, val, idx):
        top, child = hole, hole
        while child < (length - 1) // 2:
            child = 2 * (child + 1)
            if v[child] > v[child - 1]:
                child -= 1
            v[hole], ix[hole] = v[child], ix[child]
            hole = child
        if length % 2 == 0 and child == (length - 2) // 2:
            child = 2 * (child + 1)
            v[hole], ix[hole] = v[child - 1], ix[child - 1]
            hole = child - 1
        parent = (hole - 1) // 2
        while hole > top and v[parent] > val:
            v[hole], ix[hole] = v[parent], ix[parent]
            hole = parent
            parent = (hole - 1) // 2
        v[hole], ix[hole] = val, idx

    for parent in (2, 1, 0):
        adjust(parent, 6, v[parent], ix[parent])
    for j in range(6, len(neg_dist_row)):
        x = float(neg_dist_row[j])
        if x > v[0]:
            adjust(0, 6, x, j)
    for last in range(5, 0, -1):
        val, idx = v[last], ix[last]
        v[last], ix[last] = v[0], ix[0]
        adjust(0, last, val, idx)
    return ix


def test_reference_knn_restated_to_the_bit():
    """knn_point (pn_utils.py:72-83) on this host: (i) the float32 accumulation order of matmul / sum the kernel restates, (ii) the
    tie behaviour of topk - on the converged trained-like fixture (where pairs of points make dist_ii and dist_ij tie at exactly 0)
    and on rows of integers (many ties)."""
    import os
    f = np.load(os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden", "convonet_golden_trained_long.npz"))
    pc = torch.from_numpy(f["traj499_x"])
    inner = -2.0 * torch.matmul(pc, pc.transpose(2, 1))
    xx = torch.sum(pc.transpose(2, 1) ** 2, dim=1, keepdim=True)
    dist = xx + inner + xx.transpose(2, 1)
    x = pc.numpy()

    def fma(a, b, c):
        return (a.astype(np.float64) * b.astype(np.float64) + c.astype(np.float64)).astype(np.float32)
    xi, xj = x[:, :, None, :], x[:, None, :, :]
    dot = fma(xi[..., 2], xj[..., 2], fma(xi[..., 1], xj[..., 1], xi[..., 0] * xj[..., 0]))
    sq = [x[..., k] * x[..., k] for k in range(3)]
    xxn = (sq[0] + sq[1]) + sq[2]
    mine = (xxn[:, None, :] + np.float32(-2.0) * dot) + xxn[:, :, None]
    assert np.array_equal(mine, dist.numpy())                                 # (i)
    _, top = (-dist).topk(6, dim=-1)
    nd = (-dist).numpy()
    ties = 0
    for b in range(nd.shape[0]):
        for i in range(0, nd.shape[1], 1):
            row = nd[b, i]
            s = np.sort(row)[::-1][:7]
            if (s[:-1] == s[1:]).any():                                        # rows with a tie among the best seven: all of them
                ties += 1
                assert _ref_topk6_restated(row) == top[b, i].tolist(), (b, i)
            elif i % 16 == 0:
                assert _ref_topk6_restated(row) == top[b, i].tolist(), (b, i)
    assert ties >= 10, ties                                                    # (measured: 26 rows of 8192 tie at Adam t = 500)
    g = torch.Generator().manual_seed(3)
    v = torch.randint(0, 9, (48, 1024), generator=g).float()
    _, top = v.topk(6, dim=-1)
    for r in range(48):
        assert _ref_topk6_restated(v[r].numpy()) == top[r].tolist(), r


def test_reference_knn_distance_matrix_fixture_pins_the_restatement():
    """tests/golden/knn_ref_dist.npz holds the matrix the REFERENCE's knn_point handed to topk on the fixture host (captured inside
    pn_utils.py:64-83 by tests/golden/make_knn_ref_dist.py; near-tie cluster + a coincident pair included) and the indices it
    returned: the numpy restatement of the kernel's accumulation order reproduces the matrix bit for bit, and the restated
    partial_sort selection (`_ref_topk6_restated`) its neighbour indices - on any host, no BLAS involved."""
    import os
    z = np.load(os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden", "knn_ref_dist.npz"))
    x, dist, idx = z["pc"], z["dist"], z["idx"]

    def fma(a, b, c):
        return (a.astype(np.float64) * b.astype(np.float64) + c.astype(np.float64)).astype(np.float32)
    xi, xj = x[:, :, None, :], x[:, None, :, :]
    dot = fma(xi[..., 2], xj[..., 2], fma(xi[..., 1], xj[..., 1], xi[..., 0] * xj[..., 0]))
    sq = [x[..., k] * x[..., k] for k in range(3)]
    xxn = (sq[0] + sq[1]) + sq[2]
    mine = (xxn[:, None, :] + np.float32(-2.0) * dot) + xxn[:, :, None]
    assert np.array_equal(mine, dist)
    nd = -dist
    quirk = 0
    for b in range(nd.shape[0]):
        for i in range(nd.shape[1]):
            top6 = _ref_topk6_restated(nd[b, i])
            assert top6[1:] == idx[b, i].tolist(), (b, i)
            quirk += int(i in top6[1:])                 # "self" kept as a neighbour (column 0 was another point)
    assert quirk >= 1, quirk                             # the coincident pair exercises the reference's column-0 quirk
