"""The ONet-Opt oracle (oracle/onet_oracle.py) against outputs of the reference's own ONet modules
(tests/golden/onet_golden.npz, written by tests/golden/make_golden_onet.py).  CPU only."""
import os

import numpy as np
import pytest
import torch

HERE = os.path.dirname(os.path.abspath(__file__))


@pytest.fixture(scope="module")
def og():
    return np.load(os.path.join(HERE, "golden", "onet_golden.npz"))


@pytest.fixture(scope="module")
def ow():
    from oracle import onet_oracle as OO
    return OO.to_torch(OO.make_random_weights(0))


def test_weight_schema_matches_reference_checkpoint(ow):
    assert sum(v.numel() for k, v in ow.items() if not k.endswith("num_batches_tracked")) == 10379532 - 11
    assert ow["decoder.fc_p.weight"].shape == (256, 3, 1) and ow["encoder.block_4.shortcut.weight"].shape == (512, 1024)
    assert ow["decoder.block3.bn_1.conv_gamma.weight"].shape == (256, 512, 1)


def test_encoder_latent(og, ow):
    from oracle import onet_oracle as OO
    c, stages = OO.encode_latent(ow, torch.from_numpy(og["sel"]), return_stages=True)
    np.testing.assert_allclose(c.numpy(), og["c"], rtol=0, atol=1e-6)
    got = torch.stack([s[0].max(dim=0).values for s in stages]).numpy()
    np.testing.assert_allclose(got, og["enc_stage_max0"], rtol=1e-5, atol=1e-6)


def test_decoder_logits_and_gradient(og, ow, golden):
    from oracle import onet_oracle as OO
    p = torch.from_numpy(golden["init_points"][:2]).clone().requires_grad_()
    logits = OO.decode_logits(ow, p, torch.from_numpy(og["c"][:2]))
    logits.sum().backward()
    np.testing.assert_allclose(logits.detach().numpy(), og["dec_logits"], rtol=0, atol=2e-6)
    assert np.abs(p.grad.numpy() - og["dec_dlogit_dp"]).max() < 1e-5 * np.abs(og["dec_dlogit_dp"]).max()


def test_cbn_affine_equals_reference_op_order(og, ow):
    from oracle import onet_oracle as OO
    c = torch.from_numpy(og["c"][:2])
    x = torch.randn(2, 7, 256, generator=torch.Generator().manual_seed(1))
    a, b = OO.cbn_affine(ow, "decoder.block2.bn_1", c)
    ref = OO._cbn(ow, "decoder.block2.bn_1", x, c)
    np.testing.assert_allclose((a[:, None] * x + b[:, None]).numpy(), ref.numpy(), rtol=1e-5, atol=1e-6)


def test_teacher_forced_steps_and_trajectory(og, ow):
    from oracle import convonet_oracle as CO
    from oracle import onet_oracle as OO
    c = torch.from_numpy(og["c"][:2])
    for t in (0, 1, 9):
        x = torch.from_numpy(og[f"traj{t}_x"]).clone().requires_grad_()
        total, occ, rep, _ = OO.losses(ow, x, c, 500.0)
        total.backward()
        dg = np.abs(x.grad.numpy() - og[f"traj{t}_g"]) / np.abs(og[f"traj{t}_g"]).max()
        assert (dg > 2e-5).mean() < 1e-3 and dg.max() < 1e-3, dg.max()      # a near-tied neighbour may differ
        np.testing.assert_allclose([float(occ.detach()), float(rep.detach())], og[f"traj{t}_loss"], rtol=1e-5)
        xn, _, _ = CO.adam_step(x.detach(), torch.from_numpy(og[f"traj{t}_g"]), torch.from_numpy(og[f"traj{t}_m"]),
                                torch.from_numpy(og[f"traj{t}_v"]), t + 1)
        assert np.abs(xn.numpy() - og[f"traj{t}_x_next"]).max() < 1e-6
    init = torch.from_numpy(og["traj0_x"])
    x10 = OO.optimize_points(ow, init, c, iterations=9, normalize=False)
    assert (x10 - torch.from_numpy(og["traj9_x_next"])).norm(dim=-1).max() < 1e-4


def test_end_to_end_11_steps(og, ow, golden):
    from oracle import onet_oracle as OO
    out = OO.optimize_points(ow, torch.from_numpy(golden["init_points"]), torch.from_numpy(og["c"]), iterations=10)
    d = np.linalg.norm(out.numpy() - og["e2e10_out"], axis=-1)
    assert d.max() < 1e-3 and np.median(d) < 1e-5


def test_mesh_oracle_surface_sampling_and_reference_libs():
    """oracle/mesh_oracle.py: the restated trimesh.sample.sample_surface is area-uniform, and (when oracle/_ref is
    built) the reference MISE + libmcubes reproduce a sphere's area."""
    from oracle import mesh_oracle as MO
    v = np.array([[0, 0, 0], [1, 0, 0], [1, 1, 0], [0, 1, 0], [3, 0, 0], [3, 1, 0]], np.float64)
    f = np.array([[0, 1, 2], [0, 2, 3], [1, 4, 5], [1, 5, 2]])            # unit square + a 2 x 1 rectangle
    s = MO.sample_surface(v, f, 30000, np.random.default_rng(0))
    assert (s[:, 2] == 0).all() and s[:, 0].min() >= 0 and s[:, 0].max() <= 3 and s[:, 1].min() >= 0 and s[:, 1].max() <= 1
    assert abs((s[:, 0] < 1).mean() - 1 / 3) < 0.01                     # area-weighted
    try:
        mise, mcubes = MO.ref_libs()
    except ImportError:
        pytest.skip("oracle/_ref not built")
    m = mise.MISE(8, 2, 0.0)
    pts = m.query()
    while pts.shape[0]:
        m.update(pts, (0.3 - np.linalg.norm(pts / m.resolution - 0.5, axis=1)).astype(np.float64))
        pts = m.query()
    vert, tri = MO.extract_mesh(m.to_dense(), 0.0, padding=0.0)
    t = vert[tri]
    area = 0.5 * np.linalg.norm(np.cross(t[:, 1] - t[:, 0], t[:, 2] - t[:, 0]), axis=1).sum()
    assert abs(area - 4 * np.pi * 0.09) / (4 * np.pi * 0.09) < 0.03
