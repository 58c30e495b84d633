"""CPU tests of the host logic: CLI flags identical to the reference, config loading, save names, .npz schema,
shard arithmetic, and the world_size-2 gloo all-gather path."""
import os
import subprocess
import sys

import numpy as np
import pytest
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def test_cli_flags_and_defaults_match_reference():
    from ifdefense_amd.opt_defense import build_parser
    a = build_parser().parse_args([])
    ref = dict(config='configs/convonet_3plane_mn40.yaml', sample_npoint=1024, padding_scale=0.9, data_root='',
               train=False, init_sigma=0.01, iterations=200, batch_size=192, lr=0.001, rep_weight=500., sor=True,
               sor_k=2, sor_alpha=1.1)                       # ConvONet/opt_defense.py:24-53
    for k, v in ref.items():
        assert getattr(a, k) == v, k
    b = build_parser().parse_args(["--sor=False", "--train=yes", "--iterations=500", "--data_root=x.npz"])
    assert b.sor is False and b.train is True and b.iterations == 500 and b.data_root == "x.npz"


def test_config_loading(tmp_path):
    from ifdefense_amd import opt_defense as od
    cfg = od.load_config(str(tmp_path / "missing.yaml"))
    assert cfg['test']['threshold'] == 0.2 and cfg['data']['pointcloud_n'] == 600
    od.check_supported(cfg)
    (tmp_path / "default.yaml").write_text("data:\n  padding: 0.1\n  pointcloud_n: 256\ntest:\n  threshold: 0.5\n")
    (tmp_path / "c.yaml").write_text("data:\n  pointcloud_n: 600\ntest:\n  threshold: 0.2\n  model_file: w.pth\n")
    cfg = od.load_config(str(tmp_path / "c.yaml"), str(tmp_path / "default.yaml"))
    assert cfg['data'] == {'pointcloud_n': 600, 'padding': 0.1, 'dim': 3} and cfg['test']['model_file'] == 'w.pth'
    bad = od.load_config(str(tmp_path / "missing.yaml"))
    bad = {**bad, 'model': {**bad['model'], 'c_dim': 128}}
    with pytest.raises(SystemExit):
        od.check_supported(bad)


def test_onet_cli_config_and_save_name(tmp_path):
    """ONet/opt_defense.py differs from ConvONet's in the default config, the model and the save name only."""
    import ifdefense_amd as I
    from ifdefense_amd import opt_defense as od
    a = od.build_parser("onet").parse_args([])
    assert a.config == 'configs/onet_mn40.yaml' and a.iterations == 200 and a.batch_size == 192      # ONet/opt_defense.py:24-53
    cfg = od.load_config(str(tmp_path / "missing.yaml"), None, "onet")
    assert cfg['data']['pointcloud_n'] == 300 and cfg['test']['threshold'] == 0.2 and cfg['model']['z_dim'] == 0
    od.check_supported(cfg, "onet")
    with pytest.raises(SystemExit):
        od.check_supported({**cfg, 'model': {**cfg['model'], 'z_dim': 64}}, "onet")
    assert I.get_save_name(str(tmp_path / "x.npz"), "onet") == str(tmp_path / "ONet-Opt" / "onet_opt-x.npz")
    keys = I.weights.onet_canonical_keys()
    assert sum(int(np.prod(s)) for _, s in keys) == 10379521 and keys[0][0] == "decoder.fc_p.weight"
    sd = I.weights.onet_random_state_dict(1)
    assert I.weights.pack_state_dict(sd, "onet").size == 10379521
    with pytest.raises(KeyError):
        I.weights.pack_state_dict({k: v for k, v in sd.items() if k != "encoder.fc_c.bias"}, "onet")


def test_save_name_and_npz_schema(tmp_path):
    import ifdefense_amd as I
    src = tmp_path / "kNN-pointnet.npz"
    pc = np.random.default_rng(0).standard_normal((3, 64, 6)).astype(np.float32)
    np.savez(src, test_pc=pc, test_label=np.arange(3), target_label=np.array([5, 6, 7]))
    seen = {}

    def fake_defend(x):
        seen['shape'] = x.shape
        return np.zeros((len(x), 1024, 3), np.float64)

    out = I.defend_npz_test_data(None, str(src), I.DefenseArgs(), defend=fake_defend)
    assert out == str(tmp_path / "ConvONet-Opt" / "convonet_opt-kNN-pointnet.npz")       # opt_defense.py:242-252
    assert seen['shape'] == (3, 64, 3)                                                    # [..., :3] (:323)
    z = np.load(out)
    assert sorted(z.files) == ['target_label', 'test_label', 'test_pc']
    assert z['test_pc'].dtype == np.float32 and z['test_pc'].shape == (3, 1024, 3)
    assert z['test_label'].dtype == np.uint8 and z['target_label'].dtype == np.uint8
    src2 = tmp_path / "clean.npz"
    np.savez(src2, test_pc=pc, test_label=np.arange(3))
    assert sorted(np.load(I.defend_npz_test_data(None, str(src2), I.DefenseArgs(), defend=fake_defend)).files) == \
        ['test_label', 'test_pc']
    src3 = tmp_path / "hybrid.npz"
    np.savez(src3, train_pc=pc, train_label=np.arange(3), test_pc=pc, test_label=np.arange(3))
    z = np.load(I.defend_npz_train_test_data(None, str(src3), I.DefenseArgs(), defend=fake_defend))
    assert sorted(z.files) == ['test_label', 'test_pc', 'train_label', 'train_pc']


def test_shard_ranges_cover_and_are_contiguous():
    from ifdefense_amd.dist import shard_range
    for n in (1, 7, 8, 2468):
        for world in (1, 2, 4, 8):
            got = []
            for r in range(world):
                lo, hi, per = shard_range(n, r, world)
                assert hi - lo <= per and per == -(-n // world)
                got += list(range(lo, hi))
            assert got == list(range(n))


def test_reference_batch_grouping_of_the_loss_factor():
    """defend_point_cloud groups clouds by the reference batch (size 192, last one shorter) -> loss_batch."""
    import ifdefense_amd as I

    class Fake:
        device = torch.device("cpu")
        calls = []

        def sor(self, x, k, a):
            return None

        def prepare(self, xb, keep, **kw):
            return {"sel": xb, "t_per_cloud": None, "init": torch.zeros(len(xb), 4, 3)}

        def encode_inputs(self, sel, t):
            return torch.zeros(len(sel), 1)

        def optimize_points(self, init, planes, loss_batch=None, **kw):
            self.calls.append(loss_batch.tolist())
            return torch.zeros(len(init), 4, 3)

    f = Fake()
    args = I.DefenseArgs(sample_npoint=4, batch_size=192, chunk=1000)
    I.defend_point_cloud(f, np.zeros((2468, 8, 3), np.float32), args)
    lbs = sum(f.calls, [])
    assert len(lbs) == 2468 and lbs[:2304] == [192] * 2304 and lbs[2304:] == [164] * 164
    f.calls.clear()                                         # rank 1 of 2: clouds 1234..2467 of 2468
    I.defend_point_cloud(f, np.zeros((1234, 8, 3), np.float32), args, cloud_index_base=1234, total_clouds=2468)
    lbs = sum(f.calls, [])
    assert lbs == [192] * (2304 - 1234) + [164] * 164


def test_world2_gloo_shard_and_allgather(tmp_path):
    """N > 1 path on CPU: two gloo processes shard 7 'clouds' and all-gather; equals the single-process result."""
    script = tmp_path / "w2.py"
    script.write_text(
        "import os, sys, numpy as np, torch\n"
        "sys.path.insert(0, %r)\n"
        "from ifdefense_amd import dist as D\n"
        "rank, world, local = D.init_from_env('gloo')\n"
        "pc = np.arange(7 * 5 * 3, dtype=np.float32).reshape(7, 5, 3)\n"
        "f = lambda shard, base, total: torch.from_numpy(shard * 2 + base * 0 + total * 0).float()\n"
        "out = D.defend_sharded(f, pc)\n"
        "assert out.shape == (7, 5, 3) and np.array_equal(out.numpy(), pc * 2), out\n"
        "open(os.path.join(%r, 'ok_rank%%d_of_%%d' %% (rank, world)), 'w').write('ok')\n" % (ROOT, str(tmp_path)))
    import socket
    with socket.socket() as sk:                 # a free port (a fixed one fails when two test runs follow each other)
        sk.bind(("127.0.0.1", 0))
        port = str(sk.getsockname()[1])
    env = dict(os.environ, MASTER_ADDR="127.0.0.1", MASTER_PORT=port)
    r = subprocess.run([sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node=2",
                        "--master-addr", "127.0.0.1", "--master-port", port, str(script)],
                       capture_output=True, text=True, env=env, timeout=300)
    assert r.returncode == 0, r.stdout + r.stderr
    # one marker file per rank (the two ranks' stdout lines interleave character by character)
    assert (tmp_path / "ok_rank0_of_2").exists() and (tmp_path / "ok_rank1_of_2").exists(), r.stdout + r.stderr
