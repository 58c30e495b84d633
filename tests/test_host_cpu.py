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
    cfg = od.load_config('configs/convonet_3plane_mn40.yaml')          # the parser's default: shipped values, no file needed
    assert cfg['test']['threshold'] == 0.2 and cfg['data']['pointcloud_n'] == 600
    od.check_supported(cfg)
    with pytest.raises(SystemExit):                                     # a mistyped explicit path is an error, like the reference's open()
        od.load_config(str(tmp_path / "missing.yaml"))
    (tmp_path / "default.yaml").write_text("data:\n  padding: 0.1\n  pointcloud_n: 256\ntest:\n  threshold: 0.5\n")
    (tmp_path / "c.yaml").write_text("data:\n  pointcloud_n: 600\ntest:\n  threshold: 0.2\n  model_file: w.pth\n")
    cfg = od.load_config(str(tmp_path / "c.yaml"), str(tmp_path / "default.yaml"))
    assert cfg['data'] == {'pointcloud_n': 600, 'padding': 0.1, 'dim': 3} and cfg['test']['model_file'] == 'w.pth'
    bad = od.load_config('configs/convonet_3plane_mn40.yaml')
    bad = {**bad, 'model': {**bad['model'], 'c_dim': 128}}
    with pytest.raises(SystemExit):
        od.check_supported(bad)


def test_onet_cli_config_and_save_name(tmp_path):
    """ONet/opt_defense.py differs from ConvONet's in the default config, the model and the save name only."""
    import ifdefense_amd as I
    from ifdefense_amd import opt_defense as od
    a = od.build_parser("onet").parse_args([])
    assert a.config == 'configs/onet_mn40.yaml' and a.iterations == 200 and a.batch_size == 192      # ONet/opt_defense.py:24-53
    cfg = od.load_config('configs/onet_mn40.yaml', None, "onet")
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


# ------------------------------------------------------------------------------------------------
# The N > 1 path through the REAL CLI (opt_defense.main) on CPU: gloo ranks, a stand-in model whose "restoration" is a
# deterministic function of (cloud, global index, seed, loss_batch), an uneven split, fewer clouds than ranks, the
# --train two-array path, the rank-0-only write, and a failing write that must stop every rank instead of hanging them.
# ------------------------------------------------------------------------------------------------
_CLI_SCRIPT = r'''
import os, sys
import numpy as np
import torch
sys.path.insert(0, %(root)r)
from ifdefense_amd import opt_defense


class FakeRestorer:
    """The call seams of Restorer on the CPU: every output row depends on its cloud, its GLOBAL index and its 1/B group, so
    a wrong shard range, index base or gather order changes the file."""
    model_name = "convonet"

    def __init__(self, cfg, device):
        self.device = torch.device(device)

    def sor(self, x, k, alpha):
        return None

    def prepare(self, xb, keep, n_sel, n_opt, padding_scale, init_sigma, seed, cloud_index_base):
        idx = torch.arange(cloud_index_base, cloud_index_base + xb.shape[0], dtype=torch.float32)
        init = xb[:, :1, :].mean(dim=1, keepdim=True).repeat(1, n_opt, 1) + idx[:, None, None] + 0.001 * seed
        return {"sel": xb[:, :n_sel], "t_per_cloud": None, "init": init}

    def encode_inputs(self, sel, t):
        return sel.sum(dim=(1, 2))

    def optimize_points(self, init, planes, rep_weight, iterations, lr, loss_batch, normalize, **kw):
        if os.environ.get("FAKE_FAIL_RANK") == os.environ.get("RANK", "0"):      # a kernel / IfdError on ONE rank, mid-compute
            self.n_calls = getattr(self, "n_calls", 0) + 1
            if self.n_calls > int(os.environ.get("FAKE_FAIL_AFTER", "0")):
                raise RuntimeError("injected compute failure")
        return init + planes[:, None, None] * 1e-3 + loss_batch.float()[:, None, None] * 1e-2 + iterations


argv = sys.argv[1:]
rc = opt_defense.main(argv, restorer_factory=FakeRestorer, backend="gloo", device="cpu")
open(os.path.join(%(out)r, "done_rank%%s" %% os.environ.get("RANK", "0")), "w").write(str(rc))
'''


def _run_cli(tmp_path, world, argv, expect_ok=True, extra_env=None):
    import socket
    script = tmp_path / "cli_run.py"
    script.write_text(_CLI_SCRIPT % {"root": ROOT, "out": str(tmp_path)})
    for f in tmp_path.glob("done_rank*"):
        f.unlink()
    if world == 1:
        env = {k: v for k, v in os.environ.items() if k not in ("WORLD_SIZE", "RANK", "LOCAL_RANK", "MASTER_ADDR", "MASTER_PORT")}
        r = subprocess.run([sys.executable, str(script)] + argv, capture_output=True, text=True, env=env, timeout=300)
    else:
        with socket.socket() as sk:
            sk.bind(("127.0.0.1", 0))
            port = str(sk.getsockname()[1])
        env = dict(os.environ, MASTER_ADDR="127.0.0.1", MASTER_PORT=port, **(extra_env or {}))
        r = subprocess.run([sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node=%d" % world,
                            "--master-addr", "127.0.0.1", "--master-port", port, str(script)] + argv,
                           capture_output=True, text=True, env=env, timeout=300)
    if expect_ok:
        assert r.returncode == 0, r.stdout + r.stderr
        assert len(list(tmp_path.glob("done_rank*"))) == world, r.stdout + r.stderr
    return r


def test_cli_sharded_write_uneven_split_empty_shards_and_train_path(tmp_path):
    rng = np.random.default_rng(3)
    pc7 = rng.normal(size=(7, 40, 6)).astype(np.float32)            # 7 clouds (uneven over 2 and 3 ranks), normals to slice off
    pc2 = rng.normal(size=(2, 40, 3)).astype(np.float32)            # fewer clouds than ranks at world 3: an empty shard
    src = tmp_path / "adv.npz"
    np.savez(src, test_pc=pc7, test_label=np.arange(7), target_label=np.arange(7)[::-1].copy())
    small = tmp_path / "two.npz"
    np.savez(small, test_pc=pc2, test_label=np.array([3, 4]))
    both = tmp_path / "tt.npz"
    np.savez(both, train_pc=pc7[:5], train_label=np.arange(5), test_pc=pc2, test_label=np.array([3, 4]))
    flags = ["--iterations=3", "--batch_size=4", "--seed=9", "--sample_npoint=16"]
    outs = {}
    for world in (1, 2, 3):
        _run_cli(tmp_path, world, ["--data_root", str(src)] + flags)
        _run_cli(tmp_path, world, ["--data_root", str(small)] + flags)
        _run_cli(tmp_path, world, ["--data_root", str(both), "--train=True"] + flags)
        outs[world] = {n: dict(np.load(tmp_path / "ConvONet-Opt" / ("convonet_opt-" + n))) for n in ("adv.npz", "two.npz", "tt.npz")}
        for n in ("adv.npz", "two.npz", "tt.npz"):
            os.remove(tmp_path / "ConvONet-Opt" / ("convonet_opt-" + n))
    # a directory of files goes through the streamed driver (defend_stream): same files as one by one, at any world size
    d = tmp_path / "dir"
    d.mkdir()
    np.savez(d / "adv.npz", test_pc=pc7, test_label=np.arange(7), target_label=np.arange(7)[::-1].copy())
    np.savez(d / "two.npz", test_pc=pc2, test_label=np.array([3, 4]))
    for world in (1, 2):
        _run_cli(tmp_path, world, ["--data_root", str(d)] + flags)
        for n in ("adv.npz", "two.npz"):
            got = dict(np.load(d / "ConvONet-Opt" / ("convonet_opt-" + n)))
            for k in outs[1][n]:
                assert np.array_equal(got[k], outs[1][n][k]), (world, n, k)
            os.remove(d / "ConvONet-Opt" / ("convonet_opt-" + n))
        os.rmdir(d / "ConvONet-Opt")
    one = outs[1]
    assert sorted(one["adv.npz"]) == ["target_label", "test_label", "test_pc"]
    assert one["adv.npz"]["test_pc"].shape == (7, 16, 3) and one["adv.npz"]["test_pc"].dtype == np.float32
    assert one["adv.npz"]["test_label"].dtype == np.uint8 and one["adv.npz"]["target_label"].tolist() == list(range(7))[::-1]
    # the stand-in encodes the global index and the reference batch (4, the last one 3) of every cloud
    lb = np.array([4, 4, 4, 4, 3, 3, 3], np.float32)
    base = pc7[:, :1, :3].mean(axis=1) + np.arange(7, dtype=np.float32)[:, None] + 0.009
    want = base + pc7[:, :, :3].sum(axis=(1, 2))[:, None] * 1e-3 + lb[:, None] * 1e-2 + 3      # (the 600-point subset takes all 40)
    np.testing.assert_allclose(one["adv.npz"]["test_pc"][:, 0], want, rtol=1e-5, atol=1e-5)
    assert sorted(one["tt.npz"]) == ["test_label", "test_pc", "train_label", "train_pc"]
    assert one["tt.npz"]["train_pc"].shape == (5, 16, 3) and one["tt.npz"]["test_pc"].shape == (2, 16, 3)
    for world in (2, 3):                                            # P4 through the CLI: any world size writes the same file
        for n in one:
            for k in one[n]:
                assert np.array_equal(outs[world][n][k], one[n][k]), (world, n, k)


def test_cli_rejects_sizes_and_missing_files_before_loading_anything(tmp_path):
    from ifdefense_amd import opt_defense
    ok = tmp_path / "ok.npz"
    np.savez(ok, test_pc=np.zeros((2, 64, 3), np.float32), test_label=np.zeros(2))
    big = tmp_path / "big.npz"
    np.savez(big, test_pc=np.zeros((1, 10001, 3), np.float32), test_label=np.zeros(1))
    nolabel = tmp_path / "nolabel.npz"
    np.savez(nolabel, test_pc=np.zeros((1, 64, 3), np.float32))

    def boom(cfg, device):                                          # validation must come first: the model is never built
        raise AssertionError("the model was built before the arguments were validated")

    for argv, word in ((["--data_root", str(ok), "--sample_npoint", "10001"], "sample_npoint"),
                       (["--data_root", str(big)], "points per cloud"),
                       (["--data_root", str(nolabel)], "test_label"),
                       (["--data_root", str(tmp_path / "missing.npz")], "not found"),
                       (["--data_root", str(ok), "--config", str(tmp_path / "typo.yaml")], "config file not found")):
        with pytest.raises(SystemExit) as e:
            opt_defense.main(argv, restorer_factory=boom, backend="gloo", device="cpu")
        assert word in str(e.value), (argv, str(e.value))


def test_cli_failing_write_stops_every_rank(tmp_path):
    """Rank 0 cannot write its output (the directory name is taken by a file): the other rank must leave with an error
    too - within seconds, not by a collective timeout - and no rank reports success."""
    src = tmp_path / "adv.npz"
    np.savez(src, test_pc=np.zeros((5, 40, 3), np.float32), test_label=np.arange(5))
    (tmp_path / "ConvONet-Opt").write_text("in the way")
    r = _run_cli(tmp_path, 2, ["--data_root", str(src), "--iterations=1", "--sample_npoint=16"], expect_ok=False)
    assert r.returncode != 0
    assert not list(tmp_path.glob("done_rank*")), r.stdout + r.stderr
    assert "failed on" in (r.stdout + r.stderr) and "another rank failed" in (r.stdout + r.stderr), r.stdout + r.stderr


def test_cli_compute_failure_on_one_rank_stops_every_rank_before_the_gather(tmp_path):
    """A non-zero rank throws DURING its compute (a kernel error, an IfdError): its peers must not be left inside the
    all-gather it never reaches (mismatched collectives hang on RCCL) - the ranks agree on a status BEFORE every gather and
    stop together, within seconds.  Both drivers: one file, and a directory through the streamed driver (failing on the
    second file, after a first successful gather)."""
    import time
    src = tmp_path / "adv.npz"
    np.savez(src, test_pc=np.zeros((5, 40, 3), np.float32), test_label=np.arange(5))
    t0 = time.time()
    r = _run_cli(tmp_path, 2, ["--data_root", str(src), "--iterations=1", "--sample_npoint=16"], expect_ok=False,
                 extra_env={"FAKE_FAIL_RANK": "1"})
    out = r.stdout + r.stderr
    assert r.returncode != 0 and not list(tmp_path.glob("done_rank*")), out
    assert "rank 1 failed on" in out and "injected compute failure" in out and "rank 0 stops: another rank failed" in out, out
    assert not (tmp_path / "ConvONet-Opt" / "convonet_opt-adv.npz").exists()
    d = tmp_path / "dir"
    d.mkdir()
    for n in ("a.npz", "b.npz", "c.npz"):
        np.savez(d / n, test_pc=np.ones((4, 40, 3), np.float32), test_label=np.arange(4))
    r = _run_cli(tmp_path, 2, ["--data_root", str(d), "--iterations=1", "--sample_npoint=16"], expect_ok=False,
                 extra_env={"FAKE_FAIL_RANK": "1", "FAKE_FAIL_AFTER": "1"})
    out = r.stdout + r.stderr
    assert r.returncode != 0 and not list(tmp_path.glob("done_rank*")), out
    assert "rank 1 failed on" in out and "rank 0 stops: another rank failed" in out, out
    assert (d / "ConvONet-Opt" / "convonet_opt-a.npz").exists()          # the first file went through before the failure
    assert not (d / "ConvONet-Opt" / "convonet_opt-c.npz").exists()
    assert time.time() - t0 < 120, "the ranks did not stop promptly"


@pytest.mark.parametrize("scaling", ["weak", "strong"])
def test_bench_launch_line_dry_run_for_2_4_8_ranks(scaling):
    """The driver's N > 1 launch line (`python -m torch.distributed.run --nnodes=1 --nproc-per-node N ... bench.py --gpus N`)
    without N GPUs: `--dry-run` creates the process group (gloo), shards the clouds like the timed path does, pushes a
    placeholder of every rank's shard through the path's one collective and prints the JSON line with value null.  Checked
    for N = 2, 4, 8 in both scalings: the gathered array is the concatenation of the shards on every rank, and
    clouds_total / clouds_per_gpu / parallelism are what the scaling mode promises (weak: 2468 per rank; strong: ONE
    2468-cloud array - BASELINE configs #3 / #5 - over the ranks, the last shard shorter).  The reference shards its attack
    scripts the same way (baselines/attack_scripts/targeted_knn_attack.py:97-128)."""
    import json
    import socket
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    procs = []
    for n in (2, 4, 8):
        with socket.socket() as s:
            s.bind(("127.0.0.1", 0))
            port = s.getsockname()[1]
        procs.append((n, subprocess.Popen(
            [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", str(n), "--master-addr", "127.0.0.1",
             "--master-port", str(port), "bench.py", "--gpus", str(n), "--scaling", scaling, "--dry-run"],
            stdout=subprocess.PIPE, stderr=subprocess.PIPE, text=True, cwd=root,
            env=dict(os.environ, OMP_NUM_THREADS="1", MASTER_ADDR="127.0.0.1"))))
    for n, p in procs:
        out, err = p.communicate(timeout=600)
        assert p.returncode == 0, (n, out[-2000:], err[-3000:])
        lines = [l for l in out.splitlines() if l.startswith("{")]
        assert len(lines) == 1, (n, out)                               # rank 0 prints ONE line
        j = json.loads(lines[0])
        assert j["dry_run"] and j["gather_ok"] and j["value"] is None and j["n_gpus"] == n and j["scaling"] == scaling
        c = j["config"]
        assert c["parallelism"] == "shard%d+allgather" % n
        if scaling == "weak":
            assert c["clouds_per_gpu"] == 2468 and c["clouds_total"] == 2468 * n and c["shard_of_rank0"] == [0, 2468]
        else:
            per = -(-2468 // n)
            assert c["clouds_total"] == 2468 and c["clouds_per_gpu"] == per and c["shard_of_rank0"] == [0, per]


def test_init_from_env_rejects_a_local_rank_without_a_gpu(monkeypatch):
    """One process per GPU: a rank whose LOCAL_RANK has no device behind it (more ranks than visible GPUs) must say so, not
    die inside torch.cuda.set_device / the RCCL communicator."""
    from ifdefense_amd import dist as D
    monkeypatch.setenv("WORLD_SIZE", "4")
    monkeypatch.setenv("RANK", "3")
    monkeypatch.setenv("LOCAL_RANK", "3")
    monkeypatch.setenv("MASTER_ADDR", "127.0.0.1")
    monkeypatch.setenv("MASTER_PORT", "29999")
    monkeypatch.setattr(torch.cuda, "device_count", lambda: 1)
    with pytest.raises(RuntimeError, match="LOCAL_RANK=3 but this node shows 1 GPU"):
        D.init_from_env("nccl")
