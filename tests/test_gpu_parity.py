"""GPU parity tests of the hot loop (decoder fwd/bwd, kNN repulsion, fused Adam) through the C ABI,
against the golden fixtures produced by the reference and against the CPU oracle.

Tolerances (float32 path; north_star: 1e-3 per-point L2; SURVEY section 8c protocol P1-P4):
  logits / gradients   : relative 1e-4 of the tensor's max magnitude
  P1 teacher-forced    : |dx| <= 1e-6 per coordinate for one Adam step from the oracle state - every coordinate
                         (measured max 3e-8), and the hot kernel's gradient itself (recovered from Adam's first
                         moment) within 5e-6 of the reference autograd gradient's maximum (measured 4e-7 ... 6e-7)
  P2 free-running 10   : per-point L2 <= 1e-3 (the trajectory is chaotic - SURVEY F6)
  P4 sharding          : bitwise
"""
import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu

PL = ("xz", "xy", "yz")


@pytest.fixture(scope="module")
def restorer(np_weights):
    import ifdefense_amd as I
    r = I.Restorer(I.weights.pack_state_dict(np_weights), device="cuda:0")
    yield r
    r.close()


@pytest.fixture(scope="module")
def planes2(golden):
    return {pl: torch.from_numpy(golden["planes01"][:, i]) for i, pl in enumerate(PL)}


def _rel(a, b):
    return float(np.abs(a - b).max() / np.abs(b).max())


def test_decode_logits_and_input_gradient(restorer, golden, planes2):
    p = torch.from_numpy(golden["init_points"][:2])
    logits, grad = restorer.decode(p, planes2, want_grad=True)
    assert _rel(logits.cpu().numpy(), golden["dec_logits"]) < 1e-5
    assert _rel(grad.cpu().numpy(), golden["dec_dlogit_dp"]) < 1e-4
    only = restorer.decode(p, planes2)
    assert torch.equal(only, logits)


def test_decode_ragged_k(restorer, golden, planes2, oracle_weights):
    from oracle import convonet_oracle as O
    for K in (1, 31, 33, 100, 1000):
        p = torch.from_numpy(golden["init_points"][:2, :K]).clone()
        ref = O.decode_logits(oracle_weights, p, planes2).numpy()
        got = restorer.decode(p, planes2).cpu().numpy()
        assert got.shape == (2, K) and _rel(got, ref) < 1e-5, K


def test_decode_clamped_coordinates(restorer, planes2, oracle_weights):
    """Points outside the padded cube: coordinates are clamped by assignment => zero plane gradient."""
    from oracle import convonet_oracle as O
    g = torch.Generator().manual_seed(5)
    p = (torch.rand(2, 64, 3, generator=g) - 0.5) * 1.4          # up to +-0.7 > 0.55
    pr = p.clone().requires_grad_()
    ref = O.decode_logits(oracle_weights, pr, planes2)
    ref.sum().backward()
    logits, grad = restorer.decode(p, planes2, want_grad=True)
    assert _rel(logits.cpu().numpy(), ref.detach().numpy()) < 1e-5
    assert _rel(grad.cpu().numpy(), pr.grad.numpy()) < 1e-4


def test_repulsion_loss_knn_and_gradient(restorer, golden):
    from oracle import convonet_oracle as O
    p = torch.from_numpy(golden["init_points"][:2])
    loss, grad, idx = restorer.repulsion_loss(p, want_grad=True, want_idx=True)
    np.testing.assert_allclose(loss.cpu().numpy(), golden["rep_loss_b"], rtol=1e-5)
    got, ref = idx.cpu().numpy(), golden["knn_idx"]
    same = np.array([[set(got[b, i]) == set(ref[b, i]) for i in range(1024)] for b in range(2)])
    print("kNN: identical 5-NN sets %d of %d, identical order %.5f" % (same.sum(), same.size, (got == ref).mean()))
    assert same.all()                           # measured: every one of the 2048 sets
    assert (got == ref).mean() > 0.999          # and the same order (sorted by distance)
    pr = p.clone().requires_grad_()
    O.repulsion_loss(pr).sum().backward()
    assert _rel(grad.cpu().numpy(), pr.grad.numpy()) < 1e-4


def test_repulsion_edge_sizes(restorer):
    from oracle import convonet_oracle as O
    g = torch.Generator().manual_seed(3)
    for K in (6, 7, 64, 513, 1024):
        p = torch.rand(3, K, 3, generator=g) - 0.5
        ref = O.repulsion_loss(p).numpy()
        got = restorer.repulsion_loss(p).cpu().numpy()
        np.testing.assert_allclose(got, ref, rtol=2e-5, err_msg=str(K))


def test_repulsion_duplicate_points(restorer):
    """Exact duplicates (init draws indices with replacement): distance below eps => clamped, no NaN."""
    g = torch.Generator().manual_seed(4)
    p = torch.rand(1, 512, 3, generator=g) - 0.5
    p = torch.cat([p, p], dim=1)                                  # every point has an exact twin
    loss, grad = restorer.repulsion_loss(p, want_grad=True)
    assert torch.isfinite(loss).all() and torch.isfinite(grad).all()


def test_p1_teacher_forced_single_steps(restorer, golden, planes2):
    flips = 0
    for t in (0, 1, 9, 49):
        x = torch.from_numpy(golden[f"traj{t}_x"])
        state = (torch.from_numpy(golden[f"traj{t}_m"]), torch.from_numpy(golden[f"traj{t}_v"]), t)
        out, (m1, v1, t1) = restorer.optimize_points(x, planes2, rep_weight=500.0, steps=1, state=state,
                                                     normalize=False, return_state=True)
        d = np.abs(out.cpu().numpy() - golden[f"traj{t}_x_next"])
        bad = d > 1e-6
        flips += int(bad.sum())
        assert bad.sum() == 0, (t, float(d.max()), int(bad.sum()))     # measured: 0 of 6144 at every t <= 50
        assert t1 == t + 1
    print("P1: coordinates off by > 1e-6 over 4 teacher-forced steps:", flips, "of", 4 * 2 * 1024 * 3)


def test_p1_late_steps_t100_t500_and_loss_batch(restorer, golden, planes2, oracle_weights):
    """G4 at Adam t = 100 and 500 (fixtures from a 500-step run of the reference), and the 1/B loss factor at the
    reference batch size 192 (G3) against the oracle."""
    import os
    from oracle import convonet_oracle as O
    gl = np.load(os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden", "convonet_golden_long.npz"))
    for t in (99, 499):
        state = (torch.from_numpy(gl[f"traj{t}_m"]), torch.from_numpy(gl[f"traj{t}_v"]), t)
        out, loss = restorer.optimize_points(torch.from_numpy(gl[f"traj{t}_x"]), planes2, rep_weight=500.0, steps=1,
                                             state=state, normalize=False, return_loss=True)
        d = np.abs(out.cpu().numpy() - gl[f"traj{t}_x_next"])
        print("P1 t=%d: coordinates off by > 1e-6: %d of %d (max %.2e)" % (t + 1, (d > 1e-6).sum(), d.size, d.max()))
        assert (d > 1e-6).sum() == 0, (t, float(d.max()), int((d > 1e-6).sum()))     # measured: 0 of 6144, max 3e-8
        loss = loss.cpu().numpy().astype(np.float64)
        np.testing.assert_allclose(loss[:, 0].sum() / 2, gl[f"traj{t}_loss"][0], rtol=1e-5)
        np.testing.assert_allclose(loss[:, 1].mean() * 500.0, gl[f"traj{t}_loss"][1], rtol=1e-4)
    x = torch.from_numpy(golden["traj9_x"])
    m, v = torch.from_numpy(golden["traj9_m"]), torch.from_numpy(golden["traj9_v"])
    for lb in (1, 192):
        xr = x.clone().requires_grad_()
        total, _, _, _ = O.losses(oracle_weights, xr, planes2, 500.0, loss_batch=lb)
        total.backward()
        ref, _, _ = O.adam_step(x, xr.grad, m, v, 10)
        got = restorer.optimize_points(x, planes2, rep_weight=500.0, steps=1, state=(m, v, 9), normalize=False,
                                       loss_batch=lb)
        d = np.abs(got.cpu().numpy() - ref.numpy())
        assert (d > 1e-6).sum() == 0, (lb, float(d.max()), int((d > 1e-6).sum()))


def test_p2_free_running_10_and_50_steps(restorer, golden, planes2):
    init = torch.from_numpy(golden["init_points"][:2])
    x10 = restorer.optimize_points(init, planes2, rep_weight=500.0, steps=10, normalize=False)
    d10 = np.linalg.norm(x10.cpu().numpy() - golden["traj9_x_next"], axis=-1)
    assert d10.max() < 1e-3, float(d10.max())
    x50 = restorer.optimize_points(init, planes2, rep_weight=500.0, steps=50, normalize=False)
    d50 = np.linalg.norm(x50.cpu().numpy() - golden["traj49_x_next"], axis=-1)
    floor = golden["selfdiv51"]
    print("P2: 10 steps max %.2e | 50 steps max %.2e mean %.2e frac>1e-3 %.4f (reference self-divergence @51: "
          "max %.2e mean %.2e frac %.4f)" % (d10.max(), d50.max(), d50.mean(), (d50 > 1e-3).mean(), *floor))
    assert (d50 > 1e-3).mean() < 0.05 and np.median(d50) < 1e-4


def test_loss_report_matches_reference(restorer, golden, planes2):
    init = torch.from_numpy(golden["init_points"][:2])
    _, loss = restorer.optimize_points(init, planes2, rep_weight=500.0, steps=1, normalize=False, return_loss=True)
    loss = loss.cpu().numpy().astype(np.float64)
    occ, rep = golden["traj0_loss"]
    np.testing.assert_allclose(loss[:, 0].sum() / 2, occ, rtol=1e-5)         # K * mean_{B,K}
    np.testing.assert_allclose(loss[:, 1].mean() * 500.0, rep, rtol=1e-5)


def test_end_to_end_21_steps_normalised(restorer, golden, oracle_weights):
    from oracle import convonet_oracle as O
    proc = [golden["proc_pad"][b, :golden["proc_len"][b]] for b in range(4)]
    sel = torch.from_numpy(np.stack([proc[b][golden["sel_idx"][b]] for b in range(4)]))
    planes4 = O.encode_inputs(oracle_weights, sel)              # encoder from the oracle here; HIP encoder has its own test
    out = restorer.optimize_points(torch.from_numpy(golden["init_points"]), planes4, rep_weight=500.0, iterations=20)
    out = out.cpu().numpy()
    d = np.linalg.norm(out - golden["e2e20_out"], axis=-1)
    print("e2e 21 steps + normalisation: median %.2e max %.2e, points > 1e-3: %d, > 1e-2: %d of %d" %
          (np.median(d), d.max(), (d > 1e-3).sum(), (d > 1e-2).sum(), d.size))
    # measured (round 3): median 5.0e-7, max 5.5e-4, no point beyond 1e-3 - asserted at 2x, so that a regression of a few
    # flipped neighbours fails (the north-star bound, 1e-3 per point, holds for every point here)
    assert np.median(d) < 1e-6 and d.max() < 1.1e-3 and (d > 1e-3).sum() == 0, (np.median(d), d.max())
    np.testing.assert_allclose(np.linalg.norm(out, axis=-1).max(axis=1), 1.0, rtol=1e-6)
    assert np.abs(out.mean(axis=1)).max() < 1e-3


def test_normalize_unit_sphere(restorer, golden):
    from oracle import convonet_oracle as O
    x = torch.from_numpy(golden["traj_final51"])
    got = restorer.normalize_batch_pc(x).cpu().numpy()
    np.testing.assert_allclose(got, O.normalize_batch_pc(x).numpy(), rtol=0, atol=1e-6)


def test_p4_determinism_and_sharding_invariance(restorer, golden, oracle_weights):
    from oracle import convonet_oracle as O
    import ifdefense_amd as I
    proc = [golden["proc_pad"][b, :golden["proc_len"][b]] for b in range(4)]
    sel = torch.from_numpy(np.stack([proc[b][golden["sel_idx"][b]] for b in range(4)]))
    planes4 = I.planes_to_channel_last(O.encode_inputs(oracle_weights, sel)).cuda()
    init = torch.from_numpy(golden["init_points"]).cuda()
    kw = dict(rep_weight=500.0, iterations=30, loss_batch=4)
    a = restorer.optimize_points(init, planes4, **kw)
    b = restorer.optimize_points(init, planes4, **kw)
    assert torch.equal(a, b)                                     # run-to-run bitwise
    lo = restorer.optimize_points(init[:2], planes4[:2], **kw)
    hi = restorer.optimize_points(init[2:], planes4[2:], **kw)
    assert torch.equal(torch.cat([lo, hi]), a)                   # shard-and-concatenate bitwise


def test_certified_neighbour_lists_equal_exact_scan(restorer, golden, oracle_weights):
    """The O(32)-per-point certified list path must reproduce the brute-force 5-NN scan bit for bit."""
    from oracle import convonet_oracle as O
    import ifdefense_amd as I
    proc = [golden["proc_pad"][b, :golden["proc_len"][b]] for b in range(4)]
    sel = torch.from_numpy(np.stack([proc[b][golden["sel_idx"][b]] for b in range(4)]))
    planes4 = I.planes_to_channel_last(O.encode_inputs(oracle_weights, sel)).cuda()
    init = torch.from_numpy(golden["init_points"]).cuda()
    for K in (1024, 333, 40):
        a = restorer.optimize_points(init[:, :K], planes4, rep_weight=500.0, iterations=150, normalize=False)
        c = restorer.counters()
        b = restorer.optimize_points(init[:, :K], planes4, rep_weight=500.0, iterations=150, normalize=False,
                                     knn_scan_every_step=True)
        assert torch.equal(a, b), (K, float((a - b).abs().max()))
        ev = c["knn_rebuilds"] / (4 * 8)
        print("K=%d: %.1f synchronous list rebuilds per cloud over 151 steps, %d certificate failures" %
              (K, ev, c["knn_brute_scans"]))
        # lists must actually be reused across steps; wave-steps with a certificate that expired a step early are answered
        # by the exact per-point query (the soft margin is tuned for that trade, knn_device.h IFD_SOFT_SLACK): measured
        # 264 of 4832 wave-steps at K = 1024
        assert ev < 30 and c["knn_brute_scans"] <= 0.12 * 8 * 4 * 151, c


def test_split_clouds_are_bit_identical_to_one_workgroup_per_cloud(restorer, golden, oracle_weights):
    """Cooperative mode (ifd_opt_params.split): 2 or 4 workgroups - CUs - per cloud, used for the clouds of a partial round
    (a launch with fewer clouds than CUs: one GPU's shard of a file spread over 8 GPUs).  Every member owns a quarter / half
    of the points; neighbour terms cross CUs as integer atomics, positions and certificate maxima through global memory.
    Points, both Adam moments and the reported losses must equal the one-workgroup kernel's bit for bit - full and ragged
    clouds, certified lists and the exact scan, with and without the final normalisation, and resumed from a state."""
    from oracle import convonet_oracle as O
    import ifdefense_amd as I
    proc = [golden["proc_pad"][b, :golden["proc_len"][b]] for b in range(4)]
    sel = torch.from_numpy(np.stack([proc[b][golden["sel_idx"][b]] for b in range(4)]))
    planes4 = I.planes_to_channel_last(O.encode_inputs(oracle_weights, sel)).cuda()
    init = torch.from_numpy(golden["init_points"]).cuda()

    def same(a, b):
        return (torch.equal(a[0], b[0]) and torch.equal(a[1][0], b[1][0]) and torch.equal(a[1][1], b[1][1]) and
                torch.equal(a[2], b[2]))

    for K, steps in ((1024, 120), (777, 60), (300, 60)):
        for kw in (dict(normalize=True), dict(normalize=False, knn_scan_every_step=True)):
            run = lambda split, st=None, n=steps: restorer.optimize_points(
                init[:, :K].contiguous(), planes4, rep_weight=500.0, steps=n, loss_batch=192, return_state=True,
                return_loss=True, split=split, state=st, **kw)
            ref = run(1)
            for split in (2, 4, 0):
                assert same(run(split), ref), (K, kw, split)
        # resume: 20 steps, then 7 more from the returned state (teacher-forcing path of the split kernel)
        first = restorer.optimize_points(init[:, :K].contiguous(), planes4, rep_weight=500.0, steps=20, loss_batch=192,
                                         normalize=False, return_state=True, split=1)
        a = restorer.optimize_points(first[0], planes4, rep_weight=500.0, steps=7, loss_batch=192, normalize=False,
                                     state=first[1], return_state=True, return_loss=True, split=1)
        b = restorer.optimize_points(first[0], planes4, rep_weight=500.0, steps=7, loss_batch=192, normalize=False,
                                     state=first[1], return_state=True, return_loss=True, split=4)
        assert same(a, b), K
    # rep_weight 0 (no exchange of neighbour terms at all) and tiny clouds (never split)
    for K, rw in ((1024, 0.0), (100, 500.0)):
        a = restorer.optimize_points(init[:, :K].contiguous(), planes4, rep_weight=rw, steps=15, split=1)
        for split in (2, 4):
            assert torch.equal(restorer.optimize_points(init[:, :K].contiguous(), planes4, rep_weight=rw, steps=15, split=split), a)


def test_automatic_split_of_a_partial_round(restorer):
    """More clouds than CUs, not a multiple: whole rounds run one workgroup per cloud, the remainder split over 4 (up to
    a quarter of the CUs) or 2 CUs per cloud - same file, bit for bit, as with splitting disabled."""
    import bench
    n_cu = torch.cuda.get_device_properties(0).multi_processor_count
    for extra in (n_cu // 8 + 3, n_cu // 2 - 5):                 # a remainder that takes S = 4, one that takes S = 2
        n = n_cu + extra
        x = torch.from_numpy(bench.synth_clouds(n)).cuda()
        prep = restorer.prepare(x, restorer.sor(x), seed=5)
        planes = restorer.encode_inputs(prep["sel"], prep["t_per_cloud"])
        lb = torch.full((n,), 192, dtype=torch.int32, device="cuda")
        a = restorer.optimize_points(prep["init"], planes, rep_weight=500.0, steps=25, loss_batch=lb, split=1, return_loss=True)
        b = restorer.optimize_points(prep["init"], planes, rep_weight=500.0, steps=25, loss_batch=lb, split=0, return_loss=True)
        assert torch.equal(a[0], b[0]) and torch.equal(a[1], b[1]), extra
        assert bool(torch.isfinite(b[0]).all())


def test_printing_seam_matches_reference_prints(restorer, golden, planes2, capsys):
    """optimize_points(..., printing=True) (opt_defense.py:229-236): the three printed quantities at iterations 0 and 100 -
    loss terms against the reference's own values at those iterations, the mean occupancy probability against the oracle -
    and a result bit-identical to the silent run."""
    from oracle import convonet_oracle as O
    init = torch.from_numpy(golden["init_points"][:2])
    quiet = restorer.optimize_points(init, planes2, rep_weight=500.0, iterations=120)
    capsys.readouterr()
    loud = restorer.optimize_points(init, planes2, rep_weight=500.0, iterations=120, printing=True)
    out = capsys.readouterr().out
    assert torch.equal(quiet, loud)
    lines = out.strip().splitlines()
    assert [l.split(",")[0] for l in lines if l.startswith("iter")] == ["iter 0", "iter 100"], out
    occ0 = float(lines[1].split("occ loss:")[1].split(",")[0])
    rep0 = float(lines[1].split("rep loss:")[1])
    assert abs(occ0 - golden["traj0_loss"][0]) < 2e-4 * golden["traj0_loss"][0] and abs(rep0 - golden["traj0_loss"][1]) < 1e-3
    prob0 = float(lines[2].split("occ value mean:")[1])
    ref_prob = float(torch.sigmoid(torch.from_numpy(golden["dec_logits"])).mean())
    assert abs(prob0 - ref_prob) < 1e-4, (prob0, ref_prob)
    assert lines[3].startswith("iter 100") and "occ value mean:" in lines[5]


def test_repulsion_accumulators_do_not_wrap_on_a_tight_cluster(restorer, golden, planes2):
    """The optimiser accumulates the repulsion gradient in 32-bit fixed point (2^-23; x and y share a 64-bit word): a sum
    beyond +-256 would wrap silently.  Terms are bounded by 1.4 and a point is among the 5 nearest of at most 5 x 12 others
    (kissing number), so sums stay below ~90 - checked here on the worst case the loss allows: clouds drawn into a few
    blobs at the distance h where the terms are largest, against ifd_repulsion's 64-bit accumulators (2^-40)."""
    g = torch.Generator().manual_seed(3)
    centres = (torch.rand(2, 6, 3, generator=g) - 0.5) * 0.6
    which = torch.randint(0, 6, (2, 1024), generator=g)
    x = torch.gather(centres, 1, which[..., None].expand(2, 1024, 3)) + 0.03 * torch.randn(2, 1024, 3, generator=g) * 0.35
    _, grad_wide = restorer.repulsion_loss(x, want_grad=True)                 # d(sum_b loss_b)/dp, wide accumulators
    zero = {k: torch.zeros_like(v) for k, v in planes2.items()}
    base, (m0, _, _) = restorer.optimize_points(x, zero, rep_weight=0.0, steps=1, normalize=False, loss_batch=1, return_state=True, split=1)
    _, (m1, _, _) = restorer.optimize_points(x, zero, rep_weight=500.0, steps=1, normalize=False, loss_batch=1, return_state=True, split=1)
    g_rep = (m1 - m0).cpu().numpy() / 0.1 / 500.0                             # the kernel's repulsion gradient alone
    ref = grad_wide.cpu().numpy()
    scale = 1024 * 5.0                                                         # largest accumulated sum in fixed-point units
    print("tight clusters: largest accumulated repulsion sum %.1f (wraps at 256), max error %.2e of max" %
          (np.abs(ref).max() * scale, np.abs(g_rep - ref).max() / np.abs(ref).max()))
    assert np.abs(ref).max() * scale < 128.0
    assert np.abs(g_rep - ref).max() < 2e-5 * np.abs(ref).max()


def test_repulsion_overflow_is_reported_not_silent(restorer, golden, planes2):
    """Runtime detection behind the headroom argument above (return a status; the reference's failure path
    in this loop is exit(-1), defense/repulsion_loss.py:25-39).  With the reference's radius / h no launch raises the sticky
    overflow word; with RepulsionLoss(radius=50) a single neighbour term is ~1e3 - it saturates the 32-bit fixed-point
    conversion at +-256 - and ifd_optimize_status returns IFD_ERR_OVERFLOW once, for the persistent kernel, the split kernel
    and the launch-per-step path of clouds beyond 1024 points."""
    import ifdefense_amd as I
    x = torch.from_numpy(golden["init_points"][:2])
    restorer.check_status()                                                    # nothing pending from earlier tests
    restorer.optimize_points(x, planes2, rep_weight=500.0, steps=5, check=True)
    for kw in (dict(split=1), dict(split=2)):
        with pytest.raises(I.IfdError, match="-6.*fixed-point"):
            restorer.optimize_points(x, planes2, rep_weight=500.0, steps=2, rep_radius=50.0, check=True, **kw)
        restorer.check_status()                                                # reported once, then clear
    big = torch.cat([x, x + 0.003], dim=1)                                     # 2048 points: two launches per step
    restorer.optimize_points(big, planes2, rep_weight=500.0, steps=2, check=True)
    with pytest.raises(I.IfdError, match="-6"):
        restorer.optimize_points(big, planes2, rep_weight=500.0, steps=2, rep_radius=50.0, check=True)
    # an unchecked failure stays pending until somebody asks (the drivers of pipeline.py ask once per file)
    restorer.optimize_points(x, planes2, rep_weight=500.0, steps=1, rep_radius=50.0, check=False)
    restorer.optimize_points(x, planes2, rep_weight=500.0, steps=1, check=False)
    with pytest.raises(I.IfdError, match="-6"):
        restorer.check_status()


def test_split_cloud_wait_is_bounded_and_reported(restorer, golden, planes2, np_weights, monkeypatch):
    """The members of a split cloud wait for each other across CUs (knn_device.h coop_wait).  The wait is bounded: with one
    member's arrivals suppressed (test hook IFD_TEST_COOP_DROP, honoured only with IFD_ENABLE_TEST_HOOKS=1 and read when the context
    is created) and the bound lowered to 20 ms, every other member gives up, the launch ends within milliseconds instead of
    hanging the GPU and ifd_optimize_status returns IFD_ERR_TIMEOUT.  A context whose launch timed out keeps working WITHOUT the
    host having asked for the status in between (round-4 advisor: the waiters look at the current call's time-out word, which
    every optimise call clears, not at the sticky one): the next launch is clean and bit-identical to the unsplit kernel, and
    the sticky word still reports the earlier time-out afterwards."""
    import time
    import ifdefense_amd as I
    x = torch.from_numpy(golden["init_points"][:2]).cuda()
    ref = restorer.optimize_points(x, planes2, rep_weight=500.0, steps=30, split=1)
    monkeypatch.setenv("IFD_COOP_TIMEOUT_MS", "20")
    monkeypatch.setenv("IFD_ENABLE_TEST_HOOKS", "1")
    for split, drop, rw in ((2, 1, 500.0), (4, 3, 500.0), (4, 0, 0.0)):
        monkeypatch.setenv("IFD_TEST_COOP_DROP", str(drop))
        bad = I.Restorer(I.weights.pack_state_dict(np_weights), device="cuda:0")        # the hooks are read here
        monkeypatch.delenv("IFD_TEST_COOP_DROP")
        good = I.Restorer(I.weights.pack_state_dict(np_weights), device="cuda:0")       # same 20 ms bound, every member arrives
        try:
            torch.cuda.synchronize()
            t0 = time.perf_counter()
            with pytest.raises(I.IfdError, match="-5.*wait"):
                bad.optimize_points(x, planes2, rep_weight=rw, steps=300, split=split)
            dt = time.perf_counter() - t0
            print("split %d, member %d never arrives (rep_weight %g): launch ended after %.0f ms with IFD_ERR_TIMEOUT" % (split, drop, rw, dt * 1e3))
            assert dt < 2.0, dt
            assert torch.equal(good.optimize_points(x, planes2, rep_weight=500.0, steps=30, split=split), ref)
            # a timed-out launch whose status nobody fetched does not poison the context's later launches ...
            bad.optimize_points(x, planes2, rep_weight=rw, steps=300, split=split, check=False)
            torch.cuda.synchronize()
            out = bad.optimize_points(x, planes2, rep_weight=500.0, steps=30, split=1, check=False)
            assert torch.equal(out, ref)
            with pytest.raises(I.IfdError, match="-5.*wait"):       # ... and the sticky word still tells
                bad.check_status()
            bad.check_status()                                      # reported once
        finally:
            bad.close()
            good.close()
    # the hook alone, without IFD_ENABLE_TEST_HOOKS, does nothing
    monkeypatch.delenv("IFD_ENABLE_TEST_HOOKS")
    monkeypatch.setenv("IFD_TEST_COOP_DROP", "1")
    plain = I.Restorer(I.weights.pack_state_dict(np_weights), device="cuda:0")
    try:
        assert torch.equal(plain.optimize_points(x, planes2, rep_weight=500.0, steps=30, split=2), ref)
    finally:
        plain.close()


def test_rep_weight_zero_and_small_k(restorer, golden, planes2, oracle_weights):
    from oracle import convonet_oracle as O
    init = torch.from_numpy(golden["init_points"][:2, :100]).clone()
    ref = O.optimize_points(oracle_weights, init, planes2, rep_weight=0.0, iterations=4, normalize=False)
    got = restorer.optimize_points(init, planes2, rep_weight=0.0, iterations=4, normalize=False)
    assert np.abs(got.cpu().numpy() - ref.numpy()).max() < 1e-5
    ref = O.optimize_points(oracle_weights, init, planes2, rep_weight=500.0, iterations=4, normalize=False)
    got = restorer.optimize_points(init, planes2, rep_weight=500.0, iterations=4, normalize=False)
    assert np.linalg.norm(got.cpu().numpy() - ref.numpy(), axis=-1).max() < 1e-4


@pytest.mark.both_precisions
def test_large_clouds_more_than_1024_points(restorer, golden, planes2, oracle_weights, precision_mode):
    """--sample_npoint beyond the persistent kernel's 1024 points (the reference has no limit, opt_defense.py:27): the
    two-launch-per-step path against the oracle - repulsion loss / exact 5-NN, P1 (one Adam step from the oracle's
    state at t = 1, 2 and 6: |dx| <= 1e-6 per coordinate, gradient to 1e-4 of its maximum), P2 (a short free run) and
    the unit-sphere normalisation."""
    from oracle import convonet_oracle as O
    g = torch.Generator().manual_seed(17)
    for K in (1025, 2048, 3000):
        base = torch.from_numpy(golden["init_points"][:2])                               # [2,1024,3] near the surface
        idx = torch.randint(0, 1024, (2, K), generator=g)
        init = (torch.gather(base, 1, idx[..., None].expand(2, K, 3)) + 0.01 * torch.randn(2, K, 3, generator=g)).clamp(-0.45, 0.45)
        # repulsion: loss and neighbour sets
        loss, knn = restorer.repulsion_loss(init, want_idx=True)
        ref_idx = O.knn_point(5, init)
        np.testing.assert_allclose(loss.cpu().numpy(), O.repulsion_loss(init, ref_idx).numpy(), rtol=2e-5)
        same = (np.sort(knn.cpu().numpy(), -1) == np.sort(ref_idx.numpy(), -1)).all(-1)
        print("large K=%d: kNN sets equal for %d of %d points" % (K, same.sum(), same.size))
        assert (~same).sum() <= 2, K              # measured: all equal (the reference's expanded-form distances may reorder a near-tie)
        # P1: single steps from the oracle's own trajectory
        x, m, v = init.clone(), torch.zeros_like(init), torch.zeros_like(init)
        for t in range(1, 7):
            # the oracle's objective on the exact neighbour sets: the reference's expanded-form f32 distances pick
            # another neighbour at a near-tie now and then (compared above, and bounded here), which is a property of
            # its kNN, not of the step being checked
            _, idx_hip = restorer.repulsion_loss(x, want_idx=True)
            idx_ref = O.knn_point(5, x)
            assert (np.sort(idx_hip.cpu().numpy(), -1) == np.sort(idx_ref.numpy(), -1)).all(-1).mean() > 0.998, (K, t)
            xg = x.clone().requires_grad_(True)
            occ = O.losses(oracle_weights, xg, planes2, 0.0)[0]
            (occ + O.repulsion_loss(xg, idx_hip.cpu().long()).sum() / 2.0 * 500.0).backward()
            x_next, m_next, v_next = O.adam_step(x, xg.grad, m, v, t)
            if t in (1, 2, 6):
                x1, (m1, v1, _) = restorer.optimize_points(x, planes2, rep_weight=500.0, steps=1, normalize=False,
                                                           state=(m, v, t - 1), return_state=True)
                flips = int((np.abs(x1.cpu().numpy() - x_next.numpy()) > 1e-6).sum())
                g_hip = (m1.cpu().numpy() - 0.9 * m.numpy()) / 0.1
                gerr = np.abs(g_hip - xg.grad.numpy()).max() / np.abs(xg.grad.numpy()).max()
                print("large K=%d t=%d: coordinates off by > 1e-6: %d of %d, gradient error %.1e of max" % (K, t, flips, x1.numel(), gerr))
                assert flips == 0 and gerr < 1e-4, (K, t)
            x, m, v = x_next, m_next, v_next
        # P2 + normalisation: 6 free steps
        ref = O.optimize_points(oracle_weights, init, planes2, rep_weight=500.0, iterations=5, normalize=True)
        got, lv = restorer.optimize_points(init, planes2, rep_weight=500.0, iterations=5, normalize=True, return_loss=True)
        err = np.abs(got.cpu().numpy() - ref.numpy()).max(-1)
        print("large K=%d free run of 6 steps + normalisation: %d of %d points off by > 2e-5 (max %.1e)" % (K, (err > 2e-5).sum(), err.size, err.max()))
        # (a near-tie neighbour choice moves a point by ~lr per step; these clouds are crowded on purpose - up to three
        # points per source point - and the measured counts are 0 / 4 / 17 of 2050 / 4096 / 6000)
        max_off, max_err = {1025: (2, 1e-5), 2048: (8, 1e-3), 3000: (34, 8e-3)}[K]      # 2x the measured 0 / 4 / 17, 2.2e-6 / 4.8e-4 / 3.6e-3
        assert (err > 2e-5).sum() <= max_off and err.max() < max_err, (K, int((err > 2e-5).sum()), float(err.max()))
        assert abs(float(got.norm(dim=-1).max()) - 1.0) < 1e-5 and torch.isfinite(lv).all()


def test_large_clouds_stream_groups_are_bit_identical_to_one_group(restorer, np_weights, monkeypatch):
    """Clouds of more than 1024 points, 32 clouds and more per call: the batch goes as up to four groups on streams of the context
    (api.cpp large_optimize_in_groups; a launch pair ends with its slowest cloud, the groups fill each other's idle CUs).  Same
    kernels, same arithmetic per cloud: points, moments and losses bit for bit those of ONE group (a context created with the
    measurement hook), for uneven groups, own and caller-held moments, per-cloud loss batches, and a second call on the same
    context (the side streams are joined into the caller's)."""
    import ifdefense_amd as I
    import bench
    B, K = 37, 1100                                              # groups of 10, 10, 10, 7
    x = torch.from_numpy(bench.synth_clouds(B))
    prep = restorer.prepare(x, restorer.sor(x), n_sel=600, n_opt=K, seed=5)
    planes = restorer.encode_inputs(prep["sel"], prep["t_per_cloud"])
    lb = torch.tensor([3 + (i % 5) for i in range(B)], dtype=torch.int32)
    monkeypatch.setenv("IFD_ENABLE_TEST_HOOKS", "1")
    monkeypatch.setenv("IFD_TEST_LARGE_GROUPS", "1")
    one = I.Restorer(I.weights.pack_state_dict(np_weights), device="cuda:0")
    monkeypatch.delenv("IFD_TEST_LARGE_GROUPS")
    try:
        for r_ in (restorer, restorer):                          # twice: the second call reuses the side streams and the workspace
            got = r_.optimize_points(prep["init"], planes, rep_weight=500.0, steps=6, loss_batch=lb, normalize=True, return_state=True, return_loss=True)
            ref = one.optimize_points(prep["init"], planes, rep_weight=500.0, steps=6, loss_batch=lb, normalize=True, return_state=True, return_loss=True)
            assert torch.equal(got[0], ref[0]) and torch.equal(got[1][0], ref[1][0]) and torch.equal(got[1][1], ref[1][1]) and torch.equal(got[2], ref[2])
        # caller-held moments carried across two calls (t0 > 0), no normalisation
        a1 = restorer.optimize_points(prep["init"], planes, rep_weight=500.0, steps=3, normalize=False, return_state=True)
        a2 = restorer.optimize_points(a1[0], planes, rep_weight=500.0, steps=3, normalize=False, state=a1[1], return_state=True)
        b1 = one.optimize_points(prep["init"], planes, rep_weight=500.0, steps=3, normalize=False, return_state=True)
        b2 = one.optimize_points(b1[0], planes, rep_weight=500.0, steps=3, normalize=False, state=b1[1], return_state=True)
        assert torch.equal(a2[0], b2[0]) and torch.equal(a2[1][0], b2[1][0]) and torch.equal(a2[1][1], b2[1][1])
        whole = restorer.optimize_points(prep["init"], planes, rep_weight=500.0, steps=6, normalize=False)
        assert torch.equal(whole, a2[0])                          # (and the cut run equals the uncut one)
        assert float(got[0].norm(dim=-1).max()) == pytest.approx(1.0, abs=1e-5) and bool(torch.isfinite(got[0]).all())
    finally:
        one.close()


def test_large_cloud_lists_equal_exact_scan(restorer, golden):
    """Round 6: 1025 ... 4096 optimised points keep certified neighbour lists too (optimize.hip large_step_lists_kernel): per-point
    lists in global memory, a certificate per point and step, an exact wave-cooperative query (and a new list) for every point whose
    certificate does not hold.  Every path selects the five smallest (distance, index) pairs, so the launch is
    bit-identical - points, moments, losses - to the exact brute-force scan at every step (knn_scan_every_step); on clean, clustered
    (config #3-like) and heavily duplicated (K = 256 sources) clouds, ragged point counts, enough steps for every list to expire
    several times."""
    import bench
    clouds = bench.synth_clouds(6, seed=77)
    makes = (("clean", lambda c: c), ("clusters", bench.knn_attack_like), ("256 sources", lambda c: bench.subsample_like(c, 256)))
    for name, make in makes:
        x = torch.from_numpy(make(clouds)).cuda()
        for K, steps in ((2048, 150), (1500, 60), (4096, 25), (1025, 40)):
            prep = restorer.prepare(x, restorer.sor(x), n_opt=K, seed=5)
            planes = restorer.encode_inputs(prep["sel"], prep["t_per_cloud"])
            a, (ma, va, _), la = restorer.optimize_points(prep["init"], planes, rep_weight=500.0, steps=steps, normalize=False,
                                                          return_state=True, return_loss=True)
            c = restorer.counters()
            b, (mb, vb, _), lb = restorer.optimize_points(prep["init"], planes, rep_weight=500.0, steps=steps, normalize=False,
                                                          return_state=True, return_loss=True, knn_scan_every_step=True)
            frac = c["knn_exact_evals"] / (6.0 * K * steps)
            print("large lists, %s, K=%d, %d steps: %.1f whole-cloud list builds per cloud, %.2f %% of the point-steps resolved by the exact query" %
                  (name, K, steps, c["knn_rebuilds"] / 6.0, 100.0 * frac))
            assert torch.equal(a, b) and torch.equal(ma, mb) and torch.equal(va, vb) and torch.equal(la, lb), (name, K)
            assert c["knn_rebuilds"] == 6, c                                       # the lists were in use: built once per cloud, at the call's first step
            if name == "clean":
                assert frac < 0.15, (name, K, frac)                               # ... and they hold: a few per cent of the points are renewed per step
    # a resumed call (t0 > 0) builds its own lists: they never come from another call's workspace
    prep = restorer.prepare(x, restorer.sor(x), n_opt=2048, seed=6)
    planes = restorer.encode_inputs(prep["sel"], prep["t_per_cloud"])
    one = restorer.optimize_points(prep["init"], planes, rep_weight=500.0, steps=30, normalize=False)
    h, st = restorer.optimize_points(prep["init"], planes, rep_weight=500.0, steps=13, normalize=False, return_state=True)
    restorer.optimize_points(prep["init"][:2, :1100].contiguous(), planes[:2], rep_weight=500.0, steps=3)      # another call in between
    two = restorer.optimize_points(h, planes, rep_weight=500.0, steps=17, normalize=False, state=st)
    assert torch.equal(one, two)


@pytest.mark.both_precisions
def test_large_clouds_precision_and_reference_form_carried_through(restorer, golden, planes2, oracle_weights, precision_mode):
    """Round 6: ifd_opt_params.precision and .knn_reference_form apply above 1024 points as well (the occupancy half of the
    launch-per-step path runs the persistent kernel's tile, f32 or split precision; the step half can rank neighbours like the
    reference).  P1 at K = 2048 from the oracle's own state in the mode under test at the f32 bar, and the reference-form launch
    against the oracle run with the REFERENCE's kNN (no substitution of neighbour sets)."""
    from oracle import convonet_oracle as O
    g = torch.Generator().manual_seed(23)
    K = 2048
    base = torch.from_numpy(golden["init_points"][:2])
    idx = torch.randint(0, 1024, (2, K), generator=g)
    init = (torch.gather(base, 1, idx[..., None].expand(2, K, 3)) + 0.01 * torch.randn(2, K, 3, generator=g)).clamp(-0.45, 0.45)
    x, m, v = init.clone(), torch.zeros_like(init), torch.zeros_like(init)
    for t in (1, 2, 3):
        xg = x.clone().requires_grad_(True)
        total = O.losses(oracle_weights, xg, planes2, 500.0)[0]                      # the reference's own neighbour choice
        total.backward()
        x_next, m_next, v_next = O.adam_step(x, xg.grad, m, v, t)
        x1, (m1, _, _) = restorer.optimize_points(x, planes2, rep_weight=500.0, steps=1, normalize=False, state=(m, v, t - 1),
                                                  return_state=True, knn_reference_form=True)
        flips = int((np.abs(x1.cpu().numpy() - x_next.numpy()) > 1e-6).sum())
        g_hip = (m1.cpu().numpy() - 0.9 * m.numpy()) / 0.1
        gerr = np.abs(g_hip - xg.grad.numpy()).max(-1) / np.abs(xg.grad.numpy()).max()
        print("large K=%d t=%d [%s, reference-form kNN]: coordinates off by > 1e-6: %d of %d; gradient error median %.1e, points beyond 5e-6 of max: %d"
              % (K, t, precision_mode, flips, x1.numel(), np.median(gerr), int((gerr > 5e-6).sum())))
        assert np.median(gerr) < 2e-7 and (gerr > 5e-6).sum() <= 2 and flips <= 6, (t, flips)      # (<= 2 ReLU-boundary points, see test_gpu_split_precision)
        x, m, v = x_next, m_next, v_next
    # the mode really is in use above 1024 points: f32 and bf16x6 differ in the last bits, and each is reproducible
    a = restorer.optimize_points(init, planes2, rep_weight=500.0, steps=4, normalize=False)
    b = restorer.optimize_points(init, planes2, rep_weight=500.0, steps=4, normalize=False)
    other = restorer.optimize_points(init, planes2, rep_weight=500.0, steps=4, normalize=False,
                                     precision="f32" if precision_mode != "f32" else "bf16x6")
    assert torch.equal(a, b) and not torch.equal(a, other)
    assert float((a - other).abs().max()) < 1e-5


def test_large_clouds_beyond_4096_points(restorer, planes2, oracle_weights):
    """4097 ... 10,000 optimised points: the large path with its repulsion accumulators in global memory (large_step_kernel<true>;
    up to 4096 points they sit in LDS) against the oracle - loss and exact 5-NN, P1 at t = 1 and 2, a short free run with
    normalisation - and the 10,000-point limit itself."""
    from oracle import convonet_oracle as O
    g = torch.Generator().manual_seed(29)
    for K in (4097, 6000):
        init = (torch.rand(2, K, 3, generator=g) - 0.5) * 0.9
        loss, knn = restorer.repulsion_loss(init, want_idx=True)
        ref_idx = O.knn_point(5, init)
        np.testing.assert_allclose(loss.cpu().numpy(), O.repulsion_loss(init, ref_idx).numpy(), rtol=2e-5)
        same = (np.sort(knn.cpu().numpy(), -1) == np.sort(ref_idx.numpy(), -1)).all(-1)
        print("large K=%d: kNN sets equal for %d of %d points" % (K, same.sum(), same.size))
        assert (~same).sum() <= 2, K
        x, m, v = init.clone(), torch.zeros_like(init), torch.zeros_like(init)
        for t in (1, 2):
            _, idx_hip = restorer.repulsion_loss(x, want_idx=True)
            xg = x.clone().requires_grad_(True)
            occ = O.losses(oracle_weights, xg, planes2, 0.0)[0]
            (occ + O.repulsion_loss(xg, idx_hip.cpu().long()).sum() / 2.0 * 500.0).backward()
            x_next, m_next, v_next = O.adam_step(x, xg.grad, m, v, t)
            x1, (m1, v1, _) = restorer.optimize_points(x, planes2, rep_weight=500.0, steps=1, normalize=False,
                                                       state=(m, v, t - 1), return_state=True)
            flips = int((np.abs(x1.cpu().numpy() - x_next.numpy()) > 1e-6).sum())
            g_hip = (m1.cpu().numpy() - 0.9 * m.numpy()) / 0.1
            # (points drawn from the whole cube: one in a few thousand sits on a bilinear cell boundary or a ReLU kink of the
            # decoder, where the two implementations may take different sides - counted, like in the ONet test)
            gerr = np.abs(g_hip - xg.grad.numpy()).max(-1) / np.abs(xg.grad.numpy()).max()
            print("large K=%d t=%d: coordinates off by > 1e-6: %d of %d, gradient error median %.1e, points > 1e-4: %d" %
                  (K, t, flips, x1.numel(), np.median(gerr), (gerr > 1e-4).sum()))
            assert flips == 0 and np.median(gerr) < 1e-6 and (gerr > 1e-4).sum() <= 2, (K, t)
            x, m, v = x_next, m_next, v_next
        ref = O.optimize_points(oracle_weights, init, planes2, rep_weight=500.0, iterations=2, normalize=True)
        got, lv = restorer.optimize_points(init, planes2, rep_weight=500.0, iterations=2, normalize=True, return_loss=True)
        err = np.abs(got.cpu().numpy() - ref.numpy()).max(-1)
        print("large K=%d free run of 3 steps + normalisation: %d of %d points off by > 2e-5 (max %.1e)" % (K, (err > 2e-5).sum(), err.size, err.max()))
        max_off, max_err = {4097: (2, 1e-5), 6000: (4, 3e-4)}[K]                # 2x the measured 0 / 2, 4.2e-7 / 1.5e-4
        assert (err > 2e-5).sum() <= max_off and err.max() < max_err, (K, int((err > 2e-5).sum()), float(err.max()))
        assert abs(float(got.norm(dim=-1).max()) - 1.0) < 1e-5 and torch.isfinite(lv).all()
    big = (torch.rand(1, 10000, 3, generator=g) - 0.5) * 0.9                     # the limit: runs, and twice the same
    planes1 = {k: v[:1] for k, v in planes2.items()}
    a = restorer.optimize_points(big, planes1, rep_weight=500.0, iterations=1)
    b = restorer.optimize_points(big, planes1, rep_weight=500.0, iterations=1)
    assert torch.equal(a, b) and bool(torch.isfinite(a).all())


def test_bad_arguments_return_errors(restorer, planes2):
    import ifdefense_amd as I
    with pytest.raises(I.IfdError):
        restorer.optimize_points(torch.zeros(2, 10001, 3), planes2, iterations=1)     # K > 10,000
    with pytest.raises(I.IfdError):
        restorer.repulsion_loss(torch.zeros(1, 5, 3))                                  # K < 6
    with pytest.raises(I.IfdError):
        restorer.decode(torch.zeros(1, 8, 3), torch.zeros(1, 3, 64, 64, 16))


# ---------------------------------------------------------------------------------------------------
# pre-processing + encoder (rows A3-A8 of SURVEY section 8a)
# ---------------------------------------------------------------------------------------------------
def test_sor_mask_bit_exact(restorer, golden):
    keep, val = restorer.sor(torch.from_numpy(golden["raw"]), 2, 1.1, want_value=True)
    assert np.array_equal(keep.cpu().numpy().astype(bool), golden["sor_keep"])       # boolean work: bit-exact
    np.testing.assert_allclose(val.cpu().numpy(), golden["sor_value"], rtol=1e-9, atol=1e-18)


def test_sor_edge_sizes(restorer):
    from oracle import convonet_oracle as O
    g = torch.Generator().manual_seed(11)
    for K in (8, 100, 1024, 1500, 2048, 3001, 4096, 4097, 6001, 10000):      # > 4096: the float-in-LDS layout of sor_kernel
        pc = torch.randn(2 if K <= 4096 else 1, K, 3, generator=g)
        ref, ref_val = O.sor_keep_mask(pc)
        got, val = restorer.sor(pc, want_value=True)
        got = got.cpu().numpy().astype(bool)
        print("SOR K=%d: mask mismatches %d of %d" % (K, (got != ref.numpy()).sum(), got.size))
        assert np.array_equal(got, ref.numpy()), K                   # boolean work on fp64 statistics: bit-exact
        np.testing.assert_allclose(val.cpu().numpy(), ref_val.numpy(), rtol=1e-9, atol=1e-18)
    with pytest.raises(Exception):                                   # the limit: 10,000 points (prepare_kernel's LDS)
        restorer.sor(torch.zeros(1, 10001, 3))


def test_prepare_large_input_clouds(restorer):
    """10,000-point input clouds (the size of the resampled ModelNet40 clouds; the reference takes any size,
    opt_defense.py:114-146): SOR mask, centring / scaling against the oracle's numpy restatement, a 600-subset without
    replacement, init points that are cloud points + noise, and the encoder on the subset."""
    from oracle import convonet_oracle as O
    g = torch.Generator().manual_seed(21)
    v = torch.randn(2, 10000, 3, generator=g)
    pc = v / v.norm(dim=-1, keepdim=True) * (0.6 + 0.4 * torch.rand(2, 10000, 1, generator=g))
    keep = restorer.sor(pc)
    ref_keep, _ = O.sor_keep_mask(pc[:1])
    assert np.array_equal(keep[:1].cpu().numpy().astype(bool), ref_keep.numpy())
    out = restorer.prepare(pc, keep, seed=3, want_proc=True)
    proc, sel, init = out["proc"].cpu().numpy(), out["sel"].cpu().numpy(), out["init"].cpu().numpy()
    for b in range(2):
        kept = pc[b][keep[b].cpu().bool()].numpy()
        n = int(out["n_kept"][b])
        assert n == len(kept) and 8000 < n < 10000
        np.testing.assert_allclose(proc[b, :n], O.preprocess_pc(kept), rtol=0, atol=2e-7)
        rows = {tuple(np.round(r, 6)) for r in proc[b, :n]}
        got = [tuple(np.round(r, 6)) for r in sel[b]]
        assert len(set(got)) == 600 and set(got) <= rows
        d = torch.cdist(torch.from_numpy(init[b]), torch.from_numpy(proc[b, :n])).min(1).values
        assert float(d.max()) < 0.08
    assert out["t_per_cloud"].cpu().tolist() == [600, 600]
    planes = restorer.encode_inputs(out["sel"], out["t_per_cloud"])
    assert bool(torch.isfinite(planes).all())


def test_prepare_with_recorded_draws(restorer, golden):
    out = restorer.prepare(torch.from_numpy(golden["raw"]), torch.from_numpy(golden["sor_keep"].astype(np.uint8)),
                           sel_idx=torch.from_numpy(golden["sel_idx"]), init_idx=torch.from_numpy(golden["init_idx"]),
                           noise=torch.from_numpy(golden["noise"]), want_proc=True)
    assert out["n_kept"].cpu().tolist() == golden["proc_len"].tolist()
    assert out["t_per_cloud"].cpu().tolist() == [600] * 4
    proc = out["proc"].cpu().numpy()
    for b in range(4):
        n = golden["proc_len"][b]
        np.testing.assert_allclose(proc[b, :n], golden["proc_pad"][b, :n], rtol=0, atol=2e-7)
        sel_ref = golden["proc_pad"][b][golden["sel_idx"][b]]
        np.testing.assert_allclose(out["sel"][b].cpu().numpy(), sel_ref, rtol=0, atol=2e-7)
    np.testing.assert_allclose(out["init"].cpu().numpy(), golden["init_points"], rtol=0, atol=3e-7)


def test_prepare_random_draws_are_valid_and_shard_invariant(restorer, golden):
    raw = torch.from_numpy(golden["raw"])
    keep = torch.from_numpy(golden["sor_keep"].astype(np.uint8))
    a = restorer.prepare(raw, keep, seed=7, cloud_index_base=10, want_proc=True)
    b0 = restorer.prepare(raw[:1], keep[:1], seed=7, cloud_index_base=10)
    b1 = restorer.prepare(raw[1:], keep[1:], seed=7, cloud_index_base=11)
    assert torch.equal(a["sel"], torch.cat([b0["sel"], b1["sel"]]))           # draws keyed by the global cloud index
    assert torch.equal(a["init"], torch.cat([b0["init"], b1["init"]]))
    c = restorer.prepare(raw, keep, seed=8, cloud_index_base=10)
    assert not torch.equal(a["sel"], c["sel"])
    proc, sel, init = a["proc"].cpu().numpy(), a["sel"].cpu().numpy(), a["init"].cpu().numpy()
    for bb in range(4):
        n = int(a["n_kept"][bb])
        rows = {tuple(np.round(v, 6)) for v in proc[bb, :n]}
        got = [tuple(np.round(v, 6)) for v in sel[bb]]
        assert len(set(got)) == 600 and set(got) <= rows                      # a subset without replacement
    assert np.abs(init).max() <= 0.45 + 1e-7
    d = init[0][:, None, :] - proc[0][None, :int(a["n_kept"][0]), :]
    nn = np.sqrt((d ** 2).sum(-1)).min(1)                                     # every init point = a cloud point + noise
    assert 0.010 < nn.mean() < 0.03 and nn.max() < 0.08


def test_prepare_library_draws_leave_in_morton_order(restorer, golden):
    """Round 6: with the library's own draws the optimised points are written in Morton order of their coordinates (prep.hip; the
    reference's draws are i.i.d., opt_defense.py:166-176, so the row order is free): non-decreasing 30-bit Z-curve keys, the same
    multiset of points whatever the batch composition, n_opt = 1024 and a ragged 1500 (more optimised points than input points)."""
    raw = torch.from_numpy(golden["raw"])
    keep = torch.from_numpy(golden["sor_keep"].astype(np.uint8))

    def spread(v):
        v = (v | (v << 16)) & 0x030000FF
        v = (v | (v << 8)) & 0x0300F00F
        v = (v | (v << 4)) & 0x030C30C3
        return (v | (v << 2)) & 0x09249249

    for n_opt in (1024, 1500):
        a = restorer.prepare(raw, keep, seed=7, cloud_index_base=10, n_opt=n_opt, want_proc=True)
        init = a["init"].cpu().numpy()
        q = np.clip((init + np.float32(0.5)) * np.float32(1024.0), 0, 1023).astype(np.int64)
        key = spread(q[..., 0]) | (spread(q[..., 1]) << 1) | (spread(q[..., 2]) << 2)
        assert (np.diff(key, axis=1) >= 0).all(), n_opt
        # every point is still "a cloud point + noise", every source index range is covered like before
        d = init[0][:, None, :] - a["proc"].cpu().numpy()[0][None, :int(a["n_kept"][0]), :]
        nn = np.sqrt((d ** 2).sum(-1)).min(1)
        assert 0.010 < nn.mean() < 0.03 and nn.max() < 0.08
        one = restorer.prepare(raw[2:3], keep[2:3], seed=7, cloud_index_base=12, n_opt=n_opt)
        assert torch.equal(one["init"][0], a["init"][2])
    # explicit draws keep their order (the parity fixtures): rows follow init_idx
    idx = torch.from_numpy(golden["init_idx"]) if "init_idx" in golden else None
    if idx is not None:
        b = restorer.prepare(raw, keep, sel_idx=torch.from_numpy(golden["sel_idx"]), init_idx=idx, noise=torch.from_numpy(golden["noise"]))
        np.testing.assert_allclose(b["init"].cpu().numpy(), golden["init_points"], atol=3e-7)


def test_prepare_sparse_cloud_fewer_than_subset(restorer):
    """Drop-attack regime: fewer points than the 600-point encoder subset -> use them all (ragged)."""
    g = torch.Generator().manual_seed(2)
    pc = torch.randn(3, 256, 3, generator=g)
    out = restorer.prepare(pc, None)
    assert out["t_per_cloud"].cpu().tolist() == [256] * 3 and out["n_kept"].cpu().tolist() == [256] * 3
    assert float(out["sel"][:, 256:].abs().max()) == 0.0
    planes = restorer.encode_inputs(out["sel"], out["t_per_cloud"])
    assert torch.isfinite(planes).all()


def test_encoder_pointnet_and_scatter_mean(restorer, golden):
    proc = [golden["proc_pad"][b, :golden["proc_len"][b]] for b in range(4)]
    sel = torch.from_numpy(np.stack([proc[b][golden["sel_idx"][b]] for b in range(4)]))
    pre, c = restorer.encode_points(sel, want_c=True)
    assert _rel(c.cpu().numpy(), golden["enc_c"]) < 1e-5
    pre_xz0 = pre[0, 0].permute(2, 0, 1).cpu().numpy()                         # [32,64,64] like the reference
    assert _rel(pre_xz0, golden["enc_pre_xz0"]) < 1e-5
    assert (pre_xz0 != 0).any(axis=0).sum() == (golden["enc_pre_xz0"] != 0).any(axis=0).sum()   # same occupied cells


def test_encoder_planes_match_reference(restorer, golden):
    import ifdefense_amd as I
    proc = [golden["proc_pad"][b, :golden["proc_len"][b]] for b in range(4)]
    sel = torch.from_numpy(np.stack([proc[b][golden["sel_idx"][b]] for b in range(4)]))
    planes = I.planes_from_channel_last(restorer.encode_inputs(sel))
    for i, pl in enumerate(PL):
        ref = golden["planes01"][:, i]
        assert _rel(planes[pl][:2].cpu().numpy(), ref) < 2e-4, pl
        for b in range(4):
            assert abs(float(planes[pl][b].abs().mean()) - golden["planes_stats"][b, i, 1]) < 1e-4


def test_encoder_ragged_matches_dense(restorer, golden):
    """t_per_cloud < T must equal encoding the shorter cloud alone (no cross-sample ops in the encoder)."""
    proc = golden["proc_pad"][0, :golden["proc_len"][0]]
    sel = torch.zeros(2, 600, 3)
    sel[0] = torch.from_numpy(proc[:600])
    sel[1, :300] = torch.from_numpy(proc[:300])
    a = restorer.encode_points(sel, torch.tensor([600, 300]))
    b = restorer.encode_points(sel[1:, :300])
    assert torch.equal(a[1], b[0])


def test_encoder_point_half_against_oracle_large_and_crowded(restorer, oracle_weights):
    """encode_points against the oracle beyond the golden clouds: a 1000-point subset (the 1024-thread / 160 KB LDS
    configuration) and clouds whose points crowd into a few cells (cell rings of tens to hundreds of points, several
    points per cell in every plane) - the ring construction, the pooled max and the ascending-order mean."""
    from oracle import convonet_oracle as O
    g = torch.Generator().manual_seed(5)
    cases = {
        "T1000": torch.rand(2, 1000, 3, generator=g) - 0.5,
        "crowded600": torch.cat([0.02 * torch.randn(1, 600, 3, generator=g),                     # ~5 x 5 cells
                                 torch.zeros(1, 600, 3) + 0.3], 0),                              # all in ONE cell
        "clusters640": (torch.randint(0, 6, (2, 640, 3), generator=g).float() / 6 - 0.4),        # <= 216 distinct points
    }
    for name, sel in cases.items():
        pre, c = restorer.encode_points(sel, want_c=True)
        c_ref, index = O.pointnet_features(oracle_weights, sel)
        assert _rel(c.cpu().numpy(), c_ref.numpy()) < 2e-5, name
        for i, pl in enumerate(PL):
            ref = O.scatter_mean_plane(c_ref, index[pl]).numpy()                                  # [B,32,64,64]
            got = pre[:, i].permute(0, 3, 1, 2).cpu().numpy()
            assert ((got != 0).any(1) == (ref != 0).any(1)).all(), (name, pl)
            assert _rel(got, ref) < 2e-5, (name, pl)


def test_defend_point_cloud_end_to_end_and_sharding(restorer, golden):
    import ifdefense_amd as I
    args = I.DefenseArgs(iterations=30, batch_size=3, seed=3)
    raw = golden["raw"]
    full = I.defend_point_cloud(restorer, raw, args)
    assert full.shape == (4, 1024, 3) and full.dtype == np.float32 and np.isfinite(full).all()
    np.testing.assert_allclose(np.linalg.norm(full, axis=-1).max(axis=1), 1.0, rtol=1e-6)
    lo = I.defend_point_cloud(restorer, raw[:1], args, cloud_index_base=0, total_clouds=4)
    hi = I.defend_point_cloud(restorer, raw[1:], args, cloud_index_base=1, total_clouds=4)
    assert np.array_equal(np.concatenate([lo, hi]), full)                     # P4: shard + concatenate is bitwise
    small = I.defend_point_cloud(restorer, raw, I.DefenseArgs(iterations=30, batch_size=3, seed=3, chunk=1))
    assert np.array_equal(small, full)                                         # the memory knob does not change results


# ------------------------------------------------------------------------------------------------
# Attribution of out-of-tolerance points (round 4).  north_star's bound is 1e-3 per-point L2; on clustered / sparse inputs a
# handful of points per 4096 end up beyond it after 10 free-running steps.  Instead of budgeting for them, every point whose
# two trajectories separate (by more than 2e-5 in a coordinate - a fiftieth of the bound) must be EXPLAINED, by substitution
# experiments on the two trajectories themselves.  The build's trajectory: one launch per step with the Adam state carried
# (bit-equal to the one-launch run), gradient recovered from the first moment.  The oracle's: the reference's op sequence
# with the gradient recorded before every Adam update.  At the step t where a point separates, the ORACLE's gradient is
# evaluated twice more, at the BUILD's positions x_t: once as it is (G1) and once with the build's 5-NN sets substituted for
# the expanded-form kNN of pn_utils.py:76-82 (G2).  With g_b / g_o the two sides' own gradients and `noise` = 1e-5 of the
# step's largest gradient component (the hot kernel's gradient matches autograd to 5e-7 of it):
#   B  neighbour choice          |g_b - G2| <= noise < |g_b - G1|: at identical positions the two implementations differ, and
#                                they stop differing when the oracle is given the build's neighbour sets - the reference's
#                                expanded-form f32 distance ranked two candidates the other way round (its noise is ~1e-7 for
#                                coordinates of ~0.45; the kernel ranks exact differences).
#   S  the oracle's sensitivity  |g_b - G1| <= noise < |G1 - g_o|: at identical positions the two implementations agree; the
#                                ORACLE's own gradient moves by more than the noise between two sets of positions that agree to
#                                2e-5 (a ReLU boundary of the decoder, a cell edge of grid_sample, a neighbour rank inside the
#                                oracle): rounding-level differences of earlier steps amplified by the reference function itself.
#   C  propagation               as S, but a neighbour of the point (its 5-NN on either side, or a point that counts it among
#                                its 5-NN) separated at an earlier step: the repulsion term couples them.
#   A  Adam at |g| ~ 0           all three gradients agree to noise and the diverging coordinate's gradient history is inside the
#                                noise: torch.optim.Adam's update lr * m_hat / (sqrt(v_hat) + eps) is +-lr for any |g| >> eps in
#                                the first steps (opt_defense.py:207, torch/optim/adam.py).
#   R  a ReLU boundary inside   noise < |g_b - G1| = |g_b - G2|: the implementations differ at identical positions, whatever the
#      the rounding              neighbour sets.  Demonstrated on the oracle alone: nudging the point by a few ulps makes the ORACLE's
#                                own occupancy gradient jump by exactly that difference - a hidden unit of the decoder has a
#                                pre-activation within f32 rounding of zero there, and the two summation orders land on
#                                opposite sides (torch's threshold_backward, src/layers.py:39-47, has no 'almost').
# A separating point with none of these FAILS the test.  The count of points beyond 1e-3 is additionally held against the
# oracle's own sensitivity: the same oracle run from inputs moved by one ulp (five sign patterns).
# ------------------------------------------------------------------------------------------------
# The oracle's side of the protocol does not depend on the precision mode under test (the planes come from the f32 encoder in every
# mode): its traces and one-ulp floors are computed once per input and reused by the second parametrisation of a `both_precisions` test.
_ORACLE_MEMO = {}


def _memo_key(obj):
    import hashlib
    if torch.is_tensor(obj):
        obj = obj.detach().cpu().numpy()
    if isinstance(obj, np.ndarray):
        return ("a", str(obj.dtype), obj.shape, hashlib.sha1(np.ascontiguousarray(obj).tobytes()).hexdigest())
    if isinstance(obj, dict):
        return ("d",) + tuple((k, _memo_key(obj[k])) for k in sorted(obj))
    if isinstance(obj, (list, tuple)):
        return ("l",) + tuple(_memo_key(v) for v in obj)
    if isinstance(obj, (int, float, bool, str)) or obj is None:
        return ("s", repr(obj))
    return ("o", getattr(obj, "__name__", type(obj).__name__))             # (the oracle module)


def _oracle_trace(MO, w, init, cond, steps, loss_batch):
    """The oracle's optimize_points loop (opt_defense.py:205-228) with x_t and g_t kept: x [steps + 1, B, K, 3], g [steps, B, K, 3]."""
    key = ("trace", _memo_key(MO), _memo_key(w), _memo_key(init), _memo_key(cond), steps, loss_batch)
    if key not in _ORACLE_MEMO:
        _ORACLE_MEMO[key] = _oracle_trace_uncached(MO, w, init, cond, steps, loss_batch)
    xs, gs = _ORACLE_MEMO[key]
    return xs.copy(), gs.copy()


def _oracle_trace_uncached(MO, w, init, cond, steps, loss_batch):
    x = init.clone().float().requires_grad_(True)
    opt = torch.optim.Adam([x], lr=1e-3)
    xs, gs = [x.detach().clone()], []
    for _ in range(steps):
        total = MO.losses(w, x, cond, 500.0, 0.2, loss_batch)[0]
        opt.zero_grad()
        total.backward()
        gs.append(x.grad.detach().clone())
        opt.step()
        xs.append(x.detach().clone())
    return torch.stack(xs).numpy(), torch.stack(gs).numpy()


def _oracle_grad(MO, w, x, cond, loss_batch, idx=None):
    """The oracle's gradient of the step objective (opt_defense.py:212-225) at positions x [B,K,3]; idx [B,K,5]: neighbour sets
    to use instead of the reference's expanded-form kNN."""
    from oracle import convonet_oracle as O
    import torch.nn.functional as F
    p = torch.from_numpy(np.ascontiguousarray(x)).clone().requires_grad_(True)
    logits = MO.decode_logits(w, p, cond)
    bce = F.binary_cross_entropy_with_logits(logits, torch.full_like(logits, 0.2), reduction="none")
    rep = O.repulsion_loss(p, None if idx is None else torch.from_numpy(idx).long())
    (bce.sum() / float(loss_batch) + rep.sum() / float(loss_batch) * 500.0).backward()
    return p.grad.numpy()


def _oracle_kink_jumps(MO, w, x_point, cond_b, loss_batch, nudges=(0, 1, -1, 2, -2, 4, -4, 8, -8, 16, -16, 32, -32, 64, -64, 128, -128, 256, -256, 512, -512)):
    """Occupancy-gradient jumps of the oracle at one point under nudges of 1 ... 512 ulps per coordinate (512 ulps of a
    coordinate of 0.4 are 1.5e-5: below the separation threshold, and what a hidden unit's pre-activation - a sum of up to
    256 terms of O(1) in another order - can differ by between two implementations): (gradient at the nudged position -
    gradient at the position itself), [n, 3]."""
    import itertools
    import torch.nn.functional as F
    offs = list(itertools.product(nudges, repeat=3))
    bits = np.ascontiguousarray(x_point, dtype=np.float32).view(np.int32)
    sgn = np.where(bits < 0, -1, 1).astype(np.int32)                       # (moving a negative float up = fewer magnitude bits)
    pts = np.stack([(bits + sgn * np.array(o, dtype=np.int32)).view(np.float32) for o in offs])
    p = torch.from_numpy(pts)[None].clone().requires_grad_(True)         # [1, n, 3]
    logits = MO.decode_logits(w, p, cond_b)
    (F.binary_cross_entropy_with_logits(logits, torch.full_like(logits, 0.2), reduction="none").sum() / float(loss_batch)).backward()
    g = p.grad[0].numpy()
    return g - g[offs.index((0, 0, 0))]


def _hip_trace(r, init, cond, steps, loss_batch):
    """The build's trajectory, one launch per Adam step with the state carried (x_t, and g_t recovered from the first moment:
    m_t = 0.9 m_{t-1} + 0.1 g_t)."""
    x, st = init, None
    xs, gs = [init.cpu().numpy()], []
    m_prev = np.zeros_like(xs[0])
    for _ in range(steps):
        x, st = r.optimize_points(x, cond, rep_weight=500.0, steps=1, loss_batch=loss_batch, normalize=False, state=st,
                                  return_state=True)
        m = st[0].cpu().numpy()
        gs.append((m - 0.9 * m_prev) / 0.1)
        m_prev = m
        xs.append(x.cpu().numpy())
    return np.stack(xs), np.stack(gs), x


def _attribute_separations(r, MO, w, cond, loss_batch, hx, hg, ox, og, tag, onset_tol=2e-5, g_noise=1e-5):
    """hx / ox [T + 1, B, K, 3], hg / og [T, B, K, 3]: the build's / the oracle's positions and gradients.  Classifies every
    point whose trajectories separate (block comment above); fails on a point without an explanation."""
    from oracle import convonet_oracle as O
    T, B = hg.shape[0], hg.shape[1]
    sep = np.abs(hx - ox).max(-1)                              # [T + 1, B, K]
    onset = np.full(sep.shape[1:], T + 1, dtype=int)            # first t with x_{t+1} apart
    for t in range(T, 0, -1):
        onset[sep[t] > onset_tol] = t - 1
    todo = sorted(zip(*np.nonzero(onset <= T - 1)), key=lambda bi: onset[bi])
    cache = {}

    def at_step(t):
        if t not in cache:
            hip_idx = r.repulsion_loss(torch.from_numpy(hx[t]), want_idx=True)[1].cpu().numpy()     # exact 5-NN: what the kernel uses
            ref_idx = O.knn_point(5, torch.from_numpy(hx[t])).numpy()
            g1 = _oracle_grad(MO, w, hx[t], cond, loss_batch)
            g2 = _oracle_grad(MO, w, hx[t], cond, loss_batch, hip_idx)
            cache[t] = (g1, g2, hip_idx, ref_idx)
        return cache[t]

    np.testing.assert_allclose(_oracle_grad(MO, w, ox[0], cond, loss_batch), og[0], rtol=0, atol=1e-6 * np.abs(og[0]).max())
    reasons, unexplained = {}, []
    def gradient_cause(b, i, t):
        """B / R at step t for point (b, i): the implementations' gradients differ at IDENTICAL positions (None if they do not)."""
        g1, g2, hip_idx, ref_idx = at_step(t)
        noise = g_noise * np.abs(og[t, b]).max()
        d_impl = np.abs(hg[t, b, i] - g1[b, i]).max()
        d_idx = np.abs(hg[t, b, i] - g2[b, i]).max()
        if d_impl <= noise:
            return None
        if d_idx <= noise:
            return "B at step %d (gradients %.1e of max apart, %.1e with the build's 5-NN sets in the oracle)" % (t, d_impl / noise * g_noise, d_idx / noise * g_noise)
        cond_b = {k: v[b:b + 1] for k, v in cond.items()} if isinstance(cond, dict) else cond[b:b + 1]
        jumps = _oracle_kink_jumps(MO, w, hx[t, b, i], cond_b, loss_batch)
        miss = np.abs(jumps - (hg[t, b, i] - g1[b, i])[None]).max(-1).min()
        return "R at step %d (gradients %.1e of max apart; an oracle jump under a nudge of <= 512 ulps within %.1e)" % (
            t, d_impl / noise * g_noise, miss / noise * g_noise) if miss <= 3.0 * noise else None

    for b, i in todo:
        t = int(onset[b, i])
        g1, g2, hip_idx, ref_idx = at_step(t)
        noise = g_noise * np.abs(og[t, b]).max()
        d_impl = np.abs(hg[t, b, i] - g1[b, i]).max()
        d_idx = np.abs(hg[t, b, i] - g2[b, i]).max()
        d_sens = np.abs(g1[b, i] - og[t, b, i]).max()
        a = int(np.abs(hx[t + 1, b, i] - ox[t + 1, b, i]).argmax())
        why = None
        if d_idx <= noise < d_impl:
            who = [j for j in range(hip_idx.shape[1]) if (j == i or i in hip_idx[b, j] or i in ref_idx[b, j])
                   and set(hip_idx[b, j].tolist()) != set(ref_idx[b, j].tolist())]
            why = "B: same positions, gradients %.1e apart (of max), %.1e with the build's 5-NN sets in the oracle; sets differ for point(s) %s" % (
                d_impl / noise * g_noise, d_idx / noise * g_noise, who[:4])
        elif d_impl <= noise and d_sens > noise:
            nb = set(hip_idx[b, i].tolist()) | set(ref_idx[b, i].tolist()) | {j for j in range(hip_idx.shape[1]) if i in hip_idx[b, j] or i in ref_idx[b, j]}
            early = [j for j in nb if onset[b, j] < t and (b, j) in reasons]
            if early:
                why = "C: neighbour %d separated at step %d; implementations agree at the build's positions to %.1e" % (
                    early[0], onset[b, early[0]], d_impl / noise * g_noise)
            else:
                why = "S: implementations agree at the build's positions to %.1e; the oracle's gradient moves by %.1e between positions %.1e apart" % (
                    d_impl / noise * g_noise, d_sens / noise * g_noise, sep[t, b].max())
        elif d_impl <= noise:
            hist = max(np.abs(og[:t + 1, b, i, a]).max(), np.abs(hg[:t + 1, b, i, a]).max())
            # Adam's first steps move a coordinate by ~lr * sign-like(m / sqrt(v)): a RELATIVE perturbation delta of the coordinate's
            # gradient moves x by ~lr * delta per step.  Either |g| is inside the noise band outright, or the agreed noise, relative
            # to the coordinate's own |g|, accounts for at least half of the separation seen at the onset.
            amplified = 1e-3 * max(d_impl, d_sens) / max(hist, 1e-30) * (t + 1)
            if hist <= 10.0 * noise or amplified >= 0.5 * sep[t + 1, b, i]:
                why = "A: all gradients agree to %.1e; |g| <= %.1e of max over steps 0..%d (lr x relative noise x steps = %.1e against a separation of %.1e)" % (
                    max(d_impl, d_sens) / noise * g_noise, hist / noise * g_noise, t, amplified, sep[t + 1, b, i])
        else:
            cond_b = {k: v[b:b + 1] for k, v in cond.items()} if isinstance(cond, dict) else cond[b:b + 1]
            jumps = _oracle_kink_jumps(MO, w, hx[t, b, i], cond_b, loss_batch)
            miss = np.abs(jumps - (hg[t, b, i] - g1[b, i])[None]).max(-1).min()
            print("  %s: cloud %d point %d step %d: implementations %.1e apart at identical positions; nearest oracle jump under a nudge misses it by %.1e (of max)" % (
                tag, b, i, t, d_impl / noise * g_noise, miss / noise * g_noise))
            if miss <= 3.0 * noise:
                why = "R: same positions and neighbour sets, gradients %.1e apart (of max); the oracle's own occupancy gradient makes that jump (to %.1e) under a nudge of <= 512 ulps" % (
                    d_impl / noise * g_noise, miss / noise * g_noise)
        if why is None and d_impl <= noise:
            # The gradients agree at the onset step - but Adam's first steps move every coordinate by ~lr sign(g) whatever |g| is, so a
            # difference between the implementations' gradients at an EARLIER step (positions still identical) only surfaces later,
            # through the moments.  Look back for it.
            for s_ in range(t - 1, -1, -1):
                if np.abs(hx[s_, b, i] - ox[s_, b, i]).max() <= 1e-7:
                    cause = gradient_cause(b, i, s_)
                    if cause is not None:
                        why = cause[0] + ": latent - " + cause + "; it reaches the positions through Adam's moments at step %d" % t
                        break
        if why is None:
            print("  %s: UNEXPLAINED cloud %d point %d coordinate %d, onset step %d: x build %s oracle %s; g build %s oracle %s (max |g| of the cloud %.3e)" % (
                tag, b, i, a, t, hx[:t + 2, b, i, a].tolist(), ox[:t + 2, b, i, a].tolist(), hg[:t + 1, b, i, a].tolist(), og[:t + 1, b, i, a].tolist(),
                np.abs(og[t, b]).max()))
            unexplained.append((int(b), int(i), t, a, "d_impl %.1e d_idx %.1e d_sens %.1e (of max)" % (
                d_impl / noise * g_noise, d_idx / noise * g_noise, d_sens / noise * g_noise)))
        else:
            reasons[(b, i)] = (t, why)
    final_off = np.linalg.norm(hx[-1] - ox[-1], axis=-1) > 1e-3
    kinds_all, kinds_off = {}, {}
    for (b, i), (t, why) in reasons.items():
        kinds_all[why[0]] = kinds_all.get(why[0], 0) + 1
        if final_off[b, i]:
            kinds_off[why[0]] = kinds_off.get(why[0], 0) + 1
            print("  %s: cloud %d point %4d separates at step %d - %s" % (tag, b, i, t, why))
    print("%s: %d of %d points separate by > %.0e in %d steps %s; %d of them end beyond 1e-3 %s; unexplained: %d" %
          (tag, len(todo), sep[0].size, onset_tol, T, kinds_all, int(final_off.sum()), kinds_off, len(unexplained)))
    for u in unexplained:
        print("  UNEXPLAINED", tag, u)
    assert not unexplained, (tag, len(unexplained))
    return reasons


def _ulp_floor(MO, w, init, cond_fn, iterations, loss_batch, ref_out, seeds=(0, 1, 2, 3, 4), cond=None):
    """How many points of the ORACLE end beyond 1e-3 of its own run when every input coordinate moves by one ulp (one sign
    pattern per seed): the sensitivity floor any other f32 implementation is measured against.  cond: what cond_fn closes over
    (the conditioning planes / code) - given, the floor is memoised on (weights, init, cond, iterations, loss_batch, seeds, ref_out)."""
    key = None if cond is None else ("ulp", _memo_key(MO), _memo_key(w), _memo_key(init), _memo_key(cond), iterations, loss_batch,
                                     tuple(seeds), _memo_key(ref_out))
    if key is not None and key in _ORACLE_MEMO:
        return list(_ORACLE_MEMO[key])
    counts = []
    for seed in seeds:
        g = torch.Generator().manual_seed(100 + seed)
        up = torch.rand(init.shape, generator=g) < 0.5
        pert = torch.where(up, torch.nextafter(init, torch.full_like(init, 2.0)), torch.nextafter(init, torch.full_like(init, -2.0)))
        out = cond_fn(pert)
        counts.append(int((np.linalg.norm(out - ref_out, axis=-1) > 1e-3).sum()))
    if key is not None:
        _ORACLE_MEMO[key] = list(counts)
    return counts



def _oracle_restore_from_hip_draws(restorer, oracle_weights, clouds, iterations, sor=True):
    """HIP SOR + prepare (its own counter-based draws), then BOTH sides restore from those draws:
    returns (hip_out, oracle_out, prep, keep)."""
    from oracle import convonet_oracle as O
    x = torch.from_numpy(clouds).cuda()
    keep = restorer.sor(x) if sor else None
    prep = restorer.prepare(x, keep, seed=11)
    t = prep["t_per_cloud"].cpu().tolist()
    planes_hip = restorer.encode_inputs(prep["sel"], prep["t_per_cloud"])
    hip = restorer.optimize_points(prep["init"], planes_hip, rep_weight=500.0, iterations=iterations)
    outs, planes_ref = [], []
    for b in range(len(clouds)):                     # ragged subsets: the oracle encodes cloud by cloud
        sel_b = prep["sel"][b:b + 1, :t[b]].cpu()
        key = ("restore", _memo_key(oracle_weights), _memo_key(sel_b), _memo_key(prep["init"][b:b + 1]), iterations, len(clouds))
        if key not in _ORACLE_MEMO:
            planes_b = O.encode_inputs(oracle_weights, sel_b)
            _ORACLE_MEMO[key] = (planes_b, O.optimize_points(oracle_weights, prep["init"][b:b + 1].cpu(), planes_b, rep_weight=500.0,
                                                             iterations=iterations, loss_batch=len(clouds)))
        planes_b, out_b = _ORACLE_MEMO[key]
        planes_ref.append(planes_b)
        outs.append(out_b)
    prep["_planes_hip"], prep["_planes_ref"] = planes_hip, planes_ref
    return hip.cpu().numpy(), torch.cat(outs).numpy(), prep, keep


def _attribute_config(restorer, oracle_weights, prep, hip, iterations, tag):
    """Every point of a config #3 / #5 run that leaves the oracle's trajectory is attributed (see the block comment above):
    the optimiser on the build's planes against the oracle fed with the same planes (what the end-to-end figures above add is
    the two encoders' 1e-5 difference in the planes - the same mechanisms with a wider noise band), and the count of points
    beyond 1e-3 held against the oracle's own one-ulp sensitivity.  Returns (points beyond 1e-3, the ulp floor)."""
    from oracle import convonet_oracle as O
    import ifdefense_amd as I
    B, T = hip.shape[0], iterations + 1
    init = prep["init"]
    hx, hg, last = _hip_trace(restorer, init, prep["_planes_hip"], T, B)
    assert np.array_equal(restorer.normalize_batch_pc(last).cpu().numpy(), hip)          # launch per step == one launch, bitwise
    pd = I.planes_from_channel_last(prep["_planes_hip"].cpu())
    ox, og = _oracle_trace(O, oracle_weights, init.cpu(), pd, T, B)
    _attribute_separations(restorer, O, oracle_weights, pd, B, hx, hg, ox, og, tag + " (optimiser alone)")
    n_opt = int((np.linalg.norm(hx[-1] - ox[-1], axis=-1) > 1e-3).sum())
    floor = _ulp_floor(O, oracle_weights, init.cpu(),
                       lambda q: O.optimize_points(oracle_weights, q, pd, rep_weight=500.0, iterations=iterations, loss_batch=B,
                                                   normalize=False).numpy(), iterations, B, ox[-1], cond=pd)
    print("%s: points beyond 1e-3 after %d steps, optimiser alone: build vs oracle %d, oracle vs its own 1-ulp-perturbed runs %s" %
          (tag, T, n_opt, floor))
    assert n_opt <= max(floor) + 5, (tag, n_opt, floor)
    return n_opt, floor


@pytest.mark.both_precisions
def test_config3_knn_attack_like_clouds(restorer, golden, oracle_weights, precision_mode):
    """BASELINE config #3: perturbed clouds with tight clusters (the stress case of the neighbour lists)."""
    from oracle import convonet_oracle as O
    import bench
    clouds = bench.knn_attack_like(golden["raw"])
    x = torch.from_numpy(clouds)
    keep = restorer.sor(x.cuda()).cpu().numpy().astype(bool)
    ref_keep, _ = O.sor_keep_mask(x)
    assert np.array_equal(keep, ref_keep.numpy().astype(bool))                 # fp64 statistics: bit-exact mask
    hip, ref, prep, _ = _oracle_restore_from_hip_draws(restorer, oracle_weights, clouds, iterations=9)
    d = np.linalg.norm(hip - ref, axis=-1)
    # north_star: 1e-3 per-point L2.  Inside the sigma = 0.01 clusters 5th/6th-neighbour distances tie to ~1e-4
    # relative, which is the rounding noise of the reference's own |a|^2 + |b|^2 - 2ab distance: a flipped
    # neighbour moves that one point - allow 1 point in 1000, bound the rest
    print("config #3, 10 steps: max %.2e, median %.2e, points > 1e-3: %d of %d" % (d.max(), np.median(d), (d > 1e-3).sum(), d.size))
    # measured (round 3): 1 of 4096 points beyond 1e-3 (1.26e-3), median 3.2e-7
    # (round 6: the initial points leave ifd_prepare in Morton order - another sample of the same chaos: 3 of 4096 beyond 1e-3, max 6.0e-3,
    # median 1.1e-6 (the cloud's farthest point is one of the separating ones, and normalisation divides everything by its norm); what
    # holds the line is the attribution below: every separating point explained, the count inside the oracle's own 1-ulp floor + 5)
    assert (d > 1e-3).sum() <= 6 and np.median(d) < 2.5e-6 and d.max() < 1.2e-2, (d.max(), np.median(d), int((d > 1e-3).sum()))
    _attribute_config(restorer, oracle_weights, prep, hip, 9, "config #3")
    # 150 steps on the clustered clouds: lists + individual refreshes == exact scan, bit for bit
    planes = restorer.encode_inputs(prep["sel"], prep["t_per_cloud"])
    a = restorer.optimize_points(prep["init"], planes, rep_weight=500.0, iterations=150, normalize=False)
    c = restorer.counters()
    b = restorer.optimize_points(prep["init"], planes, rep_weight=500.0, iterations=150, normalize=False,
                                 knn_scan_every_step=True)
    assert torch.equal(a, b)
    assert c["knn_refresh_waves"] > 0, c                                       # clusters are refreshed individually...
    assert c["knn_rebuilds"] / (4 * 8) < 60, c                                 # ...instead of forcing whole-cloud rebuilds


@pytest.mark.both_precisions
def test_config5_sparse_inputs(restorer, golden, oracle_weights, precision_mode):
    """BASELINE config #5: Drop-200 (K = 824) and K = 256 inputs -> 1024 restored points each."""
    import bench
    for clouds in (bench.drop_like(golden["raw"]), bench.subsample_like(golden["raw"], 256)):
        for sor in (True, False):
            hip, ref, prep, keep = _oracle_restore_from_hip_draws(restorer, oracle_weights, clouds, iterations=9, sor=sor)
            assert hip.shape == (4, 1024, 3) and np.isfinite(hip).all()
            n_kept = prep["n_kept"].cpu().numpy()
            assert (n_kept <= clouds.shape[1]).all() and (sor or (n_kept == clouds.shape[1]).all())
            assert (prep["t_per_cloud"].cpu().numpy() == np.minimum(n_kept, 600)).all()
            d = np.linalg.norm(hip - ref, axis=-1)
            # 1024 points drawn with replacement from <= 256 sources + N(0, 0.01^2): micro-clusters, so a few
            # first-step Adam sign flips (|g| ~ 0, protocol P1) and neighbour near-ties are expected; each moves
            # one point by ~lr per step.  Count them, bound everything else.
            print("config #5 K=%d sor=%s: max %.2e median %.2e, points > 1e-3: %d of %d" %
                  (clouds.shape[1], sor, d.max(), np.median(d), (d > 1e-3).sum(), d.size))
            # measured (round 3): K = 824: 0 points beyond 1e-3 (max 2.9e-4 / 1.1e-4); K = 256: 13 (max 1.6e-2) with SOR,
            # 2 (max 1.9e-3) without - asserted at 2x the measured values
            max_off, max_d = {(824, True): (0, 6e-4), (824, False): (0, 3e-4), (256, True): (26, 3.2e-2), (256, False): (4, 4e-3)}[
                (clouds.shape[1], sor)]
            assert (d > 1e-3).sum() <= max_off and np.median(d) < 1e-6 and d.max() < max_d, (clouds.shape, sor, d.max(), int((d > 1e-3).sum()))
            _attribute_config(restorer, oracle_weights, prep, hip, 9, "config #5 K=%d sor=%s" % (clouds.shape[1], sor))


@pytest.mark.both_precisions
def test_attribution_on_sixteen_bench_clouds_twenty_steps(restorer, oracle_weights, precision_mode):
    """The attribution protocol (block comment above _oracle_trace) on a wider base than the 4-cloud fixtures: sixteen bench
    clouds - every shape family at least twice - through the build's own SOR / preprocess / encoder, 20 free-running Adam steps of the
    optimiser against the oracle on the same planes.  Every separating point explained, the count beyond 1e-3 within the oracle's
    own 1-ulp sensitivity."""
    from oracle import convonet_oracle as O
    import bench
    import ifdefense_amd as I
    clouds = bench.synth_clouds(16, seed=99)
    x = torch.from_numpy(clouds).cuda()
    prep = restorer.prepare(x, restorer.sor(x), seed=17)
    planes = restorer.encode_inputs(prep["sel"], prep["t_per_cloud"])
    T, B = 20, 16
    hx, hg, last = _hip_trace(restorer, prep["init"], planes, T, B)
    one = restorer.optimize_points(prep["init"], planes, rep_weight=500.0, steps=T, loss_batch=B, normalize=False)
    assert torch.equal(one, last)                                              # a launch per step == one launch, bitwise
    pd = I.planes_from_channel_last(planes.cpu())
    ox, og = _oracle_trace(O, oracle_weights, prep["init"].cpu(), pd, T, B)
    _attribute_separations(restorer, O, oracle_weights, pd, B, hx, hg, ox, og, "16 bench clouds, 20 steps")
    d = np.linalg.norm(hx[-1] - ox[-1], axis=-1)
    n_off = int((d > 1e-3).sum())
    floor = _ulp_floor(O, oracle_weights, prep["init"].cpu(),
                       lambda q: O.optimize_points(oracle_weights, q, pd, rep_weight=500.0, iterations=T - 1, loss_batch=B,
                                                   normalize=False).numpy(), T - 1, B, ox[-1], seeds=(0, 1, 2), cond=pd)
    print("16 bench clouds, 20 steps: median %.2e, points beyond 1e-3: build vs oracle %d of %d, oracle vs its 1-ulp-perturbed runs %s" %
          (np.median(d), n_off, d.size, floor))
    assert np.median(d) < 2e-6 and n_off <= max(floor) + 8, (n_off, floor)


def test_unet_matches_reference_and_is_batch_invariant(restorer, golden):
    pre = torch.from_numpy(golden["enc_pre_xz0"]).permute(1, 2, 0)[None, None].repeat(1, 3, 1, 1, 1).contiguous()
    out = restorer.unet(pre)                                                    # [1,3,64,64,32]
    got = out[0, 0].permute(2, 0, 1).cpu().numpy()
    assert _rel(got, golden["planes01"][0, 0]) < 2e-5
    big = restorer.unet(torch.cat([torch.randn(2, 3, 64, 64, 32), pre, torch.randn(3, 3, 64, 64, 32)]))
    assert torch.equal(big[2], out[0])                                          # fixed summation order: batch-size independent


def test_cli_end_to_end(tmp_path, np_weights):
    """python -m ifdefense_amd.opt_defense on a 3-key .npz: same output location / keys / dtypes as the reference."""
    import subprocess, sys, os
    wpath = tmp_path / "convonet.pth"
    torch.save({k: torch.from_numpy(v) for k, v in np_weights.items()}, wpath)
    g = dict(np.load(os.path.join(os.path.dirname(__file__), "golden", "convonet_golden.npz")))
    src = tmp_path / "kNN-pointnet.npz"
    pc6 = np.concatenate([g["raw"], np.zeros_like(g["raw"])], axis=-1)            # [N,K,6]: normals are sliced off
    np.savez(src, test_pc=pc6, test_label=np.array([0, 8, 30, 39]), target_label=np.array([1, 2, 3, 4]))
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    r = subprocess.run([sys.executable, "-m", "ifdefense_amd.opt_defense", "--data_root", str(src), "--iterations=20",
                        "--weights", str(wpath), "--seed=5"], capture_output=True, text=True, cwd=root, timeout=600)
    assert r.returncode == 0, r.stdout + r.stderr
    out = tmp_path / "ConvONet-Opt" / "convonet_opt-kNN-pointnet.npz"
    z = np.load(out)
    assert sorted(z.files) == ["target_label", "test_label", "test_pc"]
    assert z["test_pc"].shape == (4, 1024, 3) and z["test_pc"].dtype == np.float32
    assert z["test_label"].dtype == np.uint8 and z["test_label"].tolist() == [0, 8, 30, 39]
    np.testing.assert_allclose(np.linalg.norm(z["test_pc"], axis=-1).max(axis=1), 1.0, rtol=1e-6)


def test_cli_large_inputs_and_sample_npoint_2048(tmp_path, np_weights):
    """The reference takes any size (opt_defense.py:27, SOR.py:22-49); this build up to 10,000 points per input cloud and
    10,000 optimised points: 5000-point inputs through SOR (its float-in-LDS layout) / preprocess / encoder, 2048 points
    optimised on the two-launch-per-step path, the reference's output contract."""
    import subprocess, sys, os
    wpath = tmp_path / "convonet.pth"
    torch.save({k: torch.from_numpy(v) for k, v in np_weights.items()}, wpath)
    rng = np.random.default_rng(0)
    v = rng.normal(size=(3, 5000, 3)).astype(np.float32)
    pc = (v / np.linalg.norm(v, axis=-1, keepdims=True) * rng.uniform(0.5, 1.0, size=(3, 5000, 1))).astype(np.float32)
    src = tmp_path / "big.npz"
    np.savez(src, test_pc=pc, test_label=np.arange(3))
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    r = subprocess.run([sys.executable, "-m", "ifdefense_amd.opt_defense", "--data_root", str(src), "--iterations=30",
                        "--sample_npoint=2048", "--weights", str(wpath), "--seed=5"], capture_output=True, text=True,
                       cwd=root, timeout=600)
    assert r.returncode == 0, r.stdout + r.stderr
    z = np.load(tmp_path / "ConvONet-Opt" / "convonet_opt-big.npz")
    assert z["test_pc"].shape == (3, 2048, 3) and z["test_pc"].dtype == np.float32 and np.isfinite(z["test_pc"]).all()
    np.testing.assert_allclose(np.linalg.norm(z["test_pc"], axis=-1).max(axis=1), 1.0, rtol=1e-6)
    # the restored points sit on the same (seeded random-weight) surface as a 1024-point run of the same clouds
    r = subprocess.run([sys.executable, "-m", "ifdefense_amd.opt_defense", "--data_root", str(src), "--iterations=30",
                        "--weights", str(wpath), "--seed=5"], capture_output=True, text=True, cwd=root, timeout=600)
    assert r.returncode == 0, r.stdout + r.stderr
    z1 = np.load(tmp_path / "ConvONet-Opt" / "convonet_opt-big.npz")["test_pc"]
    for b in range(3):
        d = torch.cdist(torch.from_numpy(z["test_pc"][b]), torch.from_numpy(z1[b])).min(1).values
        assert float(d.mean()) < 0.08, (b, float(d.mean()))


def _chamfer(a, b):
    d = torch.cdist(torch.from_numpy(a), torch.from_numpy(b))
    return float(d.min(1).values.mean() + d.min(0).values.mean())


@pytest.mark.both_precisions
def test_p3_full_501_steps_against_oracle(restorer, golden, oracle_weights, planes2, precision_mode):
    """Protocol P3: the 501-step trajectory is chaotic, so compare what is comparable - the share of
    points within 1e-3 next to the oracle's own 1-ulp self-divergence, the final losses, and the symmetric Chamfer
    distance build<->oracle against oracle<->perturbed oracle."""
    from oracle import convonet_oracle as O
    torch.set_num_threads(min(16, torch.get_num_threads()))
    init = torch.from_numpy(golden["init_points"][:2])
    ref, snaps = O.optimize_points(oracle_weights, init, planes2, rep_weight=500.0, iterations=500, normalize=False,
                                   record=(500,))
    bumped = torch.nextafter(init, torch.full_like(init, 2.0))                     # +1 ulp on every coordinate
    ref2, snaps2 = O.optimize_points(oracle_weights, bumped, planes2, rep_weight=500.0, iterations=500, normalize=False,
                                     record=(500,))
    got, loss = restorer.optimize_points(init, planes2, rep_weight=500.0, iterations=500, normalize=False, return_loss=True)
    got = got.cpu()
    d_build = (got - ref).norm(dim=-1).numpy()
    d_self = (ref2 - ref).norm(dim=-1).numpy()
    cd_build = np.mean([_chamfer(got[b].numpy(), ref[b].numpy()) for b in range(2)])
    cd_self = np.mean([_chamfer(ref2[b].numpy(), ref[b].numpy()) for b in range(2)])
    print("P3 501 steps: within 1e-3: build %.3f (mean %.2e) | oracle self-divergence %.3f (mean %.2e); "
          "Chamfer build<->oracle %.3e, oracle<->perturbed %.3e" %
          ((d_build < 1e-3).mean(), d_build.mean(), (d_self < 1e-3).mean(), d_self.mean(), cd_build, cd_self))
    assert (d_build < 1e-3).mean() > 0.8 * (d_self < 1e-3).mean() - 0.05
    assert cd_build < 1.5 * cd_self + 1e-4
    # final losses (evaluated at the pre-update points of the last step).  The occupancy term is a smooth statistic
    # (1 %); the repulsion term is carried by a few close pairs and scatters by itself: over six +-1-ulp perturbations
    # of the initial points the oracle gives 3.878 ... 3.925 (std 0.4 %, range 1.2 %) on these two clouds, and
    # different host CPUs move the unperturbed value just as much -> 2.5 % (6 sigma), or twice the perturbed
    # oracle's own difference if that is larger
    with torch.no_grad():
        _, occ, rep, _ = O.losses(oracle_weights, snaps[500], planes2, 500.0)
        _, occ2, rep2, _ = O.losses(oracle_weights, snaps2[500], planes2, 500.0)
    loss = loss.cpu().numpy().astype(np.float64)
    tol_occ = max(1e-2, 2.0 * abs(float(occ2) - float(occ)) / float(occ))
    tol_rep = max(2.5e-2, 2.0 * abs(float(rep2) - float(rep)) / float(rep))
    print("final losses: occ %.5f (oracle %.5f, perturbed %.5f) rep %.5f (oracle %.5f, perturbed %.5f)" %
          (loss[:, 0].sum() / 2, float(occ), float(occ2), loss[:, 1].mean() * 500.0, float(rep), float(rep2)))
    np.testing.assert_allclose(loss[:, 0].sum() / 2, float(occ), rtol=tol_occ)
    np.testing.assert_allclose(loss[:, 1].mean() * 500.0, float(rep), rtol=tol_rep)


@pytest.mark.both_precisions
def test_full_size_properties(restorer, precision_mode):
    """BASELINE size (2468 clouds x 1024 points x 501 steps): size-independent properties."""
    import sys, os
    sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
    import bench
    import ifdefense_amd as I
    clouds = bench.synth_clouds(2468)
    args = I.DefenseArgs(iterations=500, seed=1234, precision=precision_mode)
    full = I.defend_point_cloud(restorer, clouds, args)
    assert full.shape == (2468, 1024, 3) and np.isfinite(full).all()
    np.testing.assert_allclose(np.linalg.norm(full, axis=-1).max(axis=1), 1.0, rtol=1e-5)       # unit sphere
    assert np.abs(full.mean(axis=1)).max() < 1e-4                                               # centred
    c = restorer.counters()
    print("2468 clouds x 501 steps: %.1f synchronous list rebuilds per cloud, %d certificate failures, %.2f passes/rebuild"
          % (c["knn_rebuilds"] / 8 / 2468, c["knn_brute_scans"], c["knn_passes"] / max(1, c["knn_rebuilds"])))
    # sharding invariance at full size: two halves, concatenated, bit-identical
    lo = I.defend_point_cloud(restorer, clouds[:1234], args, cloud_index_base=0, total_clouds=2468)
    hi = I.defend_point_cloud(restorer, clouds[1234:], args, cloud_index_base=1234, total_clouds=2468)
    assert np.array_equal(np.concatenate([lo, hi]), full)
    # Parity INSIDE the full-size launch (round-4 verdict, item 9): clouds taken out of a late whole round of the 2468-cloud launch
    # (clouds 2048 ... 2303 are its ninth round of 256) and out of the 164-wide tail round (2304 ... 2467, whose reference batch has
    # 164 members: another 1 / B) come out bit-identical when they are restored as a batch of four on their own ...
    for base in (2050, 2460):
        small = I.defend_point_cloud(restorer, clouds[base:base + 4], args, cloud_index_base=base, total_clouds=2468)
        assert np.array_equal(small, full[base:base + 4]), base
    # ... and two of each are held against the oracle over 20 free-running steps with the attribution protocol (every separating
    # point explained; the count beyond 1e-3 inside the oracle's own 1-ulp floor + 5, like the configuration tests)
    from oracle import convonet_oracle as O
    ow = O.to_torch(O.make_random_weights(0))
    x = torch.from_numpy(clouds).cuda()
    for base, B in ((2050, 192), (2460, 164)):
        xs = x[base:base + 2]
        prep = restorer.prepare(xs, restorer.sor(xs), seed=1234, cloud_index_base=base)
        planes = restorer.encode_inputs(prep["sel"], prep["t_per_cloud"])
        T = 20
        hx, hg, last = _hip_trace(restorer, prep["init"], planes, T, B)
        pd = I.planes_from_channel_last(planes.cpu())
        ox, og = _oracle_trace(O, ow, prep["init"].cpu(), pd, T, B)
        _attribute_separations(restorer, O, ow, pd, B, hx, hg, ox, og, "clouds %d, %d of the full-size launch, 20 steps" % (base, base + 1))
        n_off = int((np.linalg.norm(hx[-1] - ox[-1], axis=-1) > 1e-3).sum())
        floor = _ulp_floor(O, ow, prep["init"].cpu(),
                           lambda q: O.optimize_points(ow, q, pd, rep_weight=500.0, iterations=T - 1, loss_batch=B, normalize=False).numpy(),
                           T - 1, B, ox[-1], cond=pd)
        print("clouds %d, %d of the full-size launch: points beyond 1e-3 after 20 steps: build vs oracle %d, oracle vs its 1-ulp-perturbed runs %s"
              % (base, base + 1, n_off, floor))
        assert n_off <= max(floor) + 5, (base, n_off, floor)


# ------------------------------------------------------------------------------------------------
# ONet-Opt variant (BASELINE config #1): encoder / decoder / optimiser through ifd_onet_* against the fixtures
# generated from the reference's ONet modules and against the ONet oracle
# ------------------------------------------------------------------------------------------------
@pytest.fixture(scope="module")
def og():
    import os
    return np.load(os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden", "onet_golden.npz"))


@pytest.fixture(scope="module")
def onet():
    import ifdefense_amd as I
    r = I.OnetRestorer(I.weights.pack_state_dict(I.weights.onet_random_state_dict(0), "onet"), device="cuda:0")
    yield r
    r.close()


def test_onet_encoder_latent_code(onet, og):
    c = onet.encode_inputs(torch.from_numpy(og["sel"])).cpu().numpy()
    assert _rel(c, og["c"]) < 1e-5
    # ragged: a shorter cloud in a padded batch == that cloud alone
    sel = torch.from_numpy(og["sel"]).clone()
    sel[1, 200:] = 0
    a = onet.encode_inputs(sel, torch.tensor([300, 200, 300, 300]))
    b = onet.encode_inputs(sel[1:2, :200])
    assert torch.equal(a[1], b[0])


def test_onet_decode_logits_and_input_gradient(onet, og, golden):
    p = torch.from_numpy(golden["init_points"][:2])
    logits, grad = onet.decode(p, torch.from_numpy(og["c"][:2]), want_grad=True)
    assert _rel(logits.cpu().numpy(), og["dec_logits"]) < 1e-5
    # 22 ReLUs x 256 channels per point: a pre-activation within rounding of 0 flips one mask bit (CBN is folded to
    # a x + b here, (x - mean) / sqrt(var + eps) * gamma + beta in the reference) - count such points, bound the rest
    dg = np.abs(grad.cpu().numpy() - og["dec_dlogit_dp"]).max(-1) / np.abs(og["dec_dlogit_dp"]).max()
    assert (dg > 1e-4).sum() <= 3 and dg.max() < 2e-2 and np.median(dg) < 1e-6, (dg.max(), (dg > 1e-4).sum())
    assert torch.equal(onet.decode(p, torch.from_numpy(og["c"][:2])), logits)
    for K in (1, 17, 129, 1000):                                  # ragged point counts
        got = onet.decode(p[:, :K], torch.from_numpy(og["c"][:2])).cpu().numpy()
        assert got.shape == (2, K) and _rel(got, og["dec_logits"][:, :K]) < 1e-5, K


def test_onet_p1_teacher_forced_and_p2_free_running(onet, og):
    c = torch.from_numpy(og["c"][:2])
    for t in (0, 1, 9):
        x = torch.from_numpy(og[f"traj{t}_x"])
        state = (torch.from_numpy(og[f"traj{t}_m"]), torch.from_numpy(og[f"traj{t}_v"]), t)
        out = onet.optimize_points(x, c, rep_weight=500.0, steps=1, state=state, normalize=False)
        d = np.abs(out.cpu().numpy() - og[f"traj{t}_x_next"])
        print("ONet P1 t=%d: coordinates off by > 1e-6: %d of %d" % (t + 1, (d > 1e-6).sum(), d.size))
        assert (d > 1e-6).sum() == 0, (t, float(d.max()), int((d > 1e-6).sum()))     # measured: 0 of 6144
    x10 = onet.optimize_points(torch.from_numpy(og["traj0_x"]), c, rep_weight=500.0, steps=10, normalize=False)
    d10 = np.linalg.norm(x10.cpu().numpy() - og["traj9_x_next"], axis=-1)
    print("ONet P2: 10 steps max %.2e" % d10.max())
    assert d10.max() < 1e-3
    _, loss = onet.optimize_points(torch.from_numpy(og["traj0_x"]), c, rep_weight=500.0, steps=1, normalize=False,
                                   return_loss=True)
    loss = loss.cpu().numpy().astype(np.float64)
    np.testing.assert_allclose(loss[:, 0].sum() / 2, og["traj0_loss"][0], rtol=1e-5)
    np.testing.assert_allclose(loss[:, 1].mean() * 500.0, og["traj0_loss"][1], rtol=1e-5)


@pytest.mark.both_precisions
def test_onet_large_clouds_more_than_1024_points(onet, og, golden, precision_mode):
    """ONet-Opt beyond the persistent kernel's 1024 points (ONet/opt_defense.py:27 takes any --sample_npoint): the
    two-launch-per-step path (onet.hip onet_large_occupancy_kernel / onet_bf.hip's split-precision one + the shared step kernels)
    against the ONet oracle - teacher-forced Adam steps from the oracle's trajectory and a short free run with normalisation.
    In both precisions (f32 and the f32-equivalent bf16x6), and the two DO differ (the mode reaches this path)."""
    from oracle import convonet_oracle as O
    from oracle import onet_oracle as OO
    w = OO.to_torch(OO.make_random_weights(0))
    c = torch.from_numpy(og["c"][:2])
    g = torch.Generator().manual_seed(23)
    for K in (1025, 2048):
        base = torch.from_numpy(golden["init_points"][:2])
        idx = torch.randint(0, 1024, (2, K), generator=g)
        init = (torch.gather(base, 1, idx[..., None].expand(2, K, 3)) + 0.01 * torch.randn(2, K, 3, generator=g)).clamp(-0.45, 0.45)
        x, m, v = init.clone(), torch.zeros_like(init), torch.zeros_like(init)
        for t in range(1, 4):
            _, idx_hip = onet.repulsion_loss(x, want_idx=True)
            xg = x.clone().requires_grad_(True)
            occ = OO.losses(w, xg, c, 0.0)[0]
            (occ + O.repulsion_loss(xg, idx_hip.cpu().long()).sum() / 2.0 * 500.0).backward()
            x_next, m_next, v_next = O.adam_step(x, xg.grad, m, v, t)
            x1, (m1, v1, _) = onet.optimize_points(x, c, rep_weight=500.0, steps=1, normalize=False, state=(m, v, t - 1),
                                                   return_state=True)
            # ReLU-boundary flips of the 256-wide decoder move single points (ONet P1: <= 1 point per 2048, section 4.4)
            off = (np.abs(x1.cpu().numpy() - x_next.numpy()) > 1e-6).any(-1)
            g_hip = (m1.cpu().numpy() - 0.9 * m.numpy()) / 0.1
            gerr = np.abs(g_hip - xg.grad.numpy()).max(-1) / np.abs(xg.grad.numpy()).max()
            print("ONet large K=%d t=%d: points off by > 1e-6: %d of %d, gradient error median %.1e, points > 1e-4: %d" %
                  (K, t, off.sum(), off.size, np.median(gerr), (gerr > 1e-4).sum()))
            assert off.sum() <= 2 and np.median(gerr) < 1e-6 and (gerr > 1e-4).sum() <= 2, (K, t)
            x, m, v = x_next, m_next, v_next
        ref = OO.optimize_points(w, init, c, rep_weight=500.0, iterations=3, normalize=True)
        got = onet.optimize_points(init, c, rep_weight=500.0, iterations=3, normalize=True)
        err = np.abs(got.cpu().numpy() - ref.numpy()).max(-1)
        print("ONet large K=%d free run of 4 steps + normalisation: %d of %d points off by > 2e-5 (max %.1e)" % (K, (err > 2e-5).sum(), err.size, err.max()))
        max_off, max_err = {1025: (2, 1e-5), 2048: (8, 1.2e-3)}[K]          # 2x the measured 0 / 4, 2.6e-6 / 5.7e-4
        assert (err > 2e-5).sum() <= max_off and err.max() < max_err, (K, int((err > 2e-5).sum()), float(err.max()))
        assert abs(float(got.norm(dim=-1).max()) - 1.0) < 1e-5
        other = onet.optimize_points(init, c, rep_weight=500.0, iterations=3, normalize=True,
                                     precision="bf16x6" if precision_mode == "f32" else "f32")
        assert not torch.equal(other, got) and float((other - got).abs().max()) < 1.2e-3


def test_onet_large_clouds_beyond_4096_points(onet, og):
    """ONet-Opt with more than 4096 optimised points: the shared large_step_kernel with its accumulators in global memory."""
    from oracle import onet_oracle as OO
    w = OO.to_torch(OO.make_random_weights(0))
    c = torch.from_numpy(og["c"][:2])
    g = torch.Generator().manual_seed(31)
    init = (torch.rand(2, 4500, 3, generator=g) - 0.5) * 0.9
    ref = OO.optimize_points(w, init, c, rep_weight=500.0, iterations=2, normalize=True)
    got = onet.optimize_points(init, c, rep_weight=500.0, iterations=2, normalize=True)
    err = np.abs(got.cpu().numpy() - ref.numpy()).max(-1)
    print("ONet large K=4500 free run of 3 steps + normalisation: %d of %d points off by > 2e-5 (max %.1e)" % ((err > 2e-5).sum(), err.size, err.max()))
    assert (err > 2e-5).sum() <= 2 and err.max() < 5e-5                      # 2x the measured 1, 2.2e-5
    assert abs(float(got.norm(dim=-1).max()) - 1.0) < 1e-5
    with pytest.raises(Exception):
        onet.optimize_points(torch.zeros(1, 10001, 3), c[:1], iterations=1)


def test_onet_end_to_end_and_sharding(onet, og, golden):
    init = torch.from_numpy(golden["init_points"])
    c = torch.from_numpy(og["c"])
    out = onet.optimize_points(init, c, rep_weight=500.0, iterations=10)
    d = np.linalg.norm(out.cpu().numpy() - og["e2e10_out"], axis=-1)
    assert d.max() < 1e-3, d.max()
    lo = onet.optimize_points(init[:1], c[:1], rep_weight=500.0, iterations=10, loss_batch=4)
    hi = onet.optimize_points(init[1:], c[1:], rep_weight=500.0, iterations=10, loss_batch=4)
    assert torch.equal(torch.cat([lo, hi]), out)                  # P4: shard-and-concatenate is bitwise
    with pytest.raises(Exception):
        onet.encode_points(torch.zeros(1, 300, 3))


@pytest.mark.both_precisions
def test_onet_config1_pipeline_16_clouds(onet, precision_mode):
    """BASELINE config #1: ONet-Opt on 16 clean 1024-point clouds, 50 iterations - the whole driver
    (SOR -> preprocess / 300-point subset -> encoder -> init -> optimiser -> normalise) against the oracle fed
    with the same draws, plus sharding invariance."""
    from oracle import convonet_oracle as CO
    from oracle import onet_oracle as OO
    import bench
    import ifdefense_amd as I
    ow = OO.to_torch(OO.make_random_weights(0))
    clouds = bench.synth_clouds(16)
    x = torch.from_numpy(clouds).cuda()
    keep = onet.sor(x)
    ref_keep, _ = CO.sor_keep_mask(torch.from_numpy(clouds))
    assert np.array_equal(keep.cpu().numpy().astype(bool), ref_keep.numpy().astype(bool))
    prep = onet.prepare(x, keep, n_sel=300, seed=21)
    assert prep["t_per_cloud"].cpu().tolist() == [300] * 16
    c = onet.encode_inputs(prep["sel"], prep["t_per_cloud"])
    c_ref = OO.encode_latent(ow, prep["sel"].cpu())
    assert _rel(c.cpu().numpy(), c_ref.numpy()) < 1e-5
    hip = onet.optimize_points(prep["init"], c, rep_weight=500.0, iterations=9).cpu().numpy()
    ref = OO.optimize_points(ow, prep["init"].cpu(), c_ref, rep_weight=500.0, iterations=9).numpy()
    d = np.linalg.norm(hip - ref, axis=-1)
    print("ONet config #1, 10 steps, 16 clouds: max %.2e median %.2e, points > 1e-3: %d of %d" %
          (d.max(), np.median(d), (d > 1e-3).sum(), d.size))
    assert (d > 1e-3).mean() < 1e-3 and np.median(d) < 1e-5
    # every separating point attributed (block comment above _oracle_trace), the optimiser alone on the build's latent codes
    hx, hg, _ = _hip_trace(onet, prep["init"], c, 10, 16)
    ox, og = _oracle_trace(OO, ow, prep["init"].cpu(), c.cpu(), 10, 16)
    _attribute_separations(onet, OO, ow, c.cpu(), 16, hx, hg, ox, og, "ONet config #1 (optimiser alone)")
    n_opt = int((np.linalg.norm(hx[-1] - ox[-1], axis=-1) > 1e-3).sum())
    floor = _ulp_floor(OO, ow, prep["init"].cpu(), lambda q: OO.optimize_points(ow, q, c.cpu(), rep_weight=500.0, iterations=9,
                                                                               normalize=False).numpy(), 9, 16, ox[-1], cond=c.cpu())
    print("ONet config #1: points beyond 1e-3 after 10 steps: build vs oracle %d, oracle vs its 1-ulp-perturbed runs %s" % (n_opt, floor))
    assert n_opt <= max(floor) + 5, (n_opt, floor)
    args = I.DefenseArgs(iterations=50, input_npoint=300, seed=21, precision=precision_mode)
    full = I.defend_point_cloud(onet, clouds, args)
    assert full.shape == (16, 1024, 3) and np.isfinite(full).all()
    np.testing.assert_allclose(np.linalg.norm(full, axis=-1).max(axis=1), 1.0, rtol=1e-6)
    lo = I.defend_point_cloud(onet, clouds[:5], args, cloud_index_base=0, total_clouds=16)
    hi = I.defend_point_cloud(onet, clouds[5:], args, cloud_index_base=5, total_clouds=16)
    assert np.array_equal(np.concatenate([lo, hi]), full)


def test_onet_partial_round_first_and_overlap_are_bit_identical(onet):
    """pipeline.defend_stream(tail_first / overlap) on the ONet restorer with n > CUs (round-5 advisor): the optimiser of one pass reads
    its folded CBN coefficients for the whole launch while ifd_onet_encode of the next pass runs on the second stream - they used to
    share the encoder scratch (api.cpp onet_fold; now a buffer of its own).  Enough steps that late-starting workgroups read their
    coefficients long after the other stream's encoder has started."""
    import bench
    import ifdefense_amd as I
    n_cu = torch.cuda.get_device_properties(0).multi_processor_count
    a = bench.synth_clouds(n_cu + 44, seed=70)
    b = bench.synth_clouds(n_cu + 9, seed=71)
    args = I.DefenseArgs(iterations=5, batch_size=192, seed=3, input_npoint=300)
    ref = [o.clone() for o in I.defend_stream(onet, [a, b], args, bases=[0, 1000], totals=[len(a), 1000 + len(b)],
                                             overlap=False, tail_first=False)]
    for overlap in (False, True):
        for _ in range(2):
            got = [o.clone() for o in I.defend_stream(onet, [a, b], args, bases=[0, 1000], totals=[len(a), 1000 + len(b)],
                                                     overlap=overlap, tail_first=True)]
            assert len(got) == 2 and torch.equal(got[0], ref[0]) and torch.equal(got[1], ref[1]), overlap


def test_onet_cli_end_to_end(tmp_path):
    """python -m ifdefense_amd.onet_opt_defense: ONet/opt_defense.py's output location / keys / dtypes."""
    import subprocess, sys, os
    import ifdefense_amd as I
    wpath = tmp_path / "onet.pth"
    sd = {k: torch.from_numpy(v) for k, v in I.weights.onet_random_state_dict(0).items()}
    sd["decoder.bn.bn.num_batches_tracked"] = torch.tensor(7)                      # present in real checkpoints, ignored
    torch.save(sd, wpath)
    g = dict(np.load(os.path.join(os.path.dirname(__file__), "golden", "convonet_golden.npz")))
    src = tmp_path / "clean.npz"
    np.savez(src, test_pc=g["raw"], test_label=np.array([0, 8, 30, 39]))
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    r = subprocess.run([sys.executable, "-m", "ifdefense_amd.onet_opt_defense", "--data_root", str(src), "--iterations=10",
                        "--weights", str(wpath)], capture_output=True, text=True, cwd=root, timeout=600)
    assert r.returncode == 0, r.stdout + r.stderr
    z = np.load(tmp_path / "ONet-Opt" / "onet_opt-clean.npz")
    assert sorted(z.files) == ["test_label", "test_pc"]
    assert z["test_pc"].shape == (4, 1024, 3) and z["test_pc"].dtype == np.float32 and z["test_label"].dtype == np.uint8
    np.testing.assert_allclose(np.linalg.norm(z["test_pc"], axis=-1).max(axis=1), 1.0, rtol=1e-6)


# ------------------------------------------------------------------------------------------------
# ONet-Mesh path (BASELINE config #4): MISE occupancy grid, marching cubes, surface sampling - against the reference's
# own native libraries (oracle/_ref, built by oracle/build_ref.py from the sources under /root/reference)
# ------------------------------------------------------------------------------------------------
def _ref_libs():
    import os, sys
    d = os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "oracle", "_ref")
    if not os.path.isdir(d) or not any(f.startswith("mise") for f in os.listdir(d)):
        pytest.skip("oracle/_ref not built (python oracle/build_ref.py needs /root/reference)")
    sys.path.insert(0, d)
    import mcubes, mise
    return mise, mcubes


def _cutting_threshold(onet, c):
    """The random-weight occupancy field never crosses logit(0.2) (the box would come out "full"): use the median
    occupancy over the box as the probability threshold so that the iso-surface cuts through the volume."""
    g = torch.Generator().manual_seed(9)
    p = (torch.rand(c.shape[0], 4096, 3, generator=g) - 0.5) * 1.1
    med = float(onet.decode(p, c).median())
    return float(1.0 / (1.0 + np.exp(-med)))


def _reference_grid(onet, mise, c_row, res0, steps, threshold, padding=0.1):
    """generate_from_latent's MISE loop (generation.py:97-129) around the reference's MISE class, with the occupancy
    values coming from the HIP decoder (the point set evaluated each round is what is under test)."""
    thr = np.log(threshold) - np.log(1. - threshold)
    m = mise.MISE(res0, steps, thr)
    pts = m.query()
    rounds = 0
    while pts.shape[0] != 0:
        pf = torch.from_numpy(pts.astype(np.float32))                    # torch.FloatTensor(points)
        pf = pf / m.resolution
        pf = (1 + padding) * (pf - 0.5)
        v = onet.decode(pf[None], c_row[None])[0].cpu().numpy().astype(np.float64)
        m.update(pts, v)
        pts = m.query()
        rounds += 1
    return m.to_dense(), thr, rounds


@pytest.mark.parametrize("res0,steps", [(8, 2), (16, 1), (32, 2)])
def test_onet_mesh_grid_and_surface_match_reference_libs(onet, og, res0, steps):
    mise, mcubes = _ref_libs()
    c = torch.from_numpy(og["c"][:2])
    MESH_THRESHOLD = _cutting_threshold(onet, c)
    out = onet.mesh_sample(c, n_sample=1024, resolution0=res0, upsampling_steps=steps, seed=3, want_grid=True,
                           want_triangles=True, max_triangles=300000, threshold=MESH_THRESHOLD)
    grid = out["grid"].cpu().numpy()
    ntri = out["n_triangles"].cpu().numpy()
    P = (res0 << steps) + 1
    for b in range(2):
        ref_grid, thr, rounds = _reference_grid(onet, mise, c[b], res0, steps, MESH_THRESHOLD)
        assert ref_grid.shape == (P, P, P)
        assert np.array_equal(grid[b].astype(np.float64), ref_grid), (b, np.abs(grid[b] - ref_grid).max())     # bit-exact
        # marching cubes of the reference on the same grid (generation.py:166-176)
        v, t = mcubes.marching_cubes(np.pad(ref_grid, 1, 'constant', constant_values=-1e6), thr)
        v = 1.1 * ((v - 0.5 - 1) / (P - 1) - 0.5)
        tris = out["triangles"][b, :ntri[b]].cpu().numpy().reshape(-1, 3, 3)
        assert ntri[b] > 0 and len(t) > 0
        # the reference's triangles, one for one: same count, same order (cube by cube, table order), same winding
        ref_tris = v[t.astype(np.int64)]
        assert ntri[b] == len(t), (ntri[b], len(t))
        np.testing.assert_allclose(tris, ref_tris, rtol=0, atol=2e-6)
        # same vertices (the iso-crossings of the grid edges)
        mine = np.unique(np.round(tris.reshape(-1, 3), 5), axis=0)
        theirs = np.unique(np.round(v, 5), axis=0)
        from scipy.spatial import cKDTree
        d_mt, _ = cKDTree(theirs).query(mine)
        d_tm, _ = cKDTree(mine).query(theirs)
        assert abs(len(mine) - len(theirs)) <= 8 and d_mt.max() < 3e-5 and d_tm.max() < 3e-5, (mine.shape, theirs.shape)
        def area(tr):
            return 0.5 * np.linalg.norm(np.cross(tr[:, 1] - tr[:, 0], tr[:, 2] - tr[:, 0]), axis=1).sum()
        a_mine, a_ref = area(tris), area(v[t.astype(np.int64)])
        assert abs(a_mine - a_ref) / a_ref < 1e-6, (a_mine, a_ref)
        # watertight: every undirected edge of the soup is shared by exactly two triangles
        key = np.round(tris, 5)
        e = np.concatenate([key[:, [0, 1]], key[:, [1, 2]], key[:, [2, 0]]]).reshape(-1, 6)
        a_, b_ = e[:, :3], e[:, 3:]
        swap = np.array([tuple(x) > tuple(y) for x, y in zip(a_, b_)])
        e = np.where(swap[:, None], np.concatenate([b_, a_], 1), np.concatenate([a_, b_], 1))
        # (on the box boundary the -1e6 padding collapses several crossings onto one grid point, which this
        # coordinate-keyed count cannot tell apart: look at the edges strictly inside the box)
        inner = (np.abs(e) < 0.55 - 1e-3).all(1)
        _, cnt = np.unique(e[inner], axis=0, return_counts=True)
        # (a grid corner within ~1e-4 of the iso-value pulls the crossings of its edges onto one point: a handful)
        assert (cnt != 2).mean() < 5e-3 and (cnt % 2 == 0).all(), np.bincount(cnt)
        # the samples lie on the surface: nearest mesh vertex within one voxel diagonal
        from scipy.spatial import cKDTree
        s = out["points"][b].cpu().numpy()
        d, _ = cKDTree(theirs).query(s)
        assert d.max() < 1.1 / (P - 1) * 1.8, d.max()
        print("ONet-Mesh res0=%d steps=%d cloud %d: %d MISE rounds, %d triangles (reference %d), area %.4f vs %.4f" %
              (res0, steps, b, rounds, ntri[b], len(t), a_mine, a_ref))
    # sharding / seeding: same seed + global index -> same samples
    again = onet.mesh_sample(c[1:], n_sample=1024, resolution0=res0, upsampling_steps=steps, seed=3, cloud_index_base=1,
                             threshold=MESH_THRESHOLD)
    assert torch.equal(again["points"][0], out["points"][1])


def test_onet_mesh_driver_and_cli(onet, tmp_path):
    """remesh_point_cloud (ONet/remesh_defense.py:228-262) + the CLI: output schema, unit-sphere normalisation,
    sharding invariance, and the empty-mesh fallback."""
    import subprocess, sys, os
    import bench
    import ifdefense_amd as I
    clouds = bench.synth_clouds(6)
    args = I.DefenseArgs(input_npoint=300, seed=4)
    full = I.remesh_point_cloud(onet, clouds, args)
    assert full.shape == (6, 1024, 3) and full.dtype == np.float32 and np.isfinite(full).all()
    np.testing.assert_allclose(np.linalg.norm(full, axis=-1).max(axis=1), 1.0, rtol=1e-6)
    lo = I.remesh_point_cloud(onet, clouds[:2], args, cloud_index_base=0)
    hi = I.remesh_point_cloud(onet, clouds[2:], args, cloud_index_base=2)
    assert np.array_equal(np.concatenate([lo, hi]), full)
    # empty mesh (threshold above every occupancy): the reference falls back to the (post-SOR) input points
    onet_hi = I.OnetRestorer(I.weights.pack_state_dict(I.weights.onet_random_state_dict(0), "onet"), device="cuda:0",
                             threshold=0.999)
    fb = I.remesh_point_cloud(onet_hi, clouds[:2], I.DefenseArgs(input_npoint=300, seed=4, sor=False))
    ref = clouds[:2] - clouds[:2].mean(1, keepdims=True)
    ref = ref / np.linalg.norm(ref, axis=-1).max(1)[:, None, None]
    np.testing.assert_allclose(fb, ref, atol=1e-6)
    onet_hi.close()
    # CLI
    wpath = tmp_path / "onet.pth"
    torch.save({k: torch.from_numpy(v) for k, v in I.weights.onet_random_state_dict(0).items()}, wpath)
    src = tmp_path / "adv.npz"
    np.savez(src, test_pc=clouds[:3], test_label=np.array([1, 2, 3]), target_label=np.array([4, 5, 6]))
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    r = subprocess.run([sys.executable, "-m", "ifdefense_amd.remesh_defense", "--data_root", str(src), "--weights", str(wpath)],
                       capture_output=True, text=True, cwd=root, timeout=600)
    assert r.returncode == 0, r.stdout + r.stderr
    z = np.load(tmp_path / "ONet-Mesh" / "onet_remesh-adv.npz")
    assert sorted(z.files) == ["target_label", "test_label", "test_pc"]
    assert z["test_pc"].shape == (3, 1024, 3) and z["test_pc"].dtype == np.float32 and z["target_label"].dtype == np.uint8


def test_onet_mesh_samples_match_oracle_distribution(onet, og):
    """End of the mesh path: the surface samples are random draws (unseeded in the reference), so compare
    distributions - symmetric Chamfer distance between HIP samples and the CPU oracle's (reference MISE + libmcubes
    around the oracle decoder) against the oracle's own seed-to-seed Chamfer distance."""
    _ref_libs()
    from oracle import mesh_oracle as MO
    from oracle import onet_oracle as OO
    ow = OO.to_torch(OO.make_random_weights(0))
    c = torch.from_numpy(og["c"][:1])
    thr = _cutting_threshold(onet, c)
    grid, t = MO.occupancy_grid(ow, c[0], resolution0=16, upsampling_steps=1, threshold=thr)
    v, f = MO.extract_mesh(grid, t)
    a = MO.sample_surface(v, f, 1024, np.random.default_rng(0)).astype(np.float32)
    b = MO.sample_surface(v, f, 1024, np.random.default_rng(1)).astype(np.float32)
    hip = onet.mesh_sample(c, resolution0=16, upsampling_steps=1, threshold=thr, seed=5)["points"][0].cpu().numpy()
    cd_self, cd_hip = _chamfer(a, b), 0.5 * (_chamfer(hip, a) + _chamfer(hip, b))
    print("ONet-Mesh samples: Chamfer HIP<->oracle %.4f, oracle seed<->seed %.4f" % (cd_hip, cd_self))
    assert cd_hip < 1.15 * cd_self


# ------------------------------------------------------------------------------------------------
# The hot tile's gradient, pinned directly (round 2).  `ifd_decode` runs the stand-alone decoder tile; the optimiser
# runs the software-pipelined two-sub-tile one, which used to be checked only through Adam-normalised positions (a
# ~1 % gradient error hides behind |dx| <= 1e-6 there).  One teacher-forced step with return_state gives m1 and v1:
#   g = m0 + (m1 - m0) / (1 - beta1)      v1 = beta2 v0 + (1 - beta2) g^2
# and that g is the hot kernel's total gradient (decoder tile + repulsion) - compared with the reference's autograd
# gradient of the same step (fixtures traj{t}_g) at 1e-4 of its maximum.
# ------------------------------------------------------------------------------------------------
def _golden_file(name):
    import os
    return np.load(os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden", name))


def _hot_gradient(restorer_, fixtures, cond, t):
    x = torch.from_numpy(fixtures[f"traj{t}_x"])
    m0, v0 = fixtures[f"traj{t}_m"], fixtures[f"traj{t}_v"]
    out, (m1, v1, t1) = restorer_.optimize_points(x, cond, rep_weight=500.0, steps=1, normalize=False,
                                                  state=(torch.from_numpy(m0), torch.from_numpy(v0), t), return_state=True)
    m1, v1 = m1.cpu().numpy().astype(np.float64), v1.cpu().numpy().astype(np.float64)
    g = m0 + (m1 - m0) / 0.1
    return out.cpu().numpy(), g, v1


@pytest.mark.parametrize("fixture,steps", [("convonet_golden.npz", (0, 1, 9, 49)), ("convonet_golden_long.npz", (99, 499)),
                                           ("convonet_golden_seed1.npz", (0, 1, 9, 49, 99))])
def test_hot_tile_gradient_against_reference_autograd(np_weights, fixture, steps):
    import ifdefense_amd as I
    from oracle import convonet_oracle as O
    f = _golden_file(fixture)
    if "seed1" in fixture:
        r = I.Restorer(I.weights.pack_state_dict(O.make_random_weights(1)), device="cuda:0")
        planes = {pl: torch.from_numpy(f["planes"][:, i]) for i, pl in enumerate(PL)}        # 4 clouds, B = 4
    else:
        r = I.Restorer(I.weights.pack_state_dict(np_weights), device="cuda:0")
        g0 = _golden_file("convonet_golden.npz")
        planes = {pl: torch.from_numpy(g0["planes01"][:, i]) for i, pl in enumerate(PL)}
    try:
        for t in steps:
            x_next, g, v1 = _hot_gradient(r, f, planes, t)
            g_ref = f[f"traj{t}_g"].astype(np.float64)
            v_ref = 0.999 * f[f"traj{t}_v"].astype(np.float64) + 0.001 * g_ref ** 2
            eg = np.abs(g - g_ref).max() / np.abs(g_ref).max()
            ev = np.abs(v1 - v_ref).max() / np.abs(v_ref).max()
            flips = int((np.abs(x_next - f[f"traj{t}_x_next"]) > 1e-6).sum())
            print("%s t=%d: gradient rel. error %.2e, v1 rel. error %.2e, coordinates off by > 1e-6: %d of %d" %
                  (fixture, t, eg, ev, flips, x_next.size))
            # measured on MI355X (round 2): gradient 4e-7 ... 6e-7 of its maximum at every t and both weight seeds, v1
            # 1e-5 (t = 0, where v1 = 0.001 g^2 is recovered from a difference) ... 7e-8, and not one coordinate off by
            # more than 1e-6 after the Adam update (max 3e-8) - the bars sit ~10x above the measured values
            assert eg < 5e-6 and ev < 1e-4, (fixture, t, eg, ev)
            assert flips == 0, (fixture, t, flips)
    finally:
        r.close()


def test_hot_tile_gradient_onet(onet, og):
    c = torch.from_numpy(og["c"][:2])
    for t in (0, 1, 9):
        x_next, g, v1 = _hot_gradient(onet, og, c, t)
        g_ref = og[f"traj{t}_g"].astype(np.float64)
        # 22 ReLUs x 256 channels per point: a pre-activation within rounding of zero flips a mask bit (see
        # test_onet_decode_logits_and_input_gradient) - per point, relative to the largest gradient
        dg = np.abs(g - g_ref).max(-1) / np.abs(g_ref).max()
        print("ONet t=%d: gradient error median %.2e max %.2e, points > 1e-4: %d" % (t, np.median(dg), dg.max(), (dg > 1e-4).sum()))
        assert (dg > 1e-4).sum() <= 2 and np.median(dg) < 5e-7 and dg.max() < 2e-3, (t, dg.max())   # measured: <= 1 point, max 4e-4


def test_decode_second_weight_seed(np_weights):
    """G2 on the second parity base: logits and d(sum logits)/dp of the stand-alone tile, 4 clouds, weight seed 1."""
    import ifdefense_amd as I
    from oracle import convonet_oracle as O
    f = _golden_file("convonet_golden_seed1.npz")
    r = I.Restorer(I.weights.pack_state_dict(O.make_random_weights(1)), device="cuda:0")
    try:
        planes = {pl: torch.from_numpy(f["planes"][:, i]) for i, pl in enumerate(PL)}
        logits, grad = r.decode(torch.from_numpy(f["traj0_x"]), planes, want_grad=True)
        assert _rel(logits.cpu().numpy(), f["dec_logits"]) < 1e-5
        assert _rel(grad.cpu().numpy(), f["dec_dlogit_dp"]) < 1e-4
    finally:
        r.close()


@pytest.mark.parametrize("mode", ["bf16x6", "bf16x3"])
def test_decode_seam_in_split_precision(restorer, golden, planes2, oracle_weights, mode):
    """ifd_decode_ex: the stand-alone decoder in the arithmetic of ifd_opt_params.precision (the optimiser's split-precision tile
    evaluating sum(logits)).  bf16x6 to the f32 seam's bounds against the reference's logits and autograd gradient; bf16x3 (reduced:
    three of the six piece products) to 2e-3 of the largest value.  Ragged point counts (a tile is 32 points), clouds of more than
    1024 points, coordinates outside the cube, and the sharing of one cloud's tiles over several workgroups (few clouds) against one
    workgroup per cloud (many clouds) bit for bit."""
    from oracle import convonet_oracle as O
    tl, tg = (1e-5, 1e-4) if mode == "bf16x6" else (2e-3, 2e-3)
    p = torch.from_numpy(golden["init_points"][:2])
    logits, grad = restorer.decode(p, planes2, want_grad=True, precision=mode)
    el, eg = _rel(logits.cpu().numpy(), golden["dec_logits"]), _rel(grad.cpu().numpy(), golden["dec_dlogit_dp"])
    print("%s seam: logits %.2e, gradient %.2e of the largest value" % (mode, el, eg))
    assert el < tl and eg < tg
    assert torch.equal(restorer.decode(p, planes2, precision=mode), logits)
    f32 = restorer.decode(p, planes2, precision="f32")
    assert not torch.equal(f32, logits)                           # (another arithmetic did run)
    for K in (1, 31, 33, 100, 1000):
        q = torch.from_numpy(golden["init_points"][:2, :K]).clone()
        ref = O.decode_logits(oracle_weights, q, planes2).numpy()
        got = restorer.decode(q, planes2, precision=mode).cpu().numpy()
        assert got.shape == (2, K) and _rel(got, ref) < tl, K
    g = torch.Generator().manual_seed(5)
    q = (torch.rand(2, 1500, 3, generator=g) - 0.5) * 1.4         # beyond the cube (clamped => zero plane gradient) and K > 1024
    qr = q.clone().requires_grad_()
    ref = O.decode_logits(oracle_weights, qr, planes2)
    ref.sum().backward()
    lg, gr = restorer.decode(q, planes2, want_grad=True, precision=mode)
    assert _rel(lg.cpu().numpy(), ref.detach().numpy()) < tl
    dg = np.abs(gr.cpu().numpy() - qr.grad.numpy()).max(-1) / np.abs(qr.grad.numpy()).max()
    print("%s seam, 3000 random points: gradient median %.2e, max %.2e, beyond the bound %d" % (mode, np.median(dg), dg.max(), (dg > tg).sum()))
    if mode == "bf16x6":
        assert dg.max() < tg
    else:                                  # the reduced mode flips a ReLU mask where a pre-activation is within ~1e-3 of zero: count those
        assert (dg > tg).sum() <= 6 and dg.max() < 5e-2 and np.median(dg) < 2e-4
    # many clouds: one workgroup per cloud; two clouds: each cloud's tiles over many workgroups
    B = 300
    pm = q[:1].expand(B, -1, -1).contiguous()
    cm = {k: v[:1].expand(B, -1, -1, -1).contiguous() for k, v in planes2.items()}
    lm, gm = restorer.decode(pm, cm, want_grad=True, precision=mode)
    assert torch.equal(lm[0], lg[0]) and torch.equal(lm[B - 1], lg[0]) and torch.equal(gm[B - 1], gr[0])


@pytest.mark.parametrize("mode", ["bf16x6", "bf16x3"])
def test_onet_decode_seam_in_split_precision(onet, og, golden, mode):
    """ifd_onet_decode_ex: the ONet decoder seam on the split-precision passes of ifd_onet_optimize."""
    tl = 1e-5 if mode == "bf16x6" else 2e-3
    p = torch.from_numpy(golden["init_points"][:2])
    c = torch.from_numpy(og["c"][:2])
    logits, grad = onet.decode(p, c, want_grad=True, precision=mode)
    el = _rel(logits.cpu().numpy(), og["dec_logits"])
    dg = np.abs(grad.cpu().numpy() - og["dec_dlogit_dp"]).max(-1) / np.abs(og["dec_dlogit_dp"]).max()
    print("%s ONet seam: logits %.2e; gradient: %d points beyond 1e-4, median %.2e, max %.2e" % (mode, el, (dg > 1e-4).sum(), np.median(dg), dg.max()))
    assert el < tl
    if mode == "bf16x6":                                           # (ReLU-boundary points as in the f32 test above)
        assert (dg > 1e-4).sum() <= 3 and dg.max() < 2e-2 and np.median(dg) < 1e-6
    else:
        assert np.median(dg) < 2e-3 and dg.max() < 5e-2
    assert torch.equal(onet.decode(p, c, precision=mode), logits)
    assert not torch.equal(onet.decode(p, c, precision="f32"), logits)
    for K in (1, 17, 129, 1000):
        got = onet.decode(p[:, :K], c, precision=mode).cpu().numpy()
        assert got.shape == (2, K) and _rel(got, og["dec_logits"][:, :K]) < tl, K


def _philox4x32_10(k0, k1, c0, c1, c2, c3):
    """Philox-4x32-10 (Salmon et al. 2011) on uint32 numpy arrays - the generator of prep.hip / mesh.hip."""
    k0, k1 = np.uint64(k0), np.uint64(k1)
    c = [np.asarray(x, np.uint64) for x in np.broadcast_arrays(c0, c1, c2, c3)]
    M = np.uint64(0xffffffff)
    for _ in range(10):
        p0, p1 = np.uint64(0xD2511F53) * c[0], np.uint64(0xCD9E8D57) * c[2]
        c = [((p1 >> np.uint64(32)) ^ c[1] ^ k0) & M, p1 & M, ((p0 >> np.uint64(32)) ^ c[3] ^ k1) & M, p0 & M]
        k0, k1 = (k0 + np.uint64(0x9E3779B9)) & M, (k1 + np.uint64(0xBB67AE85)) & M
    return c


def test_onet_mesh_sampler_is_the_documented_algorithm_on_its_own_uniforms(onet, og):
    """trimesh.sample.sample_surface, pinned with explicit uniforms: the test regenerates the counter-based draws of the
    GPU sampler (Philox keyed by seed / global cloud index / sample index), feeds them to the restated documented
    algorithm (oracle/mesh_oracle.sample_surface) on the SAME triangle soup, and compares sample by sample."""
    from oracle import mesh_oracle as MO
    c = torch.from_numpy(og["c"][:2])
    thr = _cutting_threshold(onet, c)
    seed, base, n = 11, 5, 1024
    out = onet.mesh_sample(c, n_sample=n, resolution0=16, upsampling_steps=1, seed=seed, cloud_index_base=base,
                           want_triangles=True, max_triangles=200000, threshold=thr)
    for b in range(2):
        nt = int(out["n_triangles"][b])
        tris = out["triangles"][b, :nt].cpu().numpy().reshape(-1, 3, 3).astype(np.float64)
        r = _philox4x32_10(seed & 0xffffffff, seed >> 32, np.arange(n), 0x5a3f, base + b, 0)
        pick = (r[0].astype(np.float64) + r[1].astype(np.float64) * 4294967296.0) / 18446744073709551616.0
        u = (r[2].astype(np.float32) * np.float32(1.0 / 4294967296.0)).astype(np.float64)
        v = (r[3].astype(np.float32) * np.float32(1.0 / 4294967296.0)).astype(np.float64)
        ref = MO.sample_surface(tris.reshape(-1, 3), np.arange(3 * nt).reshape(-1, 3), n,
                                uniforms=np.stack([pick, u, v], 1))
        d = np.abs(out["points"][b].cpu().numpy() - ref).max(axis=1)
        # the GPU accumulates the areas from the double-precision vertices, this test from their float32 copies: a pick
        # within ~1e-7 of a face boundary may land on the neighbouring face
        print("mesh sampler cloud %d: %d of %d samples differ (max of the rest %.1e)" % (b, (d > 1e-5).sum(), n, d[d <= 1e-5].max()))
        assert (d > 1e-5).sum() <= 2, (b, int((d > 1e-5).sum()))


def test_streamed_driver_equals_serial_passes(restorer, golden):
    """pipeline.defend_stream (pre-processing of pass n + 1 on a second HIP stream under pass n's optimiser) returns,
    array by array, exactly what one defend_point_cloud call per array returns - also with a chunk size that cuts the
    arrays into several device passes."""
    import bench
    import ifdefense_amd as I
    arrays = [bench.synth_clouds(5, seed=40), golden["raw"], bench.synth_clouds(3, seed=41)]
    bases = [0, 7, 100]
    for chunk in (4096, 2):
        args = I.DefenseArgs(iterations=12, batch_size=3, seed=3, chunk=chunk)
        serial = [I.defend_point_cloud(restorer, a, args, cloud_index_base=b, total_clouds=b + len(a), return_device=True)
                  for a, b in zip(arrays, bases)]
        streamed = [o.clone() for o in I.defend_stream(restorer, arrays, args, bases=bases,
                                                       totals=[b + len(a) for a, b in zip(arrays, bases)])]
        assert len(streamed) == 3
        for s_, t_ in zip(serial, streamed):
            assert torch.equal(s_, t_)


def test_partial_round_first_is_bit_identical(restorer):
    """pipeline.defend_stream(tail_first=True) (round 5): the clouds of a file's partial last round (n mod CUs) prepared and optimised
    first, the pre-processing of the others on the second stream under that round - same output, bit for bit, as the one-pass order;
    one file and a stream of two, with and without cross-file overlap."""
    import bench
    import ifdefense_amd as I
    n_cu = torch.cuda.get_device_properties(0).multi_processor_count
    a = bench.synth_clouds(n_cu + 44, seed=60)
    b = bench.synth_clouds(2 * n_cu + 3, seed=61)
    args = I.DefenseArgs(iterations=6, batch_size=192, seed=3)
    ref = [o.clone() for o in I.defend_stream(restorer, [a, b], args, bases=[0, 1000], totals=[len(a), 1000 + len(b)],
                                             overlap=False, tail_first=False)]
    for overlap in (False, True):
        got = [o.clone() for o in I.defend_stream(restorer, [a, b], args, bases=[0, 1000], totals=[len(a), 1000 + len(b)],
                                                 overlap=overlap, tail_first=True)]
        assert len(got) == 2 and torch.equal(got[0], ref[0]) and torch.equal(got[1], ref[1]), overlap
    one = I.defend_point_cloud(restorer, a, args, cloud_index_base=0, total_clouds=len(a), return_device=True)
    assert torch.equal(one, ref[0])


def test_lists_equal_scan_on_mixed_inputs(restorer):
    """Regression net for the neighbour lists (exact ties, refreshes, ragged K): a slice of scripts/fuzz_lists_vs_scan.py."""
    import bench
    rng = np.random.default_rng(5)
    for it, make in enumerate((lambda c: c, bench.knn_attack_like, lambda c: bench.subsample_like(c, 256))):
        clouds = make(bench.synth_clouds(14, seed=300 + it))
        x = torch.from_numpy(clouds).cuda()
        prep = restorer.prepare(x, restorer.sor(x), seed=it)
        planes = restorer.encode_inputs(prep["sel"], prep["t_per_cloud"])
        for K, steps in ((1024, 200), (int(rng.integers(6, 600)), 120)):
            init = prep["init"][:, :K].contiguous()
            a = restorer.optimize_points(init, planes, rep_weight=500.0, steps=steps, normalize=False)
            b = restorer.optimize_points(init, planes, rep_weight=500.0, steps=steps, normalize=False, knn_scan_every_step=True)
            assert torch.equal(a, b), (it, K, steps)


_RCCL_SCRIPT = r'''
import os, sys
import numpy as np
import torch
import torch.distributed as dist
sys.path.insert(0, sys.argv[1])
import ifdefense_amd as I
from ifdefense_amd import dist as D

rank, world, local = D.init_from_env()
assert dist.is_initialized() and dist.get_backend() == "nccl", "torchrun launch must create the RCCL group"
dev = torch.device("cuda", local)
r = I.Restorer(I.weights.pack_state_dict(I.weights.random_state_dict(0)), device=dev)
pc = np.load(sys.argv[2])
args = I.DefenseArgs(iterations=10, seed=3)

def defend(shard, base, total):
    return I.defend_point_cloud(r, torch.from_numpy(shard).to(dev), args, cloud_index_base=base, total_clouds=total,
                                return_device=True)

out = D.defend_sharded(defend, pc)
t = torch.tensor([float(rank + 1)], device=dev)
dist.all_reduce(t, op=dist.ReduceOp.MAX)
assert float(t.item()) == float(world)
if rank == 0:
    np.save(sys.argv[3], out.cpu().numpy())
dist.barrier()
dist.destroy_process_group()
'''


def test_rccl_path_under_torchrun(tmp_path):
    """The N > 1 launch line of the bench contract with one rank: `python -m torch.distributed.run --nproc-per-node 1`
    creates the RCCL communicator on the GPU, and sharding + ncclAllGather + barrier + all-reduce run for real.  The
    gathered array is bit-identical to the in-process result; bench.py prints its one JSON line under the same launch."""
    import json, os, socket, subprocess, sys
    import ifdefense_amd as I
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    rng = np.random.default_rng(11)
    v = rng.normal(size=(5, 1024, 3)).astype(np.float32)
    pc = (v / np.linalg.norm(v, axis=-1, keepdims=True) * rng.uniform(0.3, 1.0, size=(5, 1024, 1))).astype(np.float32)
    np.save(tmp_path / "pc.npy", pc)
    (tmp_path / "rccl_run.py").write_text(_RCCL_SCRIPT)

    def launch(script_args):
        with socket.socket() as s:
            s.bind(("127.0.0.1", 0))
            port = s.getsockname()[1]
        env = dict(os.environ, HSA_ENABLE_IPC_MODE_LEGACY="0")
        return subprocess.run([sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", "1",
                               "--master-addr", "127.0.0.1", "--master-port", str(port)] + script_args,
                              capture_output=True, text=True, cwd=root, timeout=900, env=env)

    r = launch([str(tmp_path / "rccl_run.py"), root, str(tmp_path / "pc.npy"), str(tmp_path / "out.npy")])
    assert r.returncode == 0, r.stdout + r.stderr
    args = I.DefenseArgs(iterations=10, seed=3)
    r0 = I.Restorer(I.weights.pack_state_dict(I.weights.random_state_dict(0)), device="cuda:0")
    want = I.defend_point_cloud(r0, torch.from_numpy(pc).cuda(), args, cloud_index_base=0, total_clouds=5,
                                return_device=True).cpu().numpy()
    r0.close()
    assert np.array_equal(np.load(tmp_path / "out.npy"), want)

    r = launch(["bench.py", "--gpus", "1", "--clouds", "64", "--steps", "1", "--warmup", "1", "--no-cpu-baseline"])
    assert r.returncode == 0, r.stdout + r.stderr
    line = json.loads([l for l in r.stdout.splitlines() if l.startswith("{")][-1])
    assert line["n_gpus"] == 1 and line["value"] > 0 and line["config"]["parallelism"] == "shard1+allgather"
    assert 0.05 < line["roofline"]["frac"] < 1.0
    # --scaling strong (BASELINE configs #3 / #5: ONE array over the ranks) through the same launch line
    r = launch(["bench.py", "--gpus", "1", "--clouds", "70", "--steps", "1", "--warmup", "1", "--no-cpu-baseline", "--no-extras",
                "--scaling", "strong"])
    assert r.returncode == 0, r.stdout + r.stderr
    line = json.loads([l for l in r.stdout.splitlines() if l.startswith("{")][-1])
    assert line["scaling"] == "strong" and line["value"] > 0
    assert line["config"]["clouds_total"] == 70 and line["config"]["clouds_per_gpu"] == 70


@pytest.mark.both_precisions
@pytest.mark.parametrize("k", [6, 7, 16, 17, 33, 48, 49, 100, 333, 777, 1023])
def test_optimise_point_count_sweep(restorer, golden, planes2, oracle_weights, k, precision_mode):
    """Every boundary of the kernel's blocking: K = 6 is the smallest cloud with five neighbours, 16 / 32 the decoder's
    sub-tile / tile, 48 the neighbour-list length, 1023 one short of the maximum.  Six free-running steps against the
    oracle (tolerance of P2: 1e-3 after 10 steps; here 1e-4 after 6), with a teacher-forced single step at 1e-6."""
    from oracle import convonet_oracle as O
    init = torch.from_numpy(golden["init_points"][:2, :k]).clone()
    one = O.optimize_points(oracle_weights, init, planes2, rep_weight=500.0, iterations=0, normalize=False)
    got1 = restorer.optimize_points(init, planes2, rep_weight=500.0, iterations=0, normalize=False)
    assert np.abs(got1.cpu().numpy() - one.numpy()).max() <= 1e-6
    ref = O.optimize_points(oracle_weights, init, planes2, rep_weight=500.0, iterations=5, normalize=True)
    got = restorer.optimize_points(init, planes2, rep_weight=500.0, iterations=5, normalize=True)
    assert np.linalg.norm(got.cpu().numpy() - ref.numpy(), axis=-1).max() < 1e-4


_EXACT_SCRIPT = r'''
import os, sys
import numpy as np
import torch
sys.path.insert(0, sys.argv[1])
import ifdefense_amd as I
from ifdefense_amd import _lib
from oracle import convonet_oracle as O
g = np.load(os.path.join(sys.argv[1], "tests", "golden", "convonet_golden.npz"))
planes = {pl: torch.from_numpy(g["planes01"][:, i]) for i, pl in enumerate(("xz", "xy", "yz"))}
r = I.Restorer(I.weights.pack_state_dict(O.make_random_weights(0)), device="cuda:0")
out = {"lib": np.array(os.path.basename(_lib.LIB_PATH))}
for i in (0, 1, 9, 49):
    x, m, v = (torch.from_numpy(g["traj%d_%s" % (i, k)]) for k in ("x", "m", "v"))
    x1, (m1, v1, _) = r.optimize_points(x, planes, rep_weight=500.0, steps=1, normalize=False, state=(m, v, i),
                                        return_state=True, split=1)
    out["x1_%d" % i] = x1.cpu().numpy()
    out["g_%d" % i] = (m1.cpu().numpy() - 0.9 * m.numpy()) / 0.1        # the kernel's gradient, from Adam's first moment
free = r.optimize_points(torch.from_numpy(g["init_points"][:2]), planes, rep_weight=500.0, iterations=9, normalize=False)
out["free10"] = free.cpu().numpy()
np.savez(sys.argv[2], **out)
'''


def test_exact_repulsion_build(tmp_path, golden):
    """The default build takes sqrt, 1/h, 1/d and exp of the repulsion terms from the 1-ulp hardware instructions
    (knn_device.h rep_point2); -DIFD_EXACT_REP (csrc/libifd_exact.so, built next to libifd.so) uses the IEEE expansions the
    reference's kernels use (defense/repulsion_loss.py:43-54).  Both builds, each in its own process: P1 from the
    reference's states at t = 1, 2, 10, 50 with 0 coordinates off by more than 1e-6 in EITHER build, each build's gradient
    within 5e-6 of the reference's autograd gradient, the two builds' gradients within 2e-6 of each other (of the maximum),
    and 10 free-running steps of the two builds within 1e-5."""
    import os, subprocess, sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    exact = os.path.join(root, "if-defense_amd", "csrc", "libifd_exact.so")
    assert os.path.exists(exact), "libifd_exact.so is not built (python if-defense_amd/build.py)"
    (tmp_path / "run.py").write_text(_EXACT_SCRIPT)
    res = {}
    for name, lib in (("default", ""), ("exact", exact)):
        env = dict(os.environ, IFD_LIB=lib)
        r = subprocess.run([sys.executable, str(tmp_path / "run.py"), root, str(tmp_path / (name + ".npz"))],
                           capture_output=True, text=True, timeout=600, env=env)
        assert r.returncode == 0, r.stdout + r.stderr
        res[name] = dict(np.load(tmp_path / (name + ".npz")))
    assert str(res["exact"]["lib"]) == "libifd_exact.so" and str(res["default"]["lib"]) == "libifd.so"
    for i in (0, 1, 9, 49):
        g_ref, x_next = golden["traj%d_g" % i], golden["traj%d_x_next" % i]
        gmax = np.abs(g_ref).max()
        for name in ("default", "exact"):
            flips = int((np.abs(res[name]["x1_%d" % i] - x_next) > 1e-6).sum())
            gerr = np.abs(res[name]["g_%d" % i] - g_ref).max() / gmax
            print("%-7s build, t=%d: coordinates off by > 1e-6: %d, gradient error %.2e of max" % (name, i + 1, flips, gerr))
            assert flips == 0 and gerr < 5e-6, (name, i)
        between = np.abs(res["default"]["g_%d" % i] - res["exact"]["g_%d" % i]).max() / gmax
        print("default vs exact build, t=%d: gradient difference %.2e of max" % (i + 1, between))
        assert between < 2e-6, i
    d = np.linalg.norm(res["default"]["free10"] - res["exact"]["free10"], axis=-1)
    print("default vs exact build, 10 free steps: max per-point L2 %.2e" % d.max())
    assert d.max() < 1e-5


# ------------------------------------------------------------------------------------------------
# The converged-surface regime: trained-like weights (tests/golden/train_trained_like.py: the reference model trained on
# analytic occupancy of the bench shapes - a field that crosses the iso-value on a closed surface), eight clouds, B = 8.
# Fixtures from the reference's own modules AND its own driver functions (make_golden_trained.py).
# ------------------------------------------------------------------------------------------------
@pytest.fixture(scope="module")
def trained():
    import os
    import ifdefense_amd as I
    here = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden")
    z = np.load(os.path.join(here, "trained_like_f16.npz"))
    w = {k: z[k].astype(np.float32) for k in z.files}
    f = dict(np.load(os.path.join(here, "convonet_golden_trained.npz")))
    long = np.load(os.path.join(here, "convonet_golden_trained_long.npz"))       # Adam t = 300 / 500 of a 500-iteration reference run
    assert np.array_equal(long["init_points"], f["init_points"])
    f.update({k: long[k] for k in long.files if k != "init_points"})
    planes = {pl: torch.from_numpy(f["planes_f16"][:, i].astype(np.float32)) for i, pl in enumerate(PL)}
    r = I.Restorer(I.weights.pack_state_dict(w), device="cuda:0")
    yield r, f, planes, w
    r.close()


@pytest.mark.both_precisions
def test_trained_like_decoder_and_hot_gradient(trained, precision_mode):
    r, f, planes, w_np = trained
    p = torch.from_numpy(f["init_points"])
    logits, grad = r.decode(p, planes, want_grad=True)
    assert _rel(logits.cpu().numpy(), f["dec_logits"]) < 1e-5
    assert _rel(grad.cpu().numpy(), f["dec_dlogit_dp"]) < 1e-4
    # the field has a surface: logits on both sides of logit(0.2) among the initial points of every cloud
    thr = float(np.log(0.2 / 0.8))
    assert ((f["dec_logits"] > thr).any(1) & (f["dec_logits"] < thr).any(1)).all()
    from oracle import convonet_oracle as O
    ow = O.to_torch(w_np)
    quirks, nsets = {}, {}
    for t in (0, 9, 99, 299, 499):                                 # P1 + the hot tile's gradient, Adam t = 1, 10, 100, 300, 500, B = 8
        x_next, g, v1 = _hot_gradient(r, f, planes, t)
        g_ref = f[f"traj{t}_g"].astype(np.float64)
        gmax = np.abs(g_ref).max()
        x = torch.from_numpy(f[f"traj{t}_x"])
        # Where the two gradients differ at all, it is the NEIGHBOUR CHOICE and nothing else: the reference ranks candidates by
        # the expanded form |a|^2 + |b|^2 - 2ab in f32 (~1e-7 of absolute noise; pn_utils.py:76-82), the kernel by exact
        # differences.  On the converged surface optimised points pair up (1.5e-4 apart and closer), and then (i) 5th / 6th
        # candidates swap, and (ii) column 0 of the reference's top-6 - which it drops as "self" (pn_utils.py:81-82) - is the
        # OTHER point of a pair, so that self stays in as a neighbour and the pair's term is missing from the reference's own
        # gradient.  Demonstrated by substitution: the reference's objective evaluated by autograd at the same
        # points with the kernel's 5-NN sets must reproduce the kernel's gradient at EVERY point (no exclusions), and the
        # points where the reference's own gradient differs must all be touched by a differing set.
        ref_idx = O.knn_point(5, x).numpy()
        hip_idx = r.repulsion_loss(x, want_idx=True)[1].cpu().numpy()
        g_sub = _oracle_grad(O, ow, x.numpy(), planes, 8, hip_idx).astype(np.float64)
        e_pt = np.abs(g - g_sub).max(-1) / gmax                                     # per point
        # A point beyond the bar must be a ReLU boundary inside f32 rounding (category R of the attribution protocol above: a hidden
        # unit's pre-activation within rounding of zero, where two summation orders land on opposite sides; threshold_backward has no
        # 'almost'): nudging the point by <= 512 ulps makes the ORACLE's own gradient jump by that difference.  None on the f32 tile
        # with these fixtures; the bf16x6 tile (another summation order, same accuracy) meets one now and then - at most 2 per step.
        boundary = np.zeros_like(e_pt, dtype=bool)
        for b, k in np.argwhere(e_pt > 5e-6):
            jumps = _oracle_kink_jumps(O, ow, f[f"traj{t}_x"][b, k], {pl: v[b:b + 1] for pl, v in planes.items()}, 8)
            miss = np.abs(jumps - (g[b, k] - g_sub[b, k])[None, :]).max(-1).min() / gmax
            print("    trained-like t=%d point (%d, %d): kernel - reference = %.1e of max; nearest jump of the oracle's own gradient under "
                  "nudges of <= 512 ulps is %.1e of max away" % (t + 1, b, k, e_pt[b, k], miss))
            assert miss < 1e-5, (t, b, k, miss)
            boundary[b, k] = True
        assert boundary.sum() <= (0 if precision_mode == "f32" else 2), (t, int(boundary.sum()))
        e_sub = e_pt[~boundary].max()
        differ = np.array([[set(ref_idx[b, k].tolist()) != set(hip_idx[b, k].tolist()) for k in range(1024)] for b in range(8)])
        touched = differ.copy()
        for b, k in zip(*np.nonzero(differ)):
            for j in set(ref_idx[b, k].tolist()) ^ set(hip_idx[b, k].tolist()):
                touched[b, j] = True
        off = np.abs(g - g_ref).max(-1) > 5e-6 * gmax
        own = (ref_idx == np.arange(1024)[None, :, None]).any(-1)                  # the "self" quirk (ii)
        flips = (np.abs(x_next - f[f"traj{t}_x_next"]) > 1e-6).any(-1)
        print("trained-like t=%d: gradient vs the reference with the kernel's neighbour sets %.2e of max (every point); the reference's own "
              "gradient differs at %d points, all touched by one of %d differing 5-NN sets (%d of them the self-column quirk); "
              "coordinates of x_next off by > 1e-6: %d, all at such points" % (t + 1, e_sub, int(off.sum()), int(differ.sum()), int(own.sum()),
                                                                               int(flips.sum())))
        assert e_sub < 5e-6, (t, e_sub)
        touched |= boundary
        assert not (off & ~touched).any() and not (flips & ~touched).any(), (t, int((off & ~touched).sum()), int((flips & ~touched).sum()))
        quirks[t + 1], nsets[t + 1] = int(own.sum()), int(differ.sum())
    # Both effects grow with the iteration count - converged points pair up on the surface - and stay a fringe: measured
    # (round 4) at Adam t = 1 / 10 / 100 / 300 / 500, of 8192 points
    print("reference 5-NN sets that differ from the exact ones, by Adam t: %s; of which self-column quirk: %s" % (nsets, quirks))
    assert nsets[1] == 0 and nsets[10] == 0 and nsets[100] <= 8 and nsets[300] <= 40 and nsets[500] <= 60, nsets
    assert quirks[500] >= quirks[100]


@pytest.mark.both_precisions
def test_trained_like_free_running_and_losses(trained, precision_mode):
    r, f, planes, w_np = trained
    init = torch.from_numpy(f["init_points"])
    out10 = r.optimize_points(init, planes, rep_weight=500.0, iterations=9, normalize=False).cpu().numpy()
    d = np.linalg.norm(out10 - f["traj9_x_next"], axis=-1)         # P2: 10 free-running steps
    print("trained-like P2, 10 steps, 8 clouds: max %.2e median %.2e, points > 1e-3: %d of %d" % (d.max(), np.median(d), (d > 1e-3).sum(), d.size))
    assert d.max() < 1e-3 and np.median(d) < 1e-6
    x99 = torch.from_numpy(f["traj99_x"])                          # the losses the reference computed at step 99
    _, loss = r.optimize_points(x99, planes, rep_weight=500.0, steps=1, normalize=False, loss_batch=8, return_loss=True,
                                state=(torch.from_numpy(f["traj99_m"]), torch.from_numpy(f["traj99_v"]), 99))
    occ, rep = float(loss[:, 0].sum()) / 8.0, float(loss[:, 1].mean()) * 500.0
    assert abs(occ - f["traj99_loss"][0]) / f["traj99_loss"][0] < 1e-5 and abs(rep - f["traj99_loss"][1]) / abs(f["traj99_loss"][1]) < 1e-4
    # 100 steps + normalisation against the reference function's return value: the trajectory is chaotic, so
    # this is distributional - the bulk of the points agrees closely, the surface is the same
    out = r.optimize_points(init, planes, rep_weight=500.0, iterations=99).cpu().numpy()
    d = np.linalg.norm(out - f["out100_normalised"], axis=-1)
    # ... and the fraction beyond 1e-3 is held against the reference's own sensitivity: the oracle (the reference's op
    # sequence) from initial points moved by one ulp, against the same fixture
    from oracle import convonet_oracle as O
    ow = O.to_torch(w_np)
    floor = []
    for seed in (0, 1):
        g = torch.Generator().manual_seed(200 + seed)
        up = torch.rand(init.shape, generator=g) < 0.5
        pert = torch.where(up, torch.nextafter(init, torch.full_like(init, 2.0)), torch.nextafter(init, torch.full_like(init, -2.0)))
        o = O.optimize_points(ow, pert, planes, rep_weight=500.0, iterations=99).numpy()
        floor.append(float((np.linalg.norm(o - f["out100_normalised"], axis=-1) > 1e-3).mean()))
    frac = float((d > 1e-3).mean())
    print("trained-like 100 steps + normalisation: median %.2e, points > 1e-3: %.2f %% (the oracle from 1-ulp-perturbed inputs: %s %%)" %
          (np.median(d), 100.0 * frac, ["%.2f" % (100.0 * x) for x in floor]))
    assert np.median(d) < 1e-4 and frac <= max(floor) + 0.003, (frac, floor)
    np.testing.assert_allclose(np.linalg.norm(out, axis=-1).max(axis=1), 1.0, rtol=1e-6)


@pytest.mark.both_precisions
def test_trained_like_pipeline_lists_and_split(trained, precision_mode):
    """The whole path on the trained-like field from the fixture's recorded draws, and the kernel-level invariants in the
    regime where the points settle on a surface: certified lists == exact scan, split == one workgroup per cloud."""
    r, f, planes, _ = trained
    x = torch.from_numpy(f["raw"]).cuda()
    keep = r.sor(x)
    assert (keep.sum(1).cpu().numpy() == f["sor_len"]).all()       # the reference's sor_process kept as many points
    prep = r.prepare(x, keep, sel_idx=torch.from_numpy(f["sel_idx"]), init_idx=torch.from_numpy(f["init_idx"]),
                     noise=torch.from_numpy(f["noise"]))
    np.testing.assert_allclose(prep["sel"].cpu().numpy(), f["sel"], atol=2e-7)
    np.testing.assert_allclose(prep["init"].cpu().numpy(), f["init_points"], atol=3e-7)
    pl_hip = r.encode_inputs(prep["sel"], prep["t_per_cloud"]).cpu().numpy()        # [8,3,64,64,32] channel-last
    pl_ref = f["planes_f16"].astype(np.float32).transpose(0, 1, 3, 4, 2)
    err = np.abs(pl_hip - pl_ref).max() / np.abs(pl_ref).max()
    print("trained-like encoder planes vs the reference's (rounded to f16 in the fixture): %.2e of max" % err)
    assert err < 2e-3                                              # f16 rounding of the fixture: 2^-11 relative
    init = torch.from_numpy(f["init_points"]).cuda()
    a = r.optimize_points(init, planes, rep_weight=500.0, iterations=300, normalize=False, split=1)
    c = r.counters()
    b = r.optimize_points(init, planes, rep_weight=500.0, iterations=300, normalize=False, knn_scan_every_step=True, split=1)
    assert torch.equal(a, b)
    for split in (2, 4):
        assert torch.equal(r.optimize_points(init, planes, rep_weight=500.0, iterations=300, normalize=False, split=split), a)
    p0 = torch.sigmoid(r.decode(init, planes)).cpu().numpy()
    p1 = torch.sigmoid(r.decode(a, planes)).cpu().numpy()
    print("trained-like, 301 steps: |occupancy probability - 0.2| %.3f -> %.3f; %.1f list rebuilds per cloud, ring on %.2f of the wave-steps"
          % (np.abs(p0 - 0.2).mean(), np.abs(p1 - 0.2).mean(), c["knn_rebuilds"] / 64.0, c["knn_ring_evals"] / (64.0 * 301)))
    assert np.abs(p1 - 0.2).mean() < 0.5 * np.abs(p0 - 0.2).mean()          # the points moved onto the iso-surface


@pytest.mark.gpu
def test_winograd_unet_against_the_implicit_gemm_unet():
    """The 3x3 layers of the U-Net run in the Winograd F(2x2, 3x3) domain (csrc/unet.hip wino_kernel; reference:
    src/encoder/unet.py:48-57 conv3x3).  IFD_UNET_DIRECT=1 sends them through the implicit-GEMM kernel - the plain
    nine-tap sum - instead: the two must agree to float32 rounding of the transforms (measured 1.0e-6 of the planes'
    maximum; F(2x2, 3x3) has transform constants 0, +-1, 1/2 only)."""
    import subprocess, sys, os, re
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    r = subprocess.run([sys.executable, os.path.join(root, "scripts", "wino_check.py"), "16"], capture_output=True, text=True,
                       cwd=root, timeout=600)
    assert r.returncode == 0, r.stdout + r.stderr
    m = re.search(r"max \|winograd - direct\| / max \|direct\| = ([0-9.e+-]+) .*any nan (\w+)", r.stdout)
    assert m, r.stdout + r.stderr
    assert m.group(2) == "False" and float(m.group(1)) < 3e-6, r.stdout


@pytest.mark.gpu
def test_mfma_point_encoder_against_the_thread_per_point_encoder():
    """encode_points runs its linear layers on the matrix pipe (csrc/encoder.hip encode_points_mfma_kernel; reference:
    src/encoder/pointnet.py:124-168).  IFD_ENC_VALU=1 selects the thread-per-point kernel with its sequential
    fused-multiply-add order instead: same occupied cells, features equal to float32 rounding of the sums."""
    import subprocess, sys, os, re
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    r = subprocess.run([sys.executable, os.path.join(root, "scripts", "enc_check.py"), "16"], capture_output=True, text=True,
                       cwd=root, timeout=600)
    assert r.returncode == 0, r.stdout + r.stderr
    m = re.search(r"max \|mfma - valu\| / max \|valu\| = ([0-9.e+-]+) .*occupied cells equal (\w+), any nan (\w+)", r.stdout)
    assert m, r.stdout + r.stderr
    assert m.group(2) == "True" and m.group(3) == "False" and float(m.group(1)) < 5e-6, r.stdout
