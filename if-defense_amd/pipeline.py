"""The defense driver: ``defend_point_cloud`` and the .npz in/out of ConvONet/opt_defense.py:255-369.

Everything between the input array and the restored array runs on the GPU (SOR, preprocess, subset, encoder,
init, 501-step optimiser, normalisation); the host only slices batches and copies the result back.
"""
from __future__ import annotations

import os
from dataclasses import dataclass
from typing import Optional

import numpy as np
import torch

from .runtime import Restorer


@dataclass
class DefenseArgs:
    """Mirror of the reference's CLI namespace (ConvONet/opt_defense.py:21-61), same names and defaults."""
    sample_npoint: int = 1024
    padding_scale: float = 0.9
    init_sigma: float = 0.01
    iterations: int = 200
    batch_size: int = 192
    lr: float = 0.001
    rep_weight: float = 500.0
    sor: bool = True
    sor_k: int = 2
    sor_alpha: float = 1.1
    threshold: float = 0.2          # cfg['test']['threshold']
    input_npoint: int = 600         # cfg['data']['pointcloud_n']
    seed: int = 0                   # extension: the reference is unseeded
    chunk: int = 4096               # extension: clouds per device pass (memory knob, ~17 MB of scratch per cloud;
                                    # does not change results)
    printing: bool = False          # optimize_points(..., printing=True) of the reference (opt_defense.py:229-236)
    precision: str = "f32"          # extension (opt-in): "bf16x6" (f32-equivalent) / "bf16x3" (reduced) decoder layers, ifd_opt_params.precision


def _prepare_unit(r: Restorer, xb: torch.Tensor, args: DefenseArgs, base: int, total: int):
    """SOR -> preprocess / subset / init -> encoder for one device pass (opt_defense.py:262-306), on the CURRENT stream.
    Returns what the optimiser needs: (init points, planes, per-cloud 1/B group sizes)."""
    keep = r.sor(xb, args.sor_k, args.sor_alpha) if args.sor else None
    prep = r.prepare(xb, keep, n_sel=args.input_npoint, n_opt=args.sample_npoint, padding_scale=args.padding_scale,
                     init_sigma=args.init_sigma, seed=args.seed, cloud_index_base=base)
    planes = r.encode_inputs(prep["sel"], prep["t_per_cloud"])
    # the 1/B of both losses is the size of the reference batch each cloud would have been in
    bs = int(args.batch_size)
    gidx = torch.arange(base, base + xb.shape[0], device=r.device)
    start = (gidx // bs) * bs
    lb = torch.clamp(total - start, max=bs).to(torch.int32)
    return prep["init"], planes, lb


def defend_stream(r: Restorer, arrays, args: DefenseArgs, bases=None, totals=None, overlap: bool = True, tail_first: bool = True):
    """Restore a SEQUENCE of arrays (the files of a --data_root directory, the steps of a benchmark), yielding one
    restored device tensor [N_i, sample_npoint, 3] per array, in order.

    The optimiser kernel keeps one workgroup per CU busy for ~0.9 s per 2468 clouds, and its last round leaves a third
    of the CUs idle (2468 = 9.64 x 256).  With ``overlap`` the SOR / preprocess / encoder kernels of the NEXT device pass
    are enqueued on a second HIP stream as soon as the current pass's optimiser is launched, so they run on the CUs that
    the tail frees instead of after it.  Results are bit-identical to the serial order: every kernel is deterministic and
    the passes share nothing but read-only weights (the context's encoder scratch is used by one pass at a time: the
    side stream's work is ordered; the optimiser uses the separate neighbour-list scratch and - ONet - its own buffer for the
    folded CBN coefficients, api.cpp onet_fold).

    ``tail_first`` (round 5) does the same INSIDE a file: the clouds of its partial last round (n mod CUs) are prepared and
    optimised first - their round occupies that many CUs - and the pre-processing of the file's other clouds runs on the second
    stream on the CUs that round leaves idle; then the whole rounds follow.  Same kernels on the same clouds (random draws and the
    1 / B factor are keyed by the global cloud index): bit-identical output, and no work of another file is involved, so a single
    file (``overlap=False``: BASELINE configs[1]) benefits too.

    Lazy: ``arrays`` may be a generator (the CLI loads the files of a directory on demand).  Array i + 1 is pulled,
    uploaded and given its output tensor only when its first device pass is about to be prepared - one pass ahead of the
    optimiser - and array i's input is dropped once its last pass has been prepared, so host and device memory hold two
    files at a time, not the directory.
    """
    bases = None if bases is None else list(bases)
    totals = None if totals is None else list(totals)
    on_gpu = torch.device(r.device).type == "cuda"           # (the host tests drive this with a stand-in model on the CPU)
    main = torch.cuda.current_stream(r.device) if on_gpu else None
    side = None
    n_cu = torch.cuda.get_device_properties(r.device).multi_processor_count if on_gpu else 0

    def units():
        """(array index, holder of the array's input / output tensors, lo, hi, last pass of its array), one device pass each"""
        for i, pc in enumerate(arrays):
            n = int(pc.shape[0])
            h = {"pc": pc, "x": None, "out": torch.empty(n, args.sample_npoint, 3, device=r.device, dtype=torch.float32)}
            if on_gpu and torch.is_tensor(pc) and pc.is_cuda:
                # a device tensor handed over lazily was produced on the main stream, by work queued before this moment:
                # the side stream waits for exactly that (an event recorded now), not for the optimiser launches that follow
                h["ev_in"] = torch.cuda.Event()
                h["ev_in"].record(main)
            rest = n % n_cu if n_cu else 0
            # (not with --printing: the reference prints the batches of a file in order, opt_defense.py:229-236 - the partial round
            # first would print the LAST clouds' loss lines first)
            if tail_first and rest and n > n_cu and n <= int(args.chunk) and int(args.sample_npoint) <= 1024 and not args.printing:
                yield i, h, n - rest, n, False                       # the partial round first ...
                yield i, h, 0, n - rest, True                        # ... the whole rounds behind it
                continue
            los = list(range(0, max(n, 1), int(args.chunk)))
            for lo in los:
                yield i, h, lo, min(n, lo + int(args.chunk)), lo == los[-1]

    def upload(h):
        """the array's points on the device - copied on the CURRENT stream (the side stream when passes overlap: a copy
        enqueued on the main stream would wait behind the optimiser launch it is supposed to run under)"""
        if h["x"] is None:
            if side is not None and h.get("ev_in") is not None:
                side.wait_event(h["ev_in"])
            h["x"] = torch.as_tensor(h["pc"])[..., :3].to(device=r.device, dtype=torch.float32).contiguous()
            h["pc"] = None
        return h["x"]

    def launch_prepare(u, after=None):
        """``after``: the unit whose optimiser launch has just been enqueued.  All pre-processing runs on the side stream (one
        stream: the passes' uses of the context's encoder scratch stay ordered); without ``overlap`` another FILE's pre-processing
        first waits for everything on the main stream - it does not run under the previous file's optimiser, only a unit of the
        same file does (tail_first)."""
        i, h, lo, hi, last = u
        use_side = side is not None
        if use_side and not overlap and after is not None and after[0] != i:
            side.wait_stream(main)
        base = 0 if bases is None else bases[i]
        total = int(totals[i] if (totals is not None and totals[i] is not None) else base + h["out"].shape[0])
        if hi <= lo:
            return None
        if not use_side:
            res = _prepare_unit(r, upload(h)[lo:hi], args, base + lo, total) + (None,)
        else:
            with torch.cuda.stream(side):         # (side-stream order keeps the passes' use of the encoder scratch apart)
                init, planes, lb = _prepare_unit(r, upload(h)[lo:hi], args, base + lo, total)
                ev = torch.cuda.Event()
                ev.record(side)
            for t in (init, planes, lb):          # consumed on the main stream: keep the allocator from recycling them early
                t.record_stream(main)
            res = (init, planes, lb, ev)
        if last:
            h["x"] = None                         # the input is not needed beyond its last pass's pre-processing
        return res

    it = units()
    cur_u = next(it, None)
    if cur_u is None:
        return
    nxt_u = next(it, None)
    if on_gpu and (overlap or tail_first) and nxt_u is not None:
        side = torch.cuda.Stream(r.device)
        side.wait_stream(main)                    # inputs already on the device were produced on the main stream
    ready = launch_prepare(cur_u)
    while cur_u is not None:
        i, h, lo, hi, last = cur_u
        if ready is not None:
            init, planes, lb, ev = ready
            if ev is not None:
                main.wait_event(ev)
            h["out"][lo:hi] = r.optimize_points(init, planes, rep_weight=args.rep_weight, iterations=args.iterations,
                                                lr=args.lr, loss_batch=lb, normalize=True, printing=args.printing,
                                                **({"check": False, "precision": getattr(args, "precision", "f32") or "f32"}
                                                   if on_gpu and hasattr(r, "check_status") else {}))
        ready = launch_prepare(nxt_u, cur_u) if nxt_u is not None else None        # rides on the optimiser's tail
        if last:
            if on_gpu and hasattr(r, "check_status") and os.environ.get("IFD_STATUS_CHECK", "1") != "0":
                # device-side failures of this file's launches (a split cloud's wait that gave up, fixed-point sums near
                # their range) raise here, once per file, where its result is handed on.  The synchronisation costs one launch
                # latency: the next pass's pre-processing is already queued on the side stream.
                r.check_status()
            yield h["out"]
        del h, cur_u
        cur_u, nxt_u = nxt_u, (next(it, None) if nxt_u is not None else None)


def defend_point_cloud(r: Restorer, pc, args: DefenseArgs, cloud_index_base: int = 0, total_clouds: Optional[int] = None,
                       return_device: bool = False):
    """defend_point_cloud(pc) (opt_defense.py:255-314): pc [N,K,3] (numpy or torch) -> restored [N,sample_npoint,3].

    ``cloud_index_base`` / ``total_clouds`` place this call's clouds inside a larger (sharded) array: random
    draws are keyed by the global cloud index and the 1/B loss factor (opt_defense.py:215,222) is that of the
    reference batch (size ``batch_size``, the last one shorter) the cloud would have been in - so any sharding
    gives bit-identical results.
    """
    out = next(defend_stream(r, [pc], args, [cloud_index_base], [total_clouds]))
    return out if return_device else out.cpu().numpy()


def remesh_point_cloud(r, pc, args: "DefenseArgs", cloud_index_base: int = 0, return_device: bool = False,
                       normalize: bool = True):
    """ONet/remesh_defense.py:228-262 for an array of clouds: SOR -> preprocess (300-point subset) -> encode ->
    reconstruct_mesh (MISE grid + marching cubes) -> 1024 area-weighted surface samples -> normalize_pc
    (``normalize=False`` for the train split, which the reference leaves un-normalised, :208-211).

    pc [N,K,3] -> [N,sample_npoint,3].  A cloud whose mesh comes out empty gets the reference's fallback
    (remesh_defense.py:160-170): its (post-SOR) input points, zero-padded or randomly subsampled to sample_npoint.
    """
    x = torch.as_tensor(pc)[..., :3].to(device=r.device, dtype=torch.float32).contiguous()
    N, K = x.shape[:2]
    n = int(args.sample_npoint)
    out = torch.empty(N, n, 3, device=r.device, dtype=torch.float32)
    for lo in range(0, N, int(args.chunk)):
        hi = min(N, lo + int(args.chunk))
        xb = x[lo:hi]
        keep = r.sor(xb, args.sor_k, args.sor_alpha) if args.sor else None
        prep = r.prepare(xb, keep, n_sel=args.input_npoint, n_opt=n, padding_scale=args.padding_scale,
                         init_sigma=args.init_sigma, seed=args.seed, cloud_index_base=cloud_index_base + lo)
        c = r.encode_inputs(prep["sel"], prep["t_per_cloud"])
        res = r.mesh_sample(c, n_sample=n, seed=args.seed, cloud_index_base=cloud_index_base + lo,
                            precision=getattr(args, "precision", "f32") or "f32")
        pts = res["points"]
        empty = (res["n_triangles"] == 0).nonzero().flatten().tolist()
        for b in empty:                                          # rare: reconstruction failed
            ori = xb[b][keep[b].bool()] if keep is not None else xb[b]
            fb = torch.zeros(n, 3, device=r.device)
            if ori.shape[0] > n:
                g = torch.Generator().manual_seed(int(args.seed) * 1000003 + cloud_index_base + lo + b)
                ori = ori[torch.randperm(ori.shape[0], generator=g)[:n].to(r.device)]
            fb[:ori.shape[0]] = ori
            pts[b] = fb
        out[lo:hi] = r.normalize_batch_pc(pts) if normalize else pts
    return out if return_device else out.cpu().numpy()


def get_save_name(path: str, model: str = "convonet") -> str:
    """opt_defense.py:242-252: <dir>/ConvONet-Opt/convonet_opt-<basename> (ONet/opt_defense.py: ONet-Opt/onet_opt-)."""
    sub = path.split('/')
    folder_name, prefix = {"onet": ('ONet-Opt', 'onet_opt-'), "onet-mesh": ('ONet-Mesh', 'onet_remesh-')}.get(
        model, ('ConvONet-Opt', 'convonet_opt-'))                 # ONet/remesh_defense.py:173-184
    folder = os.path.join(path[:path.rindex(sub[-1])], folder_name)
    os.makedirs(folder, exist_ok=True)
    return os.path.join(folder, prefix + sub[-1])


def defend_npz_test_data(r: Restorer, path: str, args: DefenseArgs, defend=None, save_model=None) -> str:
    """opt_defense.py:317-344: test_pc / test_label (/ target_label) in, same keys out (float32 / uint8)."""
    npz = np.load(path)
    test_pc = npz['test_pc'][..., :3]
    test_label = npz['test_label']
    target_label = npz['target_label'] if 'target_label' in npz.files else None
    fn = defend or (lambda a: defend_point_cloud(r, a, args))
    out = fn(test_pc)
    save_path = get_save_name(path, save_model or getattr(r, 'model_name', 'convonet'))
    kw = dict(test_pc=out.astype(np.float32), test_label=test_label.astype(np.uint8))
    if target_label is not None:
        kw['target_label'] = target_label.astype(np.uint8)
    np.savez(save_path, **kw)
    print('defense result saved to {}'.format(save_path))
    return save_path


def defend_npz_train_test_data(r: Restorer, path: str, args: DefenseArgs, defend=None, save_model=None) -> str:
    """opt_defense.py:347-369 (--train=True): train_pc/train_label/test_pc/test_label."""
    npz = np.load(path)
    fn = defend or (lambda a: defend_point_cloud(r, a, args))
    def_train = fn(npz['train_pc'][..., :3])
    def_test = fn(npz['test_pc'][..., :3])
    save_path = get_save_name(path, save_model or getattr(r, 'model_name', 'convonet'))
    np.savez(save_path, train_pc=def_train.astype(np.float32), train_label=npz['train_label'].astype(np.uint8),
             test_pc=def_test.astype(np.float32), test_label=npz['test_label'].astype(np.uint8))
    print('defense result saved to {}'.format(save_path))
    return save_path
